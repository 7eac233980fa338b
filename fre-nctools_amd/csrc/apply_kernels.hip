// apply_kernels.hip -- the weight-apply sweep (do_scalar_conserve_interp) as a CSR
// SpMV-style gather on gfx950, plus the kernels that build the CSR layout.
//
// The reference scatters  out[dst] += (f[src] + gx*di + gy*dj) * area  sequentially over
// the exchange cells (tools/fregrid/conserve_interp.c:593-614, :785-811) and then divides
// by the accumulated area (:831-839).  Sorting the exchange cells by destination cell with
// a STABLE order (ascending exchange-cell index inside a row) turns that into one private
// sum per destination cell that adds in exactly the reference's order -- no FP atomics and
// bitwise the same result as the serial reference for the same weights.
//
// HBM-bound: per level the kernel streams the CSR entries (32 B each for order 2) and
// gathers 8..24 B per entry from the source fields (L2/Infinity-Cache resident).
#include "xgrid_device.h"

static inline int nblk(long n, int t) { return (int)((n + t - 1) / t); }

__global__ __launch_bounds__(256) void k_csr_count(long nx, const int *x_dst, int *row_cnt)
{
  long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n < nx) atomicAdd(&row_cnt[x_dst[n]], 1);
}

__global__ __launch_bounds__(256) void k_csr_fill(long nx, const int *x_dst, const int *row_ptr, int *row_fill, int *perm)
{
  long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nx) return;
  int d = x_dst[n];
  int pos = atomicAdd(&row_fill[d], 1);
  perm[row_ptr[d] + pos] = (int)n;
}

// restore ascending exchange-cell order inside each row (rows are short: ~4 entries)
__global__ __launch_bounds__(256) void k_csr_sort_rows(int ndst, const int *row_ptr, int *perm)
{
  int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= ndst) return;
  int b = row_ptr[d], e = row_ptr[d + 1];
  for (int i = b + 1; i < e; i++) {
    int v = perm[i], j = i - 1;
    while (j >= b && perm[j] > v) { perm[j + 1] = perm[j]; j--; }
    perm[j + 1] = v;
  }
}

template <int ORDER>
__global__ __launch_bounds__(256) void k_csr_gather(long nx, const int *perm, const int *x_src, const double *x_area,
                                                     const double *x_c1, const double *x_c2, const int *src_idx_f, FgCsr csr)
{
  long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nx) return;
  int n = perm[e];
  int s = x_src[n];
  csr.idx_g[e] = s;
  csr.idx_f[e] = src_idx_f[s];
  csr.area[e] = x_area[n];
  if (ORDER == 2) { csr.di[e] = x_c1[n]; csr.dj[e] = x_c2[n]; }
}

// index of source cell s inside one level of the field array: order 1 fields have no halo
// (index == s); order 2 fields carry a 1-cell halo per tile (fregrid_util.c:2137-2145)
__global__ __launch_bounds__(256) void k_src_field_index(int order, const FgTile *tiles, int ntiles, int nsrc, int *src_idx_f)
{
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsrc) return;
  if (order != 2) { src_idx_f[s] = s; return; }
  int t = 0, foff = 0;
  while (t + 1 < ntiles && s >= tiles[t + 1].cell_off) { foff += (tiles[t].nx + 2) * (tiles[t].ny + 2); t++; }
  int loc = s - tiles[t].cell_off;
  int i = loc % tiles[t].nx, j = loc / tiles[t].nx;
  src_idx_f[s] = foff + (j + 1) * (tiles[t].nx + 2) + i + 1;
}

// one thread per (destination cell, level)
template <int ORDER, bool MISSING>
__global__ __launch_bounds__(256) void k_apply(int ndst, FgCsr csr, const double *data, const double *gx, const double *gy,
                                                const int *gmask, double missing, long f_stride, long g_stride,
                                                double *out, double *row_sum)
{
  int d = blockIdx.x * blockDim.x + threadIdx.x;
  int k = blockIdx.y;
  if (d >= ndst) return;
  const double *f = data + (size_t)k * f_stride;
  const double *px = (ORDER == 2) ? gx + (size_t)k * g_stride : nullptr;
  const double *py = (ORDER == 2) ? gy + (size_t)k * g_stride : nullptr;
  int b = csr.row_ptr[d], e = csr.row_ptr[d + 1];
  double acc = 0.0, asum = 0.0;
  int touched = 0;
  for (int q = b; q < e; q++) {
    double a = csr.area[q];
    double v = f[csr.idx_f[q]];
    if (MISSING) { if (v == missing) continue; }
    if (ORDER == 2) {
      int g = csr.idx_g[q];
      bool flatgrad = false;
      if (MISSING) flatgrad = gmask[g] != 0;
      if (!flatgrad) v = (v + px[g] * csr.di[q] + py[g] * csr.dj[q]);
    }
    acc += v * a;
    asum += a;
    touched = 1;
  }
  size_t o = (size_t)k * ndst + d;
  if (row_sum) row_sum[o] = (asum > 0) ? acc : 0.0;          // conserve_interp.c:815-819
  double r;                                                   // :831-839
  if (asum > 0) r = acc / asum;
  else if (touched) r = 0.0;
  else r = missing;
  out[o] = r;
}

// interp.c:262-305 (conserve_interp): weights are xarea / (sum of xarea in the destination cell)
__global__ __launch_bounds__(256) void k_apply_frac(int ndst, FgCsr csr, const double *data, double *out)
{
  int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= ndst) return;
  int b = csr.row_ptr[d], e = csr.row_ptr[d + 1];
  double asum = 0.0, acc = 0.0;
  for (int q = b; q < e; q++) asum += csr.area[q];
  for (int q = b; q < e; q++) {
    double frac = csr.area[q] / asum;
    acc += data[csr.idx_f[q]] * frac;
  }
  out[d] = acc;
}

// deterministic two-stage sum
__global__ __launch_bounds__(256) void k_reduce_partial(const double *v, long n, double *partial)
{
  __shared__ double sh[256];
  double s = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += v[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(256) void k_reduce_final(const double *partial, int np, double *result)
{
  __shared__ double sh[256];
  double s = 0;
  for (int i = threadIdx.x; i < np; i += 256) s += partial[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) *result = sh[0];
}

#define REDUCE_BLOCKS 512

void fgd_csr_count(long nx, const int *x_dst, int *row_cnt, hipStream_t st)
{
  if (nx > 0) k_csr_count<<<nblk(nx, 256), 256, 0, st>>>(nx, x_dst, row_cnt);
}
void fgd_csr_fill(long nx, const int *x_dst, const int *row_ptr, int *row_fill, int *perm, hipStream_t st)
{
  if (nx > 0) k_csr_fill<<<nblk(nx, 256), 256, 0, st>>>(nx, x_dst, row_ptr, row_fill, perm);
}
void fgd_csr_sort_rows(int ndst, const int *row_ptr, int *perm, hipStream_t st)
{
  if (ndst > 0) k_csr_sort_rows<<<nblk(ndst, 256), 256, 0, st>>>(ndst, row_ptr, perm);
}
void fgd_csr_gather(int order, long nx, const int *perm, const int *x_src, const double *x_area, const double *x_c1,
                    const double *x_c2, const int *src_idx_f, FgCsr csr, hipStream_t st)
{
  if (nx <= 0) return;
  if (order == 2) k_csr_gather<2><<<nblk(nx, 256), 256, 0, st>>>(nx, perm, x_src, x_area, x_c1, x_c2, src_idx_f, csr);
  else            k_csr_gather<1><<<nblk(nx, 256), 256, 0, st>>>(nx, perm, x_src, x_area, x_c1, x_c2, src_idx_f, csr);
}
void fgd_src_field_index(int order, const FgTile *tiles_dev, int ntiles, int nsrc, int *src_idx_f, hipStream_t st)
{
  if (nsrc > 0) k_src_field_index<<<nblk(nsrc, 256), 256, 0, st>>>(order, tiles_dev, ntiles, nsrc, src_idx_f);
}
void fgd_apply(int order, int ndst, FgCsr csr, const double *data, const double *gx, const double *gy,
               const int *gmask, int has_missing, double missing, int nz, long f_stride, long g_stride,
               double *out, double *row_sum, hipStream_t st)
{
  if (ndst <= 0 || nz <= 0) return;
  dim3 grid(nblk(ndst, 256), nz);
  if (order == 2) {
    if (has_missing) k_apply<2, true><<<grid, 256, 0, st>>>(ndst, csr, data, gx, gy, gmask, missing, f_stride, g_stride, out, row_sum);
    else             k_apply<2, false><<<grid, 256, 0, st>>>(ndst, csr, data, gx, gy, gmask, missing, f_stride, g_stride, out, row_sum);
  } else {
    if (has_missing) k_apply<1, true><<<grid, 256, 0, st>>>(ndst, csr, data, gx, gy, gmask, missing, f_stride, g_stride, out, row_sum);
    else             k_apply<1, false><<<grid, 256, 0, st>>>(ndst, csr, data, gx, gy, gmask, missing, f_stride, g_stride, out, row_sum);
  }
}
void fgd_apply_frac(int ndst, FgCsr csr, const double *data, double *out, hipStream_t st)
{
  if (ndst > 0) k_apply_frac<<<nblk(ndst, 256), 256, 0, st>>>(ndst, csr, data, out);
}
void fgd_reduce_sum(const double *v, long n, double *partial, double *result, hipStream_t st)
{
  k_reduce_partial<<<REDUCE_BLOCKS, 256, 0, st>>>(v, n, partial);
  k_reduce_final<<<1, 256, 0, st>>>(partial, REDUCE_BLOCKS, result);
}
