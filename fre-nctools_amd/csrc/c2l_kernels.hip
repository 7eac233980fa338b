// c2l_kernels.hip -- order-2 input preparation on the device: halo fill across tile contacts,
// the cubed-sphere gradient grad_c2l and the missing-value gradient mask.
//
//   k_pack_interior   copy [nz][ncells] fields into the halo'd layout [nz][F]   (fregrid_util.c:2137-2145)
//   k_halo_gather     update_halo for every tile at once through the gather map built by fg_halo_map
//                     (fregrid_util.c:2168-2182, :2614-2658)
//   k_grad_c2l        a2b_ord2 + Green's-theorem gradient + projection, fused: one thread per cell evaluates the
//                     four B-grid corner values it needs (gradient_c2l.c:58-118, :124-195).  Pure add/mul/div in
//                     the reference's order => bit-identical to the CPU reference (compiled -ffp-contract=off).
//   k_grad_mask       3x3 missing-value stencil (fregrid_util.c:2203-2216)
#include "xgrid_device.h"

struct C2lTile {
  int nx, ny;
  long cell_off;      // first cell of the tile in [ncells] arrays
  long f_off;         // first element of the tile in halo'd [F] arrays
  long dx_off, dy_off;  // offsets into dx / en_n ([ny+1][nx]) and dy / en_e ([ny][nx+1]) concatenations
  long ew_off, es_off;  // offsets into edge_w/e ([ny+1]) and edge_s/n ([nx+1]) concatenations
};

struct C2lGeom {
  const double *dx, *dy, *area, *edge_w, *edge_e, *edge_s, *edge_n, *en_n, *en_e, *vlon, *vlat;
};

__device__ __forceinline__ int d_find_tile(const C2lTile *tiles, int ntiles, long cell)
{
  int t = 0;
  while (t + 1 < ntiles && cell >= tiles[t + 1].cell_off) t++;
  return t;
}

__global__ __launch_bounds__(256) void k_pack_interior(const C2lTile *tiles, int ntiles, long ncells, long F, int nz,
                                                        const double *src, double *dst)
{
  long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int k = blockIdx.y;
  if (c >= ncells) return;
  const C2lTile T = tiles[d_find_tile(tiles, ntiles, c)];
  long loc = c - T.cell_off;
  int i = (int)(loc % T.nx), j = (int)(loc / T.nx);
  dst[(size_t)k * F + T.f_off + (long)(j + 1) * (T.nx + 2) + i + 1] = src[(size_t)k * ncells + c];
}

__global__ __launch_bounds__(256) void k_halo_gather(long F, int nz, const int *map, double *data)
{
  long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int k = blockIdx.y;
  if (e >= F) return;
  int src = map[e];
  if (src >= 0) data[(size_t)k * F + e] = data[(size_t)k * F + src];   // sources are interior cells: no read/write overlap
}

// B-grid value at corner (i, j) of a tile from the four A-grid cells around it, all four tile edges treated as cubed-sphere
// edges (gradient_c2l.c:124-195).  cXY = halo'd cell (i+X, j+Y); the sums keep the reference's term order.
__device__ __forceinline__ double d_a2b4(double c00, double c10, double c01, double c11, int nx, int ny, int i, int j,
                                         const double *edge_w, const double *edge_e, const double *edge_s, const double *edge_n)
{
  const double r3 = 1. / 3.;
  if (i == 0 && j == 0)   return r3 * (c11 + c01 + c10);
  if (i == nx && j == 0)  return r3 * (c01 + c00 + c11);
  if (i == nx && j == ny) return r3 * (c00 + c10 + c01);
  if (i == 0 && j == ny)  return r3 * (c10 + c00 + c11);
  if (i == 0) {
    double a = 0.5 * (c00 + c10), b = 0.5 * (c01 + c11);
    return edge_w[j] * a + (1 - edge_w[j]) * b;
  }
  if (i == nx) {
    double a = 0.5 * (c00 + c10), b = 0.5 * (c01 + c11);
    return edge_e[j] * a + (1 - edge_e[j]) * b;
  }
  if (j == 0) {
    double a = 0.5 * (c00 + c01), b = 0.5 * (c10 + c11);
    return edge_s[i] * a + (1 - edge_s[i]) * b;
  }
  if (j == ny) {
    double a = 0.5 * (c00 + c01), b = 0.5 * (c10 + c11);
    return edge_n[i] * a + (1 - edge_n[i]) * b;
  }
  return 0.25 * (c00 + c10 + c01 + c11);
}

// grad_c2l of one cell: the level-independent geometry is loaded once (ctor), then one call per level.  The expression order
// is the reference's (gradient_c2l.c:84-117), so both kernels below give its bits.
struct C2lCell {
  int nx, ny, i, j;
  const double *ew, *ee, *es, *en;
  double dxs, dxn, dyw, dye, area;
  double ens[3], enn3[3], enw[3], ene3[3], vlo[3], vla[3];
  __device__ C2lCell(const C2lTile &T, long c, const C2lGeom &g)
  {
    nx = T.nx; ny = T.ny;
    const int nxp = nx + 1;
    const long loc = c - T.cell_off;
    i = (int)(loc % nx); j = (int)(loc / nx);
    ew = g.edge_w + T.ew_off; ee = g.edge_e + T.ew_off; es = g.edge_s + T.es_off; en = g.edge_n + T.es_off;
    const double *dx = g.dx + T.dx_off, *dy = g.dy + T.dy_off;
    const double *enn = g.en_n + 3 * T.dx_off, *ene = g.en_e + 3 * T.dy_off;
    const long ms = (long)j * nx + i, mn = (long)(j + 1) * nx + i;       // south / north edges of the cell
    const long mw = (long)j * nxp + i, me = mw + 1;                       // west / east edges
    dxs = dx[ms]; dxn = dx[mn]; dyw = dy[mw]; dye = dy[me];
#pragma unroll
    for (int n = 0; n < 3; n++) {
      ens[n] = enn[3 * ms + n]; enn3[n] = enn[3 * mn + n]; enw[n] = ene[3 * mw + n]; ene3[n] = ene[3 * me + n];
      vlo[n] = g.vlon[3 * c + n]; vla[n] = g.vlat[3 * c + n];
    }
    area = g.area[c];
  }
  // q = the tile's halo'd level
  __device__ __forceinline__ void level(const double *q, double *gx_out, double *gy_out) const
  {
    const int w = nx + 2;
    double v[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; dy++)
#pragma unroll
      for (int dx = 0; dx < 3; dx++) v[dy][dx] = q[(size_t)(j + dy) * w + i + dx];
    stencil(v, gx_out, gy_out);
  }
  // v[dy][dx] = halo'd cell (i + dx, j + dy): the cell itself is v[1][1]
  __device__ __forceinline__ void stencil(const double (&v)[3][3], double *gx_out, double *gy_out) const
  {
    const double b00 = d_a2b4(v[0][0], v[0][1], v[1][0], v[1][1], nx, ny, i, j, ew, ee, es, en);
    const double b10 = d_a2b4(v[0][1], v[0][2], v[1][1], v[1][2], nx, ny, i + 1, j, ew, ee, es, en);
    const double b01 = d_a2b4(v[1][0], v[1][1], v[2][0], v[2][1], nx, ny, i, j + 1, ew, ee, es, en);
    const double b11 = d_a2b4(v[1][1], v[1][2], v[2][1], v[2][2], nx, ny, i + 1, j + 1, ew, ee, es, en);
    double g3[3];
#pragma unroll
    for (int n = 0; n < 3; n++) {
      double pdx_s = 0.5 * (b00 + b10) * dxs * ens[n];
      double pdx_n = 0.5 * (b01 + b11) * dxn * enn3[n];
      double pdy_w = 0.5 * (b00 + b01) * dyw * enw[n];
      double pdy_e = 0.5 * (b10 + b11) * dye * ene3[n];
      g3[n] = pdx_n - pdx_s - pdy_w + pdy_e;
    }
    double gx = (vlo[0] * g3[0] + vlo[1] * g3[1] + vlo[2] * g3[2]) / area;
    gx *= 6371000.;
    double gy = (vla[0] * g3[0] + vla[1] * g3[1] + vla[2] * g3[2]) / area;
    gy *= 6371000.;
    *gx_out = gx; *gy_out = gy;
  }
};

__global__ __launch_bounds__(256) void k_grad_c2l(const C2lTile *tiles, int ntiles, long ncells, long F, int nz,
                                                   const double *data, C2lGeom g, double *grad_x, double *grad_y)
{
  long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncells) return;
  const C2lTile T = tiles[d_find_tile(tiles, ntiles, c)];
  const C2lCell cell(T, c, g);
  for (int k = 0; k < nz; k++) {
    double gx, gy;
    cell.level(data + (size_t)k * F + T.f_off, &gx, &gy);
    grad_x[(size_t)k * ncells + c] = gx;
    grad_y[(size_t)k * ncells + c] = gy;
  }
}

// Block-wide store of CB records held in registers (row[3*NB] per thread, thread t = cell c0 + t) as one contiguous run, through
// an LDS tile of CB/PASSES rows: with all 64 rows of a wave staged at once (12.8 KB at NB = 8) LDS alone would cap a CU at
// three waves per SIMD, so the rows go in PASSES batches.
template <int NB, int CB, int PASSES>
__device__ __forceinline__ void d_store_records(const double *row, bool valid, long c0, long ncells, double *rec)
{
  constexpr int R = 3 * NB, H = CB / PASSES;
  __shared__ double tile[H * (R + 1)];
#pragma unroll
  for (int p = 0; p < PASSES; p++) {
    if (p) __syncthreads();
    if (valid && (int)threadIdx.x / H == p) {
      double *dst = tile + ((int)threadIdx.x % H) * (R + 1);
#pragma unroll
      for (int q = 0; q < R; q++) dst[q] = row[q];
    }
    __syncthreads();
    const long first = c0 + (long)p * H;
    const long left = ncells - first;
    const long cnt = (left < H ? (left < 0 ? 0 : left) : H) * R;
    for (long e = threadIdx.x; e < cnt; e += CB) rec[(size_t)first * R + e] = tile[(e / R) * (R + 1) + (e % R)];
  }
}

// The same gradients written as the sweep's merged records rec[cell][3][NB] = {field, grad_x, grad_y} x levels (zero padded
// beyond nz): what fg_plan_apply would otherwise build from the level-major arrays with k_merge3.  One lane per cell computes
// its record into LDS; the block then stores the CB records as one contiguous run.
template <int NB, int CB>
__global__ __launch_bounds__(CB) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_grad_c2l_rec(const C2lTile *tiles, int ntiles, long ncells, long F, int nz,
                                                     const double *data, C2lGeom g, double *rec)
{
  constexpr int R = 3 * NB;
  const long c0 = (long)blockIdx.x * CB;
  const long c = c0 + threadIdx.x;
  double row[R];
  if (c < ncells) {
    const C2lTile T = tiles[d_find_tile(tiles, ntiles, c)];
    const C2lCell cell(T, c, g);
    const long fc = T.f_off + (long)(cell.j + 1) * (T.nx + 2) + cell.i + 1;
#pragma unroll
    for (int k = 0; k < NB; k++) {
      double f = 0.0, gx = 0.0, gy = 0.0;
      if (k < nz) {
        f = data[(size_t)k * F + fc];
        cell.level(data + (size_t)k * F + T.f_off, &gx, &gy);
      }
      row[k] = f; row[NB + k] = gx; row[2 * NB + k] = gy;
    }
  }
  d_store_records<NB, CB, (NB >= 8 ? 2 : 1)>(row, c < ncells, c0, ncells, rec);
}

// Unpadded levels src[nz][ncells] -> the sweep's records in one pass: no halo'd copy, no halo fill.  cell_of[e] is, for
// every element e of the halo'd layout, the (unpadded) cell whose value update_halo would leave there: the cell itself in
// the interior, the neighbour tile's cell in the halo, -1 where init_halo's zero stays (fg_c2l_create builds it from the same
// gather map k_halo_gather uses).  Same stencil, same arithmetic as k_grad_c2l_rec on filled halo'd data.
template <int NB, int CB>
__global__ __launch_bounds__(CB) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_c2l_records(const C2lTile *tiles, int ntiles, long ncells, int nz, const double *src,
                                                    const int *cell_of, C2lGeom g, double *rec)
{
  constexpr int R = 3 * NB;
  const long c0 = (long)blockIdx.x * CB;
  const long c = c0 + threadIdx.x;
  double row[R];
  if (c < ncells) {
    const C2lTile T = tiles[d_find_tile(tiles, ntiles, c)];
    const C2lCell cell(T, c, g);
    int at[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; dy++)
#pragma unroll
      for (int dx = 0; dx < 3; dx++) at[dy][dx] = cell_of[T.f_off + (long)(cell.j + dy) * (T.nx + 2) + cell.i + dx];
#pragma unroll
    for (int k = 0; k < NB; k++) {
      double f = 0.0, gx = 0.0, gy = 0.0;
      if (k < nz) {
        const double *lev = src + (size_t)k * ncells;
        double v[3][3];
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
          for (int dx = 0; dx < 3; dx++) v[dy][dx] = (at[dy][dx] >= 0) ? lev[at[dy][dx]] : 0.0;
        f = v[1][1];
        cell.stencil(v, &gx, &gy);
      }
      row[k] = f; row[NB + k] = gx; row[2 * NB + k] = gy;
    }
  }
  d_store_records<NB, CB, (NB >= 8 ? 2 : 1)>(row, c < ncells, c0, ncells, rec);
}

__global__ __launch_bounds__(256) void k_grad_mask(const C2lTile *tiles, int ntiles, long ncells, long F, int nz,
                                                    const double *data, double missing, int *mask)
{
  long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int k = blockIdx.y;
  if (c >= ncells) return;
  const C2lTile T = tiles[d_find_tile(tiles, ntiles, c)];
  const int w = T.nx + 2;
  long loc = c - T.cell_off;
  const int ii = (int)(loc % T.nx) + 1, jj = (int)(loc / T.nx) + 1;
  const double *q = data + (size_t)k * F + T.f_off;
  int m = 0;
  for (int dj = -1; dj <= 1; dj++)
    for (int di = -1; di <= 1; di++) {
      if (di == 0 && dj == 0) continue;
      if (q[(jj + dj) * w + ii + di] == missing) m = 1;
    }
  mask[(size_t)k * ncells + c] = m;
}

static inline int nblk(long n, int t) { return (int)((n + t - 1) / t); }

void fgd_pack_interior(const void *tiles, int ntiles, long ncells, long F, int nz, const double *src, double *dst, hipStream_t st)
{
  if (ncells > 0 && nz > 0) k_pack_interior<<<dim3(nblk(ncells, 256), nz), 256, 0, st>>>((const C2lTile *)tiles, ntiles, ncells, F, nz, src, dst);
}
void fgd_halo_gather(long F, int nz, const int *map, double *data, hipStream_t st)
{
  if (F > 0 && nz > 0) k_halo_gather<<<dim3(nblk(F, 256), nz), 256, 0, st>>>(F, nz, map, data);
}
void fgd_grad_c2l(const void *tiles, int ntiles, long ncells, long F, int nz, const double *data, const double *const *geom,
                  double *grad_x, double *grad_y, hipStream_t st)
{
  if (ncells <= 0 || nz <= 0) return;
  C2lGeom g{geom[0], geom[1], geom[2], geom[3], geom[4], geom[5], geom[6], geom[7], geom[8], geom[9], geom[10]};
  k_grad_c2l<<<nblk(ncells, 256), 256, 0, st>>>((const C2lTile *)tiles, ntiles, ncells, F, nz, data, g, grad_x, grad_y);
}
void fgd_grad_c2l_rec(const void *tiles, int ntiles, long ncells, long F, int nz, int nb_pad, const double *data,
                      const double *const *geom, double *rec, hipStream_t st)
{
  if (ncells <= 0 || nz <= 0) return;
  C2lGeom g{geom[0], geom[1], geom[2], geom[3], geom[4], geom[5], geom[6], geom[7], geom[8], geom[9], geom[10]};
  const C2lTile *T = (const C2lTile *)tiles;
  if (nb_pad == 8) k_grad_c2l_rec<8, 64><<<nblk(ncells, 64), 64, 0, st>>>(T, ntiles, ncells, F, nz, data, g, rec);
  else if (nb_pad == 4) k_grad_c2l_rec<4, 128><<<nblk(ncells, 128), 128, 0, st>>>(T, ntiles, ncells, F, nz, data, g, rec);
  else k_grad_c2l_rec<2, 128><<<nblk(ncells, 128), 128, 0, st>>>(T, ntiles, ncells, F, nz, data, g, rec);
}
void fgd_c2l_records(const void *tiles, int ntiles, long ncells, int nz, int nb_pad, const double *src, const int *cell_of,
                     const double *const *geom, double *rec, hipStream_t st)
{
  if (ncells <= 0 || nz <= 0) return;
  C2lGeom g{geom[0], geom[1], geom[2], geom[3], geom[4], geom[5], geom[6], geom[7], geom[8], geom[9], geom[10]};
  const C2lTile *T = (const C2lTile *)tiles;
  if (nb_pad == 8) k_c2l_records<8, 64><<<nblk(ncells, 64), 64, 0, st>>>(T, ntiles, ncells, nz, src, cell_of, g, rec);
  else if (nb_pad == 4) k_c2l_records<4, 128><<<nblk(ncells, 128), 128, 0, st>>>(T, ntiles, ncells, nz, src, cell_of, g, rec);
  else k_c2l_records<2, 128><<<nblk(ncells, 128), 128, 0, st>>>(T, ntiles, ncells, nz, src, cell_of, g, rec);
}
void fgd_grad_mask(const void *tiles, int ntiles, long ncells, long F, int nz, const double *data, double missing, int *mask, hipStream_t st)
{
  if (ncells > 0 && nz > 0) k_grad_mask<<<dim3(nblk(ncells, 256), nz), 256, 0, st>>>((const C2lTile *)tiles, ntiles, ncells, F, nz, data, missing, mask);
}
size_t fgd_c2l_tile_size(void) { return sizeof(C2lTile); }
void fgd_c2l_tile_fill(void *dst, int idx, int nx, int ny, long cell_off, long f_off, long dx_off, long dy_off, long ew_off, long es_off)
{
  C2lTile *t = (C2lTile *)dst + idx;
  t->nx = nx; t->ny = ny; t->cell_off = cell_off; t->f_off = f_off; t->dx_off = dx_off; t->dy_off = dy_off; t->ew_off = ew_off; t->es_off = es_off;
}
