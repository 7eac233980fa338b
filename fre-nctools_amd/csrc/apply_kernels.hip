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
#include <type_traits>

static inline int nblk(long n, int t) { return (int)((n + t - 1) / t); }

__global__ __launch_bounds__(256) void k_csr_count(long nx, const int *x_dst, int *row_cnt)
{
  long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n < nx) atomicAdd(&row_cnt[x_dst[n]], 1);
}

// Slot of every exchange cell in its destination row.  Exchange cells are ordered by source cell, so in a fine -> coarse remap
// long runs of consecutive cells fall into the same row (C768 -> 1 deg: ~8): one atomic per run of equal rows within a wave
// instead of one per cell (the per-cell version spent 0.4 ms there on same-address atomics); the cells of a run take
// consecutive slots in ascending order.
__global__ __launch_bounds__(256) void k_csr_fill(long nx, const int *x_dst, const int *row_ptr, int *row_fill, int *perm)
{
  const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int d = (n < nx) ? x_dst[n] : -1;
  const int prev = __shfl_up(d, 1);
  const bool head = (lane == 0) || (d != prev);
  const unsigned long long heads = __ballot(head);
  const int start = 63 - __clzll((long long)(heads & ((2ull << lane) - 1ull)));     // head of this lane's run
  const unsigned long long above = (start == 63) ? 0ull : (heads & ~((2ull << start) - 1ull));
  const int end = above ? (__ffsll((long long)above) - 1) : 64;
  int base = 0;
  if (lane == start && d >= 0) base = atomicAdd(&row_fill[d], end - start);
  base = __shfl(base, start);
  if (d >= 0) perm[row_ptr[d] + base + (lane - start)] = (int)n;
}

// the same with slots already taken while the search compacted its exchange cells (k_compact): no atomics.  Launched for
// the capacity of the exchange-cell arrays when the host does not know the count yet (nx_dev)
__global__ __launch_bounds__(256) void k_csr_fill_pos(long nx, const unsigned long long *nx_dev, const int *x_dst, const int *row_ptr,
                                                       const int *x_rowpos, int *perm)
{
  if (nx_dev) { const unsigned long long nd = *nx_dev; if (nd < (unsigned long long)nx) nx = (long)nd; }
  const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n < nx) perm[row_ptr[x_dst[n]] + x_rowpos[n]] = (int)n;
}

// Rows of `perm` into ascending exchange-cell order, then the packed CSR records -- one kernel.  A block owns RPW consecutive
// rows, i.e. one contiguous run of perm: it is staged in LDS, every lane sorts its own row there by insertion (rows are short
// when the grids are of similar resolution, ~4 entries); rows longer than SHORT (fine -> coarse remaps: 50-1000 entries) are
// sorted by the whole wave, every lane placing its elements at their rank (the exchange-cell numbers are distinct).  The
// records then leave in one coalesced sweep over the run.  RPW = 64 when rows are short, 16 when the mean row is long, so that
// four times as many waves are in flight.  Runs beyond the staging capacity are sorted in place in global memory.
// DIST (order 2): x_c1/x_c2 still hold the centroid integrals; di = clon/area - cen_lon, dj = clat/area - cen_lat
// (conserve_interp.c:256-257,355-356) are formed here, the operations k_distances applies to the arrays themselves.
// rank[k] += number of keys[0..n) below v[k]: one walk over the keys (LDS) for N values -- a walk per value waits for LDS once
// per comparison
template <int N>
__device__ __forceinline__ void d_rank_count(const int *keys, int n, const int (&v)[N], int (&rank)[N])
{
  int j = 0;
  for (; j + 4 <= n; j += 4) {
    const int x0 = keys[j], x1 = keys[j + 1], x2 = keys[j + 2], x3 = keys[j + 3];
#pragma unroll
    for (int k = 0; k < N; k++) rank[k] += ((x0 < v[k]) ? 1 : 0) + ((x1 < v[k]) ? 1 : 0) + ((x2 < v[k]) ? 1 : 0) + ((x3 < v[k]) ? 1 : 0);
  }
  for (; j < n; j++) {
    const int x = keys[j];
#pragma unroll
    for (int k = 0; k < N; k++) rank[k] += (x < v[k]) ? 1 : 0;
  }
}

template <int ORDER, int RPW, bool DIST, int BT>
__global__ __launch_bounds__(BT) void k_csr_sortgather(int ndst, int *perm, const int *x_src, const double *x_area, const double *x_c1,
                                                        const double *x_c2, const int *src_idx_f, const double *cen, int nsrc, FgCsr csr,
                                                        int *tmp, long ntmp)
{
  constexpr int SHORT = 12, CAP = (RPW >= 64) ? 512 : 2048;                // 4 KB of LDS per wave in the short-row case
  // (BT: 64 lanes for 64 short rows; 256 for 16 long rows -- their ~1000 records leave through three dependent gathers each, and
  // with one wave that was 18 rounds of them per block: C384 -> 2 deg finalize 0.186 ms)
  __shared__ int sh[CAP], sh2[CAP];
  __shared__ int long_b[RPW], long_n[RPW];
  __shared__ int nlong;
  if (threadIdx.x == 0) nlong = 0;
  __syncthreads();
  const int d0 = blockIdx.x * RPW, d1 = min(d0 + RPW, ndst);
  const int q0 = csr.row_ptr[d0], q1 = csr.row_ptr[d1], nq = q1 - q0;
  const bool staged = nq <= CAP;
  const int d = d0 + threadIdx.x;
  int b = 0, e = 0;
  if ((int)threadIdx.x < RPW && d < ndst) { b = csr.row_ptr[d]; e = csr.row_ptr[d + 1]; }
  if (staged) {
    for (int i = threadIdx.x; i < nq; i += BT) sh[i] = perm[q0 + i];
    __syncthreads();
    if (e - b > SHORT) { const int q = atomicAdd(&nlong, 1); long_b[q] = b - q0; long_n[q] = e - b; }
    else
      for (int i = b - q0 + 1; i < e - q0; i++) {
        int v = sh[i], j = i - 1;
        while (j >= b - q0 && sh[j] > v) { sh[j + 1] = sh[j]; j--; }
        sh[j + 1] = v;
      }
    __syncthreads();
    const int nl = nlong;
    for (int q = 0; q < nl; q++) {
      const int rb = long_b[q], n = long_n[q];
      // every lane ranks up to CAP / BT = 8 elements of the row in ONE walk over its keys (a walk per element waits for LDS
      // once per comparison: 0.82 ms for the 1 850-entry rows of C384 -> 10 deg)
      constexpr int OWN = CAP / BT;
      if (n <= BT) {                                       // (block-uniform) one element per lane at most
        if ((int)threadIdx.x < n) {
          const int v = sh[rb + threadIdx.x];
          int rank = 0;
          for (int j = 0; j < n; j++) rank += (sh[rb + j] < v) ? 1 : 0;
          sh2[rb + rank] = v;
        }
      } else {
        int v[OWN], rank[OWN];
#pragma unroll
        for (int k = 0; k < OWN; k++) { const int i = threadIdx.x + BT * k; v[k] = (i < n) ? sh[rb + i] : 0x7fffffff; rank[k] = 0; }
        d_rank_count<OWN>(sh + rb, n, v, rank);
#pragma unroll
        for (int k = 0; k < OWN; k++) if ((int)threadIdx.x + BT * k < n) sh2[rb + rank[k]] = v[k];
      }
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += BT) sh[rb + i] = sh2[rb + i];
      __syncthreads();
    }
  } else {
    // beyond the staging capacity: per row in global memory (lane-serial insertion, or the wave through LDS for long rows)
    if (e - b > SHORT) { const int q = atomicAdd(&nlong, 1); long_b[q] = b; long_n[q] = e - b; }
    else
      for (int i = b + 1; i < e; i++) {
        int v = perm[i], j = i - 1;
        while (j >= b && perm[j] > v) { perm[j + 1] = perm[j]; j--; }
        perm[j + 1] = v;
      }
    __syncthreads();
    const int nl = nlong;
    for (int q = 0; q < nl; q++) {
      const int rb = long_b[q], n = long_n[q];
      if (n <= CAP) {
        for (int i = threadIdx.x; i < n; i += BT) sh[i] = perm[rb + i];
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += BT) {
          const int v = sh[i];
          int rank = 0;
          for (int j = 0; j < n; j++) rank += (sh[j] < v) ? 1 : 0;
          perm[rb + rank] = v;
        }
        __syncthreads();
      } else if (tmp) {
        // a row longer than the staging buffer (C384 -> 10 deg: 1 850 exchange cells): its keys pass through LDS a buffer-full at
        // a time, every element counts the smaller keys of each tile into its rank (kept in tmp), then the row is written out in
        // rank order (second half of tmp) and copied back.  (Until round 3 one lane sorted such a row by insertion, in global
        // memory: 1.9 SECONDS for that remap's finalize.)
        int *rk = tmp + rb, *srt = tmp + ntmp + rb;
        for (int t0 = 0; t0 < n; t0 += CAP) {
          const int m = min(CAP, n - t0);
          for (int j = threadIdx.x; j < m; j += BT) sh[j] = perm[rb + t0 + j];
          __syncthreads();
          for (int i0 = threadIdx.x; i0 < n; i0 += 8 * BT) {           // eight elements per lane and walk over the tile
            int v[8], r[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { const int i = i0 + BT * k; v[k] = (i < n) ? perm[rb + i] : 0x7fffffff; r[k] = (i < n && t0) ? rk[i] : 0; }
            d_rank_count<8>(sh, m, v, r);
#pragma unroll
            for (int k = 0; k < 8; k++) { const int i = i0 + BT * k; if (i < n) rk[i] = r[k]; }
          }
          __syncthreads();
        }
        for (int i = threadIdx.x; i < n; i += BT) srt[rk[i]] = perm[rb + i];
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += BT) perm[rb + i] = srt[i];
        __syncthreads();
      } else if (threadIdx.x == 0) {                       // serial, as a last resort (no scratch: the plan's mean row is short)
        for (int i = rb + 1; i < rb + n; i++) {
          int v = perm[i], j = i - 1;
          while (j >= rb && perm[j] > v) { perm[j + 1] = perm[j]; j--; }
          perm[j + 1] = v;
        }
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < nq; i += BT) {
    const int n = staged ? sh[i] : perm[q0 + i];
    const int s = x_src[n];
    if (ORDER == 2) {
      FgCsrEntry2 E;
      E.idx_f = src_idx_f[s]; E.idx_g = s; E.area = x_area[n];
      if (DIST) {
        double di = x_c1[n] / E.area, dj = x_c2[n] / E.area;
        di -= cen[s]; dj -= cen[nsrc + s];
        E.di = di; E.dj = dj;
      } else { E.di = x_c1[n]; E.dj = x_c2[n]; }
      csr.e2[q0 + i] = E;
    } else {
      FgCsrEntry1 E;
      E.idx_f = src_idx_f[s]; E.pad = 0; E.area = x_area[n];
      csr.e1[q0 + i] = E;
    }
  }
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

// Block -> destination-row mapping.  The hardware deals blocks round-robin over the 8 XCDs (block b runs on XCD b % 8), each
// with its own 4 MiB L2.  With the identity mapping the blocks resident on one XCD hold every 8th run of 64 rows, so two
// destination rows that are neighbours in latitude -- they gather the same source records -- never share an L2.  g_xcd_band:
// XCD x sweeps ONE contiguous band of rows (tile = start_x + b / 8), so that its ~160 resident blocks cover ~7 adjacent grid
// rows and the second use of a source record is an L2 hit.  (PMC, round 2: the identity mapping reads 538 MB per launch through
// the fabric where the algorithm needs 303 MB; TCC hit rate 41 %.)  Measured (scripts/apply_ab.py, same box, 8 levels on records):
// identity 0.096 ms, banded 0.110 ms -- slower although it re-reads less: the second and third XCD that need a record find it in
// the Infinity Cache (the fabric counters count those hits, HBM does not see them), and eight bands stream eight separate windows
// of the CSR.  The identity mapping stays the default; fg_set_apply_xcd(1) selects the banded one.  Non-temporal stores of the
// output changed nothing.
__device__ __forceinline__ int d_xcd_block(int b, int nb, int band)
{
  if (!band) return b;
  if (band == 1) {
    const int x = b & 7, k = b >> 3, q = nb >> 3, r = nb & 7;
    return x * q + min(x, r) + k;                     // XCD x owns q + (x < r) consecutive tiles
  }
  // band = C >= 2: chunks of C consecutive tiles go to one XCD, chunks round-robin over the XCDs: neighbouring rows share an L2
  // (a chunk spans several grid rows) while all eight XCDs still sweep the same window of the arrays at any time
  const int C = band, per = 8 * C, full = nb / per * per;
  if (b >= full) return b;
  const int x = b & 7, k = b >> 3;
  return ((k / C) * 8 + x) * C + (k % C);
}

// Single level, level-major fields (also the has_missing path): one thread per destination cell.
template <int ORDER, bool MISSING>
__global__ __launch_bounds__(256) void k_apply1(int ndst, FgCsr csr, const double *f, const double *px, const double *py,
                                                 const int *gmask, double missing, double *out, double *row_sum)
{
  int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= ndst) return;
  int b = csr.row_ptr[d], e = csr.row_ptr[d + 1];
  double acc = 0.0, asum = 0.0;
  int touched = 0;
  for (int q = b; q < e; q++) {
    double a, v;
    if (ORDER == 2) {
      const FgCsrEntry2 E = csr.e2[q];
      a = E.area; v = f[E.idx_f];
      if (MISSING) { if (v == missing) continue; }
      bool flatgrad = false;
      if (MISSING) flatgrad = gmask[E.idx_g] != 0;
      if (!flatgrad) v = (v + px[E.idx_g] * E.di + py[E.idx_g] * E.dj);
    } else {
      const FgCsrEntry1 E = csr.e1[q];
      a = E.area; v = f[E.idx_f];
      if (MISSING) { if (v == missing) continue; }
    }
    acc += v * a;
    asum += a;
    touched = 1;
  }
  if (row_sum) row_sum[d] = (asum > 0) ? acc : 0.0;           // conserve_interp.c:815-819
  double r;                                                   // :831-839
  if (asum > 0) r = acc / asum;
  else if (touched) r = 0.0;
  else r = missing;
  out[d] = r;
}

// The single-level sweep as fregrid's level loop calls it (fregrid.c:1045-1061: get_input_data, do_scalar_conserve_interp(..., 1),
// write_field_data per level; also every field with missing values), entry-parallel like k_apply_ep8: a tile of ROWS rows walks
// its run of CSR records in chunks of CAP: the chunk is staged with coalesced loads, a lane per exchange cell issues the chunk's
// gathers (field, two gradients, the gradient mask) in one round, the products area * (f + gx di + gy dj) go to LDS and the lane
// of each row adds its part of the chunk in CSR order, carrying its sums from chunk to chunk -- so a row may be as long as it
// likes (fine -> coarse remaps: ROWS shrinks with the mean row length, a row of thousands of exchange cells is one tile walking
// many chunks).  An exchange cell whose source value is missing adds +0.0 to both sums (x + 0.0 == x bit for bit) and does not
// count as touched: k_apply1's `continue`.  k_apply1 (a lane walking its row: a record load and a gather round per exchange cell)
// took 0.137 ms for one level of C384 -> 0.25 deg second order -- more than two levels together -- and 0.142 ms for C384 -> 2 deg.
template <int ORDER, bool MISSING, int TPB, int CAP, int ROWS>
__global__ __launch_bounds__(TPB) void k_apply_ep1(int ndst, FgCsr csr, const double *f, const double *px, const double *py, const int *gmask,
                                                    double missing, double *out, double *row_sum, int xcd_band)
{
  typedef typename std::conditional<ORDER == 2, FgCsrEntry2, FgCsrEntry1>::type Entry;
  typedef unsigned int u4v __attribute__((ext_vector_type(4)));
  constexpr int W = sizeof(Entry) / 16, PASS = CAP / TPB;
  static_assert(ROWS <= TPB && CAP % TPB == 0, "tile shape");
  __shared__ __attribute__((aligned(16))) Entry sh_e[CAP];
  double *sh_p = reinterpret_cast<double *>(sh_e), *sh_a = sh_p + CAP;       // the products take the records' place (2 * 8 <= sizeof(Entry))
  __shared__ unsigned char sh_fl[CAP];
  const int t = threadIdx.x;
  const int d0 = d_xcd_block(blockIdx.x, gridDim.x, xcd_band) * ROWS;
  const int dl = min(d0 + ROWS, ndst);
  const int q0 = csr.row_ptr[d0], q1 = csr.row_ptr[dl];
  const int d = d0 + t, dc = min(d, ndst - 1);
  int b = 0, e = 0;
  if (t < ROWS) { b = csr.row_ptr[dc]; e = csr.row_ptr[dc + 1]; }
  const Entry *src = (ORDER == 2) ? (const Entry *)csr.e2 : (const Entry *)csr.e1;
  double acc = 0.0, asum = 0.0;
  int touched = 0;
  for (int c0 = q0; c0 < q1; c0 += CAP) {                  // (block-uniform)
    const int n = min(CAP, q1 - c0);
    if (c0 > q0) __syncthreads();                          // the previous chunk's products have been added
    {
      // every 16-byte word of the chunk in flight at once (a loop would wait for each load before it issues the next), no
      // branches around the loads
      const u4v *g = reinterpret_cast<const u4v *>(src + c0);
      u4v *l = reinterpret_cast<u4v *>(sh_e);
      constexpr int WPL = CAP * W / TPB;
      const int nw = n * W;
      u4v w[WPL];
#pragma unroll
      for (int j = 0; j < WPL; j++) w[j] = __builtin_nontemporal_load(g + min(t + TPB * j, nw - 1));
#pragma unroll
      for (int j = 0; j < WPL; j++) if (t + TPB * j < nw) l[t + TPB * j] = w[j];
    }
    __syncthreads();
    Entry E[PASS];
    double v[PASS], gxv[PASS], gyv[PASS];
    int gm[PASS];
#pragma unroll
    for (int j = 0; j < PASS; j++) {
      const int i = min(t + TPB * j, n - 1);
      E[j] = sh_e[i]; v[j] = 0.0; gxv[j] = 0.0; gyv[j] = 0.0; gm[j] = 0;
      if (TPB * j < n) {                                    // (block-uniform)
        v[j] = f[E[j].idx_f];
        if constexpr (ORDER == 2) {
          gxv[j] = px[E[j].idx_g]; gyv[j] = py[E[j].idx_g];
          if (MISSING) gm[j] = gmask[E[j].idx_g];
        }
      }
    }
    __syncthreads();                                       // the records are in registers: the buffer becomes the product table
#pragma unroll
    for (int j = 0; j < PASS; j++) {
      const int i = t + TPB * j;
      if (i < n) {
        double val = v[j], a = E[j].area, p;
        unsigned char fl = 1;
        if (MISSING && val == missing) { p = 0.0; a = 0.0; fl = 0; }
        else {
          if constexpr (ORDER == 2) { if (!(MISSING && gm[j] != 0)) val = (val + gxv[j] * E[j].di + gyv[j] * E[j].dj); }
          p = val * a;
        }
        sh_p[i] = p; sh_a[i] = a; sh_fl[i] = fl;
      }
    }
    __syncthreads();
    if (t < ROWS) {
      const int qa = max(b, c0) - c0, qb = min(e, c0 + n) - c0;
      int q = qa;
      for (; q + 8 <= qb; q += 8) {                        // (long rows: eight LDS reads in flight, then the adds in CSR order)
        double pp[8], aa[8];
        unsigned ff = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { pp[k] = sh_p[q + k]; aa[k] = sh_a[q + k]; ff |= sh_fl[q + k]; }
#pragma unroll
        for (int k = 0; k < 8; k++) { acc += pp[k]; asum += aa[k]; }
        touched |= (int)ff;
      }
      for (; q < qb; q++) { acc += sh_p[q]; asum += sh_a[q]; touched |= sh_fl[q]; }
    }
  }
  if (t >= ROWS || d >= ndst) return;
  if (row_sum) row_sum[d] = (asum > 0) ? acc : 0.0;           // conserve_interp.c:815-819
  double r;                                                   // :831-839
  if (asum > 0) r = acc / asum;
  else if (touched) r = 0.0;
  else r = missing;
  out[d] = r;
}

// NB levels at once, fields interleaved [cell][NB]: every CSR entry is read once for NB levels and
// each gather is NB*8 contiguous bytes (a full 64-byte sector for NB = 8).  A row is served by NB/V
// adjacent lanes, each owning V consecutive levels (V = 2: one 16-byte load per field and entry):
// the lanes of a row read one CSR record (same address, one request) and then one contiguous
// NB*8-byte segment per field, so a wave instruction touches 64*V/NB full segments instead of 64
// scattered lines.  Every (row, level) sum adds in ascending exchange-cell order, exactly the
// reference's order.  No missing values (has_missing requires nz == 1, conserve_interp.c:544).
// Every option of do_scalar_conserve_interp on one level; the per-entry operation order is the reference's
// (weight, then the missing test, then sum / cell_measures scaling) so sums are bit-identical.
// MONO: values come limited from xdata[] (the monotone branch never sets out_miss, :720-739).
template <int ORDER, bool MONO>
__global__ __launch_bounds__(256) void k_apply_ex(int ndst, FgCsr csr, const double *f, const double *px, const double *py,
                                                   FgApplyEx o, double *out, double *row_sum, int *err)
{
  int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= ndst) return;
  int b = csr.row_ptr[d], e = csr.row_ptr[d + 1];
  double acc = 0.0, asum = 0.0, asum_t = 0.0;
  int touched = 0;
  for (int q = b; q < e; q++) {
    double a, v, di = 0, dj = 0;
    int s, jf;
    if (ORDER == 2) { const FgCsrEntry2 E = csr.e2[q]; a = E.area; jf = E.idx_f; s = E.idx_g; di = E.di; dj = E.dj; }
    else            { const FgCsrEntry1 E = csr.e1[q]; a = E.area; jf = E.idx_f; s = E.idx_f; }
    if (o.cell_area_out) {                                       // :845-860 (plain exchange-cell area, no weight)
      if (o.field_area) asum_t += (a * o.field_area[s] / o.cell_area[s]);
      else asum_t += a;
    }
    if (MONO) {
      v = o.xdata[q];
      if (v == o.missing) continue;
      if (o.weight) a *= o.weight[s];
    } else {
      if (o.weight) a *= o.weight[s];
      v = f[jf];
      if (o.has_missing && v == o.missing) continue;
    }
    if (o.sum) a /= o.cell_area[s];
    else if (o.field_area) {
      const double fa = o.field_area[s];
      if (!MONO && o.has_missing && fa == o.area_missing) { atomicOr(err, FG_XERR_AREA_MISSING); continue; }
      a *= (fa / o.cell_area[s]);
    }
    if (ORDER == 2 && !MONO) {
      bool flat = o.has_missing && o.gmask && o.gmask[s] != 0;
      if (!flat) v = (v + px[s] * di + py[s] * dj);
    }
    acc += v * a;
    asum += a;
    if (!MONO) touched = 1;
  }
  if (row_sum) row_sum[d] = (asum > 0) ? acc : 0.0;             // :815-819
  double r = acc;
  if (o.sum) {                                                  // :821-830
    if (asum == 0) r = touched ? 0.0 : o.missing;
  } else {
    if (asum > 0) r = acc / asum;                               // :832-839
    else if (touched) r = 0.0;
    else r = o.missing;
    if (o.cell_area_out && r != o.missing) r *= (asum_t / o.cell_area_out[d]);
  }
  out[d] = r;
}

// k_apply_ex entry-parallel, the way k_apply_ep1 is k_apply1: per exchange cell the three numbers the row's sums take -- its term of
// the --target_grid area sum (taken before the missing test, as above), the product value * area' and area' (weight, cell_methods /
// cell_measures scaling applied in the reference's order) -- go through LDS, a cell the loop above leaves with `continue` adds +0.0
// to the last two and does not count as touched.  Rows of any length (chunks of CAP records, ROWS rows per tile).
template <int ORDER, bool MONO, int TPB, int CAP, int ROWS>
__global__ __launch_bounds__(TPB) void k_apply_epx(int ndst, FgCsr csr, const double *f, const double *px, const double *py,
                                                    FgApplyEx o, double *out, double *row_sum, int *err, int xcd_band)
{
  typedef typename std::conditional<ORDER == 2, FgCsrEntry2, FgCsrEntry1>::type Entry;
  typedef unsigned int u4v __attribute__((ext_vector_type(4)));
  constexpr int W = sizeof(Entry) / 16, PASS = CAP / TPB;
  static_assert(ROWS <= TPB && CAP % TPB == 0, "tile shape");
  __shared__ __attribute__((aligned(16))) double sh_raw[CAP * 4];     // CSR records (<= 32 B each), then [3][CAP] products
  Entry *sh_e = reinterpret_cast<Entry *>(sh_raw);
  double *sh_t = sh_raw, *sh_p = sh_raw + CAP, *sh_a = sh_raw + 2 * CAP;
  __shared__ unsigned char sh_fl[CAP];
  const int t = threadIdx.x;
  const int d0 = d_xcd_block(blockIdx.x, gridDim.x, xcd_band) * ROWS;
  const int dl = min(d0 + ROWS, ndst);
  const int q0 = csr.row_ptr[d0], q1 = csr.row_ptr[dl];
  const int d = d0 + t, dc = min(d, ndst - 1);
  int b = 0, e = 0;
  if (t < ROWS) { b = csr.row_ptr[dc]; e = csr.row_ptr[dc + 1]; }
  const Entry *src = (ORDER == 2) ? (const Entry *)csr.e2 : (const Entry *)csr.e1;
  double acc = 0.0, asum = 0.0, asum_t = 0.0;
  int touched = 0;
  for (int c0 = q0; c0 < q1; c0 += CAP) {                  // (block-uniform)
    const int n = min(CAP, q1 - c0);
    if (c0 > q0) __syncthreads();
    {
      const u4v *g = reinterpret_cast<const u4v *>(src + c0);
      u4v *l = reinterpret_cast<u4v *>(sh_e);
      constexpr int WPL = CAP * W / TPB;
      const int nw = n * W;
      u4v w[WPL];
#pragma unroll
      for (int j = 0; j < WPL; j++) w[j] = __builtin_nontemporal_load(g + min(t + TPB * j, nw - 1));
#pragma unroll
      for (int j = 0; j < WPL; j++) if (t + TPB * j < nw) l[t + TPB * j] = w[j];
    }
    __syncthreads();
    double tq[PASS], pp[PASS], aa[PASS];
    unsigned char fl[PASS];
#pragma unroll
    for (int j = 0; j < PASS; j++) {
      const int i = min(t + TPB * j, n - 1);
      double a, v, di = 0, dj = 0;
      int s, jf;
      if constexpr (ORDER == 2) { const FgCsrEntry2 E = ((const FgCsrEntry2 *)sh_e)[i]; a = E.area; jf = E.idx_f; s = E.idx_g; di = E.di; dj = E.dj; }
      else                      { const FgCsrEntry1 E = ((const FgCsrEntry1 *)sh_e)[i]; a = E.area; jf = E.idx_f; s = E.idx_f; }
      // every value the cell may need, in one round
      const double wgt = o.weight ? o.weight[s] : 1.0;
      const double ca = (o.sum || o.field_area) ? o.cell_area[s] : 1.0;
      const double fa = o.field_area ? o.field_area[s] : 0.0;
      const double gxv = (ORDER == 2 && !MONO) ? px[s] : 0.0, gyv = (ORDER == 2 && !MONO) ? py[s] : 0.0;
      const int gm = (ORDER == 2 && !MONO && o.has_missing && o.gmask) ? o.gmask[s] : 0;
      v = MONO ? o.xdata[c0 + i] : f[jf];
      double tt = 0.0;
      if (o.cell_area_out) tt = o.field_area ? (a * fa / ca) : a;            // :845-860 (plain exchange-cell area, no weight)
      bool skip = false;
      if (MONO) { if (v == o.missing) skip = true; else if (o.weight) a *= wgt; }
      else { if (o.weight) a *= wgt; if (o.has_missing && v == o.missing) skip = true; }
      if (!skip) {
        if (o.sum) a /= ca;
        else if (o.field_area) {
          if (!MONO && o.has_missing && fa == o.area_missing) { if (t + TPB * j < n) atomicOr(err, FG_XERR_AREA_MISSING); skip = true; }
          else a *= (fa / ca);
        }
      }
      if (!skip && ORDER == 2 && !MONO) { if (gm == 0) v = (v + gxv * di + gyv * dj); }
      tq[j] = tt; pp[j] = skip ? 0.0 : v * a; aa[j] = skip ? 0.0 : a; fl[j] = (skip || MONO) ? 0 : 1;
    }
    __syncthreads();                                       // the records have been read: the buffer becomes the product table
#pragma unroll
    for (int j = 0; j < PASS; j++) {
      const int i = t + TPB * j;
      if (i < n) { sh_t[i] = tq[j]; sh_p[i] = pp[j]; sh_a[i] = aa[j]; sh_fl[i] = fl[j]; }
    }
    __syncthreads();
    if (t < ROWS) {
      const int qa = max(b, c0) - c0, qb = min(e, c0 + n) - c0;
      for (int q = qa; q < qb; q++) { asum_t += sh_t[q]; acc += sh_p[q]; asum += sh_a[q]; touched |= sh_fl[q]; }
    }
  }
  if (t >= ROWS || d >= ndst) return;
  if (row_sum) row_sum[d] = (asum > 0) ? acc : 0.0;             // :815-819
  double r = acc;
  if (o.sum) {                                                  // :821-830
    if (asum == 0) r = touched ? 0.0 : o.missing;
  } else {
    if (asum > 0) r = acc / asum;                               // :832-839
    else if (touched) r = 0.0;
    else r = o.missing;
    if (o.cell_area_out && r != o.missing) r *= (asum_t / o.cell_area_out[d]);
  }
  out[d] = r;
}

// :622-645: bounds of the 3x3 halo'd neighbourhood, ignoring missing values
__global__ __launch_bounds__(256) void k_mono_bounds(const FgTile *tiles, int ntiles, int nsrc, const int *src_idx_f, const double *f,
                                                      double missing, double *fbmax, double *fbmin, double *fmax, double *fmin)
{
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsrc) return;
  int t = 0;
  while (t + 1 < ntiles && s >= tiles[t + 1].cell_off) t++;
  const int ld = tiles[t].nx + 2, c = src_idx_f[s];
  double mx = -1.e20, mn = 1.e20;
  for (int jj = -1; jj <= 1; jj++)
    for (int ii = -1; ii <= 1; ii++) {
      double v = f[c + jj * ld + ii];
      if (v != missing) { if (v > mx) mx = v; if (v < mn) mn = v; }
    }
  fbmax[s] = mx; fbmin[s] = mn; fmax[s] = -1.e20; fmin[s] = 1.e20;
}

// :647-669: second-order value of every exchange cell (CSR order) and its extremes per source cell
__global__ __launch_bounds__(256) void k_mono_xdata(long nx, FgCsr csr, const double *f, const double *px, const double *py,
                                                     const int *gmask, double missing, double *xdata, double *fmax, double *fmin)
{
  long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nx) return;
  const FgCsrEntry2 E = csr.e2[q];
  double v = f[E.idx_f], x;
  if (v != missing) {
    if (gmask && gmask[E.idx_g]) x = v;
    else x = v + px[E.idx_g] * E.di + py[E.idx_g] * E.dj;
    // (the hardware's own FP64 max / min, no value returned: a compare-and-swap loop per value -- a read, then a CAS round, twice per
    // exchange cell -- made this kernel 0.415 of the 0.485 ms of a monotone sweep)
    (void)atomicMax(fmax + E.idx_g, x);
    (void)atomicMin(fmin + E.idx_g, x);
  } else
    x = missing;
  xdata[q] = x;
}

// :679-716
__global__ __launch_bounds__(256) void k_mono_limit(long nx, FgCsr csr, const double *f, double missing, const double *fbmax,
                                                     const double *fbmin, const double *fmax, const double *fmin, double *xdata, int *err)
{
  long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nx) return;
  double x = xdata[q];
  if (x == missing) return;
  const FgCsrEntry2 E = csr.e2[q];
  const int s = E.idx_g;
  const double f_bar = f[E.idx_f];
  if (fmax[s] > fbmax[s]) {
    x = f_bar + ((x - f_bar) / (fmax[s] - f_bar)) * (fbmax[s] - f_bar);
    if (x > fbmax[s]) {
      if (x - fbmax[s] < 1.e-10) x = fbmax[s];
      if (x > fbmax[s]) atomicOr(err, FG_XERR_ABOVE);
    }
  } else if (fmin[s] < fbmin[s]) {
    x = f_bar + ((x - f_bar) / (fmin[s] - f_bar)) * (fbmin[s] - f_bar);
    if (x < fbmin[s]) {
      if (fbmin[s] - x < 1.e-10) x = fbmin[s];
      if (x < fbmin[s]) atomicOr(err, FG_XERR_BELOW);
    }
  }
  xdata[q] = x;
}

template <int V> struct VecD;
template <> struct VecD<1> { double v[1]; };
template <> struct __attribute__((aligned(16))) VecD<2> { double v[2]; };
template <> struct __attribute__((aligned(16))) VecD<4> { double v[4]; };
template <> struct __attribute__((aligned(16))) VecD<8> { double v[8]; };


// MERGED (order 2): f points at records [source cell][3][NB] = {field, grad_x, grad_y} of the interior cell idx_g
// (k_merge3): one contiguous 3*NB*8-byte segment per entry instead of three NB*8-byte ones in three arrays -- 13 % (NB = 8)
// to 23 % (NB = 4) faster on MI355X; px, py unused.
template <int ORDER, int NB, int V, bool MERGED = false, int UNR = 1>
__global__ __launch_bounds__(256) void k_apply_il(int ndst, FgCsr csr, const double *f, const double *px, const double *py,
                                                   double missing, double *out, double *row_sum, long out_ld, int nb_valid, int xcd_band)
{
  constexpr int LPR = NB / V;                        // lanes per row
  constexpr int ROWS = 256 / LPR;                    // rows per block
  constexpr int APPLY_STAGE = ROWS * 6;              // CSR records staged in LDS (mean row has ~4)
  typedef typename std::conditional<ORDER == 2, FgCsrEntry2, FgCsrEntry1>::type Entry;
  __shared__ Entry sh_e[APPLY_STAGE];
  const int lev = (threadIdx.x % LPR) * V;
  const int d0 = d_xcd_block(blockIdx.x, gridDim.x, xcd_band) * ROWS;
  const int d = d0 + threadIdx.x / LPR;
  // the block's rows own one contiguous run of CSR records: stage it with coalesced 16-byte loads
  const int dl = min(d0 + ROWS, ndst);
  const int q0 = csr.row_ptr[d0], q1 = csr.row_ptr[dl];
  const int dc = min(d, ndst - 1);
  const int b = csr.row_ptr[dc], e = csr.row_ptr[dc + 1];   // (issued with q0, q1: one memory round trip less per wave)
  {
    const Entry *src = (ORDER == 2) ? (const Entry *)csr.e2 : (const Entry *)csr.e1;
    const int nstage = min(q1 - q0, APPLY_STAGE);
    constexpr int W = sizeof(Entry) / 16;            // 16-byte words per record
    const uint4 *g = reinterpret_cast<const uint4 *>(src + q0);
    uint4 *l = reinterpret_cast<uint4 *>(sh_e);
    // the CSR records are used once per launch: non-temporal loads keep them from evicting the gathered source records
    // (measured on MI355X, 8 levels: 0.1071 -> 0.0963 ms on records, 0.1162 -> 0.1049 ms on interleaved arrays)
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    for (int i = threadIdx.x; i < nstage * W; i += 256) ((u4v *)l)[i] = __builtin_nontemporal_load((const u4v *)g + i);
  }
  __syncthreads();
  if (d >= ndst) return;
  double acc[V], asum = 0.0;
#pragma unroll
  for (int k = 0; k < V; k++) acc[k] = 0.0;
  if (MERGED && UNR > 1) {
    // UNR entries of the row in flight per lane: all their gathers are issued before the first is consumed (a row has ~4
    // entries; the sums keep the CSR order).  Entries past the row's end repeat its last one (same lines, no new traffic).
    for (int q = b; q < e; q += UNR) {
      FgCsrEntry2 E[UNR];
      VecD<V> fv[UNR], gxv[UNR], gyv[UNR];
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        const int qq = min(q + u, e - 1), ql = qq - q0;
        E[u] = (ql < APPLY_STAGE) ? ((const FgCsrEntry2 *)sh_e)[ql] : csr.e2[qq];
      }
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        const double *pf = f + (size_t)E[u].idx_g * (3 * NB) + lev;
        fv[u] = *reinterpret_cast<const VecD<V> *>(pf);
        gxv[u] = *reinterpret_cast<const VecD<V> *>(pf + NB);
        gyv[u] = *reinterpret_cast<const VecD<V> *>(pf + 2 * NB);
      }
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        if (q + u < e) {
#pragma unroll
          for (int k = 0; k < V; k++) {
            double v = (fv[u].v[k] + gxv[u].v[k] * E[u].di + gyv[u].v[k] * E[u].dj);
            acc[k] += v * E[u].area;
          }
          asum += E[u].area;
        }
      }
    }
  } else
  for (int q = b; q < e; q++) {
    const int ql = q - q0;
    if (ORDER == 2) {
      const FgCsrEntry2 E = (ql < APPLY_STAGE) ? ((const FgCsrEntry2 *)sh_e)[ql] : csr.e2[q];
      const double *pf = MERGED ? f + (size_t)E.idx_g * (3 * NB) + lev : f + (size_t)E.idx_f * NB + lev;
      const double *pgx = MERGED ? pf + NB : px + (size_t)E.idx_g * NB + lev;
      const double *pgy = MERGED ? pf + 2 * NB : py + (size_t)E.idx_g * NB + lev;
      const VecD<V> fv = *reinterpret_cast<const VecD<V> *>(pf);
      const VecD<V> gxv = *reinterpret_cast<const VecD<V> *>(pgx);
      const VecD<V> gyv = *reinterpret_cast<const VecD<V> *>(pgy);
#pragma unroll
      for (int k = 0; k < V; k++) {
        double v = (fv.v[k] + gxv.v[k] * E.di + gyv.v[k] * E.dj);
        acc[k] += v * E.area;
      }
      asum += E.area;
    } else {
      const FgCsrEntry1 E = (ql < APPLY_STAGE) ? ((const FgCsrEntry1 *)sh_e)[ql] : csr.e1[q];
      const VecD<V> fv = *reinterpret_cast<const VecD<V> *>(f + (size_t)E.idx_f * NB + lev);
#pragma unroll
      for (int k = 0; k < V; k++) acc[k] += fv.v[k] * E.area;
      asum += E.area;
    }
  }
  VecD<V> r, rs;
#pragma unroll
  for (int k = 0; k < V; k++) {
    rs.v[k] = (asum > 0) ? acc[k] : 0.0;
    if (asum > 0) r.v[k] = acc[k] / asum;
    else if (e > b) r.v[k] = 0.0;
    else r.v[k] = missing;
  }
  if (row_sum) *reinterpret_cast<VecD<V> *>(row_sum + (size_t)d * NB + lev) = rs;
  if (out_ld > 0) {
    // level-major output out[level][d] (fg_plan_apply): lanes of one level group hold consecutive rows, so each store
    // instruction writes 16 consecutive doubles per level -- the separate de-interleave pass is not needed
#pragma unroll
    for (int k = 0; k < V; k++) if (lev + k < nb_valid) out[(size_t)(lev + k) * out_ld + d] = r.v[k];
  } else
    *reinterpret_cast<VecD<V> *>(out + (size_t)d * NB + lev) = r;
}

// Entry-parallel order-2 sweep on merged records, 8 levels: the gathers of ALL of a tile's exchange cells are issued in one go
// (a lane pair per exchange cell, up to three per lane pair), the products area * (f + gx di + gy dj) go to LDS, and a lane group
// per destination row adds them in CSR order -- the same operations in the same order as k_apply_il, so the same bits, but a
// wave's dependent chain is row pointers -> CSR staging -> ONE round of gathers -> LDS sums instead of ~7 gather rounds.
// A tile whose rows hold more than EP_CAP exchange cells (fine -> coarse remaps) takes the row-serial loop.
// (A persistent version -- blocks walking over tiles, the next tile's row pointers and records prefetched into registers while
//  the current tile's gathers fly, no LDS staging -- was measured: 0.119 ms against 0.0833.  Its 108 VGPRs halve the occupancy
//  (4 waves per SIMD against 7), and the occupancy is what hides the gather latency.)
#ifndef EP_ROWS
#define EP_ROWS 32
#endif
template <int TPB, int CAP>
__global__ __launch_bounds__(TPB) void k_apply_ep8(int ndst, FgCsr csr, const double *rec, double missing, double *out, double *row_sum,
                                                    long out_ld, int nb_valid, int xcd_band)
{
  constexpr int NB = 8;
  constexpr int LPR = NB, LV = 1;                          // sum phase: a lane per (row, level); EP_ROWS * NB <= TPB lanes take part
  static_assert(EP_ROWS * NB <= TPB, "rows per tile");
  constexpr int EPP = TPB / 2, PASS = CAP / EPP;          // gather phase: a lane pair per exchange cell, EPP cells per pass
  // one buffer: the staged CSR records first, then (once every lane holds its records in registers) the products and areas
  __shared__ __attribute__((aligned(16))) double sh_raw[CAP * (NB + 1)];
  FgCsrEntry2 *sh_e = reinterpret_cast<FgCsrEntry2 *>(sh_raw);
  double *sh_p = sh_raw, *sh_a = sh_raw + CAP * NB;
  const int t = threadIdx.x;
  const int d0 = d_xcd_block(blockIdx.x, gridDim.x, xcd_band) * EP_ROWS;
  const int dl = min(d0 + EP_ROWS, ndst);
  const int q0 = csr.row_ptr[d0], q1 = csr.row_ptr[dl];
  const int d = d0 + t / LPR, lev = (t % LPR) * LV;
  const int dc = min(d, ndst - 1);
  const int b = csr.row_ptr[dc], e = csr.row_ptr[dc + 1];
  const int n = q1 - q0, nst = min(n, CAP);
  {
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    const u4v *g = reinterpret_cast<const u4v *>(csr.e2 + q0);
    u4v *l = reinterpret_cast<u4v *>(sh_e);
    for (int i = t; i < nst * 2; i += TPB) l[i] = __builtin_nontemporal_load(g + i);
  }
  __syncthreads();
  double acc[LV], asum = 0.0;
#pragma unroll
  for (int k = 0; k < LV; k++) acc[k] = 0.0;
  if (n <= CAP) {
    const int h = (t & 1) * 4;                             // four levels per lane of the pair
    VecD<4> fv[PASS], gxv[PASS], gyv[PASS];
    FgCsrEntry2 E[PASS];
#pragma unroll
    for (int j = 0; j < PASS; j++) {
      if (EPP * j < n) {                                   // (block-uniform)
        const int i = min(t / 2 + EPP * j, n - 1);
        E[j] = sh_e[i];
        const double *pf = rec + (size_t)E[j].idx_g * (3 * NB) + h;
        fv[j] = *reinterpret_cast<const VecD<4> *>(pf);
        gxv[j] = *reinterpret_cast<const VecD<4> *>(pf + NB);
        gyv[j] = *reinterpret_cast<const VecD<4> *>(pf + 2 * NB);
      }
    }
    __syncthreads();                                       // the records are in registers: the buffer becomes the product table
#pragma unroll
    for (int j = 0; j < PASS; j++) {
      const int i = t / 2 + EPP * j;
      if (EPP * j < n && i < n) {
        VecD<4> pv;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          double v = (fv[j].v[k] + gxv[j].v[k] * E[j].di + gyv[j].v[k] * E[j].dj);
          pv.v[k] = v * E[j].area;
        }
        *reinterpret_cast<VecD<4> *>(sh_p + i * NB + h) = pv;
        if (h == 0) sh_a[i] = E[j].area;
      }
    }
    __syncthreads();
    if (d >= ndst || t >= EP_ROWS * NB) return;
    for (int q = b - q0; q < e - q0; q++) {
      const VecD<LV> pv = *reinterpret_cast<const VecD<LV> *>(sh_p + q * NB + lev);
#pragma unroll
      for (int k = 0; k < LV; k++) acc[k] += pv.v[k];
      asum += sh_a[q];
    }
  } else {
    if (d >= ndst || t >= EP_ROWS * NB) return;
    for (int q = b; q < e; q++) {
      const int ql = q - q0;
      const FgCsrEntry2 E = (ql < CAP) ? sh_e[ql] : csr.e2[q];
      const double *pf = rec + (size_t)E.idx_g * (3 * NB) + lev;
      const VecD<LV> fv = *reinterpret_cast<const VecD<LV> *>(pf);
      const VecD<LV> gxv = *reinterpret_cast<const VecD<LV> *>(pf + NB);
      const VecD<LV> gyv = *reinterpret_cast<const VecD<LV> *>(pf + 2 * NB);
#pragma unroll
      for (int k = 0; k < LV; k++) {
        double v = (fv.v[k] + gxv.v[k] * E.di + gyv.v[k] * E.dj);
        acc[k] += v * E.area;
      }
      asum += E.area;
    }
  }
  VecD<LV> r, rs;
#pragma unroll
  for (int k = 0; k < LV; k++) {
    rs.v[k] = (asum > 0) ? acc[k] : 0.0;
    if (asum > 0) r.v[k] = acc[k] / asum;
    else if (e > b) r.v[k] = 0.0;
    else r.v[k] = missing;
  }
  if (row_sum) *reinterpret_cast<VecD<LV> *>(row_sum + (size_t)d * NB + lev) = rs;
  if (out_ld > 0) {
#pragma unroll
    for (int k = 0; k < LV; k++) if (lev + k < nb_valid) out[(size_t)(lev + k) * out_ld + d] = r.v[k];
  } else
    *reinterpret_cast<VecD<LV> *>(out + (size_t)d * NB + lev) = r;
}

// The same for plans with longer rows (fine -> coarse remaps; k_apply_ep8 above is for rows of a few exchange cells): ROWS rows per
// tile, fewer the longer the mean row, and the tile's run of CSR records walked in chunks of CAP, the (row, level) lanes carrying
// their sums from chunk to chunk: any row length, every gather of a chunk in one round.  (Such plans took the row-serial
// k_apply_il before: C384 -> 2 deg 0.117 ms per 8 levels on records, now 0.037.)  As a loop the kernel needs more registers than
// k_apply_ep8 -- with 32 rows per tile on C384 -> 0.25 deg it runs in 0.117 ms against 0.083 -- so short rows keep the kernel above.
// (ORDER 1: `rec` is the interleaved field [cell][8], a 16-byte CSR record and one 64-byte gather per exchange cell.)
template <int ORDER, int TPB, int CAP, int ROWS>
__global__ __launch_bounds__(TPB) void k_apply_ep8g(int ndst, FgCsr csr, const double *rec, double missing, double *out, double *row_sum,
                                                    long out_ld, int nb_valid, int xcd_band)
{
  constexpr int NB = 8;
  constexpr int EPP = TPB / 2, PASS = CAP / EPP;          // gather phase: a lane pair per exchange cell, EPP cells per pass
  static_assert(ROWS * NB <= TPB, "a lane per (row, level)");
  // one buffer: the staged CSR records first, then (once every lane holds its records in registers) the products and areas
  __shared__ __attribute__((aligned(16))) double sh_raw[CAP * (NB + 1)];
  typedef typename std::conditional<ORDER == 2, FgCsrEntry2, FgCsrEntry1>::type Entry;
  constexpr int W = sizeof(Entry) / 16;
  Entry *sh_e = reinterpret_cast<Entry *>(sh_raw);
  double *sh_p = sh_raw, *sh_a = sh_raw + CAP * NB;
  const int t = threadIdx.x;
  const int d0 = d_xcd_block(blockIdx.x, gridDim.x, xcd_band) * ROWS;
  const int dl = min(d0 + ROWS, ndst);
  const int q0 = csr.row_ptr[d0], q1 = csr.row_ptr[dl];
  const bool sumlane = t < ROWS * NB;                      // lane (row, level) of the sum phase
  const int d = d0 + t / NB, lev = t % NB;
  const int dc = min(d, ndst - 1);
  int b = 0, e = 0;
  if (sumlane) { b = csr.row_ptr[dc]; e = csr.row_ptr[dc + 1]; }
  double acc = 0.0, asum = 0.0;
  for (int c0 = q0; c0 < q1; c0 += CAP) {                  // (block-uniform)
    const int n = min(CAP, q1 - c0);
    if (c0 > q0) __syncthreads();                          // the previous chunk's products have been added
    {
      typedef unsigned int u4v __attribute__((ext_vector_type(4)));
      const Entry *src = (ORDER == 2) ? (const Entry *)csr.e2 : (const Entry *)csr.e1;
      const u4v *g = reinterpret_cast<const u4v *>(src + c0);
      u4v *l = reinterpret_cast<u4v *>(sh_e);
      for (int i = t; i < n * W; i += TPB) l[i] = __builtin_nontemporal_load(g + i);
    }
    __syncthreads();
    const int h = (t & 1) * 4;                             // four levels per lane of the pair
    VecD<4> fv[PASS], gxv[PASS], gyv[PASS];
    Entry E[PASS];
#pragma unroll
    for (int j = 0; j < PASS; j++) {
      if (EPP * j < n) {                                   // (block-uniform)
        const int i = min(t / 2 + EPP * j, n - 1);
        E[j] = sh_e[i];
        if constexpr (ORDER == 2) {
          const double *pf = rec + (size_t)E[j].idx_g * (3 * NB) + h;
          fv[j] = *reinterpret_cast<const VecD<4> *>(pf);
          gxv[j] = *reinterpret_cast<const VecD<4> *>(pf + NB);
          gyv[j] = *reinterpret_cast<const VecD<4> *>(pf + 2 * NB);
        } else
          fv[j] = *reinterpret_cast<const VecD<4> *>(rec + (size_t)E[j].idx_f * NB + h);
      }
    }
    __syncthreads();                                       // the records are in registers: the buffer becomes the product table
#pragma unroll
    for (int j = 0; j < PASS; j++) {
      const int i = t / 2 + EPP * j;
      if (EPP * j < n && i < n) {
        VecD<4> pv;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          double v = fv[j].v[k];
          if constexpr (ORDER == 2) v = (v + gxv[j].v[k] * E[j].di + gyv[j].v[k] * E[j].dj);
          pv.v[k] = v * E[j].area;
        }
        *reinterpret_cast<VecD<4> *>(sh_p + i * NB + h) = pv;
        if (h == 0) sh_a[i] = E[j].area;
      }
    }
    __syncthreads();
    if (sumlane) {
      const int qa = max(b, c0) - c0, qb = min(e, c0 + n) - c0;
      int q = qa;
      for (; q + 8 <= qb; q += 8) {                        // (long rows: eight LDS reads in flight, then the adds in CSR order)
        double pp[8], aa[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { pp[k] = sh_p[(q + k) * NB + lev]; aa[k] = sh_a[q + k]; }
#pragma unroll
        for (int k = 0; k < 8; k++) { acc += pp[k]; asum += aa[k]; }
      }
      for (; q < qb; q++) { acc += sh_p[q * NB + lev]; asum += sh_a[q]; }
    }
  }
  if (!sumlane || d >= ndst) return;
  const double rs = (asum > 0) ? acc : 0.0;
  double r;
  if (asum > 0) r = acc / asum;
  else if (e > b) r = 0.0;
  else r = missing;
  if (row_sum) row_sum[(size_t)d * NB + lev] = rs;
  if (out_ld > 0) { if (lev < nb_valid) out[(size_t)lev * out_ld + d] = r; }
  else out[(size_t)d * NB + lev] = r;
}

// [nb][n] (level-major, row stride ld) <-> [n][NB] interleaved
template <int NB>
__global__ __launch_bounds__(256) void k_interleave(long n, const double *in, long ld, int nb, double *out)
{
  long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  double v[NB];
#pragma unroll
  for (int k = 0; k < NB; k++) v[k] = (k < nb) ? in[(size_t)k * ld + c] : 0.0;
#pragma unroll
  for (int k = 0; k < NB; k++) out[(size_t)c * NB + k] = v[k];
}
template <int NB>
__global__ __launch_bounds__(256) void k_deinterleave(long n, const double *in, long ld, int nb, double *out)
{
  long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  double v[NB];
#pragma unroll
  for (int k = 0; k < NB; k++) v[k] = in[(size_t)c * NB + k];
#pragma unroll
  for (int k = 0; k < NB; k++) if (k < nb) out[(size_t)k * ld + c] = v[k];
}

// Up to three level-major arrays -> interleaved in one launch (blockIdx.y picks the array), transposed through LDS so
// both the global reads ([level][cell], consecutive cells per level) and the global writes ([cell][NB], consecutive
// addresses across the block) are fully coalesced.  The per-thread version above wrote 64-byte strided chunks and reached
// only ~0.5 TB/s.
struct FgIl3 { const double *in[3]; double *out[3]; long n[3]; long ld[3]; };
template <int NB>
__global__ __launch_bounds__(256) void k_interleave3(FgIl3 a, int nb)
{
  __shared__ double tile[256 * (NB + 1)];
  const int w = blockIdx.y;
  const long n = a.n[w];
  const long c0 = (long)blockIdx.x * 256;
  if (c0 >= n) return;
  const double *in = a.in[w];
  double *out = a.out[w];
  const long ld = a.ld[w];
  const long c = c0 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < NB; k++) tile[threadIdx.x * (NB + 1) + k] = (k < nb && c < n) ? in[(size_t)k * ld + c] : 0.0;
  __syncthreads();
  const long cnt = ((n - c0) < 256 ? (n - c0) : 256) * NB;       // doubles this block writes
#pragma unroll
  for (int i = 0; i < NB; i++) {
    const long e = (long)i * 256 + threadIdx.x;                  // consecutive threads -> consecutive addresses
    if (e < cnt) out[(size_t)c0 * NB + e] = tile[(e / NB) * (NB + 1) + (e % NB)];
  }
}

// Level-major field (with halo, gathered through src_idx_f) and gradients of one chunk of levels -> records
// out[source cell][3][NB] for the MERGED sweep; zero padded beyond nb levels.  CB cells per block through LDS so that both the
// loads (consecutive cells of one level) and the stores (consecutive doubles of the records) are coalesced.
template <int NB, int CB>
__global__ __launch_bounds__(256) void k_merge3(long n, const int *src_idx_f, const double *f, long ld_f, const double *gx,
                                                const double *gy, long ld_g, int nb, double *out)
{
  constexpr int R = 3 * NB;                                      // doubles per record
  __shared__ double tile[CB * (R + 1)];
  const long c0 = (long)blockIdx.x * CB;
  const int t = threadIdx.x % CB;                                // 256 is a multiple of CB: a thread keeps its cell
  const long c = c0 + t;
  const long cf = (c < n) ? (long)src_idx_f[c] : 0;
  for (int e = threadIdx.x; e < CB * R; e += 256) {
    const int wk = e / CB, w = wk / NB, k = wk % NB;             // array, level; cell fastest
    double v = 0.0;
    if (k < nb && c < n) v = (w == 0) ? f[(size_t)k * ld_f + cf] : ((w == 1) ? gx[(size_t)k * ld_g + c] : gy[(size_t)k * ld_g + c]);
    tile[t * (R + 1) + wk] = v;
  }
  __syncthreads();
  const long cnt = ((n - c0) < CB ? (n - c0) : CB) * R;          // doubles this block writes
  for (long e = threadIdx.x; e < cnt; e += 256) out[(size_t)c0 * R + e] = tile[(e / R) * (R + 1) + (e % R)];
}

// flattened source / destination cell numbers -> (tile, i, j) and (i, j) for fg_plan_get_xgrid
__global__ __launch_bounds__(256) void k_xgrid_indices(long nx, const int *x_src, const int *x_dst, const FgTile *tiles, int ntiles,
                                                        int nx_out, int *t_in, int *i_in, int *j_in, int *i_out, int *j_out)
{
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nx) return;
  const int s = x_src[k], d = x_dst[k];
  int m = 0;
  while (m + 1 < ntiles && s >= tiles[m + 1].cell_off) m++;
  const int loc = s - tiles[m].cell_off, nxt = tiles[m].nx;
  t_in[k] = m; i_in[k] = loc % nxt; j_in[k] = loc / nxt;
  i_out[k] = d % nx_out; j_out[k] = d / nx_out;
}
void fgd_xgrid_indices(long nx, const int *x_src, const int *x_dst, const FgTile *tiles_dev, int ntiles, int nx_out,
                       int *t_in, int *i_in, int *j_in, int *i_out, int *j_out, hipStream_t st)
{
  if (nx > 0) k_xgrid_indices<<<nblk(nx, 256), 256, 0, st>>>(nx, x_src, x_dst, tiles_dev, ntiles, nx_out, t_in, i_in, j_in, i_out, j_out);
}

// interp.c:262-305 (conserve_interp): weights are xarea / (sum of xarea in the destination cell)
__global__ __launch_bounds__(256) void k_apply_frac(int ndst, FgCsr csr, const double *data, double *out)
{
  int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= ndst) return;
  int b = csr.row_ptr[d], e = csr.row_ptr[d + 1];
  double asum = 0.0, acc = 0.0;
  for (int q = b; q < e; q++) asum += csr.e1[q].area;
  for (int q = b; q < e; q++) {
    double frac = csr.e1[q].area / asum;
    acc += data[csr.e1[q].idx_f] * frac;
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
void fgd_csr_fill_pos(long nx_cap, const unsigned long long *nx_dev, const int *x_dst, const int *row_ptr, const int *x_rowpos, int *perm,
                      hipStream_t st)
{
  if (nx_cap > 0) k_csr_fill_pos<<<nblk(nx_cap, 256), 256, 0, st>>>(nx_cap, nx_dev, x_dst, row_ptr, x_rowpos, perm);
}
void fgd_csr_sortgather(int order, int ndst, long nx, const int *perm, const int *x_src, const double *x_area, const double *x_c1,
                        const double *x_c2, const int *src_idx_f, const double *cen, int nsrc, FgCsr csr, hipStream_t st, int *tmp, long ntmp,
                        int long_rows)
{
  if (ndst <= 0) return;
  int *pm = const_cast<int *>(perm);                       // sorted in place only for runs beyond the LDS staging capacity
  // rows per block: 64 short rows for one wave; 16 long ones, or a single very long one (fine -> very coarse), for four waves
  // (long_rows: the caller knows of a region of long rows under a short mean -- the cells round the pole of a curvilinear target)
  const int mode = nx > 256 * (long)ndst ? 2 : ((nx > 8 * (long)ndst || long_rows) ? 1 : 0);
#define SG(O_, R_, D_) k_csr_sortgather<O_, R_, D_, (R_ >= 64 ? 64 : 256)><<<nblk(ndst, R_), (R_ >= 64 ? 64 : 256), 0, st>>>(ndst, pm, x_src, x_area, x_c1, x_c2, src_idx_f, cen, nsrc, csr, tmp, ntmp)
#define SGM(O_, D_) do { if (mode == 2) SG(O_, 1, D_); else if (mode == 1) SG(O_, 16, D_); else SG(O_, 64, D_); } while (0)
  if (order == 2) { if (cen) SGM(2, true); else SGM(2, false); }
  else SGM(1, false);
#undef SGM
#undef SG
}
void fgd_src_field_index(int order, const FgTile *tiles_dev, int ntiles, int nsrc, int *src_idx_f, hipStream_t st)
{
  if (nsrc > 0) k_src_field_index<<<nblk(nsrc, 256), 256, 0, st>>>(order, tiles_dev, ntiles, nsrc, src_idx_f);
}
extern int g_apply_xcd, g_apply_ep;
void fgd_apply1(int order, int ndst, FgCsr csr, const double *f, const double *gx, const double *gy, const int *gmask,
                int has_missing, double missing, double *out, double *row_sum, hipStream_t st, long nx)
{
  if (ndst <= 0) return;
  if (g_apply_ep && nx >= 0) {                             // the entry-parallel single-level kernel; rows per tile by the mean row length
    const long m = nx / ndst;
    const int xb = g_apply_xcd;
#define EP1(O_, M_, R_) k_apply_ep1<O_, M_, 256, 512, R_><<<nblk(ndst, R_), 256, 0, st>>>(ndst, csr, f, gx, gy, gmask, missing, out, row_sum, xb)
#define EP1R(O_, M_) do { if (m <= 6) EP1(O_, M_, 64); else if (m <= 24) EP1(O_, M_, 16); else if (m <= 96) EP1(O_, M_, 4); else EP1(O_, M_, 1); } while (0)
    if (order == 2) { if (has_missing) EP1R(2, true); else EP1R(2, false); }
    else            { if (has_missing) EP1R(1, true); else EP1R(1, false); }
#undef EP1R
#undef EP1
    return;
  }
  int grid = nblk(ndst, 256);
  if (order == 2) {
    if (has_missing) k_apply1<2, true><<<grid, 256, 0, st>>>(ndst, csr, f, gx, gy, gmask, missing, out, row_sum);
    else             k_apply1<2, false><<<grid, 256, 0, st>>>(ndst, csr, f, gx, gy, gmask, missing, out, row_sum);
  } else {
    if (has_missing) k_apply1<1, true><<<grid, 256, 0, st>>>(ndst, csr, f, gx, gy, gmask, missing, out, row_sum);
    else             k_apply1<1, false><<<grid, 256, 0, st>>>(ndst, csr, f, gx, gy, gmask, missing, out, row_sum);
  }
}

extern int g_apply_xcd;
template <int NB, int V>
static void apply_il_nb(int order, int ndst, FgCsr csr, const double *f, const double *gx, const double *gy, double missing,
                        double *out, double *row_sum, long out_ld, int nb_valid, hipStream_t st)
{
  int grid = nblk(ndst, 256 / (NB / V));
  if (order == 2) k_apply_il<2, NB, V><<<grid, 256, 0, st>>>(ndst, csr, f, gx, gy, missing, out, row_sum, out_ld, nb_valid, g_apply_xcd);
  else            k_apply_il<1, NB, V><<<grid, 256, 0, st>>>(ndst, csr, f, gx, gy, missing, out, row_sum, out_ld, nb_valid, g_apply_xcd);
}
// nb in {2, 4, 8, 16}: interleaved fields [cell][nb]; out_ld > 0: write out level-major, out[level][cell] with row stride
// out_ld for the first nb_valid levels (else interleaved [cell][nb])
int g_apply_xcd = 64;  // d_xcd_block: 0 identity, 1 one band per XCD, C >= 2 chunks of C tiles per XCD (measured on the 1440x720 sweep,
                       // 8 levels merged records: 0.0936 ms with C = 32..256, 0.0966 identity, 0.1105 banded)
int g_apply_ep = 1;    // entry-parallel kernel for 8-level sweeps on records (k_apply_ep8): 0.0832 against 0.0872 ms (1440x720 x 8 levels)
int g_apply_vec = 0;   // levels per lane: 0 = auto (4 for nb >= 8, else 2: with chunked tiles 0.0986 against 0.1010 ms on interleaved arrays,
                       // 0.0875 against 0.0937 on records, 1440x720 x 8 levels), or force 1 / 2 / 4
void fgd_apply_il(int order, int nb, int ndst, FgCsr csr, const double *f, const double *gx, const double *gy, double missing,
                  double *out, double *row_sum, long out_ld, int nb_valid, hipStream_t st, long nx)
{
  if (ndst <= 0) return;
  if (order == 1 && nb == 8 && g_apply_ep && nx > 6L * ndst) {   // first order, long rows (fine -> coarse): chunked entry-parallel tiles
    const long m = nx / ndst;
    const int xb = g_apply_xcd >= 2 ? 2 * g_apply_xcd : g_apply_xcd;
#define EP8O1(R_) k_apply_ep8g<1, 256, 256, R_><<<nblk(ndst, R_), 256, 0, st>>>(ndst, csr, f, missing, out, row_sum, out_ld, nb_valid, xb)
    if (m <= 24) EP8O1(8); else if (m <= 96) EP8O1(2); else EP8O1(1);
#undef EP8O1
    return;
  }
  const int v = g_apply_vec ? g_apply_vec : (nb >= 8 ? 4 : 2);
#define AP(NB_) do { if (v >= 4 && NB_ >= 4) apply_il_nb<NB_, (NB_ >= 4 ? 4 : 2)>(order, ndst, csr, f, gx, gy, missing, out, row_sum, out_ld, nb_valid, st); \
                     else if (v >= 2) apply_il_nb<NB_, 2>(order, ndst, csr, f, gx, gy, missing, out, row_sum, out_ld, nb_valid, st); \
                     else apply_il_nb<NB_, 1>(order, ndst, csr, f, gx, gy, missing, out, row_sum, out_ld, nb_valid, st); } while (0)
  if (nb == 16) AP(16);
  else if (nb == 8) AP(8);
  else if (nb == 4) AP(4);
  else AP(2);
#undef AP
}
// order-2 sweep on merged records (k_merge3); same arguments otherwise
void fgd_apply_il_merged(int nb, int ndst, long nx, FgCsr csr, const double *rec, double missing, double *out, double *row_sum, long out_ld,
                         int nb_valid, hipStream_t st)
{
  if (ndst <= 0) return;
#define APM(NB_, V_) k_apply_il<2, NB_, V_, true><<<nblk(ndst, 256 / (NB_ / V_)), 256, 0, st>>>(ndst, csr, rec, nullptr, nullptr, missing, out, row_sum, out_ld, nb_valid, g_apply_xcd)
  if (nb == 16) APM(16, 4);
  else if (nb == 8 && g_apply_ep && nx <= 6L * ndst)       // rows of ~4 exchange cells: a tile's cells fit the product table
    k_apply_ep8<256, 256><<<nblk(ndst, EP_ROWS), 256, 0, st>>>(ndst, csr, rec, missing, out, row_sum, out_ld, nb_valid, g_apply_xcd >= 2 ? 2 * g_apply_xcd : g_apply_xcd);
  else if (nb == 8 && g_apply_ep) {                        // longer rows: chunked tiles, rows per tile by the mean row length
    const long m = nx / ndst;
    const int xb = g_apply_xcd >= 2 ? 2 * g_apply_xcd : g_apply_xcd;
#define EP8(R_) k_apply_ep8g<2, 256, 256, R_><<<nblk(ndst, R_), 256, 0, st>>>(ndst, csr, rec, missing, out, row_sum, out_ld, nb_valid, xb)
    if (m <= 24) EP8(8); else if (m <= 96) EP8(2); else EP8(1);
#undef EP8
  }
  else if (nb == 8) { if (g_apply_vec == 2) APM(8, 2); else APM(8, 4); }   // 4 levels per lane: 0.0875 ms against 0.0936 with 2 (1440x720, 8 levels, chunked tiles)
  else if (nb == 4) APM(4, 2);
  else APM(2, 2);
#undef APM
}
void fgd_merge3(int nb_pad, long n, const int *src_idx_f, const double *f, long ld_f, const double *gx, const double *gy, long ld_g,
                int nb_valid, double *out, hipStream_t st)
{
  if (n <= 0) return;
  if (nb_pad == 16) k_merge3<16, 64><<<nblk(n, 64), 256, 0, st>>>(n, src_idx_f, f, ld_f, gx, gy, ld_g, nb_valid, out);
  else if (nb_pad == 8) k_merge3<8, 64><<<nblk(n, 64), 256, 0, st>>>(n, src_idx_f, f, ld_f, gx, gy, ld_g, nb_valid, out);
  else if (nb_pad == 4) k_merge3<4, 128><<<nblk(n, 128), 256, 0, st>>>(n, src_idx_f, f, ld_f, gx, gy, ld_g, nb_valid, out);
  else k_merge3<2, 128><<<nblk(n, 128), 256, 0, st>>>(n, src_idx_f, f, ld_f, gx, gy, ld_g, nb_valid, out);
}
// level-major [nb_valid][n] (row stride ld) -> interleaved [n][nb_pad] (zero padded), and back
void fgd_interleave(int nb_pad, long n, const double *in, long ld, int nb_valid, double *out, hipStream_t st)
{
  if (n <= 0) return;
  if (nb_pad == 16) k_interleave<16><<<nblk(n, 256), 256, 0, st>>>(n, in, ld, nb_valid, out);
  else if (nb_pad == 8) k_interleave<8><<<nblk(n, 256), 256, 0, st>>>(n, in, ld, nb_valid, out);
  else if (nb_pad == 4) k_interleave<4><<<nblk(n, 256), 256, 0, st>>>(n, in, ld, nb_valid, out);
  else k_interleave<2><<<nblk(n, 256), 256, 0, st>>>(n, in, ld, nb_valid, out);
}
// three arrays at once (f, grad_x, grad_y of one chunk of levels); narr = 1 or 3
void fgd_interleave3(int nb_pad, int narr, const double *const *in, const long *ld, const long *n, double *const *out, int nb_valid,
                     hipStream_t st)
{
  FgIl3 a;
  long nmax = 0;
  for (int w = 0; w < 3; w++) {
    a.in[w] = (w < narr) ? in[w] : nullptr; a.out[w] = (w < narr) ? out[w] : nullptr;
    a.n[w] = (w < narr) ? n[w] : 0; a.ld[w] = (w < narr) ? ld[w] : 0;
    if (a.n[w] > nmax) nmax = a.n[w];
  }
  if (nmax <= 0) return;
  dim3 grid(nblk(nmax, 256), narr);
  if (nb_pad == 16) k_interleave3<16><<<grid, 256, 0, st>>>(a, nb_valid);
  else if (nb_pad == 8) k_interleave3<8><<<grid, 256, 0, st>>>(a, nb_valid);
  else if (nb_pad == 4) k_interleave3<4><<<grid, 256, 0, st>>>(a, nb_valid);
  else k_interleave3<2><<<grid, 256, 0, st>>>(a, nb_valid);
}
void fgd_deinterleave(int nb_pad, long n, const double *in, long ld, int nb_valid, double *out, hipStream_t st)
{
  if (n <= 0) return;
  if (nb_pad == 16) k_deinterleave<16><<<nblk(n, 256), 256, 0, st>>>(n, in, ld, nb_valid, out);
  else if (nb_pad == 8) k_deinterleave<8><<<nblk(n, 256), 256, 0, st>>>(n, in, ld, nb_valid, out);
  else if (nb_pad == 4) k_deinterleave<4><<<nblk(n, 256), 256, 0, st>>>(n, in, ld, nb_valid, out);
  else k_deinterleave<2><<<nblk(n, 256), 256, 0, st>>>(n, in, ld, nb_valid, out);
}
void fgd_apply_ex(int order, int ndst, FgCsr csr, const double *f, const double *gx, const double *gy, FgApplyEx o,
                  double *out, double *row_sum, int *err, hipStream_t st, long nx)
{
  if (ndst <= 0) return;
  if (g_apply_ep && nx >= 0) {                             // entry-parallel; rows per tile by the mean row length
    const long m = nx / ndst;
    const int xb = g_apply_xcd;
#define EPX(O_, M_, R_) k_apply_epx<O_, M_, 256, 512, R_><<<nblk(ndst, R_), 256, 0, st>>>(ndst, csr, f, gx, gy, o, out, row_sum, err, xb)
#define EPXR(O_, M_) do { if (m <= 6) EPX(O_, M_, 64); else if (m <= 24) EPX(O_, M_, 16); else if (m <= 96) EPX(O_, M_, 4); else EPX(O_, M_, 1); } while (0)
    if (order == 2) { if (o.xdata) EPXR(2, true); else EPXR(2, false); }
    else EPXR(1, false);
#undef EPXR
#undef EPX
    return;
  }
  int grid = nblk(ndst, 256);
  if (order == 2) {
    if (o.xdata) k_apply_ex<2, true><<<grid, 256, 0, st>>>(ndst, csr, f, gx, gy, o, out, row_sum, err);
    else         k_apply_ex<2, false><<<grid, 256, 0, st>>>(ndst, csr, f, gx, gy, o, out, row_sum, err);
  } else
    k_apply_ex<1, false><<<grid, 256, 0, st>>>(ndst, csr, f, gx, gy, o, out, row_sum, err);
}
void fgd_mono_bounds(const FgTile *tiles_dev, int ntiles, int nsrc, const int *src_idx_f, const double *f, double missing,
                     double *fbmax, double *fbmin, double *fmax, double *fmin, hipStream_t st)
{
  if (nsrc > 0) k_mono_bounds<<<nblk(nsrc, 256), 256, 0, st>>>(tiles_dev, ntiles, nsrc, src_idx_f, f, missing, fbmax, fbmin, fmax, fmin);
}
void fgd_mono_xdata(long nx, FgCsr csr, const double *f, const double *gx, const double *gy, const int *gmask, double missing,
                    double *xdata, double *fmax, double *fmin, hipStream_t st)
{
  if (nx > 0) k_mono_xdata<<<nblk(nx, 256), 256, 0, st>>>(nx, csr, f, gx, gy, gmask, missing, xdata, fmax, fmin);
}
void fgd_mono_limit(long nx, FgCsr csr, const double *f, double missing, const double *fbmax, const double *fbmin,
                    const double *fmax, const double *fmin, double *xdata, int *err, hipStream_t st)
{
  if (nx > 0) k_mono_limit<<<nblk(nx, 256), 256, 0, st>>>(nx, csr, f, missing, fbmax, fbmin, fmax, fmin, xdata, err);
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
