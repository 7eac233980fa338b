// xgrid_kernels.hip -- exchange-grid search kernels for gfx950 (MI355X).
//
// Pipeline for one destination tile against all source tiles:
//
//   k_cell_struct      per cell: lat min/max, fix_lon, lon min/max/avg, <=8 vertices, area
//                      [get_grid_cell_struct semantics, create_xgrid.c:991-1016; get_grid_area :66-88]
//   k_bin_build        destination cells -> uniform (lat x lon mod 2pi) bins, one 48-byte record per
//                      cell stored in bin order (single insertion; wide cells in per-row lists)
//   k_candidates(+_heavy)  per source cell: scan the <= 2 contiguous record ranges per bin row,
//                      apply the reference's exact bounding-box rejects (create_xgrid.c:1055-1079)
//                      -> candidate pairs [get_upbound_nxcells_2dx2d semantics]; count pass + fill
//                      pass; source cells at the poles (huge longitude range) get a whole wave
//   k_clip_quad        one lane per candidate pair, quad x quad fast path: Sutherland-Hodgman
//                      clip (create_xgrid.c:1266-1341) with the polygon staged in LDS
//                      [vertex][lane], then area / centroid integrals and the 1e-6 area test
//   k_clip_general     same for pairs with pole-fixed cells (5..8 vertices) or fast-path overflow
//   (accepted pairs are counted per source cell inside the clip kernels: wave-segmented ballot, one atomic per run)
//   k_scatter_xcells   compaction into the reference's canonical order (source cell ascending,
//                      destination cell index ascending) via per-source-cell rank; counts the CSR row sizes
//   k_cell_sums, k_centroids, k_distances     order-2 centroid pass (conserve_interp.c:216-221,319-358)
//
// No MFMA: this is FP64 VALU + irregular gather work.  Every floating-point operation uses the reference's expression
// trees, sin/cos included (geom.hip.h, sincos_glibc.h): lists, areas and centroid integrals are the reference's bits.
// Launched either for exact sizes or for fixed capacities with the true counts read from device memory (np_dev, cap).
#include "xgrid_device.h"
// The 3.5 KB sin/cos table of sincos_glibc.h is copied into LDS by every kernel of this file that takes sines: the lookups
// are per-lane gathers (index = latitude * 128), and from LDS they cost the clip kernel 0.42 ms instead of 0.48 ms from global
// memory, although the extra LDS and registers lower its occupancy from 5 to 4 waves per SIMD.
static __shared__ double fgs_lds_tab[112 * 4];
#define FGS_TAB(k, j) fgs_lds_tab[(k) * 4 + (j)]
#include "geom.hip.h"
__device__ __forceinline__ void d_load_trig_table()
{
  for (int i = threadIdx.x; i < 112 * 4; i += blockDim.x) fgs_lds_tab[i] = FG_SINCOS_TAB[i >> 2][i & 3];
  __syncthreads();
}

// ---------------------------------------------------------------------------------------
// exclusive scan of int32 counts (3 kernels: block sums, top-level, apply)
// ---------------------------------------------------------------------------------------
#define SCAN_THREADS 256
#define SCAN_ITEMS   8
#define SCAN_CHUNK   (SCAN_THREADS * SCAN_ITEMS)

__device__ __forceinline__ unsigned wave_incl_scan(unsigned v, int lane)
{
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    unsigned t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// inclusive scan across a 256-thread block; returns this thread's inclusive value and the block total
__device__ __forceinline__ unsigned block_incl_scan(unsigned v, unsigned *total)
{
  __shared__ unsigned wsum[4];
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned inc = wave_incl_scan(v, lane);
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) { if (k < w) base += wsum[k]; tot += wsum[k]; }
  __syncthreads();
  *total = tot;
  return inc + base;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_block_sums(const int *in, long n, unsigned long long *bsum)
{
  long base = (long)blockIdx.x * SCAN_CHUNK;
  unsigned s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    long idx = base + (long)k * SCAN_THREADS + threadIdx.x;
    if (idx < n) s += (unsigned)in[idx];
  }
  unsigned tot;
  block_incl_scan(s, &tot);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// out[i] = exclusive prefix for i in [0, n] (n inputs, n+1 outputs; 32-bit, the host checks the 64-bit total fits).
// Each block adds up the sums of the blocks before it by itself (at most a few thousand 8-byte values from L2) instead of
// waiting for a separate single-block pass over them: one launch less per scan, and the search runs four scans.
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const int *in, long n, const unsigned long long *bsum, int *out,
                                                              unsigned long long *total_out)
{
  __shared__ unsigned tile[SCAN_CHUNK];
  __shared__ unsigned long long part[SCAN_THREADS / 64];
  long base = (long)blockIdx.x * SCAN_CHUNK;
  // 64-bit sum of the preceding blocks' totals
  unsigned long long before = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += SCAN_THREADS) before += bsum[i];
#pragma unroll
  for (int o = 32; o; o >>= 1) before += __shfl_xor(before, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = before;
  // thread t owns items t*SCAN_ITEMS .. +SCAN_ITEMS-1 of the chunk (blocked arrangement via LDS)
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    int li = k * SCAN_THREADS + threadIdx.x;
    long idx = base + li;
    tile[li] = (idx < n) ? (unsigned)in[idx] : 0u;
  }
  __syncthreads();
  unsigned long long block_base = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; w++) block_base += part[w];
  unsigned loc[SCAN_ITEMS];
  unsigned s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) { loc[k] = tile[threadIdx.x * SCAN_ITEMS + k]; s += loc[k]; }
  unsigned tot;
  unsigned inc = block_incl_scan(s, &tot);
  unsigned run = (unsigned)block_base + inc - s;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) { tile[threadIdx.x * SCAN_ITEMS + k] = run; run += loc[k]; }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    int li = k * SCAN_THREADS + threadIdx.x;
    long idx = base + li;
    if (idx <= n) out[idx] = (int)tile[li];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = block_base + tot;
}

int fgd_exclusive_scan(const int *in, long n, int *out, unsigned long long *bsum_ws,
                       unsigned long long *total_dev, hipStream_t st)
{
  // n inputs -> n+1 outputs (out[n] = total); blocks cover n+1 positions, inputs beyond n read as 0
  if (n < 0) n = 0;
  int nb = (int)((n + 1 + SCAN_CHUNK - 1) / SCAN_CHUNK);
  k_scan_block_sums<<<nb, SCAN_THREADS, 0, st>>>(in, n, bsum_ws);
  k_scan_apply<<<nb, SCAN_THREADS, 0, st>>>(in, n, bsum_ws, out, total_dev);
  return 0;
}

long fgd_scan_ws_elems(long n) { return (n + 1 + SCAN_CHUNK - 1) / SCAN_CHUNK + 1; }

// ---------------------------------------------------------------------------------------
// per-cell records
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cell_struct(const FgTile *tiles, int ntiles, int ncells, FgCells c, unsigned *err)
{
  // the 16 vertex doubles of a cell record are staged so that the block stores its 256 records as one contiguous run
  // (a lane writing its own 128-byte record makes sixteen 8-byte stores at a 128-byte stride)
  __shared__ double vtile[256 * 17];
  d_load_trig_table();
  const int s0 = blockIdx.x * blockDim.x;
  const int s = s0 + threadIdx.x;
  double *row = vtile + threadIdx.x * 17;
#pragma unroll
  for (int k = 0; k < 16; k++) row[k] = 0.0;
  if (s < ncells) {
    int t = 0;
    while (t + 1 < ntiles && s >= tiles[t + 1].cell_off) t++;
    const FgTile T = tiles[t];
    int loc = s - T.cell_off;
    int i = loc % T.nx, j = loc / T.nx;
    int nxp = T.nx + 1;
    int n0 = j * nxp + i, n1 = n0 + 1, n3 = n0 + nxp, n2 = n3 + 1;
    double x[G_FIXCAP], y[G_FIXCAP];
    x[0] = T.lon[n0]; y[0] = T.lat[n0];
    x[1] = T.lon[n1]; y[1] = T.lat[n1];
    x[2] = T.lon[n2]; y[2] = T.lat[n2];
    x[3] = T.lon[n3]; y[3] = T.lat[n3];
    double lmin = y[0], lmax = y[0];
#pragma unroll
    for (int k = 1; k < 4; k++) { if (y[k] < lmin) lmin = y[k]; if (y[k] > lmax) lmax = y[k]; }
    c.lat_min[s] = lmin; c.lat_max[s] = lmax;
    if (!(lmin >= -G_HPI - 1.e-6) || !(lmax <= G_HPI + 1.e-6)) atomicOr(err, G_ERRBIT_BADLAT);   // also catches NaN
    int n = d_fix_lon(x, y, 4, G_PI);
    if (n < 0 || n > G_MAXV) {
      atomicOr(err, G_ERRBIT_MAXV);
      c.nv[s] = 0; c.lon_min[s] = 0; c.lon_max[s] = 0; c.lon_avg[s] = 0; c.area[s] = 0;
    } else {
      double xmin = x[0], xmax = x[0], xs = 0;
      for (int k = 1; k < n; k++) { if (x[k] < xmin) xmin = x[k]; if (x[k] > xmax) xmax = x[k]; }
      for (int k = 0; k < n; k++) xs += x[k];
      xs /= n;
      c.lon_min[s] = xmin; c.lon_max[s] = xmax; c.lon_avg[s] = xs;
      c.nv[s] = n;
      for (int k = 0; k < G_MAXV; k++) {
        row[k] = (k < n) ? x[k] : 0.0;
        row[8 + k] = (k < n) ? y[k] : 0.0;
      }
      c.area[s] = d_poly_area<1>(x, y, n);
    }
  }
  __syncthreads();
  const long cnt = (long)min(256, ncells - s0) * 16;
  double *out = c.verts + (size_t)s0 * 16;
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const long e = (long)q * 256 + threadIdx.x;
    if (e < cnt) out[e] = vtile[(e >> 4) * 17 + (e & 15)];
  }
}

// ---------------------------------------------------------------------------------------
// binning of destination cells
// ---------------------------------------------------------------------------------------
// Uniform bins over latitude x (longitude mod 2pi), sized ~1.5x the mean destination cell.
// A destination cell whose bounding box covers at most 2x2 bins ("regular") is stored ONCE, in
// the bin of its (lat_min, lon_min) corner; a source cell therefore finds every regular cell
// that can overlap it by scanning rows [r0-1, r1] x columns [c0-1, c1] of its own box -- and
// because bins of one row are adjacent in the table, that is at most two contiguous entry
// ranges per row.  Cells with a larger footprint (pole caps of a cubed-sphere target, ...) go to
// a per-latitude-row "wide" list that every source cell of that row tests.
// Margins of 1e-9 rad make the scan a superset of the pairs that pass the reference's tests,
// including its +-2pi shifts (create_xgrid.c:1064-1074); see DESIGN.md §3.2.
__device__ __forceinline__ void d_cell_box(double lat_min, double lat_max, double lon_min, double lon_max, FgBins b,
                                           int *r0, int *r1, long long *l0, long long *l1)
{
  const double eps = 1.e-9;
  int a0 = (int)floor((lat_min - eps + G_HPI) * b.inv_wlat);
  int a1 = (int)floor((lat_max + eps + G_HPI) * b.inv_wlat);
  *r0 = max(0, min(b.nblat - 1, a0));
  *r1 = max(0, min(b.nblat - 1, a1));
  *l0 = (long long)floor((lon_min - eps) * b.inv_wlon);
  *l1 = (long long)floor((lon_max + eps) * b.inv_wlon);
}
__device__ __forceinline__ int d_colmod(long long l, int nb) { return (int)(((l % nb) + nb) % nb); }

// one atomic per run of consecutive lanes that target the same slot (key >= 0; lanes with key < 0 sit out): returns this
// lane's position in the slot.  Consecutive cells of a regular grid share a bin (2.25 cells per bin), and same-address
// atomics that return a value are the expensive part of the fill pass.
__device__ __forceinline__ int d_slot_position(int *slot_cnt, int key)
{
  const int lane = threadIdx.x & 63;
  const int prev = __shfl_up(key, 1);
  const bool head = (lane == 0) || (key != prev);
  const unsigned long long heads = __ballot(head);
  const int start = 63 - __clzll((long long)(heads & ((2ull << lane) - 1ull)));
  const unsigned long long above = (start == 63) ? 0ull : (heads & ~((2ull << start) - 1ull));
  const int end = above ? (__ffsll((long long)above) - 1) : 64;
  int base = 0;
  if (lane == start && key >= 0) base = atomicAdd(&slot_cnt[key], end - start);
  base = __shfl(base, start);
  return base + (lane - start);
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_bin_build(int ncells, FgCells c, FgBins b, int *slot_cnt, const int *slot_start,
                                                    FgBinEntry *entries, int cap)
{
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = d < ncells && c.nv[d] != 0;      // (no early return: the slot positions are a wave-wide operation)
  FgBinEntry E;
  int r0 = 0, r1 = 0; long long l0 = 0, l1 = 0;
  if (live) {
    E.lat_min = c.lat_min[d]; E.lat_max = c.lat_max[d];
    E.lon_min = c.lon_min[d]; E.lon_max = c.lon_max[d]; E.lon_avg = c.lon_avg[d];
    E.d = d;
    d_cell_box(E.lat_min, E.lat_max, E.lon_min, E.lon_max, b, &r0, &r1, &l0, &l1);
    E.row0 = r0;
  }
  const int nbins = b.nblat * b.nblon;
  const bool regular = live && r1 - r0 <= 1 && l1 - l0 <= 1;
  const int slot = regular ? r0 * b.nblon + d_colmod(l0, b.nblon) : -1;
  if (FILL) {
    const int pos = d_slot_position(slot_cnt, slot);
    if (regular) { const int at = slot_start[slot] + pos; if (at < cap) entries[at] = E; }
  } else if (regular)
    atomicAdd(&slot_cnt[slot], 1);
  // wide cells go into the per-row lists of every row they span.  Neighbouring cells of a grid row span the same rows, so a
  // whole wave often targets one list: the rows are walked in lockstep and each step takes one atomic per run of equal lists
  // (the great-circle search, whose cap-derived boxes are wide near the poles, spent 0.45 ms here on same-address atomics).
  const int nrows = (live && !regular) ? (r1 - r0 + 1) : 0;
  int maxrows = nrows;
#pragma unroll
  for (int o = 32; o; o >>= 1) maxrows = max(maxrows, __shfl_xor(maxrows, o));
  for (int it = 0; it < maxrows; it++) {
    const int key = (it < nrows) ? nbins + r0 + it : -1;
    const int pos = d_slot_position(slot_cnt, key);
    if (FILL && key >= 0) { const int at = slot_start[key] + pos; if (at < cap) entries[at] = E; }
  }
}

// ---------------------------------------------------------------------------------------
// candidate pairs
// ---------------------------------------------------------------------------------------
struct SrcQuery {
  int ra, rb;          // rows holding regular cells that may overlap: [max(0, r0-1), r1]
  int r0, r1;          // the source cell's own rows (wide lists)
  int c_start, n0, n1; // columns [c_start, c_start+n0) and, wrapped, [0, n1)
};

__device__ __forceinline__ SrcQuery d_src_query(double lat_min, double lat_max, double lon_min, double lon_max, FgBins b)
{
  SrcQuery q;
  long long l0, l1;
  d_cell_box(lat_min, lat_max, lon_min, lon_max, b, &q.r0, &q.r1, &l0, &l1);
  q.ra = max(0, q.r0 - 1); q.rb = q.r1;
  long long count = (l1 - l0 + 1) + 1;
  if (count >= b.nblon) { q.c_start = 0; q.n0 = b.nblon; q.n1 = 0; }
  else {
    q.c_start = d_colmod(l0 - 1, b.nblon);
    q.n0 = (int)min((long long)(b.nblon - q.c_start), count);
    q.n1 = (int)count - q.n0;
  }
  return q;
}

// size of the query: bins scanned (a bin holds ~2 cells: bins are 1.5x the mean cell) plus the entries of
// the wide lists of its rows (two table loads).  Only used to route huge queries to the wave-per-cell kernel.
__device__ __forceinline__ int d_query_size(const SrcQuery &q, FgBins b, const int *slot_start)
{
  const int nbins = b.nblat * b.nblon;
  return (q.rb - q.ra + 1) * (q.n0 + q.n1) + (slot_start[nbins + q.r1 + 1] - slot_start[nbins + q.r0]);
}

// the reference's two bounding-box rejects, create_xgrid.c:1055 and :1062-1079
__device__ __forceinline__ bool d_box_pass(const FgBinEntry &E, double lat_in_min, double lat_in_max,
                                           double lon_in_min, double lon_in_max, double lon_in_avg)
{
  if (E.lat_min >= lat_in_max || E.lat_max <= lat_in_min) return false;
  double lon_out_min = E.lon_min, lon_out_max = E.lon_max;
  double dx = E.lon_avg - lon_in_avg;
  if (dx < -G_PI)     { lon_out_min += G_TPI; lon_out_max += G_TPI; }
  else if (dx > G_PI) { lon_out_min -= G_TPI; lon_out_max -= G_TPI; }
  if (lon_out_min >= lon_in_max || lon_out_max <= lon_in_min) return false;
  return true;
}

#define HEAVY_ENTRIES 32      // scanned bins + wide entries above which a source cell gets a whole wave (measured: 16 / 24 / 32 / 64 / 128
                              // give 0.31 / 0.27 / 0.28 / 0.305 / 0.38 ms for the candidate phase at C384 -> 0.25 deg; the lanes with the
                              // longest scans set the duration of the four-lanes-per-cell kernel)
#define CAND_G 4          // lanes per source cell in the candidate scan (one bin row each)

// CAND_G lanes per source cell, each scanning every CAND_G-th bin row of the cell's query.  Counts and
// offsets are kept per lane (index s*CAND_G + sub) so the fill pass needs no second counting sweep:
//   FILL == false: cand_cnt[s*G+sub] = number of destination cells passing the rejects in this lane's rows;
//                  cells whose query touches more than HEAVY_ENTRIES table entries are appended to
//                  heavy_list instead (k_candidates_heavy gives them a whole wave and writes cand_cnt[s*G]).
//   FILL == true:  write the pairs at cand_off[s*G+sub]...
//   stage[s*G+sub]: the first four destination cells the counting pass found, so that a lane with at most four (the common
//                  case: 5.2 pairs per source cell over four lanes) copies them in the fill pass instead of scanning the
//                  bins again; x = -2 marks a heavy cell.  Lanes with more than four rescan, in the same order, so the pair
//                  list is the one the plain two-pass scheme writes.  (Staging a whole cell's list and sending every
//                  overflow to the wave-per-cell kernel was measured too: no faster at C384 -> 0.25 deg, 45 % slower for
//                  coarse -> fine grids, where most cells overflow.)
// (one wave per block: a block gives its slots back when its slowest wave is done, and the scan lengths vary a lot --
// measured 256 / 128 / 64 threads: 0.240 / 0.230 / 0.224 ms for the phase)
template <bool FILL>
__global__ __launch_bounds__(64) void k_candidates(int nsrc, FgCells S, const double *mask, FgBins b,
                                                     const int *slot_start, const FgBinEntry *entries,
                                                     int *cand_cnt, const int *cand_off, int *pair_src, int *pair_dst,
                                                     int *heavy_list, int *heavy_cnt, int cap, int4 *stage, int ecap)
{
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int s = (int)(t / CAND_G), sub = (int)(t % CAND_G);
  if (s >= nsrc) return;
  int cnt = 0;
  int id0 = -1, id1 = -1, id2 = -1, id3 = -1;
  if (FILL) {
    const int4 sg = stage[t];
    if (sg.x == -2) return;                              // heavy cell: k_candidates_heavy writes its pairs
    const int n = cand_cnt[t];
    if (n == 0) return;
    if (n <= 4) {
      const int wbase = cand_off[t];
      const int ids[4] = {sg.x, sg.y, sg.z, sg.w};
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (k < n && wbase + k < cap) { pair_src[wbase + k] = s; pair_dst[wbase + k] = ids[k]; }
      return;
    }
  }
  bool active = S.nv[s] > 0;
  if (active && mask) active = mask[s] > 0.5;          // MASK_THRESH, create_xgrid.c:1030
  if (active) {
    const double lat_in_min = S.lat_min[s], lat_in_max = S.lat_max[s];
    const double lon_in_min = S.lon_min[s], lon_in_max = S.lon_max[s], lon_in_avg = S.lon_avg[s];
    SrcQuery q = d_src_query(lat_in_min, lat_in_max, lon_in_min, lon_in_max, b);
    if (!FILL && d_query_size(q, b, slot_start) > HEAVY_ENTRIES) {
      if (sub == 0) { int h = atomicAdd(heavy_cnt, 1); heavy_list[h] = s; }
      else cand_cnt[t] = 0;                            // cand_cnt[s*G] comes from the heavy kernel
      stage[t] = make_int4(-2, 0, 0, 0);
      return;
    }
    const int wbase = FILL ? cand_off[t] : 0;
    const int nbins = b.nblat * b.nblon;
    for (int r = q.ra + sub; r <= q.rb; r += CAND_G) {
      int base = r * b.nblon;
      for (int seg = 0; seg < 2; seg++) {
        if (seg && !q.n1) break;
        int e0 = seg ? slot_start[base] : slot_start[base + q.c_start];
        int e1 = seg ? slot_start[base + q.n1] : slot_start[base + q.c_start + q.n0];
        e1 = min(e1, ecap);                              // a single-sync search may have outgrown its record buffer (it is then repeated)
        for (int e = e0; e < e1; e++) {
          const FgBinEntry E = entries[e];
          if (!d_box_pass(E, lat_in_min, lat_in_max, lon_in_min, lon_in_max, lon_in_avg)) continue;
          if (FILL && wbase + cnt < cap) { pair_src[wbase + cnt] = s; pair_dst[wbase + cnt] = E.d; }
          if (!FILL) { id0 = cnt == 0 ? E.d : id0; id1 = cnt == 1 ? E.d : id1; id2 = cnt == 2 ? E.d : id2; id3 = cnt == 3 ? E.d : id3; }
          cnt++;
        }
      }
    }
    for (int r = q.r0 + sub; r <= q.r1; r += CAND_G) {
      int e0 = slot_start[nbins + r], e1 = min(slot_start[nbins + r + 1], ecap);
      for (int e = e0; e < e1; e++) {
        const FgBinEntry E = entries[e];
        if (r != max(q.r0, E.row0)) continue;          // a wide cell sits in every row it spans
        if (!d_box_pass(E, lat_in_min, lat_in_max, lon_in_min, lon_in_max, lon_in_avg)) continue;
        if (FILL && wbase + cnt < cap) { pair_src[wbase + cnt] = s; pair_dst[wbase + cnt] = E.d; }
        if (!FILL) { id0 = cnt == 0 ? E.d : id0; id1 = cnt == 1 ? E.d : id1; id2 = cnt == 2 ? E.d : id2; id3 = cnt == 3 ? E.d : id3; }
        cnt++;
      }
    }
  }
  if (!FILL) { cand_cnt[t] = cnt; stage[t] = make_int4(id0, id1, id2, id3); }
}

// One wave per heavy source cell (pole caps of the source grid: their longitude range covers
// hundreds of bins); lanes stride the contiguous entry ranges, ballot + popcount compacts.
template <bool FILL>
__global__ __launch_bounds__(64) void k_candidates_heavy(FgCells S, FgBins b, const int *slot_start, const FgBinEntry *entries,
                                                          int *cand_cnt, const int *cand_off, int *pair_src, int *pair_dst,
                                                          const int *heavy_list, const int *heavy_cnt, int cap, int ecap)
{
  const int lane = threadIdx.x;
  const int nheavy = *heavy_cnt;
  const int nbins = b.nblat * b.nblon;
  for (int h = blockIdx.x; h < nheavy; h += gridDim.x) {
    const int s = heavy_list[h];
    const double lat_in_min = S.lat_min[s], lat_in_max = S.lat_max[s];
    const double lon_in_min = S.lon_min[s], lon_in_max = S.lon_max[s], lon_in_avg = S.lon_avg[s];
    SrcQuery q = d_src_query(lat_in_min, lat_in_max, lon_in_min, lon_in_max, b);
    const int wbase = FILL ? cand_off[s * CAND_G] : 0;
    int cnt = 0;
    const int nrows_reg = q.rb - q.ra + 1, nrows_wide = q.r1 - q.r0 + 1;
    for (int it = 0; it < 2 * nrows_reg + nrows_wide; it++) {
      int e0, e1, wide_row = -1;
      if (it < 2 * nrows_reg) {
        int r = q.ra + (it >> 1), seg = it & 1, base = r * b.nblon;
        if (seg && !q.n1) continue;
        e0 = seg ? slot_start[base] : slot_start[base + q.c_start];
        e1 = seg ? slot_start[base + q.n1] : slot_start[base + q.c_start + q.n0];
      } else {
        wide_row = q.r0 + (it - 2 * nrows_reg);
        e0 = slot_start[nbins + wide_row]; e1 = slot_start[nbins + wide_row + 1];
      }
      e1 = min(e1, ecap);
      for (int eb = e0; eb < e1; eb += 64) {
        int e = eb + lane;
        bool pass = false;
        int dcell = 0;
        if (e < e1) {
          const FgBinEntry E = entries[e];
          dcell = E.d;
          pass = d_box_pass(E, lat_in_min, lat_in_max, lon_in_min, lon_in_max, lon_in_avg);
          if (wide_row >= 0 && wide_row != max(q.r0, E.row0)) pass = false;
        }
        unsigned long long m = __ballot(pass);
        if (FILL && pass) {
          int pos = wbase + cnt + __popcll(m & ((1ull << lane) - 1ull));
          if (pos < cap) { pair_src[pos] = s; pair_dst[pos] = dcell; }
        }
        cnt += __popcll(m);
      }
    }
    if (!FILL && lane == 0) cand_cnt[s * CAND_G] = cnt;
  }
}

// ---------------------------------------------------------------------------------------
// clip + area (+ centroid integrals)
// ---------------------------------------------------------------------------------------
struct ClipOut { double area, clon, clat; };

// area test and integrals on the clipped polygon held at px/py (stride S); returns area or -1
template <int ORDER, int S>
__device__ __forceinline__ void d_finish_pair(const double *px, const double *py, int n_out, double maskv,
                                              double area_in, double area_out, double lon_in_avg,
                                              ClipOut *o, unsigned long long *stats)
{
  double pa, clon = 0, clat = 0;
  if (ORDER == 2) d_poly_area_ctr<S>(px, py, n_out, lon_in_avg, &pa, &clon, &clat);
  else pa = d_poly_area<S>(px, py, n_out);
  double xarea = pa * maskv;                                   // create_xgrid.c:1083
  double min_area = (area_in < area_out) ? area_in : area_out; // :1084
  double ratio = xarea / min_area;
  if (fabs(ratio - 1.e-6) < 1.e-15) atomicAdd(&stats[FG_STAT_BORDERLINE], 1ull);
  if (ratio > 1.e-6) { o->area = xarea; o->clon = clon; o->clat = clat; }
  else o->area = -2.0;                                         // non-empty clip, below the area threshold
}

#define CLIP_THREADS 256

// Quad x quad fast path.  LDS: polygon [8][256] double2 (32 KiB); the cutting quad lives in registers.
// Returns false if the pair must go to the general kernel (more than 4 vertices on a side, or more than 8
// in an intermediate polygon); otherwise *o holds the result (area >= 0 accepted, -1 empty, -2 below threshold).
template <int ORDER>
__device__ __forceinline__ bool d_clip_quad_pair(double2 (*sh_poly)[CLIP_THREADS], const int tid, const int s, const int d,
                                                 const FgCells &S, const double *mask, const FgCells &D,
                                                 ClipOut *o_out, unsigned long long *stats, unsigned *err)
{
  const int n1 = S.nv[s], n2 = D.nv[d];
  if (n1 > 4 || n2 > 4) return false;

  const double *sv = S.verts + (size_t)s * 16, *dv = D.verts + (size_t)d * 16;
  const double lon_in_avg = S.lon_avg[s];
  double shift = 0.0;
  {
    double dx = D.lon_avg[d] - lon_in_avg;           // create_xgrid.c:1064-1074
    if (dx < -G_PI) shift = G_TPI; else if (dx > G_PI) shift = -G_TPI;
  }
  double x1[4], y1[4], x2[4], y2[4];
  bool wrap = false;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    x1[k] = sv[k]; y1[k] = sv[8 + k];
    x2[k] = dv[k]; y2[k] = dv[8 + k];
    if (shift != 0.0) x2[k] += shift;
    if (k < n1 && (x1[k] > G_TPI || x1[k] < 0.0)) wrap = true;  // create_xgrid.c:1282
  }
  if (wrap) {                                          // :1290
#pragma unroll
    for (int k = 0; k < 4; k++) { x1[k] = d_pimod1(x1[k]); x2[k] = d_pimod1(x2[k]); }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) sh_poly[k][tid] = make_double2(x1[k], y1[k]);
  // the cutting quad stays in registers; its vertex e is picked with selects (n2 <= 4)
  // vertex e of the cutting quad, picked with bit masks (a ?: chain on e is turned into a scratch-memory
  // table by the optimizer)
  const long long b0x = __double_as_longlong(x2[0]), b1x = __double_as_longlong(x2[1]);
  const long long b2x = __double_as_longlong(x2[2]), b3x = __double_as_longlong(x2[3]);
  const long long b0y = __double_as_longlong(y2[0]), b1y = __double_as_longlong(y2[1]);
  const long long b2y = __double_as_longlong(y2[2]), b3y = __double_as_longlong(y2[3]);
  auto cut = [=](int e) -> double2 {
    const long long m0 = -(long long)(e == 0), m1 = -(long long)(e == 1), m2 = -(long long)(e == 2), m3 = -(long long)(e == 3);
    return make_double2(__longlong_as_double((b0x & m0) | (b1x & m1) | (b2x & m2) | (b3x & m3)),
                        __longlong_as_double((b0y & m0) | (b1y & m1) | (b2y & m2) | (b3y & m3)));
  };
  int n_cur = n1;
  bool overflow = false, parallel = false;
  double2 e0 = cut(n2 - 1);
  for (int e = 0; e < n2 && n_cur > 0; e++) {
    double2 e1 = cut(e);
    const double x2_0 = e0.x, y2_0 = e0.y, x2_1 = e1.x, y2_1 = e1.y;
    double2 c[8];
#pragma unroll
    for (int k = 0; k < 8; k++) if (k < n_cur) c[k] = sh_poly[k][tid];
    double2 lastv = sh_poly[n_cur - 1][tid];
    double x1_0 = lastv.x, y1_0 = lastv.y;
    int inside_last = d_inside_edge(x2_0, y2_0, x2_1, y2_1, x1_0, y1_0);
    int n_new = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (k < n_cur) {
        double x1_1 = c[k].x, y1_1 = c[k].y;
        int inside = d_inside_edge(x2_0, y2_0, x2_1, y2_1, x1_1, y1_1);
        if (inside != inside_last) {
          double dy1 = y1_1 - y1_0;
          double dy2 = y2_1 - y2_0;
          double dx1 = x1_1 - x1_0;
          double dx2 = x2_1 - x2_0;
          double ds1 = y1_0 * x1_1 - y1_1 * x1_0;
          double ds2 = y2_0 * x2_1 - y2_1 * x2_0;
          double determ = dy2 * dx1 - dy1 * dx2;
          if (fabs(determ) < 1.0e-30) parallel = true;
          if (n_new < 8) sh_poly[n_new][tid] = make_double2((dx2 * ds1 - dx1 * ds2) / determ,
                                                            (dy2 * ds1 - dy1 * ds2) / determ);
          else overflow = true;
          n_new++;
        }
        if (inside) {
          if (n_new < 8) sh_poly[n_new][tid] = make_double2(x1_1, y1_1);
          else overflow = true;
          n_new++;
        }
        x1_0 = x1_1; y1_0 = y1_1; inside_last = inside;
      }
    }
    n_cur = n_new;
    if (overflow) break;
    e0 = e1;
  }
  if (overflow) return false;
  if (parallel) atomicOr(err, G_ERRBIT_PARALLEL);
  ClipOut o; o.area = -1.0; o.clon = 0; o.clat = 0;
  if (n_cur > 0) {
    const double *px = (const double *)&sh_poly[0][tid];
    d_finish_pair<ORDER, 2 * CLIP_THREADS>(px, px + 1, n_cur, mask ? mask[s] : 1.0, S.area[s], D.area[d],
                                            lon_in_avg, &o, stats);
  }
  *o_out = o;
  return true;
}

// Result encoding shared by the clip kernels and the compaction: an accepted pair keeps pair_dst[p] = d and
// gets tmp_area/clon/clat[p]; a rejected pair gets pair_dst[p] = -1.  nacc[s] counts the accepted pairs of
// source cell s: lanes are pair-ordered, so one atomic per (wave, source cell) run does it.
template <int ORDER>
__global__ __launch_bounds__(CLIP_THREADS) void k_clip_quad(int npairs, const int *pair_src, int *pair_dst,
                                                            FgCells S, const double *mask, FgCells D,
                                                            double *tmp_area, double *tmp_clon, double *tmp_clat,
                                                            int *nacc, int *defer_list, int *defer_cnt,
                                                            unsigned long long *stats, unsigned *err, const unsigned long long *np_dev)
{
  __shared__ double2 sh_poly[8][CLIP_THREADS];
  d_load_trig_table();
  const int tid = threadIdx.x, lane = tid & 63;
  const int p = blockIdx.x * CLIP_THREADS + tid;
  if (np_dev) { const unsigned long long nd = *np_dev; if (nd < (unsigned long long)npairs) npairs = (int)nd; }   // launched for the capacity
  int s = -1;
  bool acc = false;
  if (p < npairs) {
    s = pair_src[p];
    const int d = pair_dst[p];
    ClipOut o;
    if (!d_clip_quad_pair<ORDER>(sh_poly, tid, s, d, S, mask, D, &o, stats, err)) {
      int q = atomicAdd(defer_cnt, 1); defer_list[q] = p;      // rare; the general kernel finishes this pair
    } else if (o.area >= 0) {
      acc = true;
      tmp_area[p] = o.area;
      if (ORDER == 2) { tmp_clon[p] = o.clon; tmp_clat[p] = o.clat; }
    } else {
      pair_dst[p] = -1;
      if (o.area == -2.0) atomicAdd(&stats[FG_STAT_BELOW], 1ull);   // rare (slivers below the 1e-6 ratio)
    }
  }
  // segmented count of accepted lanes per run of equal s inside the wave
  const int s_prev = __shfl_up(s, 1, 64);
  const bool head = (lane == 0) || (s != s_prev);
  const unsigned long long hm = __ballot(head), am = __ballot(acc);
  if (head && s >= 0) {
    const unsigned long long above = (lane == 63) ? 0ull : (hm >> (lane + 1)) << (lane + 1);
    const int end = above ? (__ffsll((long long)above) - 1) : 64;
    const unsigned long long upto = (end == 64) ? ~0ull : ((1ull << end) - 1ull);
    const int cnt = __popcll(am & upto & ~((1ull << lane) - 1ull));
    if (cnt) atomicAdd(&nacc[s], cnt);
  }
}

// General path: up to 8 x 8 vertices, intermediate polygons up to 16.  One wave per block,
// LDS: two [16][64] double2 ping-pong buffers + cutter [8][64] double2 = 40 KiB.
#define GEN_THREADS 64
#define GEN_CAP 16
template <int ORDER>
__global__ __launch_bounds__(GEN_THREADS) void k_clip_general(const int *defer_list, const int *defer_cnt,
                                                              const int *pair_src, int *pair_dst,
                                                              FgCells S, const double *mask, FgCells D,
                                                              double *tmp_area, double *tmp_clon, double *tmp_clat, int *nacc,
                                                              unsigned long long *stats, unsigned *err)
{
  __shared__ double2 sh_a[GEN_CAP][GEN_THREADS];
  __shared__ double2 sh_b[GEN_CAP][GEN_THREADS];
  __shared__ double2 sh_cut[G_MAXV][GEN_THREADS];
  d_load_trig_table();
  const int tid = threadIdx.x;
  const int ndefer = *defer_cnt;
  for (int q = blockIdx.x * GEN_THREADS + tid; q < ndefer; q += gridDim.x * GEN_THREADS) {
    const int p = defer_list[q];
    const int s = pair_src[p], d = pair_dst[p];
    const int n1 = S.nv[s], n2 = D.nv[d];
    const double *sv = S.verts + (size_t)s * 16, *dv = D.verts + (size_t)d * 16;
    const double lon_in_avg = S.lon_avg[s];
    double shift = 0.0;
    {
      double dx = D.lon_avg[d] - lon_in_avg;
      if (dx < -G_PI) shift = G_TPI; else if (dx > G_PI) shift = -G_TPI;
    }
    bool wrap = false;
    for (int k = 0; k < n1; k++) { double xv = sv[k]; if (xv > G_TPI || xv < 0.0) wrap = true; }
    for (int k = 0; k < n1; k++) {
      double xv = sv[k]; if (wrap) xv = d_pimod1(xv);
      sh_a[k][tid] = make_double2(xv, sv[8 + k]);
    }
    for (int k = 0; k < n2; k++) {
      double xv = dv[k]; if (shift != 0.0) xv += shift; if (wrap) xv = d_pimod1(xv);
      sh_cut[k][tid] = make_double2(xv, dv[8 + k]);
    }
    double2 (*cur)[GEN_THREADS] = sh_a;
    double2 (*nxt)[GEN_THREADS] = sh_b;
    int n_cur = n1;
    bool overflow = false, parallel = false;
    double2 e0 = sh_cut[n2 - 1][tid];
    for (int e = 0; e < n2 && n_cur > 0 && !overflow; e++) {
      double2 e1 = sh_cut[e][tid];
      const double x2_0 = e0.x, y2_0 = e0.y, x2_1 = e1.x, y2_1 = e1.y;
      double2 lastv = cur[n_cur - 1][tid];
      double x1_0 = lastv.x, y1_0 = lastv.y;
      int inside_last = d_inside_edge(x2_0, y2_0, x2_1, y2_1, x1_0, y1_0);
      int n_new = 0;
      for (int k = 0; k < n_cur; k++) {
        double2 cv = cur[k][tid];
        double x1_1 = cv.x, y1_1 = cv.y;
        int inside = d_inside_edge(x2_0, y2_0, x2_1, y2_1, x1_1, y1_1);
        if (inside != inside_last) {
          double dy1 = y1_1 - y1_0;
          double dy2 = y2_1 - y2_0;
          double dx1 = x1_1 - x1_0;
          double dx2 = x2_1 - x2_0;
          double ds1 = y1_0 * x1_1 - y1_1 * x1_0;
          double ds2 = y2_0 * x2_1 - y2_1 * x2_0;
          double determ = dy2 * dx1 - dy1 * dx2;
          if (fabs(determ) < 1.0e-30) parallel = true;
          if (n_new < GEN_CAP) nxt[n_new][tid] = make_double2((dx2 * ds1 - dx1 * ds2) / determ,
                                                              (dy2 * ds1 - dy1 * ds2) / determ);
          else overflow = true;
          n_new++;
        }
        if (inside) {
          if (n_new < GEN_CAP) nxt[n_new][tid] = make_double2(x1_1, y1_1);
          else overflow = true;
          n_new++;
        }
        x1_0 = x1_1; y1_0 = y1_1; inside_last = inside;
      }
      n_cur = n_new;
      double2 (*t)[GEN_THREADS] = cur; cur = nxt; nxt = t;
      e0 = e1;
    }
    if (overflow) { atomicOr(err, G_ERRBIT_OVERFLOW); pair_dst[p] = -1; continue; }
    if (parallel) atomicOr(err, G_ERRBIT_PARALLEL);
    ClipOut o; o.area = -1.0; o.clon = 0; o.clat = 0;
    if (n_cur > 0) {
      const double *px = (const double *)&cur[0][tid];
      d_finish_pair<ORDER, 2 * GEN_THREADS>(px, px + 1, n_cur, mask ? mask[s] : 1.0, S.area[s], D.area[d],
                                            lon_in_avg, &o, stats);
    }
    if (o.area >= 0) {
      tmp_area[p] = o.area;
      if (ORDER == 2) { tmp_clon[p] = o.clon; tmp_clat[p] = o.clat; }
      atomicAdd(&nacc[s], 1);
    } else {
      pair_dst[p] = -1;
      if (o.area == -2.0) atomicAdd(&stats[FG_STAT_BELOW], 1ull);
    }
  }
}

// ---------------------------------------------------------------------------------------
// compaction into canonical order
// ---------------------------------------------------------------------------------------
// Rank of every accepted pair of a heavy source cell among that cell's accepted pairs, by destination index (the canonical
// order inside a source cell).  k_scatter_xcells ranks a pair by comparing it with all the others of its cell, which is
// quadratic: fine for 5 pairs, but a source cell at a pole of the target grid holds thousands (great-circle search: ~3000 in
// each of ~600 cells, which was 0.4 ms).  Here a block marks the cell's destination indices in an LDS bitmap over their span
// and reads each rank off as a prefix population count: linear.  rank = -1 where the span does not fit (scatter falls back).
#define RANK_WORDS 2048      // 131072 destination indices: 91 rows of a 1440-column grid
#define RANK_MIN 64          // pairs of a heavy-list cell from which the bitmap pays (the list also holds cells with few pairs)
__global__ __launch_bounds__(256) void k_rank_heavy(const int *heavy_list, const int *heavy_cnt, const int *cand_off,
                                                    const int *pair_dst, int *pair_rank, int cap)
{
  __shared__ unsigned long long bits[RANK_WORDS];
  __shared__ int pref[RANK_WORDS];
  __shared__ int smin, smax;
  const int nheavy = *heavy_cnt;
  for (int h = blockIdx.x; h < nheavy; h += gridDim.x) {
    const int s = heavy_list[h];
    const int o = cand_off[s * CAND_G];
    int c = cand_off[(s + 1) * CAND_G] - o;
    if (o + c > cap) c = max(0, cap - o);                 // (an overflowing fast search is repeated anyway)
    if (c <= RANK_MIN) continue;                          // short lists are ranked in place by k_scatter_xcells (block-uniform)
    if (threadIdx.x == 0) { smin = 0x7fffffff; smax = -1; }
    __syncthreads();
    int lmin = 0x7fffffff, lmax = -1;
    for (int k = threadIdx.x; k < c; k += 256) { const int d = pair_dst[o + k]; if (d >= 0) { lmin = min(lmin, d); lmax = max(lmax, d); } }
    if (lmax >= 0) { atomicMin(&smin, lmin); atomicMax(&smax, lmax); }
    __syncthreads();
    const int dmin = smin, dmax = smax;
    const long span = (long)dmax - dmin + 1;
    if (dmax < 0 || span > (long)RANK_WORDS * 64) {
      for (int k = threadIdx.x; k < c; k += 256) pair_rank[o + k] = -1;
      __syncthreads();
      continue;
    }
    const int nw = (int)((span + 63) >> 6);
    for (int w = threadIdx.x; w < nw; w += 256) bits[w] = 0ull;
    __syncthreads();
    for (int k = threadIdx.x; k < c; k += 256) {
      const int d = pair_dst[o + k];
      if (d >= 0) atomicOr(&bits[(d - dmin) >> 6], 1ull << ((d - dmin) & 63));
    }
    __syncthreads();
    // exclusive prefix of the word populations: each thread owns a contiguous chunk of words
    const int per = (nw + 255) / 256, w0 = threadIdx.x * per, w1 = min(nw, w0 + per);
    int mine = 0;
    for (int w = w0; w < w1; w++) mine += __popcll(bits[w]);
    unsigned tot;
    const int before = (int)block_incl_scan((unsigned)mine, &tot) - mine;
    int run = before;
    for (int w = w0; w < w1; w++) { pref[w] = run; run += __popcll(bits[w]); }
    __syncthreads();
    for (int k = threadIdx.x; k < c; k += 256) {
      const int d = pair_dst[o + k];
      if (d >= 0) {
        const int w = (d - dmin) >> 6, b = (d - dmin) & 63;
        pair_rank[o + k] = pref[w] + __popcll(bits[w] & ((1ull << b) - 1ull));
      }
    }
    __syncthreads();
  }
}

template <int ORDER>
__global__ __launch_bounds__(256) void k_scatter_xcells(int npairs, const int *pair_src, const int *pair_dst,
                                                         const int *cand_off, const int *xoff,
                                                         const double *tmp_area, const double *tmp_clon, const double *tmp_clat,
                                                         int *x_src, int *x_dst, double *x_area, double *x_c1, double *x_c2,
                                                         int *row_cnt, int *x_rowpos, const unsigned long long *np_dev,
                                                         const int4 *stage, const int *pair_rank)
{
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int cap = npairs;                           // entries the pair arrays hold
  if (np_dev) { const unsigned long long nd = *np_dev; if (nd < (unsigned long long)npairs) npairs = (int)nd; }
  if (p >= npairs) return;
  const int d = pair_dst[p];                        // -1: rejected by the clip kernels
  if (d < 0) return;
  const int s = pair_src[p];
  int o = cand_off[s * CAND_G], c = cand_off[(s + 1) * CAND_G] - o;
  if (o + c > cap) c = cap - o;                       // a single-sync search that outgrew its buffers is repeated; stay inside them
  int rank = (c > RANK_MIN && stage[(size_t)s * CAND_G].x == -2) ? pair_rank[p] : -1;    // long lists: ranked by k_rank_heavy
  if (rank < 0) {
    rank = 0;
    for (int k = 0; k < c; k++)                       // destination index ascending == the reference's ij loop
      rank += ((unsigned)pair_dst[o + k] < (unsigned)d) ? 1 : 0;    // rejected entries are 0xffffffff
  }
  int pos = xoff[s] + rank;
  x_src[pos] = s; x_dst[pos] = d; x_area[pos] = tmp_area[p];
  // destination-row sizes for the CSR build (fg_plan_finalize), and this cell's slot in its row: the value-returning atomic
  // costs little here, where it overlaps the gathers, and saves the CSR fill pass its own (k_csr_fill_pos)
  x_rowpos[pos] = atomicAdd(&row_cnt[d], 1);
  if (ORDER == 2) { x_c1[pos] = tmp_clon[p]; x_c2[pos] = tmp_clat[p]; }
}

// ---------------------------------------------------------------------------------------
// order-2 centroid pass
// ---------------------------------------------------------------------------------------
// sums[0..2][nsrc] over this plan's exchange cells, in exchange-cell order (conserve_interp.c:216-221)
__global__ __launch_bounds__(256) void k_cell_sums(int nsrc, const int *xoff, const double *x_area,
                                                    const double *x_c1, const double *x_c2, double *sums)
{
  // The xcells of a block's 256 consecutive source cells are one contiguous
  // range: stage it through LDS with coalesced loads, then each thread adds
  // its own cell's entries in canonical order (same order, same sums, as the
  // direct loop kept below for ranges that do not fit).
  constexpr int CAP = 2048;
  __shared__ double sh[3][CAP];
  int b0 = blockIdx.x * blockDim.x;
  int b1 = min(b0 + (int)blockDim.x, nsrc);
  int e0 = xoff[b0], n = xoff[b1] - e0;
  int s = b0 + threadIdx.x;
  bool staged = n <= CAP;
  if (staged) {
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
      sh[0][k] = x_area[e0 + k]; sh[1][k] = x_c1[e0 + k]; sh[2][k] = x_c2[e0 + k];
    }
    __syncthreads();
  }
  if (s >= nsrc) return;
  double a = 0, l = 0, t = 0;
  int o = xoff[s], c = xoff[s + 1] - o;
  if (staged) {
    int q = o - e0;
    int k = 0;
    for (; k + 8 <= c; k += 8) {                     // reads ahead of the ordered additions, as in the direct loop below
      double va[8], vl[8], vt[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { va[u] = sh[0][q + k + u]; vl[u] = sh[1][q + k + u]; vt[u] = sh[2][q + k + u]; }
#pragma unroll
      for (int u = 0; u < 8; u++) { a += va[u]; l += vl[u]; t += vt[u]; }
    }
    for (; k < c; k++) { a += sh[0][q + k]; l += sh[1][q + k]; t += sh[2][q + k]; }
  } else {
    // ranges that do not fit hold the cells around a pole of the target grid, with hundreds of exchange cells each: their
    // serial chains set the duration of the whole launch, so the loads go out 16 entries ahead of the (ordered) additions
    int k = 0;
    for (; k + 16 <= c; k += 16) {
      double va[16], vl[16], vt[16];
#pragma unroll
      for (int u = 0; u < 16; u++) { va[u] = x_area[o + k + u]; vl[u] = x_c1[o + k + u]; vt[u] = x_c2[o + k + u]; }
#pragma unroll
      for (int u = 0; u < 16; u++) { a += va[u]; l += vl[u]; t += vt[u]; }
    }
    for (; k < c; k++) { a += x_area[o + k]; l += x_c1[o + k]; t += x_c2[o + k]; }
  }
  sums[s] = a; sums[nsrc + s] = l; sums[2 * (size_t)nsrc + s] = t;
}

// cen[0][s], cen[1][s] = centroid lon/lat of source cell s (conserve_interp.c:327-348)
__global__ __launch_bounds__(256) void k_centroids(int nsrc, FgCells S, const double *sums, double *cen)
{
  d_load_trig_table();
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsrc) return;
  double a = sums[s], cl = 0, ct = 0;
  if (a > 0) {
    double ca = S.area[s];
    if (fabs(a - ca) / ca < 1.e-3) {
      cl = sums[nsrc + s] / a;
      ct = sums[2 * (size_t)nsrc + s] / a;
    } else {
      double x[G_MAXV], y[G_MAXV];
      int n = S.nv[s];
      const double *vp = S.verts + (size_t)s * 16;
      for (int k = 0; k < G_MAXV; k++) { x[k] = vp[k]; y[k] = vp[8 + k]; }
      cl = d_poly_ctrlon<1>(x, y, n, S.lon_avg[s]) / ca;
      ct = d_poly_ctrlat<1>(x, y, n) / ca;
    }
  }
  cen[s] = cl; cen[nsrc + s] = ct;
}

// di = clon/area - cen_lon, dj = clat/area - cen_lat (conserve_interp.c:256-257,355-356)
__global__ __launch_bounds__(256) void k_distances(long nx, int nsrc, const int *x_src, const double *x_area,
                                                    const double *cen, double *x_c1, double *x_c2)
{
  long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nx) return;
  int s = x_src[n];
  double a = x_area[n];
  double di = x_c1[n] / a, dj = x_c2[n] / a;
  di -= cen[s]; dj -= cen[nsrc + s];
  x_c1[n] = di; x_c2[n] = dj;
}

// ---------------------------------------------------------------------------------------
// host-callable launchers
// ---------------------------------------------------------------------------------------
static inline int nblk(long n, int t) { return (int)((n + t - 1) / t); }

void fgd_cell_struct(const FgTile *tiles_dev, int ntiles, int ncells, FgCells c, unsigned *err, hipStream_t st)
{
  if (ncells > 0) k_cell_struct<<<nblk(ncells, 256), 256, 0, st>>>(tiles_dev, ntiles, ncells, c, err);
}

void fgd_bin_build(bool fill, int ncells, FgCells c, FgBins b, int *slot_cnt, const int *slot_start, FgBinEntry *entries, int cap,
                   hipStream_t st)
{
  if (ncells <= 0) return;
  if (fill) k_bin_build<true><<<nblk(ncells, 256), 256, 0, st>>>(ncells, c, b, slot_cnt, slot_start, entries, cap);
  else      k_bin_build<false><<<nblk(ncells, 256), 256, 0, st>>>(ncells, c, b, slot_cnt, slot_start, entries, cap);
}

void fgd_candidates(bool fill, int nsrc, FgCells S, const double *mask, FgBins b, const int *slot_start,
                    const FgBinEntry *entries, int *cand_cnt, const int *cand_off, int *pair_src, int *pair_dst,
                    int *heavy_list, int *heavy_cnt, int cap, int *stage, int ecap, hipStream_t st)
{
  if (nsrc <= 0) return;
  int hgrid = nblk(nsrc, 64); if (hgrid > 8192) hgrid = 8192;
  if (fill) {
    k_candidates<true><<<nblk((long)nsrc * CAND_G, 64), 64, 0, st>>>(nsrc, S, mask, b, slot_start, entries, cand_cnt, cand_off, pair_src, pair_dst, heavy_list, heavy_cnt, cap, (int4 *)stage, ecap);
    k_candidates_heavy<true><<<hgrid, 64, 0, st>>>(S, b, slot_start, entries, cand_cnt, cand_off, pair_src, pair_dst, heavy_list, heavy_cnt, cap, ecap);
  } else {
    k_candidates<false><<<nblk((long)nsrc * CAND_G, 64), 64, 0, st>>>(nsrc, S, mask, b, slot_start, entries, cand_cnt, cand_off, pair_src, pair_dst, heavy_list, heavy_cnt, cap, (int4 *)stage, ecap);
    k_candidates_heavy<false><<<hgrid, 64, 0, st>>>(S, b, slot_start, entries, cand_cnt, cand_off, pair_src, pair_dst, heavy_list, heavy_cnt, cap, ecap);
  }
}

void fgd_clip_quad(int order, int npairs, const int *pair_src, int *pair_dst, FgCells S, const double *mask, FgCells D,
                   double *tmp_area, double *tmp_clon, double *tmp_clat, int *nacc, int *defer_list, int *defer_cnt,
                   unsigned long long *stats, unsigned *err, const unsigned long long *np_dev, hipStream_t st)
{
  if (npairs <= 0) return;
  if (order == 2)
    k_clip_quad<2><<<nblk(npairs, CLIP_THREADS), CLIP_THREADS, 0, st>>>(npairs, pair_src, pair_dst, S, mask, D, tmp_area, tmp_clon, tmp_clat, nacc, defer_list, defer_cnt, stats, err, np_dev);
  else
    k_clip_quad<1><<<nblk(npairs, CLIP_THREADS), CLIP_THREADS, 0, st>>>(npairs, pair_src, pair_dst, S, mask, D, tmp_area, tmp_clon, tmp_clat, nacc, defer_list, defer_cnt, stats, err, np_dev);
}

void fgd_clip_general(int order, int npairs, const int *pair_src, int *pair_dst, FgCells S, const double *mask, FgCells D,
                      double *tmp_area, double *tmp_clon, double *tmp_clat, int *nacc, int *defer_list, int *defer_cnt,
                      unsigned long long *stats, unsigned *err, hipStream_t st)
{
  if (npairs <= 0) return;
  int grid = nblk(npairs, GEN_THREADS); if (grid > 1024) grid = 1024;
  if (order == 2)
    k_clip_general<2><<<grid, GEN_THREADS, 0, st>>>(defer_list, defer_cnt, pair_src, pair_dst, S, mask, D, tmp_area, tmp_clon, tmp_clat, nacc, stats, err);
  else
    k_clip_general<1><<<grid, GEN_THREADS, 0, st>>>(defer_list, defer_cnt, pair_src, pair_dst, S, mask, D, tmp_area, tmp_clon, tmp_clat, nacc, stats, err);
}

int fgd_cand_group(void) { return CAND_G; }

void fgd_scatter_xcells(int order, int npairs, const int *pair_src, const int *pair_dst, const int *cand_off,
                        const int *xoff, const double *tmp_area, const double *tmp_clon,
                        const double *tmp_clat, int *x_src, int *x_dst, double *x_area, double *x_c1, double *x_c2,
                        int *row_cnt, int *x_rowpos, const unsigned long long *np_dev, const int *heavy_list, const int *heavy_cnt,
                        const int *stage, int *pair_rank, hipStream_t st)
{
  if (npairs <= 0) return;
  k_rank_heavy<<<1024, 256, 0, st>>>(heavy_list, heavy_cnt, cand_off, pair_dst, pair_rank, npairs);
  if (order == 2) k_scatter_xcells<2><<<nblk(npairs, 256), 256, 0, st>>>(npairs, pair_src, pair_dst, cand_off, xoff, tmp_area, tmp_clon, tmp_clat, x_src, x_dst, x_area, x_c1, x_c2, row_cnt, x_rowpos, np_dev, (const int4 *)stage, pair_rank);
  else            k_scatter_xcells<1><<<nblk(npairs, 256), 256, 0, st>>>(npairs, pair_src, pair_dst, cand_off, xoff, tmp_area, tmp_clon, tmp_clat, x_src, x_dst, x_area, x_c1, x_c2, row_cnt, x_rowpos, np_dev, (const int4 *)stage, pair_rank);
}

void fgd_cell_sums(int nsrc, const int *xoff, const double *x_area, const double *x_c1,
                   const double *x_c2, double *sums, hipStream_t st)
{
  if (nsrc > 0) k_cell_sums<<<nblk(nsrc, 256), 256, 0, st>>>(nsrc, xoff, x_area, x_c1, x_c2, sums);
}

void fgd_centroids(int nsrc, FgCells S, const double *sums, double *cen, hipStream_t st)
{
  if (nsrc > 0) k_centroids<<<nblk(nsrc, 256), 256, 0, st>>>(nsrc, S, sums, cen);
}

void fgd_distances(long nx, int nsrc, const int *x_src, const double *x_area, const double *cen, double *x_c1,
                   double *x_c2, hipStream_t st)
{
  if (nx > 0) k_distances<<<nblk(nx, 256), 256, 0, st>>>(nx, nsrc, x_src, x_area, cen, x_c1, x_c2);
}
