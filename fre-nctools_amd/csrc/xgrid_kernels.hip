// xgrid_kernels.hip -- exchange-grid search kernels for gfx950 (MI355X).
//
// Pipeline for one destination tile against all source tiles:
//
//   k_cell_struct2     per cell (source tiles and destination tile in one launch): lat min/max, fix_lon, lon
//                      min/max/avg, <=8 vertices, area [get_grid_cell_struct semantics, create_xgrid.c:991-1016;
//                      get_grid_area :66-88]; destination cells are counted into their bins on the way
//   k_scan1            single-pass exclusive scan (decoupled look-back) of the bin counts / row counts
//   k_bin_fill         destination cells -> uniform (lat x lon mod 2pi) bins, one 48-byte record per
//                      cell stored in bin order (single insertion; wide cells in per-row lists); the same launch lists
//                      the source cells whose query is large (poles: huge longitude range)
//   k_candidates1      per source cell: scan the <= 2 contiguous record ranges per bin row, apply the reference's
//                      exact bounding-box rejects (create_xgrid.c:1055-1079) -> candidate pairs
//                      [get_upbound_nxcells_2dx2d semantics], appended to the pair list in ONE pass (a wave reserves
//                      its range with one atomic on a region counter); listed cells get a whole wave, same launch
//   k_clip_quad        one lane per candidate pair, quad x quad fast path: Sutherland-Hodgman
//                      clip (create_xgrid.c:1266-1341) with the polygon staged in LDS
//                      [vertex][lane], then area / centroid integrals and the 1e-6 area test
//   k_clip_general     same for pairs with pole-fixed cells (5..8 vertices) or fast-path overflow
//   k_compact(+_big)   compaction into the reference's canonical order (source cell ascending, destination cell index
//                      ascending): a block owns 256 consecutive source cells, counts their accepted pairs, takes its
//                      offset by look-back, ranks, writes the exchange cells coalesced, takes the destination-row slots
//                      and adds up the per-source-cell sums in exchange-cell order (conserve_interp.c:216-221)
//   k_centroids, k_distances     order-2 centroid pass (conserve_interp.c:319-358)
//
// No MFMA: this is FP64 VALU + irregular gather work.  Every floating-point operation uses the reference's expression
// trees, sin/cos included (geom.hip.h, sincos_glibc.h): lists, areas and centroid integrals are the reference's bits.
// Launched either for exact sizes or for fixed capacities with the true counts read from device memory (np_dev, cap).
#include <algorithm>
#include "xgrid_device.h"
#ifndef FG_EXP
#define FG_EXP 0      // timing experiments (scripts/exp_build.sh): 1 = no integrals, 2 = no clip loop
#endif
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
// single-pass exclusive scan (decoupled look-back) and the look-back itself
// ---------------------------------------------------------------------------------------
// A tile publishes its aggregate in one 64-bit word {flag:2, value:62} and walks back over its predecessors until it meets
// one whose inclusive prefix is known.  Tiles are numbered by a ticket taken when the block starts, so every predecessor a
// block waits for is already running: the wait ends whatever order the hardware dispatches blocks in.  The spin is bounded
// all the same (G_ERRBIT_LOOKBACK).  Value and flag share the word, so relaxed agent-scope accesses are all the protocol needs.
#define SCAN_THREADS 256
#define SCAN_ITEMS   8
#define SCAN_CHUNK   (SCAN_THREADS * SCAN_ITEMS)
#define LB_AGG   (1ull << 62)
#define LB_INC   (2ull << 62)
#define LB_MASK  ((1ull << 62) - 1ull)
#define LB_SPIN_LIMIT (1u << 22)

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

__device__ __forceinline__ unsigned long long lb_load(const unsigned long long *p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lb_store(unsigned long long *p, unsigned long long v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by all 64 lanes of ONE wave of the block owning `tile`; returns the sum of the aggregates of tiles 0..tile-1 and
// publishes this tile's inclusive prefix.  Each round inspects 64 predecessors at once.
__device__ inline unsigned long long d_lookback_wave(unsigned long long *status, int tile, unsigned long long agg, unsigned *err)
{
  const int lane = threadIdx.x & 63;
  if (tile == 0) { if (lane == 0) lb_store(&status[0], LB_INC | (agg & LB_MASK)); return 0ull; }
  if (lane == 0) lb_store(&status[tile], LB_AGG | (agg & LB_MASK));
  unsigned long long excl = 0ull;
  int base = tile - 1;
  unsigned spins = 0;
  for (;;) {
    const int t = base - lane;
    const unsigned long long v = (t >= 0) ? lb_load(&status[t]) : LB_INC;      // "tiles" before the first: prefix 0, known
    const unsigned flag = (unsigned)(v >> 62);
    const unsigned long long inc = __ballot(flag == 2u), zero = __ballot(flag == 0u);
    const int first_inc = inc ? (__ffsll((long long)inc) - 1) : 63;
    const unsigned long long need = (first_inc >= 63) ? ~0ull : ((2ull << first_inc) - 1ull);   // lanes 0..first_inc
    if (zero & need) {                                  // a predecessor this round depends on has not published yet
      if (++spins > LB_SPIN_LIMIT) { if (lane == 0) atomicOr(err, G_ERRBIT_LOOKBACK); break; }
      __builtin_amdgcn_s_sleep(1);
      continue;
    }
    unsigned long long val = ((need >> lane) & 1ull) ? (v & LB_MASK) : 0ull;
#pragma unroll
    for (int o = 32; o; o >>= 1) val += __shfl_xor(val, o);
    excl += val;
    if (inc) break;
    base -= 64;
  }
  if (lane == 0) lb_store(&status[tile], LB_INC | ((excl + agg) & LB_MASK));
  return excl;
}

// out[i] = exclusive prefix for i in [0, n] (n inputs, n+1 outputs; 32-bit offsets, 64-bit total for the host's checks)
__global__ __launch_bounds__(SCAN_THREADS) void k_scan1(const int *in, long n, int *out, unsigned long long *status, unsigned *ticket,
                                                        unsigned long long *total_out, unsigned *err, const unsigned long long *base_in)
{
  __shared__ unsigned tile[SCAN_CHUNK];
  __shared__ int sh_t;
  __shared__ unsigned long long sh_excl;
  // ticket == null: the grid is small enough to be resident all at once, so no tile can wait for one that has not started
  if (ticket) { if (threadIdx.x == 0) sh_t = (int)atomicAdd(ticket, 1u); __syncthreads(); }
  const int t = ticket ? sh_t : (int)blockIdx.x;
  const long base = (long)t * SCAN_CHUNK;
  // thread k owns items k*SCAN_ITEMS .. +SCAN_ITEMS-1 of the chunk (blocked arrangement via LDS, coalesced global accesses)
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    const int li = k * SCAN_THREADS + threadIdx.x;
    const long idx = base + li;
    tile[li] = (idx < n) ? (unsigned)in[idx] : 0u;
  }
  __syncthreads();
  unsigned loc[SCAN_ITEMS];
  unsigned s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) { loc[k] = tile[threadIdx.x * SCAN_ITEMS + k]; s += loc[k]; }
  unsigned tot;
  const unsigned inc = block_incl_scan(s, &tot);
  if (threadIdx.x < 64) {
    const unsigned long long e = d_lookback_wave(status, t, (unsigned long long)tot, err);
    if (threadIdx.x == 0) sh_excl = e + (base_in ? *base_in : 0ull);
  }
  __syncthreads();
  const unsigned long long excl = sh_excl;
  unsigned run = (unsigned)excl + inc - s;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) { tile[threadIdx.x * SCAN_ITEMS + k] = run; run += loc[k]; }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    const int li = k * SCAN_THREADS + threadIdx.x;
    const long idx = base + li;
    if (idx <= n) out[idx] = (int)tile[li];
  }
  if (threadIdx.x == 0 && n >= base && n < base + SCAN_CHUNK) *total_out = excl + tot;      // the tile holding out[n]
}

// two independent scans in one launch (blockIdx.y picks the job): the accept-count scan and the destination-row scan of a search
// depend on the same kernel and feed the same one -- a launch boundary less on a path that is launch bound for small bands
struct FgScanJob { const int *in; long n; int *out; unsigned long long *status; unsigned *ticket; unsigned long long *total; };
__global__ __launch_bounds__(SCAN_THREADS) void k_scan2(FgScanJob a, FgScanJob b, unsigned *err, int by_ticket)
{
  const FgScanJob j = blockIdx.y ? b : a;
  const long nt = (j.n + 1 + SCAN_CHUNK - 1) / SCAN_CHUNK;
  if ((long)blockIdx.x >= nt) return;
  __shared__ unsigned tile[SCAN_CHUNK];
  __shared__ int sh_t;
  __shared__ unsigned long long sh_excl;
  // by_ticket == 0: the whole grid is resident at once, no tile can wait for one that has not started
  if (by_ticket) { if (threadIdx.x == 0) sh_t = (int)atomicAdd(j.ticket, 1u); __syncthreads(); }
  const int t = by_ticket ? sh_t : (int)blockIdx.x;
  const long base = (long)t * SCAN_CHUNK;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    const int li = k * SCAN_THREADS + threadIdx.x;
    const long idx = base + li;
    tile[li] = (idx < j.n) ? (unsigned)j.in[idx] : 0u;
  }
  __syncthreads();
  unsigned loc[SCAN_ITEMS];
  unsigned s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) { loc[k] = tile[threadIdx.x * SCAN_ITEMS + k]; s += loc[k]; }
  unsigned tot;
  const unsigned inc = block_incl_scan(s, &tot);
  if (threadIdx.x < 64) {
    const unsigned long long e = d_lookback_wave(j.status, t, (unsigned long long)tot, err);
    if (threadIdx.x == 0) sh_excl = e;
  }
  __syncthreads();
  const unsigned long long excl = sh_excl;
  unsigned run = (unsigned)excl + inc - s;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) { tile[threadIdx.x * SCAN_ITEMS + k] = run; run += loc[k]; }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    const int li = k * SCAN_THREADS + threadIdx.x;
    const long idx = base + li;
    if (idx <= j.n) j.out[idx] = (int)tile[li];
  }
  if (threadIdx.x == 0 && j.n >= base && j.n < base + SCAN_CHUNK) *j.total = excl + tot;
}

long fgd_scan_tiles(long n) { if (n < 0) n = 0; return (n + 1 + SCAN_CHUNK - 1) / SCAN_CHUNK; }
void fgd_exclusive_scan2(const int *in_a, long n_a, int *out_a, unsigned long long *status_a, unsigned *ticket_a, unsigned long long *total_a,
                         const int *in_b, long n_b, int *out_b, unsigned long long *status_b, unsigned *ticket_b, unsigned long long *total_b,
                         unsigned *err, hipStream_t st)
{
  if (n_a < 0) n_a = 0;
  if (n_b < 0) n_b = 0;
  const long nt = std::max(fgd_scan_tiles(n_a), fgd_scan_tiles(n_b));
  k_scan2<<<dim3((unsigned)nt, 2), SCAN_THREADS, 0, st>>>(FgScanJob{in_a, n_a, out_a, status_a, ticket_a, total_a},
                                                         FgScanJob{in_b, n_b, out_b, status_b, ticket_b, total_b}, err, 2 * nt > 1024 ? 1 : 0);
}

void fgd_exclusive_scan1(const int *in, long n, int *out, unsigned long long *status, unsigned *ticket,
                         unsigned long long *total_dev, unsigned *err, hipStream_t st, const unsigned long long *base_dev)
{
  if (n < 0) n = 0;
  const int nt = (int)fgd_scan_tiles(n);
  k_scan1<<<nt, SCAN_THREADS, 0, st>>>(in, n, out, status, nt <= 1024 ? nullptr : ticket, total_dev, err, base_dev);
}

// ---------------------------------------------------------------------------------------
// per-cell records
// ---------------------------------------------------------------------------------------
// (d_ord_key / d_ord_val: xgrid_device.h)

// body shared by the two kernels below: record of cell s0 + threadIdx.x of `ncells` cells described by `tiles`; the 16 vertex
// doubles are staged so that the block stores its 256 records as one contiguous run (a lane writing its own 128-byte
// record makes sixteen 8-byte stores at a 128-byte stride).  Returns the vertex count (0: no record) and the box.
// cull != null: {max key of lat_max, max of ~key of lat_min} over the destination cells (this rank's band): a cell whose
// latitude range cannot meet the band -- the reference's strict test, create_xgrid.c:1055, would reject every pair -- gets
// nv = 0 and area 0 and nothing else; a block without a live cell skips its vertex records altogether.
__device__ __forceinline__ int d_cell_record(const FgTile *tiles, int ntiles, int ncells, const FgCells &c, unsigned *err, int s0,
                                             double *vtile, double *box /* lat_min, lat_max, lon_min, lon_max */, int *tile_of,
                                             const unsigned long long *cull = nullptr, double tlon = G_PI)
{
  const int s = s0 + threadIdx.x;
  double *row = vtile + threadIdx.x * 17;
#pragma unroll
  for (int k = 0; k < 16; k++) row[k] = 0.0;
  int nvert = 0;
  if (s < ncells) {
    int t = 0;
    while (t + 1 < ntiles && s >= tiles[t + 1].cell_off) t++;
    *tile_of = t;
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
    box[0] = lmin; box[1] = lmax;
    if (!(lmin >= -G_HPI - 1.e-6) || !(lmax <= G_HPI + 1.e-6)) atomicOr(err, G_ERRBIT_BADLAT);   // also catches NaN
    bool out_of_band = false;
    if (cull && cull[0]) {
      const double bmax = d_ord_val(cull[0]), bmin = d_ord_val(~cull[1]);
      out_of_band = (lmax <= bmin) || (lmin >= bmax);
    }
    int n = out_of_band ? 0 : d_fix_lon_quad_fast(x, y, tlon);
    if (n < 0) n = d_fix_lon(x, y, 4, tlon);             // cells with a pole vertex / a half-turn edge: the general routine
    if (out_of_band) { c.nv[s] = 0; c.area[s] = 0; }
    else if (n < 0 || n > G_MAXV) {
      atomicOr(err, G_ERRBIT_MAXV);
      c.nv[s] = 0; c.lon_min[s] = 0; c.lon_max[s] = 0; c.lon_avg[s] = 0; c.area[s] = 0;
    } else {
      double xmin = x[0], xmax = x[0], xs = 0;
      for (int k = 1; k < n; k++) { if (x[k] < xmin) xmin = x[k]; if (x[k] > xmax) xmax = x[k]; }
      for (int k = 0; k < n; k++) xs += x[k];
      xs /= n;
      c.lon_min[s] = xmin; c.lon_max[s] = xmax; c.lon_avg[s] = xs;
      box[2] = xmin; box[3] = xmax; box[4] = xs;
      c.nv[s] = n;
      nvert = n;
      for (int k = 0; k < G_MAXV; k++) {
        row[k] = (k < n) ? x[k] : 0.0;
        row[8 + k] = (k < n) ? y[k] : 0.0;
      }
      c.area[s] = d_poly_area<1>(x, y, n);
    }
  }
  if (!__syncthreads_or(nvert != 0) && cull) return 0;        // no live cell in this block: no vertex records to store
  const long cnt = (long)min(256, ncells - s0) * 16;
  double *out = c.verts + (size_t)s0 * 16;
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const long e = (long)q * 256 + threadIdx.x;
    if (e < cnt) out[e] = vtile[(e >> 4) * 17 + (e & 15)];
  }
  return nvert;
}

__global__ __launch_bounds__(256) void k_cell_struct(const FgTile *tiles, int ntiles, int ncells, FgCells c, unsigned *err)
{
  __shared__ double vtile[256 * 17];
  d_load_trig_table();
  double box[5]; int tl = 0;
  (void)d_cell_record(tiles, ntiles, ncells, c, err, blockIdx.x * 256, vtile, box, &tl);
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

#define BIN_COPIES 16      // columns a cell may cover and still live in the bins (one copy per column but the last)
#define BIN_ROW_COPIES 4   // rows, likewise
static_assert(BIN_COPIES >= 1 && BIN_COPIES <= 16, "a bin record keeps the copy number and copies - 1 in four bits each");

// Is this bin record the first copy of its cell inside the query's window of rows [ra, rb] x columns [c_start, c_start + count)?
// Columns: p = the record's place in the window taken round the ring of nblon columns, o = the copy's number, flen = copies of the
// cell.  A copy o >= 1 has its predecessor one column earlier -- inside the window unless p == 0; copy 0 is preceded by a later
// copy only if the cell's run of columns passes the window's start going round the ring (p + flen > nblon).  Rows do not wrap: a
// copy that is not its cell's first has its predecessor one row earlier, inside the window unless this is the window's first row.
// Single-copy records always pass.
__device__ __forceinline__ bool d_copy_first(int word, int c_start, int nblon, bool first_row)
{
  const int col = word & 0xffff, o = (word >> 16) & 15, flen = ((word >> 20) & 15) + 1;
  if (((word >> 24) & 1) && !first_row) return false;
  int p = col - c_start;
  if (p < 0) p += nblon;
  return p == 0 || (o == 0 && p + flen <= nblon);
}

// Count (FILL = false) or store (FILL = true) destination cell d in its bin / wide lists.  Wave-wide: every lane of the
// wave must call it (live = false for lanes without a cell).
template <bool FILL>
__device__ __forceinline__ void d_bin_insert(bool live, int d, double lat_min, double lat_max, double lon_min, double lon_max, double lon_avg,
                                             FgBins b, int *slot_cnt, const int *slot_start, FgBinEntry *entries, int cap)
{
  FgBinEntry E;
  int r0 = 0, r1 = 0; long long l0 = 0, l1 = 0;
  if (live) {
    E.lat_min = lat_min; E.lat_max = lat_max; E.lon_min = lon_min; E.lon_max = lon_max; E.lon_avg = lon_avg;
    E.d = d;
    d_cell_box(lat_min, lat_max, lon_min, lon_max, b, &r0, &r1, &l0, &l1);
    E.row0 = r0;
  }
  const int nbins = b.nblat * b.nblon;
  // A cell whose box covers rows r0..r1 and columns l0..l1 is stored at rows r0 .. r1 - 1 x columns l0 .. l1 - 1 (at least one of
  // each): with r1 - r0 <= 1 and l1 - l0 <= 1 that is the single record described above; cells that are "long" -- the rings of a
  // cubed-sphere tile around its pole (up to BIN_COPIES columns), its cells along the diagonals, which lie at 45 degrees to the bin rows
  // (up to BIN_ROW_COPIES + 1 rows) -- get one copy per row and column, and a query takes the FIRST copy its window meets (d_copy_first).  Before round 3
  // such cells went to the wide lists, which every source cell of the row scans whatever its longitude (0.25 deg -> C384 polar tile:
  // 21 k wide cells, 188 k source cells listed for the wave-per-cell path, 2.1 ms against 0.7 ms for an equatorial tile).
  const int maxcopies = (b.nblon <= 0xffff) ? min(BIN_COPIES, b.nblon) : 1, maxrows_c = (b.nblon <= 0xffff) ? BIN_ROW_COPIES : 1;
  const bool regular = live && r1 - r0 <= maxrows_c && l1 - l0 <= maxcopies;
  const int ccopies = regular ? max(1, (int)(l1 - l0)) : 0, rcopies = regular ? max(1, r1 - r0) : 0;
  const int ncopies = ccopies * rcopies;
  int maxc = ncopies;
#pragma unroll
  for (int o = 32; o; o >>= 1) maxc = max(maxc, __shfl_xor(maxc, o));
  for (int k = 0; k < maxc; k++) {
    const bool on = k < ncopies;
    const int kr = on ? k / ccopies : 0, kc = on ? k - kr * ccopies : 0;
    const int col = on ? d_colmod(l0 + kc, b.nblon) : 0;
    const int slot = on ? (r0 + kr) * b.nblon + col : -1;
    if (FILL) {
      const int pos = d_slot_position(slot_cnt, slot);
      if (on) {
        const int at = slot_start[slot] + pos;
        // bin records: column, column-copy number, column copies - 1, "not the first row copy" (wide records: first row)
        E.row0 = (col & 0xffff) | (kc << 16) | ((ccopies - 1) << 20) | ((kr > 0) << 24);
        if ((unsigned)at < (unsigned)cap) entries[at] = E;
      }
    } else if (on)
      atomicAdd(&slot_cnt[slot], 1);
  }
  E.row0 = r0;
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
    if (FILL && key >= 0) { const int at = slot_start[key] + pos; if ((unsigned)at < (unsigned)cap) entries[at] = E; }
  }
}

// Source tiles (blocks [0, nbS)) and the destination tile (the rest) in one launch.  Source blocks also store the index of
// each cell inside one level of the field array (order 1: no halo, index == s; order 2: 1-cell halo per tile,
// fregrid_util.c:2137-2145); destination blocks count their cells into the bins; block 0 stores the tile descriptors.
__global__ __launch_bounds__(256) void k_cell_struct2(FgTileSet ts, const FgTile *tiles_in, FgTile *tiles_out, int ntiles, int nsrc, int ndst,
                                                       int nbS, FgCells S, FgCells D, FgBins b, int *slot_cnt, int order, int *src_idx_f,
                                                       double *sums, unsigned *err, unsigned long long *band_keys, int cull, double dst_tlon)
{
  __shared__ double vtile[256 * 17];
  __shared__ FgTile sh_tiles[FG_TILESET_MAX];
  if (ts.n && (int)threadIdx.x < ts.n) sh_tiles[threadIdx.x] = ts.t[threadIdx.x];
  const bool isD = (int)blockIdx.x >= nbS;
  if (cull && !isD && band_keys[0]) {
    // a rank of a banded job: most source blocks lie wholly outside the band -- find that out from the corner latitudes alone
    // and leave before the trig table is loaded (nv = 0 is all the later kernels look at)
    __syncthreads();
    const FgTile *tl0 = ts.n ? sh_tiles : tiles_in;
    const int s = blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    int idx_f = s;                                        // the cell's place in the field arrays: the sweep's record merge reads it for EVERY cell
    if (s < nsrc) {
      int t = 0;
      while (t + 1 < ntiles && s >= tl0[t + 1].cell_off) t++;
      const int loc = s - tl0[t].cell_off, i = loc % tl0[t].nx, j = loc / tl0[t].nx, nxp = tl0[t].nx + 1;
      if (order == 2) {
        int foff = 0;
        for (int m = 0; m < t; m++) foff += (tl0[m].nx + 2) * (tl0[m].ny + 2);
        idx_f = foff + (j + 1) * (tl0[t].nx + 2) + i + 1;
      }
      const int n0 = j * nxp + i;
      const double y0 = tl0[t].lat[n0], y1 = tl0[t].lat[n0 + 1], y2 = tl0[t].lat[n0 + nxp + 1], y3 = tl0[t].lat[n0 + nxp];
      const double lmin = fmin(fmin(y0, y1), fmin(y2, y3)), lmax = fmax(fmax(y0, y1), fmax(y2, y3));
      const double bmax = d_ord_val(band_keys[0]), bmin = d_ord_val(~band_keys[1]);
      keep = !((lmax <= bmin) || (lmin >= bmax)) || !(lmin == lmin);        // NaN goes on to the full path (it reports the error)
    }
    if (!__syncthreads_or(keep)) {
      if (s < nsrc) {
        S.nv[s] = 0; S.area[s] = 0;
        if (src_idx_f) src_idx_f[s] = idx_f;
        if (sums) { sums[s] = 0.0; sums[nsrc + s] = 0.0; sums[2 * (size_t)nsrc + s] = 0.0; }
      }
      if (ts.n && blockIdx.x == 0 && (int)threadIdx.x < ts.n) tiles_out[threadIdx.x] = sh_tiles[threadIdx.x];
      return;
    }
  }
  d_load_trig_table();                                   // (barrier inside)
  const FgTile *tiles = ts.n ? sh_tiles : tiles_in;
  if (ts.n && blockIdx.x == 0 && (int)threadIdx.x < ts.n) tiles_out[threadIdx.x] = sh_tiles[threadIdx.x];
  double box[5] = {0, 0, 0, 0, 0};
  int tl = 0;
  if (!isD) {
    const int s0 = blockIdx.x * 256, s = s0 + threadIdx.x;
    (void)d_cell_record(tiles, ntiles, nsrc, S, err, s0, vtile, box, &tl, cull ? band_keys : nullptr);
    if (s < nsrc && sums) { sums[s] = 0.0; sums[nsrc + s] = 0.0; sums[2 * (size_t)nsrc + s] = 0.0; }   // k_compact visits cells with exchange cells only
    if (s < nsrc && src_idx_f) {
      if (order != 2) src_idx_f[s] = s;
      else {
        int foff = 0;
        for (int t = 0; t < tl; t++) foff += (tiles[t].nx + 2) * (tiles[t].ny + 2);
        const int loc = s - tiles[tl].cell_off, i = loc % tiles[tl].nx, j = loc / tiles[tl].nx;
        src_idx_f[s] = foff + (j + 1) * (tiles[tl].nx + 2) + i + 1;
      }
    }
  } else {
    const int d0 = ((int)blockIdx.x - nbS) * 256, d = d0 + threadIdx.x;
    const int nv = d_cell_record(tiles + ntiles, 1, ndst, D, err, d0, vtile, box, &tl, nullptr, dst_tlon);
    d_bin_insert<false>(d < ndst && nv != 0, d, box[0], box[1], box[2], box[3], box[4], b, slot_cnt, nullptr, nullptr, 0);
    if (band_keys && cull != 2) {                       // latitude range of the destination cells, for the culling of a later launch
      // one pair of atomics per BLOCK, and only when it would change the value: thousands of same-address atomics serialise
      // at ~12 ns each (a wave-level version of this cost the kernel 0.37 ms)
      __shared__ unsigned long long sh_k[2][4];
      unsigned long long kmax = (d < ndst && nv != 0) ? d_ord_key(box[1]) : 0ull, kmin = (d < ndst && nv != 0) ? ~d_ord_key(box[0]) : 0ull;
#pragma unroll
      for (int o = 32; o; o >>= 1) { kmax = max(kmax, __shfl_xor(kmax, o)); kmin = max(kmin, __shfl_xor(kmin, o)); }
      if ((threadIdx.x & 63) == 0) { sh_k[0][threadIdx.x >> 6] = kmax; sh_k[1][threadIdx.x >> 6] = kmin; }
      __syncthreads();
      if (threadIdx.x == 0) {
        kmax = max(max(sh_k[0][0], sh_k[0][1]), max(sh_k[0][2], sh_k[0][3]));
        kmin = max(max(sh_k[1][0], sh_k[1][1]), max(sh_k[1][2], sh_k[1][3]));
        if (kmax > __hip_atomic_load(&band_keys[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&band_keys[0], kmax);
        if (kmin > __hip_atomic_load(&band_keys[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&band_keys[1], kmin);
      }
    }
  }
}

// latitude range of the destination grid from its corner array alone (= the range over its cells' lat_min / lat_max): lets the
// culling search keep the ONE fused record launch instead of a destination launch followed by a source launch
__global__ __launch_bounds__(256) void k_band_keys(const double *lat, long n, unsigned long long *band_keys)
{
  __shared__ unsigned long long sh_k[2][4];
  unsigned long long kmax = 0ull, kmin = 0ull;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const unsigned long long k = d_ord_key(lat[i]);
    kmax = max(kmax, k); kmin = max(kmin, ~k);
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) { kmax = max(kmax, __shfl_xor(kmax, o)); kmin = max(kmin, __shfl_xor(kmin, o)); }
  if ((threadIdx.x & 63) == 0) { sh_k[0][threadIdx.x >> 6] = kmax; sh_k[1][threadIdx.x >> 6] = kmin; }
  __syncthreads();
  if (threadIdx.x == 0) {
    kmax = max(max(sh_k[0][0], sh_k[0][1]), max(sh_k[0][2], sh_k[0][3]));
    kmin = max(max(sh_k[1][0], sh_k[1][1]), max(sh_k[1][2], sh_k[1][3]));
    atomicMax(&band_keys[0], kmax);
    atomicMax(&band_keys[1], kmin);
  }
}
void fgd_band_keys(const double *lat, long n, unsigned long long *band_keys, hipStream_t st)
{
  if (n > 0) k_band_keys<<<(int)min(64L, (n + 2047) / 2048), 256, 0, st>>>(lat, n, band_keys);
}

// stand-alone bin count for searches whose cell records come from another kernel (great circle)
__global__ __launch_bounds__(256) void k_bin_count(int ncells, FgCells c, FgBins b, int *slot_cnt)
{
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = d < ncells && c.nv[d] != 0;      // (no early return: the slot positions are a wave-wide operation)
  double v[5] = {0, 0, 0, 0, 0};
  if (live) { v[0] = c.lat_min[d]; v[1] = c.lat_max[d]; v[2] = c.lon_min[d]; v[3] = c.lon_max[d]; v[4] = c.lon_avg[d]; }
  d_bin_insert<false>(live, d, v[0], v[1], v[2], v[3], v[4], b, slot_cnt, nullptr, nullptr, 0);
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

// size of the query: bins scanned (a bin holds ~2 cells: bins are 1.25x the mean cell) plus the entries of
// the wide lists of its rows (two table loads).  Only used to route huge queries to the wave-per-cell path.
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
                              // longest scans set the duration of the four-lanes-per-cell path)
#define CAND_G 4          // lanes per source cell in the candidate scan (one bin row each)
#define CP_SMALL 128      // pairs per cell up to which the lanes of k_compact rank by comparison; cells with more are "big"
                          // (32 until round 3: every cell of a coarse -> fine remap -- ~100 pairs at C48 -> 0.25 deg -- then took a whole
                          // block of the big-cell role: compaction 0.31 ms of a 0.83 ms search; 128: 0.17 of 0.67; 256: 0.14, but the
                          // 256 halo lanes of every lane-per-pair block then cost the similar-resolution case 2 %)
#define CAND_CHUNK 1     // consecutive waves (of 16 cells) that append to the same region: 1 = round robin, the best balance
                          // (16 was tried for the locality of the clip: no faster, and the great-circle search, whose pairs
                          // pile up around the poles, then overflowed a region and had to be repeated)
#define HEAVY_BLOCKS 2048 // waves serving the listed cells, appended to the grid of the four-lanes-per-cell blocks

__device__ __forceinline__ bool d_src_active(const FgCells &S, const double *mask, int s)
{
  bool active = S.nv[s] > 0;
  if (active && mask) active = mask[s] > 0.5;          // MASK_THRESH, create_xgrid.c:1030
  return active;
}

// Bin fill (blocks [0, nbD): destination cells -> their bin records) and, in the same launch, the list of the source cells
// whose query touches more than HEAVY_ENTRIES table entries (blocks [nbD, ...): one atomic per wave that has any).
__global__ __launch_bounds__(256) void k_bin_fill(int ndst, int nbD, FgCells D, FgBins b, int *slot_fill, const int *slot_start,
                                                   FgBinEntry *entries, int cap, int nsrc, FgCells S, const double *mask,
                                                   int *heavy_list, int *heavy_cnt)
{
  if ((int)blockIdx.x < nbD) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    const bool live = d < ndst && D.nv[d] != 0;
    double v[5] = {0, 0, 0, 0, 0};
    if (live) { v[0] = D.lat_min[d]; v[1] = D.lat_max[d]; v[2] = D.lon_min[d]; v[3] = D.lon_max[d]; v[4] = D.lon_avg[d]; }
    d_bin_insert<true>(live, d, v[0], v[1], v[2], v[3], v[4], b, slot_fill, slot_start, entries, cap);
    return;
  }
  const int s = ((int)blockIdx.x - nbD) * 256 + threadIdx.x, lane = threadIdx.x & 63;
  bool heavy = false;
  if (s < nsrc && d_src_active(S, mask, s)) {
    const SrcQuery q = d_src_query(S.lat_min[s], S.lat_max[s], S.lon_min[s], S.lon_max[s], b);
    heavy = d_query_size(q, b, slot_start) > HEAVY_ENTRIES;
  }
  const unsigned long long m = __ballot(heavy);
  if (m) {
    int base = 0;
    if (lane == 0) base = atomicAdd(heavy_cnt, __popcll(m));
    base = __shfl(base, 0);
    if (heavy) heavy_list[base + __popcll(m & ((1ull << lane) - 1ull))] = s;
  }
}

// One wave per listed source cell (pole caps of the source grid: their longitude range covers hundreds of bins); lanes
// stride the contiguous entry ranges, ballot + popcount compacts.  FILL = false counts, FILL = true writes at most `limit`
// pairs from position wbase on.
template <bool FILL>
__device__ __forceinline__ int d_heavy_scan(const SrcQuery &q, FgBins b, const int *slot_start, const FgBinEntry *entries, int ecap,
                                            double lat_in_min, double lat_in_max, double lon_in_min, double lon_in_max, double lon_in_avg,
                                            int s, int *pair_src, int *pair_dst, int wbase, int limit)
{
  const int lane = threadIdx.x & 63;
  const int nbins = b.nblat * b.nblon;
  int cnt = 0;
  const int nrows_reg = q.rb - q.ra + 1, nrows_wide = q.r1 - q.r0 + 1;
  for (int it = 0; it < 2 * nrows_reg + nrows_wide; it++) {
    int e0, e1, wide_row = -1;
    const bool first_row = it < 2;
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
        if (wide_row >= 0 ? wide_row != max(q.r0, E.row0) : !d_copy_first(E.row0, q.c_start, b.nblon, first_row)) pass = false;
      }
      unsigned long long m = __ballot(pass);
      if (FILL && pass) {
        const int k = cnt + __popcll(m & ((1ull << lane) - 1ull));
        if (k < limit) { pair_src[wbase + k] = s; pair_dst[wbase + k] = dcell; }
      }
      cnt += __popcll(m);
    }
  }
  return cnt;
}

// One lane's share of a source cell's query: every CAND_G-th bin row, the <= 2 contiguous record ranges of each row, then the
// wide lists of its rows.  Records are loaded four at a time ahead of their tests: the loop is a chain of L2 round trips
// otherwise (one per record).  FILL = false counts and keeps the first four hits in ids[]; FILL = true writes the pairs
// (same order) at psrc/pdst[wloc ...] while inside the region.
template <bool FILL>
__device__ __forceinline__ int d_lane_scan(const SrcQuery &q, int sub, FgBins b, const int *slot_start, const FgBinEntry *entries, int ecap,
                                           double lat_in_min, double lat_in_max, double lon_in_min, double lon_in_max, double lon_in_avg,
                                           int *ids, int s, int *psrc, int *pdst, unsigned wloc, unsigned regcap)
{
  const int nbins = b.nblat * b.nblon;
  int cnt = 0;
  auto visit = [&](const FgBinEntry &E, bool ok) {
    if (!ok || !d_box_pass(E, lat_in_min, lat_in_max, lon_in_min, lon_in_max, lon_in_avg)) return;
    if (FILL) { if (wloc + cnt < regcap) { psrc[wloc + cnt] = s; pdst[wloc + cnt] = E.d; } }
    else { ids[0] = cnt == 0 ? E.d : ids[0]; ids[1] = cnt == 1 ? E.d : ids[1]; ids[2] = cnt == 2 ? E.d : ids[2]; ids[3] = cnt == 3 ? E.d : ids[3]; }
    cnt++;
  };
  auto range = [&](int e0, int e1, int wide_row, bool first_row) {
    for (int eb = e0; eb < e1; eb += 4) {
      FgBinEntry E[4];
#pragma unroll
      for (int k = 0; k < 4; k++) E[k] = entries[min(eb + k, e1 - 1)];
#pragma unroll
      for (int k = 0; k < 4; k++)            // a wide cell sits in every row it spans, a long one in every column but its last
        visit(E[k], eb + k < e1 && (wide_row < 0 ? d_copy_first(E[k].row0, q.c_start, b.nblon, first_row) : wide_row == max(q.r0, E[k].row0)));
    }
  };
  for (int r = q.ra + sub; r <= q.rb; r += CAND_G) {
    const int base = r * b.nblon;
    const int a0 = slot_start[base + q.c_start], a1 = slot_start[base + q.c_start + q.n0];
    const int b0 = q.n1 ? slot_start[base] : 0, b1 = q.n1 ? slot_start[base + q.n1] : 0;
    range(a0, min(a1, ecap), -1, r == q.ra);             // a search may have outgrown its record buffer (it is then repeated)
    range(b0, min(b1, ecap), -1, r == q.ra);
  }
  for (int r = q.r0 + sub; r <= q.r1; r += CAND_G) range(slot_start[nbins + r], min(slot_start[nbins + r + 1], ecap), r, false);
  return cnt;
}

// Blocks [0, nbR): CAND_G lanes per source cell, each scanning every CAND_G-th bin row of the cell's query.  A lane keeps
// the first four destination cells it finds (the common case: 5.2 pairs per source cell over four lanes); the wave then adds
// up its lanes' counts, reserves that many entries of its region of the pair list with ONE atomic, and every lane writes its
// pairs -- from registers, or, with more than four, by scanning its rows again (same order).  The pairs of a source cell
// are contiguous: pair_beg / pair_cnt.  Blocks [nbR, nbR + HEAVY_BLOCKS): a wave per listed cell (count, reserve, fill).
// (one wave per block: a block gives its slots back when its slowest wave is done, and the scan lengths vary a lot --
// measured 256 / 128 / 64 threads: 0.240 / 0.230 / 0.224 ms for the old count pass)
// The listed cells come FIRST in the grid: their waves run long and should start with the others, not after them.
__global__ __launch_bounds__(64) void k_candidates1(int c0, int c1, int H, FgCells S, const double *mask, FgBins b, const int *slot_start,
                                                     const FgBinEntry *entries, int ecap, FgPairSpace ps, int *pair_beg, int *pair_cnt,
                                                     const int *heavy_list, const int *heavy_cnt, int *big_list, int *big_cnt)
{
  const int lane = threadIdx.x;
  if ((int)blockIdx.x < H) {
    const int nheavy = *heavy_cnt;
    for (int h = blockIdx.x; h < nheavy; h += H) {
      const int s = heavy_list[h];
      if (s < c0 || s >= c1) continue;                    // another chunk's
      const double lat_in_min = S.lat_min[s], lat_in_max = S.lat_max[s];
      const double lon_in_min = S.lon_min[s], lon_in_max = S.lon_max[s], lon_in_avg = S.lon_avg[s];
      const SrcQuery q = d_src_query(lat_in_min, lat_in_max, lon_in_min, lon_in_max, b);
      const int cnt = d_heavy_scan<false>(q, b, slot_start, entries, ecap, lat_in_min, lat_in_max, lon_in_min, lon_in_max, lon_in_avg,
                                          s, nullptr, nullptr, 0, 0);
      // region by a hash of the cell number -- not by list position (the list order varies from run to run), and not s % nreg
      // (the cells around a pole share a few residues when the tile width is a multiple of nreg: 20 regions took all their pairs)
      const int r = (int)(((unsigned)s * 2654435761u >> 12) % (unsigned)ps.nreg);
      unsigned base = 0;
      if (lane == 0 && cnt) base = atomicAdd(&ps.fill[r * FG_FILL_STRIDE], (unsigned)cnt);
      base = __shfl(base, 0);
      const int loc0 = (int)min(base, (unsigned)ps.regcap), n_ok = min(cnt, ps.regcap - loc0);
      const int wbase = r * ps.regcap + loc0;
      if (n_ok > 0)
        (void)d_heavy_scan<true>(q, b, slot_start, entries, ecap, lat_in_min, lat_in_max, lon_in_min, lon_in_max, lon_in_avg,
                                 s, ps.src, ps.dst, wbase, n_ok);
      if (lane == 0) { pair_beg[s] = wbase; pair_cnt[s] = n_ok; if (n_ok > CP_SMALL) big_list[atomicAdd(big_cnt, 1)] = s; }
    }
    return;
  }
  const int bR = (int)blockIdx.x - H;                  // block among the four-lanes-per-cell blocks
  const long t = (long)bR * 64 + lane;
  const int s = c0 + (int)(t / CAND_G), sub = (int)(t % CAND_G);
  const int nsrc = c1;
  int cnt = 0;
  int ids[4] = {-1, -1, -1, -1};
  bool heavy = false;
  double lat_in_min = 0, lat_in_max = 0, lon_in_min = 0, lon_in_max = 0, lon_in_avg = 0;
  SrcQuery q{};
  // (Issuing the loads of each link of the chain -- cell box, slot offsets, first records of every range -- together, ahead of
  // the branches that may not need them, was measured: 223 us against 165.  The kernel is bound by the number of 48-byte record
  // requests its lanes make, not by their latency, and speculation adds requests.  Staging the UNION of the 16 cells' bin records
  // of a wave in LDS with coalesced loads and testing every cell against all of them there: 195 us -- 2.4x fewer record loads, but
  // 16 x 80 box tests per wave instead of 16 x 12.)
  if (s < nsrc && d_src_active(S, mask, s)) {
    lat_in_min = S.lat_min[s]; lat_in_max = S.lat_max[s];
    lon_in_min = S.lon_min[s]; lon_in_max = S.lon_max[s]; lon_in_avg = S.lon_avg[s];
    q = d_src_query(lat_in_min, lat_in_max, lon_in_min, lon_in_max, b);
    heavy = d_query_size(q, b, slot_start) > HEAVY_ENTRIES;       // listed by k_bin_fill; a whole wave writes its pairs
    if (!heavy)
      cnt = d_lane_scan<false>(q, sub, b, slot_start, entries, ecap, lat_in_min, lat_in_max, lon_in_min, lon_in_max, lon_in_avg,
                               ids, s, nullptr, nullptr, 0u, 0u);
  }
  // the wave's range in its region of the pair list
  const unsigned incl = wave_incl_scan((unsigned)cnt, lane);
  const unsigned total = __shfl(incl, 63);
  const unsigned excl = incl - (unsigned)cnt;
  // region by a hash of the wave's number: plain round robin leaves whole residue classes to the dense cells around a pole when
  // the tile width is a multiple of the region count (the great-circle search then overflowed a region and was repeated)
  const int r = (int)((((unsigned)(bR / CAND_CHUNK)) * 2654435761u >> 12) % (unsigned)ps.nreg);
  unsigned base = 0;
  if (lane == 0 && total) base = atomicAdd(&ps.fill[r * FG_FILL_STRIDE], total);
  base = __shfl(base, 0);
  // per source cell: first pair and number of pairs (entries beyond the region are dropped; the search is then repeated)
  int c4 = cnt;
  c4 += __shfl_xor(c4, 1); c4 += __shfl_xor(c4, 2);
  const unsigned excl0 = __shfl(excl, lane & ~(CAND_G - 1));
  if (sub == 0 && s < nsrc && !heavy) {
    const unsigned first = base + excl0;
    const int loc0 = (int)min(first, (unsigned)ps.regcap);
    pair_beg[s] = r * ps.regcap + loc0;
    pair_cnt[s] = min(c4, ps.regcap - loc0);
  }
  {                                                     // cells for the big-cell blocks of the compaction (rare here)
    const bool bigc = sub == 0 && s < nsrc && !heavy && min(c4, ps.regcap - (int)min(base + excl0, (unsigned)ps.regcap)) > CP_SMALL;
    const unsigned long long bm = __ballot(bigc);
    if (bm) {
      int q = 0;
      if (lane == 0) q = atomicAdd(big_cnt, __popcll(bm));
      q = __shfl(q, 0);
      if (bigc) big_list[q + __popcll(bm & ((1ull << lane) - 1ull))] = s;
    }
  }
  if (cnt == 0) return;
  const unsigned wloc = base + excl;                    // this lane's first entry within the region
  int *psrc = ps.src + (size_t)r * ps.regcap, *pdst = ps.dst + (size_t)r * ps.regcap;
  if (cnt <= 4) {
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (k < cnt && wloc + k < (unsigned)ps.regcap) { psrc[wloc + k] = s; pdst[wloc + k] = ids[k]; }
    return;
  }
  (void)d_lane_scan<true>(q, sub, b, slot_start, entries, ecap, lat_in_min, lat_in_max, lon_in_min, lon_in_max, lon_in_avg,
                          ids, s, psrc, pdst, wloc, (unsigned)ps.regcap);
}


// ---------------------------------------------------------------------------------------
// rectilinear destination grid
// ---------------------------------------------------------------------------------------
// lon_out depends on the column only and lat_out on the row only, bit for bit: what get_output_grid_by_size
// (fregrid_util.c:588-654) makes for every --nlon/--nlat target, global or regional, whole or a rank's band.  Then
//   * fix_lon's result for a cell depends on its column only (no lone pole vertex, no |dlon| = pi edge: checked), so a cell is
//     four values of a per-column record + two values of the latitude axis -- no 128-byte vertex record, no per-cell boxes;
//   * the reference's latitude reject (create_xgrid.c:1055) depends on the row only and its longitude reject (:1062-1079) on the
//     column only: the candidates of a source cell are (rows found on the latitude axis) x (columns that pass the exact
//     longitude test), found by index arithmetic -- no bins, no 48-byte record scans -- and come out in ascending destination
//     index, the reference's order.
// k_rect_tables VERIFIES the property on the device (every corner against its axis value, bitwise) in the same stream; every
// later kernel of the rectilinear path leaves at once when the check failed, and the host repeats the search with the generic
// path (plan.hip).  Results are the generic path's, bit for bit (tests/test_gpu_rect.py, scripts/legacy_fuzz.py).
// (RECT_COLW = 8 doubles per column record, xgrid_device.h: x'[0..3] after fix_lon (SW, SE, NE, NW), lon_min, lon_max, lon_avg, width)
#define RECT_HDR 8            // header of the tables: lon[0], nx / (lon[nx] - lon[0]), lat[0], ny / (lat[ny] - lat[0])
#define RECT_EPS 1.e-9
#define RECT_HEAVY 48         // rows x window columns above which a source cell's candidates are made by a whole wave

__global__ __launch_bounds__(256) void k_rect_tables(const double *lon, const double *lat, int nx, int ny, double *hdr, double *lat_ax,
                                                      double *lon_ax, double *col, double *row, unsigned *bad, unsigned *err, double dst_tlon)
{
  d_load_trig_table();                                   // (barrier inside)
  const long np = (long)(nx + 1) * (ny + 1);
  const long gid = (long)blockIdx.x * 256 + threadIdx.x, gsz = (long)gridDim.x * 256;
  bool b = false;
  for (long e = gid; e < np; e += gsz) {
    const long j = e / (nx + 1); const int i = (int)(e - j * (nx + 1));
    if (__double_as_longlong(lon[e]) != __double_as_longlong(lon[i]) ||
        __double_as_longlong(lat[e]) != __double_as_longlong(lat[j * (nx + 1)])) b = true;
  }
  if (gid <= ny) {
    const double y = lat[gid * (nx + 1)];
    lat_ax[gid] = y;
    if (!(y >= -G_HPI - 1.e-6) || !(y <= G_HPI + 1.e-6)) atomicOr(err, G_ERRBIT_BADLAT);       // (also NaN)
    if (gid < ny && !(lat[(gid + 1) * (nx + 1)] > y)) b = true;                                 // strictly ascending rows
    if (gid > 0 && gid < ny && d_is_pole(y)) b = true;                                          // a pole only as the outer edge of an outer row
    // what d_poly_area evaluates on the cells of row gid (d_rect_area below): the sines of its two flat edges (lat1 == lat2: the
    // mid latitude 0.5 * (y + y) is y) and of its two meridian edges (mid latitude 0.5 * (yb + y), half height 0.5 * (yb - y))
    double *r = row + (size_t)gid * 4;
    r[0] = d_sin_lat(0.5 * (y + y));
    if (gid < ny) {
      const double yb = lat[(gid + 1) * (nx + 1)];
      const bool flat = fabs(yb - y) < G_SMALL;
      const double dy = 0.5 * (yb - y);
      r[1] = d_sin_lat(0.5 * (yb + y)); r[2] = flat ? 1.0 : d_sin_lat(dy) / dy; r[3] = flat ? 1.0 : 0.0;
    } else { r[1] = 0.0; r[2] = 1.0; r[3] = 1.0; }
  }
  if (gid <= nx) lon_ax[gid] = lon[gid];                                                        // (a copy: the caller's array may go away)
  if (gid < nx) {
    const double x0 = lon[gid], x1 = lon[gid + 1];
    if (!(x1 > x0) || !(x1 - x0 < G_PI - 1.e-6)) b = true;                                      // ascending columns narrower than pi
    double x[G_FIXCAP], y[G_FIXCAP];
    x[0] = x0; x[1] = x1; x[2] = x1; x[3] = x0;
    y[0] = 0.0; y[1] = 0.0; y[2] = 0.1; y[3] = 0.1;
    const int n = d_fix_lon(x, y, 4, dst_tlon);
    double *c = col + (size_t)gid * RECT_COLW;
    if (n != 4) { b = true; for (int k = 0; k < RECT_COLW; k++) c[k] = 0.0; }
    else {
      double xmin = x[0], xmax = x[0], xs = 0;                                                  // as in d_cell_record
      for (int k = 1; k < 4; k++) { if (x[k] < xmin) xmin = x[k]; if (x[k] > xmax) xmax = x[k]; }
      for (int k = 0; k < 4; k++) xs += x[k];
      xs /= 4;
      c[0] = x[0]; c[1] = x[1]; c[2] = x[2]; c[3] = x[3]; c[4] = xmin; c[5] = xmax; c[6] = xs; c[7] = x1 - x0;
    }
  }
  if (gid == 0) {
    if (!(lon[nx] - lon[0] <= G_TPI + 1.e-9)) b = true;                                         // one turn at most
    hdr[0] = lon[0]; hdr[1] = nx / (lon[nx] - lon[0]); hdr[2] = lat[0]; hdr[3] = ny / (lat[(long)ny * (nx + 1)] - lat[0]);
  }
  if (__ballot(b) && (threadIdx.x & 63) == 0) atomicOr(bad, 1u);
}

// smallest k in [0, n] with ax[k] > v (STRICT) / ax[k] >= v (!STRICT); ax ascending.  (x0, inv): the uniform-axis guess that
// is right for the regular grids this path mostly sees (two loads); a binary search otherwise.
template <bool STRICT>
__device__ __forceinline__ int d_axis_first(const double *ax, int n, double v, double x0, double inv)
{
  const double t = fmin(fmax(floor((v - x0) * inv), -1.0), (double)(n - 1));
  const int g = (int)t;
  auto above = [&](double a) { return STRICT ? a > v : a >= v; };
  const bool lo_ok = (g < 0) || !above(ax[g]);
  const bool hi_ok = (g + 1 >= n) || above(ax[g + 1]);
  if (lo_ok && hi_ok) return g + 1;
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (above(ax[mid])) hi = mid; else lo = mid + 1; }
  return lo;
}

struct RectQuery { int j0, j1; int nw; int wa[3], wb[3]; };   // rows [j0, j1]; nw column windows [wa, wb], disjoint, ascending

__device__ __forceinline__ RectQuery d_rect_query(const FgRect &R, double lat_in_min, double lat_in_max, double lon_in_min, double lon_in_max)
{
  RectQuery q;
  const double lon0 = R.hdr[0], inv_lon = R.hdr[1], lat0 = R.hdr[2], inv_lat = R.hdr[3];
  // rows that pass create_xgrid.c:1055: lat[j+1] > lat_in_min && lat[j] < lat_in_max
  q.j0 = max(d_axis_first<true>(R.lat_ax, R.ny + 1, lat_in_min, lat0, inv_lat) - 1, 0);
  q.j1 = min(d_axis_first<false>(R.lat_ax, R.ny + 1, lat_in_max, lat0, inv_lat) - 1, R.ny - 1);
  q.nw = 0;
  if (lon_in_max - lon_in_min > G_PI) { q.nw = 1; q.wa[0] = 0; q.wb[0] = R.nx - 1; return q; }   // pole caps: every column is tested
  // a column passes only if its raw interval, moved by a whole number of turns, meets (lon_in_min, lon_in_max): fix_lon moves the
  // cell by 0 / +-2pi, the reference's test by another 0 / +-2pi.  The windows for different turns are disjoint (columns are
  // narrower than pi, the source range is at most pi, the axis spans one turn at most); the exact test decides inside them.
  const double axl = R.lon_ax[0], axh = R.lon_ax[R.nx];
#pragma unroll
  for (int t = 2; t >= -2; t--) {                      // descending shift = ascending columns
    const double T = t * G_TPI;
    const double lo = lon_in_min - RECT_EPS - T, hi = lon_in_max + RECT_EPS - T;
    if (!(axh > lo) || !(axl < hi)) continue;
    const int a = max(d_axis_first<true>(R.lon_ax, R.nx + 1, lo, lon0, inv_lon) - 1, 0);
    const int b = min(d_axis_first<false>(R.lon_ax, R.nx + 1, hi, lon0, inv_lon) - 1, R.nx - 1);
    if (a <= b && q.nw < 3) { q.wa[q.nw] = a; q.wb[q.nw] = b; q.nw++; }
  }
  return q;
}
__device__ __forceinline__ int d_rect_ncols(const RectQuery &q)
{
  int n = 0;
  for (int w = 0; w < q.nw; w++) n += q.wb[w] - q.wa[w] + 1;
  return n;
}
// "Heavy" = the candidates are made by a whole wave instead of one lane.  Decided from the cell's box and the MEAN axis spacings
// alone (no axis search), by the record kernel that lists such cells and by the candidate kernel that skips them: the two must
// agree, nothing else depends on the estimate (the lane path handles any count, only slowly).
__device__ __forceinline__ bool d_rect_heavy(const FgRect &R, double lat_in_min, double lat_in_max, double lon_in_min, double lon_in_max)
{
  const double nr = (lat_in_max - lat_in_min) * R.hdr[3] + 2.0;
  const double nc = (lon_in_max - lon_in_min > G_PI) ? (double)R.nx : (lon_in_max - lon_in_min) * R.hdr[1] + 2.0;
  return nr * nc > (double)RECT_HEAVY;
}
// column k of the query's concatenated windows
__device__ __forceinline__ int d_rect_col(const RectQuery &q, int k)
{
  int i = q.wa[0] + k;
  const int l0 = q.wb[0] - q.wa[0] + 1;
  if (q.nw > 1 && k >= l0) { i = q.wa[1] + (k - l0); const int l1 = q.wb[1] - q.wa[1] + 1; if (q.nw > 2 && k - l0 >= l1) i = q.wa[2] + (k - l0 - l1); }
  return i;
}
// the reference's longitude reject for column record c (create_xgrid.c:1062-1079)
__device__ __forceinline__ bool d_rect_col_pass(const double *c, double lon_in_min, double lon_in_max, double lon_in_avg)
{
  double lon_out_min = c[4], lon_out_max = c[5];
  const double dx = c[6] - lon_in_avg;
  if (dx < -G_PI)     { lon_out_min += G_TPI; lon_out_max += G_TPI; }
  else if (dx > G_PI) { lon_out_min -= G_TPI; lon_out_max -= G_TPI; }
  return !(lon_out_min >= lon_in_max || lon_out_max <= lon_in_min);
}

// poly_area (mosaic_util.c:417-459, d_poly_area) of the cell in column record c and row record r: the same terms in the same
// order -- edge 0 flat at the row's lower latitude, edge 1 a meridian (lat1 = upper, lat2 = lower), edge 2 flat at the upper
// latitude, edge 3 the other meridian (lat1 = lower, lat2 = upper: dy and sin(dy) change sign together, the mid latitude is the
// same sum) -- with the four sines and the quotient taken from the row table instead of evaluated per cell.
__device__ __forceinline__ double d_rect_area(const double *c, const double *r)
{
  const double s_lo = r[0], s_hi = r[4], s_mid = r[1], dat = r[2];
  const bool flat_m = r[3] != 0.0;
  double area = 0.0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    double dx = c[(k + 1) & 3] - c[k];
    if (dx > G_PI)  dx = dx - 2.0 * G_PI;
    if (dx < -G_PI) dx = dx + 2.0 * G_PI;
    if (fabs(dx + G_PI) < G_SMALL || fabs(dx - G_PI) < G_SMALL) { area += G_PI; continue; }
    if (k == 0) area -= dx * s_lo;
    else if (k == 2) area -= dx * s_hi;
    else if (flat_m) area -= dx * s_mid;
    else area -= dx * s_mid * dat;
  }
  if (area < 0) return -area * G_RADIUS * G_RADIUS;
  return area * G_RADIUS * G_RADIUS;
}

// Rectilinear twin of k_cell_struct2: source blocks make the full records (and list the cells whose candidates a whole wave
// will make), destination blocks store the cell AREA only.
__global__ __launch_bounds__(256) void k_cell_struct2r(FgTileSet ts, const FgTile *tiles_in, FgTile *tiles_out, int ntiles, int nsrc, int ndst,
                                                        int nbS, FgCells S, double *area_out, FgRect R, const double *mask, int order, int *src_idx_f,
                                                        double *sums, unsigned *err, unsigned long long *band_keys, int cull,
                                                        int *heavy_list, int *heavy_cnt)
{
  __shared__ double vtile[256 * 17];
  __shared__ FgTile sh_tiles[FG_TILESET_MAX];
  if (*R.bad) return;
  if (ts.n && (int)threadIdx.x < ts.n) sh_tiles[threadIdx.x] = ts.t[threadIdx.x];
  const bool isD = (int)blockIdx.x >= nbS;
  // the band of a culling search: the latitude axis gives it (rows ascend) -- no reduction over the corner array, no launch for it
  __shared__ unsigned long long sh_band[2];
  if (cull) {
    if (threadIdx.x == 0) { sh_band[0] = d_ord_key(R.lat_ax[R.ny]); sh_band[1] = ~d_ord_key(R.lat_ax[0]); }
    __syncthreads();
    band_keys = sh_band;
  }
  if (cull && !isD && band_keys[0]) {                      // (as in k_cell_struct2: blocks wholly outside the band leave early)
    __syncthreads();
    const FgTile *tl0 = ts.n ? sh_tiles : tiles_in;
    const int s = blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    int idx_f = s;
    if (s < nsrc) {
      int t = 0;
      while (t + 1 < ntiles && s >= tl0[t + 1].cell_off) t++;
      const int loc = s - tl0[t].cell_off, i = loc % tl0[t].nx, j = loc / tl0[t].nx, nxp = tl0[t].nx + 1;
      if (order == 2) {
        int foff = 0;
        for (int m = 0; m < t; m++) foff += (tl0[m].nx + 2) * (tl0[m].ny + 2);
        idx_f = foff + (j + 1) * (tl0[t].nx + 2) + i + 1;
      }
      const int n0 = j * nxp + i;
      const double y0 = tl0[t].lat[n0], y1 = tl0[t].lat[n0 + 1], y2 = tl0[t].lat[n0 + nxp + 1], y3 = tl0[t].lat[n0 + nxp];
      const double lmin = fmin(fmin(y0, y1), fmin(y2, y3)), lmax = fmax(fmax(y0, y1), fmax(y2, y3));
      const double bmax = d_ord_val(band_keys[0]), bmin = d_ord_val(~band_keys[1]);
      keep = !((lmax <= bmin) || (lmin >= bmax)) || !(lmin == lmin);
    }
    if (!__syncthreads_or(keep)) {
      if (s < nsrc) {
        S.nv[s] = 0; S.area[s] = 0;
        if (src_idx_f) src_idx_f[s] = idx_f;
        if (sums) { sums[s] = 0.0; sums[nsrc + s] = 0.0; sums[2 * (size_t)nsrc + s] = 0.0; }
      }
      if (ts.n && blockIdx.x == 0 && (int)threadIdx.x < ts.n) tiles_out[threadIdx.x] = sh_tiles[threadIdx.x];
      return;
    }
  }
  d_load_trig_table();                                   // (barrier inside)
  const FgTile *tiles = ts.n ? sh_tiles : tiles_in;
  if (ts.n && blockIdx.x == 0 && (int)threadIdx.x < ts.n) tiles_out[threadIdx.x] = sh_tiles[threadIdx.x];
  if (!isD) {
    double box[5] = {0, 0, 0, 0, 0};
    int tl = 0;
    const int s0 = blockIdx.x * 256, s = s0 + threadIdx.x, lane = threadIdx.x & 63;
    const int nv = d_cell_record(tiles, ntiles, nsrc, S, err, s0, vtile, box, &tl, cull ? band_keys : nullptr);
    if (s < nsrc && sums) { sums[s] = 0.0; sums[nsrc + s] = 0.0; sums[2 * (size_t)nsrc + s] = 0.0; }
    if (s < nsrc && src_idx_f) {
      if (order != 2) src_idx_f[s] = s;
      else {
        int foff = 0;
        for (int t = 0; t < tl; t++) foff += (tiles[t].nx + 2) * (tiles[t].ny + 2);
        const int loc = s - tiles[tl].cell_off, i = loc % tiles[tl].nx, j = loc / tiles[tl].nx;
        src_idx_f[s] = foff + (j + 1) * (tiles[tl].nx + 2) + i + 1;
      }
    }
    bool heavy = false;
    if (s < nsrc && nv > 0 && (!mask || mask[s] > 0.5)) heavy = d_rect_heavy(R, box[0], box[1], box[2], box[3]);
    const unsigned long long m = __ballot(heavy);
    if (m) {
      int base = 0;
      if (lane == 0) base = atomicAdd(heavy_cnt, __popcll(m));
      base = __shfl(base, 0);
      if (heavy) heavy_list[base + __popcll(m & ((1ull << lane) - 1ull))] = s;
    }
  } else {
    const int d = ((int)blockIdx.x - nbS) * 256 + threadIdx.x;
    if (d < ndst) {
      const int j = d / R.nx, i = d - j * R.nx;
      area_out[d] = d_rect_area(R.col + (size_t)i * RECT_COLW, R.row + (size_t)j * 4);
    }
  }
}

__global__ __launch_bounds__(256) void k_polylist_records(FgPolyList P, FgCells S, int *src_idx_f, double *sums, FgRect R, int rect,
                                                           int *heavy_list, int *heavy_cnt, unsigned *err)
{
  if (rect && *R.bad) return;
  const int s = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  bool heavy = false;
  if (s < P.npoly) {
    const int n = P.n[s];
    double *v = S.verts + (size_t)s * 16;
    if (n < 3 || n > G_MAXV) {
      if (n > G_MAXV) atomicOr(err, G_ERRBIT_MAXV);
      S.nv[s] = 0; S.area[s] = 0; S.lat_min[s] = 0; S.lat_max[s] = 0; S.lon_min[s] = 0; S.lon_max[s] = 0; S.lon_avg[s] = 0;
      for (int k = 0; k < 16; k++) v[k] = 0.0;
    } else {
      double xmin = 0, xmax = 0, ymin = 0, ymax = 0;
      for (int k = 0; k < G_MAXV; k++) {
        const double x = (k < n) ? P.lon[(size_t)s * G_MAXV + k] : 0.0, y = (k < n) ? P.lat[(size_t)s * G_MAXV + k] : 0.0;
        v[k] = x; v[8 + k] = y;
        if (k == 0) { xmin = xmax = x; ymin = ymax = y; }
        else if (k < n) { xmin = fmin(xmin, x); xmax = fmax(xmax, x); ymin = fmin(ymin, y); ymax = fmax(ymax, y); }
      }
      if (!(ymin >= -G_HPI - 1.e-6) || !(ymax <= G_HPI + 1.e-6)) atomicOr(err, G_ERRBIT_BADLAT);
      S.nv[s] = n; S.area[s] = P.area[s]; S.lon_avg[s] = P.lon_avg[s];
      S.lat_min[s] = ymin; S.lat_max[s] = ymax; S.lon_min[s] = xmin; S.lon_max[s] = xmax;
      if (rect) heavy = d_rect_heavy(R, ymin, ymax, xmin, xmax);
    }
    if (src_idx_f) src_idx_f[s] = s;
    if (sums) { sums[s] = 0.0; sums[P.npoly + s] = 0.0; sums[2 * (size_t)P.npoly + s] = 0.0; }
  }
  if (rect) {
    const unsigned long long m = __ballot(heavy);
    if (m) {
      int base = 0;
      if (lane == 0) base = atomicAdd(heavy_cnt, __popcll(m));
      base = __shfl(base, 0);
      if (heavy) heavy_list[base + __popcll(m & ((1ull << lane) - 1ull))] = s;
    }
  }
}

// the full destination records of a rectilinear plan, for fg_plan_get_cell_struct (tests): what k_cell_struct2 would have stored
__global__ __launch_bounds__(256) void k_rect_materialize(int ndst, FgRect R, FgCells D)
{
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= ndst) return;
  const int j = d / R.nx, i = d - j * R.nx;
  const double *c = R.col + (size_t)i * RECT_COLW;
  const double ya = R.lat_ax[j], yb = R.lat_ax[j + 1];
  D.lat_min[d] = ya; D.lat_max[d] = yb; D.lon_min[d] = c[4]; D.lon_max[d] = c[5]; D.lon_avg[d] = c[6]; D.nv[d] = 4;
  double *v = D.verts + (size_t)d * 16;
  for (int k = 0; k < 16; k++) v[k] = 0.0;
  v[0] = c[0]; v[1] = c[1]; v[2] = c[2]; v[3] = c[3];
  v[8] = ya; v[9] = ya; v[10] = yb; v[11] = yb;
}

// Candidates on a rectilinear destination grid.  Blocks [0, H): a wave per listed source cell (pole caps: every column);
// blocks [H, ...): one lane per source cell.  Pairs of a cell are contiguous and in ascending destination index.
__global__ __launch_bounds__(64) void k_candidates_rect(int nsrc, int H, FgCells S, const double *mask, FgRect R, FgPairSpace ps,
                                                         int *pair_beg, int *pair_cnt, const int *heavy_list, const int *heavy_cnt,
                                                         int *big_list, int *big_cnt)
{
  if (*R.bad) return;
  const int lane = threadIdx.x;
  if ((int)blockIdx.x < H) {
    const int nheavy = *heavy_cnt;
    for (int h = blockIdx.x; h < nheavy; h += H) {
      const int s = heavy_list[h];
      const double lon_in_min = S.lon_min[s], lon_in_max = S.lon_max[s], lon_in_avg = S.lon_avg[s];
      const RectQuery q = d_rect_query(R, S.lat_min[s], S.lat_max[s], lon_in_min, lon_in_max);
      const int nwc = d_rect_ncols(q), nr = q.j1 - q.j0 + 1;
      // pass 1: the columns that pass (the same set for every row of the query); pass 2 writes row after row -- the flags are
      // recomputed per 64-column chunk instead of listed (a list in LDS would cost every block of the launch its occupancy)
      int nc = 0;
      for (int kb = 0; kb < nwc; kb += 64) {
        const int k = kb + lane;
        const bool pass = k < nwc && d_rect_col_pass(R.col + (size_t)d_rect_col(q, k) * RECT_COLW, lon_in_min, lon_in_max, lon_in_avg);
        nc += __popcll(__ballot(pass));
      }
      const long cnt_l = (long)nr * nc;
      const int cnt = (int)min(cnt_l, 0x7fffffffL);
      const int r = (int)(((unsigned)s * 2654435761u >> 12) % (unsigned)ps.nreg);
      unsigned base = 0;
      if (lane == 0 && cnt) base = atomicAdd(&ps.fill[r * FG_FILL_STRIDE], (unsigned)cnt);
      base = __shfl(base, 0);
      const int loc0 = (int)min(base, (unsigned)ps.regcap), n_ok = min(cnt, ps.regcap - loc0);
      const int wbase = r * ps.regcap + loc0;
      int cbase = 0;
      for (int kb = 0; kb < nwc && n_ok > 0; kb += 64) {
        const int k = kb + lane;
        int i = 0; bool pass = false;
        if (k < nwc) { i = d_rect_col(q, k); pass = d_rect_col_pass(R.col + (size_t)i * RECT_COLW, lon_in_min, lon_in_max, lon_in_avg); }
        const unsigned long long m = __ballot(pass);
        const int at = cbase + __popcll(m & ((1ull << lane) - 1ull));
        if (pass)
          for (int jr = 0; jr < nr; jr++) {
            const long kk = (long)jr * nc + at;
            if (kk < n_ok) { ps.src[wbase + kk] = s; ps.dst[wbase + kk] = (q.j0 + jr) * R.nx + i; }
          }
        cbase += __popcll(m);
      }
      if (lane == 0) { pair_beg[s] = wbase; pair_cnt[s] = n_ok; if (n_ok > CP_SMALL) big_list[atomicAdd(big_cnt, 1)] = s; }
    }
    return;
  }
  const int bR = (int)blockIdx.x - H;
  const int s = bR * 64 + lane;
  int cnt = 0, nr = 0, nwc = 0;
  unsigned long long cmask = 0ull;
  RectQuery q{};
  bool heavy = false;
  double lon_in_min = 0, lon_in_max = 0, lon_in_avg = 0;
  if (s < nsrc && d_src_active(S, mask, s)) {
    const double lat_in_min = S.lat_min[s], lat_in_max = S.lat_max[s];
    lon_in_min = S.lon_min[s]; lon_in_max = S.lon_max[s]; lon_in_avg = S.lon_avg[s];
    heavy = d_rect_heavy(R, lat_in_min, lat_in_max, lon_in_min, lon_in_max);   // listed by k_cell_struct2r; a whole wave writes its pairs
    if (!heavy) {
      q = d_rect_query(R, lat_in_min, lat_in_max, lon_in_min, lon_in_max);
      nr = max(q.j1 - q.j0 + 1, 0);
      nwc = d_rect_ncols(q);
      int nc = 0;
      if (nr > 0) {
        // the columns that pass, as a bit mask when the windows are narrow enough (the usual case), else only counted here and
        // re-tested row by row below
        for (int k = 0; k < nwc; k++)
          if (d_rect_col_pass(R.col + (size_t)d_rect_col(q, k) * RECT_COLW, lon_in_min, lon_in_max, lon_in_avg)) { nc++; if (k < 64) cmask |= 1ull << k; }
      }
      cnt = (int)min((long)nr * nc, 0x3fffffffL);
    }
  }
  const unsigned incl = wave_incl_scan((unsigned)cnt, lane);
  const unsigned total = __shfl(incl, 63);
  const unsigned excl = incl - (unsigned)cnt;
  const int r = (int)((((unsigned)bR) * 2654435761u >> 12) % (unsigned)ps.nreg);
  unsigned base = 0;
  if (lane == 0 && total) base = atomicAdd(&ps.fill[r * FG_FILL_STRIDE], total);
  base = __shfl(base, 0);
  int n_ok = 0;
  const unsigned first = base + excl;
  if (s < nsrc && !heavy) {
    const int loc0 = (int)min(first, (unsigned)ps.regcap);
    n_ok = min(cnt, ps.regcap - loc0);
    pair_beg[s] = r * ps.regcap + loc0;
    pair_cnt[s] = n_ok;
  }
  {
    const bool bigc = n_ok > CP_SMALL;
    const unsigned long long bm = __ballot(bigc);
    if (bm) {
      int qq = 0;
      if (lane == 0) qq = atomicAdd(big_cnt, __popcll(bm));
      qq = __shfl(qq, 0);
      if (bigc) big_list[qq + __popcll(bm & ((1ull << lane) - 1ull))] = s;
    }
  }
  if (n_ok <= 0) return;
  int *psrc = ps.src + (size_t)r * ps.regcap + first, *pdst = ps.dst + (size_t)r * ps.regcap + first;
  int w = 0;
  for (int j = q.j0; j <= q.j1 && w < n_ok; j++) {
    if (nwc <= 64) {
      unsigned long long m = cmask;
      while (m && w < n_ok) {
        const int k = __ffsll((long long)m) - 1; m &= m - 1ull;
        psrc[w] = s; pdst[w] = j * R.nx + d_rect_col(q, k);
        w++;
      }
    } else
      for (int k = 0; k < nwc && w < n_ok; k++) {
        const int i = d_rect_col(q, k);
        if (d_rect_col_pass(R.col + (size_t)i * RECT_COLW, lon_in_min, lon_in_max, lon_in_avg)) { psrc[w] = s; pdst[w] = j * R.nx + i; w++; }
      }
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

#ifndef CLIP_THREADS
#define CLIP_THREADS 256
#endif
#ifndef CLIP_SLOTS
#define CLIP_SLOTS 8        // vertices a polygon of the quad kernel may have (more: general kernel)
#endif
#ifndef CLIP_WAVES
#define CLIP_WAVES 4        // waves per SIMD the quad kernel is compiled for
#endif
#ifndef CLIP_PAD_LDS
#define CLIP_PAD_LDS 0      // timing experiment: extra LDS bytes per clip block (occupancy sensitivity)
#endif

// Quad x quad fast path, first half: the Sutherland-Hodgman pass.  LDS: polygon [8][T] double2 (32 KiB per 256 lanes); the
// cutting quad lives in registers.  Returns -1 if the pair must go to the general kernel (more than 4 vertices on a side, or
// more than 8 in an intermediate polygon); otherwise the vertex count of the clipped polygon left in column tid (0: empty).
// RECT: the destination cell is four values of its column record and two of the latitude axis (FgRect) -- D holds areas only.
template <bool RECT, int T>
__device__ __forceinline__ int d_clip_quad_sh(double2 (*sh_poly)[T], const int tid, const int s, const int d,
                                              const FgCells &S, const FgCells &D, const FgRect &R, unsigned *err)
{
  // RECT: the cutting cell is always a quad, so source cells of up to 8 vertices (the pole-fixed ones) fit this kernel too --
  // the general kernel then only sees the rare pair whose intermediate polygon outgrows 8 vertices
  constexpr int NV1 = RECT ? CLIP_SLOTS : 4;
  const int n1 = S.nv[s], n2 = RECT ? 4 : D.nv[d];
  if (n1 > NV1 || n2 > 4) return -1;

  const double *sv = S.verts + (size_t)s * 16;
  const double lon_in_avg = S.lon_avg[s];
  double x1[NV1], y1[NV1], x2[4], y2[4];
  double lon_out_avg;
  if (RECT) {
    const int j = d / R.nx, i = d - j * R.nx;
    const double *c = R.col + (size_t)i * RECT_COLW;
    const double ya = R.lat_ax[j], yb = R.lat_ax[j + 1];
    x2[0] = c[0]; x2[1] = c[1]; x2[2] = c[2]; x2[3] = c[3]; lon_out_avg = c[6];
    y2[0] = ya; y2[1] = ya; y2[2] = yb; y2[3] = yb;
  } else {
    const double *dv = D.verts + (size_t)d * 16;
#pragma unroll
    for (int k = 0; k < 4; k++) { x2[k] = dv[k]; y2[k] = dv[8 + k]; }
    lon_out_avg = D.lon_avg[d];
  }
  double shift = 0.0;
  {
    double dx = lon_out_avg - lon_in_avg;            // create_xgrid.c:1064-1074
    if (dx < -G_PI) shift = G_TPI; else if (dx > G_PI) shift = -G_TPI;
  }
  bool wrap = false;
#pragma unroll
  for (int k = 0; k < NV1; k++) {
    if (k < 4 || k < n1) { x1[k] = sv[k]; y1[k] = sv[8 + k]; } else { x1[k] = 0.0; y1[k] = 0.0; }
    if (k < n1 && (x1[k] > G_TPI || x1[k] < 0.0)) wrap = true;  // create_xgrid.c:1282
  }
#pragma unroll
  for (int k = 0; k < 4; k++) if (shift != 0.0) x2[k] += shift;
  if (wrap) {                                          // :1290
#pragma unroll
    for (int k = 0; k < NV1; k++) x1[k] = d_pimod1(x1[k]);
#pragma unroll
    for (int k = 0; k < 4; k++) x2[k] = d_pimod1(x2[k]);
  }
#pragma unroll
  for (int k = 0; k < NV1; k++) if (k < 4 || k < n1) sh_poly[k][tid] = make_double2(x1[k], y1[k]);
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
  // Sutherland-Hodgman against one cutting edge at a time, restructured so that the expensive part -- the intersection (two
  // FP64 divisions) -- is not inside the per-vertex loop: a convex polygon crosses the line of a cutting edge at most twice, so
  // (1) the inside flags of all vertices are collected into bit masks (cheap, every lane busy), (2) the at most two crossings
  // are computed by code that runs twice per cutting edge instead of once per vertex slot with a third of the lanes active
  // (PMC, profiles/r03_summary.md: the old per-vertex loop kept 32.6 of 64 lanes busy), (3) the new polygon is written at
  // positions counted from the masks.  Same intersection formulas on the same operands, same output order as the reference's
  // loop (for every k: the crossing of edge (k-1, k) if there is one, then vertex k if it is inside, create_xgrid.c:1301-1333).
  // A polygon with more than two crossings of one line (non-convex source cell) goes to the general kernel.
  int n_cur = n1;
  bool overflow = false, parallel = false;
  double2 e0 = cut(n2 - 1);
#if FG_EXP == 2
  for (int e = 0; e < 0; e++) {
#else
  for (int e = 0; e < n2 && n_cur > 0; e++) {
#endif
    double2 e1 = cut(e);
    const double x2_0 = e0.x, y2_0 = e0.y, x2_1 = e1.x, y2_1 = e1.y;
    double2 c[CLIP_SLOTS];
    unsigned inmask = 0u;
#pragma unroll
    for (int k = 0; k < CLIP_SLOTS; k++)
      if (k < n_cur) {
        c[k] = sh_poly[k][tid];
        inmask |= (unsigned)d_inside_edge(x2_0, y2_0, x2_1, y2_1, c[k].x, c[k].y) << k;
      }
    const unsigned full = (1u << n_cur) - 1u;
    const unsigned prevmask = ((inmask << 1) | (inmask >> (n_cur - 1))) & full;     // bit k: vertex k-1 (cyclic) is inside
    const unsigned tmask = inmask ^ prevmask;                                        // bit k: edge (k-1, k) crosses the line
    const int ncross = __popc(tmask);
    if (ncross > 2 || ncross + __popc(inmask) > CLIP_SLOTS) { overflow = true; break; }
    if (tmask) {
      const int ka = __ffs((int)tmask) - 1, kb = 31 - __clz((int)tmask);
      double2 I[2];
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int k = j ? kb : ka;
        const double2 p0 = sh_poly[k ? k - 1 : n_cur - 1][tid], p1 = sh_poly[k][tid];
        const double x1_0 = p0.x, y1_0 = p0.y, x1_1 = p1.x, y1_1 = p1.y;
        double dy1 = y1_1 - y1_0;
        double dy2 = y2_1 - y2_0;
        double dx1 = x1_1 - x1_0;
        double dx2 = x2_1 - x2_0;
        double ds1 = y1_0 * x1_1 - y1_1 * x1_0;
        double ds2 = y2_0 * x2_1 - y2_1 * x2_0;
        double determ = dy2 * dx1 - dy1 * dx2;
        if (fabs(determ) < 1.0e-30) parallel = true;
        I[j] = make_double2((dx2 * ds1 - dx1 * ds2) / determ, (dy2 * ds1 - dy1 * ds2) / determ);
      }
      // (all reads of the old polygon are done: c[] and I[] are registers)
#pragma unroll
      for (int k = 0; k < CLIP_SLOTS; k++)
        if (k < n_cur && ((inmask >> k) & 1u))
          sh_poly[__popc(inmask & ((1u << k) - 1u)) + __popc(tmask & ((2u << k) - 1u))][tid] = c[k];
      sh_poly[__popc(inmask & ((1u << ka) - 1u))][tid] = I[0];                       // no crossing before the first one
      sh_poly[__popc(inmask & ((1u << kb) - 1u)) + 1][tid] = I[1];
    }
    // (no crossing: every vertex inside -> the polygon stays as it is; every vertex outside -> it is empty)
    n_cur = ncross + __popc(inmask);
    e0 = e1;
  }
  if (overflow) return -1;
  if (parallel) atomicOr(err, G_ERRBIT_PARALLEL);
  return n_cur;
}

// Result encoding shared by the clip kernels and the compaction: an accepted pair keeps pair_dst[p] = d and
// gets tmp_area/clon/clat[p]; a rejected pair gets pair_dst[p] = -1.
// (The integrals loop over the polygon's edges and a wave runs as many iterations as its largest polygon has.  Re-binning the
// block's 256 polygons by vertex count through LDS before the integrals, so that a wave sees polygons of equal size, was
// measured in round 1: 502 us against 482 -- the extra barriers and the scattered LDS columns cost more than the divergence.
// CLIP_COMPACT=1, round 3: the block's non-empty polygons moved to its first lanes before the integrals, order kept.  On the
// rectilinear path 4.16 M of 4.59 M candidate pairs are accepted, so there is next to nothing to move: 0.444 ms against 0.433.)
// row_cnt / tmp_rowpos (may be null): an accepted pair takes its slot in its destination row HERE -- a value-returning atomic whose
// latency hides behind the clip arithmetic -- instead of in the compaction, a kernel that does little else than wait on memory; the
// destination-row scan can then run before the compaction, which stores the row lists (perm) itself.
#ifndef CLIP_COMPACT
#define CLIP_COMPACT 0
#endif
__device__ __forceinline__ int d_row_slot(int *row_cnt, bool take, int d, int lane);   // (below, with the compaction)
template <int ORDER, bool RECT>
__global__ __launch_bounds__(CLIP_THREADS) __attribute__((amdgpu_waves_per_eu(CLIP_WAVES, CLIP_WAVES))) void k_clip_quad(FgPairSpace ps, FgCells S, const double *mask, FgCells D, FgRect R,
                                                 double *tmp_area, double *tmp_clon, double *tmp_clat, int *nacc,
                                                 int *defer_list, int *defer_cnt,
                                                 unsigned long long *stats, unsigned *err, int *row_cnt, int *tmp_rowpos)
{
  constexpr int T = CLIP_THREADS;
  __shared__ double2 sh_poly[CLIP_SLOTS][T];
#if CLIP_PAD_LDS
  __shared__ char sh_pad[CLIP_PAD_LDS];
  if (nacc == (int *)16) sh_pad[threadIdx.x * 7 % CLIP_PAD_LDS] = 1;
#endif
#if CLIP_COMPACT
  __shared__ unsigned short sh_perm[T];
  __shared__ unsigned char sh_n[T];
  __shared__ int sh_wtot[T / 64];
#endif
  if (RECT && *R.bad) return;
  {                                                   // the launch covers the regions' CAPACITY: blocks beyond a region's fill leave at once
    const unsigned first = blockIdx.x * T, r = first / (unsigned)ps.regcap;
    if (first - r * (unsigned)ps.regcap >= ps.fill[r * FG_FILL_STRIDE]) return;
  }
  d_load_trig_table();
  const int tid = threadIdx.x, lane = tid & 63;
  const int p0 = blockIdx.x * T;
  // ---- first half: clip.  Lane = candidate pair; 4 of 10 pairs of a cubed-sphere x lat-lon job come out empty.
  bool defer = false;
  int n_cur = 0;
  if (d_pair_live(ps, p0 + tid)) {
    n_cur = d_clip_quad_sh<RECT, T>(sh_poly, tid, ps.src[p0 + tid], ps.dst[p0 + tid], S, D, R, err);
    if (n_cur < 0) { defer = true; n_cur = 0; }        // rare; the general kernel finishes this pair
    else if (n_cur == 0) ps.dst[p0 + tid] = -1;
  }
  // ---- the block's non-empty polygons move to its first lanes (order kept): the integrals below cost 2/3 of this kernel's
  // arithmetic, and a wave of them costs the same with 38 live lanes as with 64.  The polygons stay where they are in LDS:
  // lane l works on column sh_perm[l].
  int src = tid;
#if CLIP_COMPACT
  {
    const int wave = tid >> 6;
    const unsigned long long hm = __ballot(n_cur > 0);
    if (lane == 0) sh_wtot[wave] = __popcll(hm);
    sh_n[tid] = (unsigned char)n_cur;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < T / 64; w++) { const int c = sh_wtot[w]; if (w < wave) base += c; total += c; }
    if (n_cur > 0) sh_perm[base + __popcll(hm & ((1ull << lane) - 1ull))] = (unsigned short)tid;
    __syncthreads();
    if (tid < total) { src = sh_perm[tid]; n_cur = sh_n[src]; } else n_cur = 0;
  }
#endif
  // ---- second half: area (+ centroid integrals) of polygon `src`
  bool below = false, acc = false;
  int s = -1, d_acc = -1;
  if (n_cur > 0) {
    const int p = p0 + src;
    s = ps.src[p];
    const int d = ps.dst[p];
    ClipOut o; o.area = -1.0; o.clon = 0; o.clat = 0;
    const double *px = (const double *)&sh_poly[0][src];
#if FG_EXP == 1
    o.area = n_cur + px[0] + px[1];
#else
    d_finish_pair<ORDER, 2 * T>(px, px + 1, n_cur, mask ? mask[s] : 1.0, S.area[s], D.area[d], S.lon_avg[s], &o, stats);
#endif
    if (o.area >= 0) {
      acc = true;
      tmp_area[p] = o.area;
      if (ORDER == 2) { tmp_clon[p] = o.clon; tmp_clat[p] = o.clat; }
      d_acc = d;
    } else {
      ps.dst[p] = -1;
      below = (o.area == -2.0);                                  // rare (slivers below the 1e-6 ratio)
    }
  }
  // the destination-row slot of an accepted pair.  Over a coarse target the pairs of a wave belong to a few destination cells
  // (C384 -> 10 deg: 1 450 exchange cells per row) and value-returning atomics on one address queue up behind each other (that
  // clip ran SEVEN times slower per pair than the headline's, with or without its arithmetic): one atomic per run of lanes with
  // the same destination cell (a lane that was not accepted ends a run).
  if (row_cnt) {
    const int slot = d_row_slot(row_cnt, acc, d_acc, lane);
    if (acc) tmp_rowpos[p0 + src] = slot;
  }
  // nacc[s]: accepted pairs of source cell s.  Lanes are pair-ordered (before and after the move), so one atomic per
  // (wave, source cell) run does it.
  if (__ballot(s >= 0)) {
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
  // one atomic per wave that has deferred pairs / slivers (value-returning same-address atomics serialise at ~12 ns each)
  const unsigned long long dm = __ballot(defer), bm = __ballot(below);
  if (dm) {
    int q = 0;
    if (lane == 0) q = atomicAdd(defer_cnt, __popcll(dm));
    q = __shfl(q, 0);
    if (defer) defer_list[q + __popcll(dm & ((1ull << lane) - 1ull))] = p0 + tid;
  }
  if (bm && lane == 0) atomicAdd(&stats[FG_STAT_BELOW], (unsigned long long)__popcll(bm));
}

// General path: up to 8 x 8 vertices, intermediate polygons up to 16.  One wave per block,
// LDS: two [16][64] double2 ping-pong buffers + cutter [8][64] double2 = 40 KiB.
#define GEN_THREADS 64
#define GEN_CAP 16
template <int ORDER, bool RECT>
__global__ __launch_bounds__(GEN_THREADS) void k_clip_general(const int *defer_list, const int *defer_cnt,
                                                              const int *pair_src, int *pair_dst,
                                                              FgCells S, const double *mask, FgCells D, FgRect R,
                                                              double *tmp_area, double *tmp_clon, double *tmp_clat, int *nacc,
                                                              unsigned long long *stats, unsigned *err, int *row_cnt, int *tmp_rowpos)
{
  __shared__ double2 sh_a[GEN_CAP][GEN_THREADS];
  __shared__ double2 sh_b[GEN_CAP][GEN_THREADS];
  __shared__ double2 sh_cut[G_MAXV][GEN_THREADS];
  if (RECT && *R.bad) return;
  d_load_trig_table();
  const int tid = threadIdx.x;
  const int ndefer = *defer_cnt;
  for (int q = blockIdx.x * GEN_THREADS + tid; q < ndefer; q += gridDim.x * GEN_THREADS) {
    const int p = defer_list[q];
    const int s = pair_src[p], d = pair_dst[p];
    const int n1 = S.nv[s], n2 = RECT ? 4 : D.nv[d];
    const double *sv = S.verts + (size_t)s * 16;
    const double lon_in_avg = S.lon_avg[s];
    double dvx[G_MAXV], dvy[G_MAXV], lon_out_avg;
    if (RECT) {
      const int j = d / R.nx, i = d - j * R.nx;
      const double *c = R.col + (size_t)i * RECT_COLW;
      const double ya = R.lat_ax[j], yb = R.lat_ax[j + 1];
      dvx[0] = c[0]; dvx[1] = c[1]; dvx[2] = c[2]; dvx[3] = c[3]; lon_out_avg = c[6];
      dvy[0] = ya; dvy[1] = ya; dvy[2] = yb; dvy[3] = yb;
#pragma unroll
      for (int k = 4; k < G_MAXV; k++) { dvx[k] = 0.0; dvy[k] = 0.0; }
    } else {
      const double *dv = D.verts + (size_t)d * 16;
#pragma unroll
      for (int k = 0; k < G_MAXV; k++) { dvx[k] = dv[k]; dvy[k] = dv[8 + k]; }
      lon_out_avg = D.lon_avg[d];
    }
    double shift = 0.0;
    {
      double dx = lon_out_avg - lon_in_avg;
      if (dx < -G_PI) shift = G_TPI; else if (dx > G_PI) shift = -G_TPI;
    }
    bool wrap = false;
    for (int k = 0; k < n1; k++) { double xv = sv[k]; if (xv > G_TPI || xv < 0.0) wrap = true; }
    for (int k = 0; k < n1; k++) {
      double xv = sv[k]; if (wrap) xv = d_pimod1(xv);
      sh_a[k][tid] = make_double2(xv, sv[8 + k]);
    }
#pragma unroll
    for (int k = 0; k < G_MAXV; k++)
      if (k < n2) {
        double xv = dvx[k]; if (shift != 0.0) xv += shift; if (wrap) xv = d_pimod1(xv);
        sh_cut[k][tid] = make_double2(xv, dvy[k]);
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
      if (row_cnt) tmp_rowpos[p] = atomicAdd(&row_cnt[d], 1);
    } else {
      pair_dst[p] = -1;
      if (o.area == -2.0) atomicAdd(&stats[FG_STAT_BELOW], 1ull);
    }
  }
}

// ---------------------------------------------------------------------------------------
// compaction into canonical order + per-source-cell sums
// ---------------------------------------------------------------------------------------
// The clip kernels count the accepted pairs of every source cell (nacc, one atomic per run of equal cells in a wave, hidden
// behind the clip arithmetic); a scan of nacc gives xoff[s], the reference's running nxgrid.  k_compact then works with ONE
// LANE PER PAIR, so that every array of the pair list is read once, fully coalesced:
//   * a block takes the source cells whose FIRST pair lies among its 256 pairs; such a cell (at most CP_SMALL pairs) ends
//     within the next CP_SMALL pairs, which the first lanes of the block handle as well ("halo");
//   * the destination indices of those 256 + CP_SMALL pairs are staged in LDS; a pair's rank among the accepted pairs of its
//     source cell by destination index (the reference's ij loop) is a handful of LDS reads;
//   * the exchange cell is stored at xoff[s] + rank (neighbouring lanes store neighbouring cells) and takes its slot in its
//     destination row (x_rowpos), which also counts the CSR row sizes;
//   * order 2: the values go to LDS at the cell's first pair + rank, and the lane holding the cell's first pair adds them up
//     in that order = exchange-cell order (conserve_interp.c:216-221).
// Cells with more than CP_SMALL pairs ("big": the cells around a pole of the target grid, every cell of a coarse -> fine
// remap; listed by the candidate kernel) are done by the big-cell blocks of the same launch.  Nothing depends on the order in which the candidate kernel filled the pair list.
// (Measured at C384 -> 0.25 deg: 109 us.  A first version gave each block 256 source cells and found xoff by look-back inside the
// kernel: 320-430 us -- its per-cell loads of the pair list are strided, and a chain of dependent round trips per block does not
// hide behind five blocks per CU; a lane-per-pair version that left the cells cut by a block boundary to one lane: 245 us.)
#define CP_SPAN (256 + CP_SMALL)
static_assert(CP_SMALL % 64 == 0 && CP_SMALL <= 256, "the halo lanes of k_compact are whole waves of its 256-lane block");

struct CpPair { int s, d, beg, cnt, li, p; long x0; int na; double a, l, t; bool mine; };

// Slot of an exchange cell in its destination row: atomicAdd(&row_cnt[d], 1) for every lane with take -- or, where many lanes of
// the wave share a destination cell (a coarse target: value-returning atomics on one address queue up), one atomic per group of
// lanes with the same d.  Every lane of the wave must call it.
__device__ __forceinline__ int d_row_slot(int *row_cnt, bool take, int d, int lane)
{
  const int key = take ? d : -1 - lane;                    // (distinct for the lanes that take no slot)
  const int prev = __shfl_up(key, 1, 64);
  const bool head = (lane == 0) || (key != prev);
  const unsigned long long hm = __ballot(head), am = __ballot(take);
  int slot = 0;
  if (!am) return 0;
  // similar resolutions: neighbouring lanes hardly ever share a cell -- a lane an atomic.  (Adjacent equal keys are the cheap sign
  // of sharing; with rejected pairs in between -- the great-circle search -- a third of the takers still have an equal neighbour.)
  if (4 * __popcll(hm & am) > 3 * __popcll(am)) {          // (wave-uniform)
    if (take) slot = atomicAdd(&row_cnt[d], 1);
    return slot;
  }
  unsigned long long rem = am;
  for (int it = 0; rem && it < 8; it++) {                  // groups by VALUE (not only runs), the first eight distinct cells
    const int lead = __ffsll((long long)rem) - 1;
    const int d0 = __shfl(d, lead);
    const unsigned long long grp = __ballot(take && d == d0) & rem;
    int base = 0;
    if (lane == lead) base = atomicAdd(&row_cnt[d0], __popcll(grp));
    base = __shfl(base, lead);
    if ((grp >> lane) & 1ull) slot = base + __popcll(grp & ((1ull << lane) - 1ull));
    rem &= ~grp;
  }
  if ((rem >> lane) & 1ull) slot = atomicAdd(&row_cnt[d], 1);       // (more than eight distinct cells in the wave: the rest one by one)
  return slot;
}

template <int ORDER>
__device__ __forceinline__ CpPair d_cp_load(const FgPairSpace &ps, const FgCompactIo &io, int p, int p0, unsigned live_end, bool halo)
{
  CpPair c;
  c.s = 0; c.d = -1; c.beg = 0; c.cnt = 0; c.x0 = 0; c.na = 0; c.a = 0; c.l = 0; c.t = 0; c.mine = false; c.li = p - p0; c.p = p;
  if ((unsigned)p < live_end) {
    c.s = ps.src[p]; c.d = ps.dst[p];
    c.beg = io.pair_beg[c.s]; c.cnt = io.pair_cnt[c.s];
    // this block's cell: its first pair is one of the block's 256 (halo lanes: ... and it is not a big cell's)
    c.mine = c.cnt <= CP_SMALL && c.beg >= p0 && c.beg < p0 + 256;
    if (c.mine) {
      c.x0 = io.xoff[c.s]; c.na = io.xoff[c.s + 1] - (int)c.x0;
      if (c.d >= 0) { c.a = io.tmp_area[p]; if (ORDER == 2) { c.l = io.tmp_clon[p]; c.t = io.tmp_clat[p]; } }
    }
  }
  (void)halo;
  return c;
}

// returns the exchange cell's position if its row slot is still to be taken (no slots from the clip kernels), else -1
template <int ORDER>
__device__ __forceinline__ long d_cp_place(const FgCompactIo &io, const CpPair &c, int p0, const int *sh_d, double (*sh_v)[CP_SPAN])
{
  if (!c.mine || c.d < 0 || c.x0 + c.na > io.xcap) return -1;
  int rank = 0;
  const int lb = c.beg - p0;
  for (int j = 0; j < c.cnt; j++) rank += ((unsigned)sh_d[lb + j] < (unsigned)c.d) ? 1 : 0;       // rejected entries are 0xffffffff
  const long pos = c.x0 + rank;
  io.x_src[pos] = c.s; io.x_dst[pos] = c.d; io.x_area[pos] = c.a;
  if (ORDER == 2) { io.x_c1[pos] = c.l; io.x_c2[pos] = c.t; sh_v[0][lb + rank] = c.a; sh_v[1][lb + rank] = c.l; sh_v[2][lb + rank] = c.t; }
  if (io.tmp_rowpos) io.perm[io.row_ptr[c.d] + io.tmp_rowpos[c.p]] = (int)pos;        // the row slot was taken by the clip kernel
  else return pos;
  return -1;
}

#define RANK_WORDS 1024      // 65536 destination indices: 45 rows of a 1440-column grid
#define BIG_STAGE 512
#define BIG_BLOCKS 512       // blocks of the big-cell role, in front of the lane-per-pair blocks
#define CP_LDS_DOUBLES (RANK_WORDS + RANK_WORDS / 2 + 3 * BIG_STAGE)   // the larger of the two roles' LDS needs (24 KB)

// The big cells: a block per cell.  Ranking a pair by comparing it with all the others of its cell is quadratic (great-circle
// search: ~3000 pairs in each of ~600 cells); here the block marks the cell's destination indices in an LDS bitmap over their
// span and reads each rank off as a prefix population count: linear.  Spans that do not fit fall back to the comparison.
template <int ORDER>
__device__ __forceinline__ void d_compact_big(int nsrc, const FgPairSpace &ps, const FgCompactIo &io, double *lds)
{
  unsigned long long *bits = (unsigned long long *)lds;              // [RANK_WORDS]
  int *pref = (int *)(lds + RANK_WORDS);                             // [RANK_WORDS]
  double (*sval)[BIG_STAGE] = (double (*)[BIG_STAGE])(lds + RANK_WORDS + RANK_WORDS / 2);   // [3][BIG_STAGE]
  __shared__ int smin, smax;
  if (io.fill_all && blockIdx.x == 0 && threadIdx.x < 64) {           // candidate totals for the host's capacity checks
    unsigned long long f = 0; unsigned mx = 0;
    for (int r = threadIdx.x; r < io.nreg_all; r += 64) { const unsigned v = io.fill_all[r * FG_FILL_STRIDE]; f += v; mx = max(mx, v); }
#pragma unroll
    for (int o = 32; o; o >>= 1) { f += __shfl_xor(f, o); mx = max(mx, (unsigned)__shfl_xor((int)mx, o)); }
    if (threadIdx.x == 0) { io.dc->total[1] = f; io.dc->total[3] = mx; }
  }
  const int nb = *io.big_cnt;
  for (int h = blockIdx.x; h < nb; h += BIG_BLOCKS) {
    const int s = io.big_list[h];
    const int o = io.pair_beg[s], c = io.pair_cnt[s];
    const long x0 = io.xoff[s];
    const int na = io.xoff[s + 1] - (int)x0;
    if (threadIdx.x == 0) { smin = 0x7fffffff; smax = -1; }
    __syncthreads();
    int lmin = 0x7fffffff, lmax = -1;
    for (int k = threadIdx.x; k < c; k += 256) { const int d = ps.dst[o + k]; if (d >= 0) { lmin = min(lmin, d); lmax = max(lmax, d); } }
    if (lmax >= 0) { atomicMin(&smin, lmin); atomicMax(&smax, lmax); }
    __syncthreads();
    const int dmin = smin, dmax = smax;
    const long span = (long)dmax - dmin + 1;
    const bool bitmap = dmax >= 0 && span <= (long)RANK_WORDS * 64;
    if (bitmap) {
      const int nw = (int)((span + 63) >> 6);
      for (int w = threadIdx.x; w < nw; w += 256) bits[w] = 0ull;
      __syncthreads();
      for (int k = threadIdx.x; k < c; k += 256) {
        const int d = ps.dst[o + k];
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
    }
    // exchange cells to their places; order 2: the values also go to LDS in rank order, BIG_STAGE ranks at a time, where three
    // lanes (one array each) add them up in exchange-cell order -- hundreds to thousands of ordered additions per cell, which
    // from global memory (even with the loads 16 ahead) set the duration of the whole compaction
    double acc = 0;
    for (int c0 = 0; c0 < max(na, 1); c0 += BIG_STAGE) {
      for (int k = threadIdx.x; k < c; k += 256) {
        const int p = o + k;
        const int d = ps.dst[p];
        if (d < 0) continue;
        int rank;
        if (bitmap) { const int w = (d - dmin) >> 6, b = (d - dmin) & 63; rank = pref[w] + __popcll(bits[w] & ((1ull << b) - 1ull)); }
        else { rank = 0; for (int j = 0; j < c; j++) rank += ((unsigned)ps.dst[o + j] < (unsigned)d) ? 1 : 0; }
        if (rank < c0 || rank >= c0 + BIG_STAGE) continue;
        const long pos = x0 + rank;
        if (pos >= io.xcap) continue;
        const double a = io.tmp_area[p];
        io.x_src[pos] = s; io.x_dst[pos] = d; io.x_area[pos] = a;
        if (ORDER == 2) {
          const double l = io.tmp_clon[p], t = io.tmp_clat[p];
          io.x_c1[pos] = l; io.x_c2[pos] = t;
          sval[0][rank - c0] = a; sval[1][rank - c0] = l; sval[2][rank - c0] = t;
        }
        if (io.tmp_rowpos) io.perm[io.row_ptr[d] + io.tmp_rowpos[p]] = (int)pos;
        else io.x_rowpos[pos] = atomicAdd(&io.row_cnt[d], 1);
      }
      __syncthreads();
      if (ORDER == 2 && threadIdx.x < 3 && x0 + na <= io.xcap) {
        const double *v = sval[threadIdx.x];
        const int n = min(BIG_STAGE, na - c0);
        int k = 0;
        for (; k + 8 <= n; k += 8) {
          double t[8];
#pragma unroll
          for (int u = 0; u < 8; u++) t[u] = v[k + u];
#pragma unroll
          for (int u = 0; u < 8; u++) acc += t[u];
        }
        for (; k < n; k++) acc += v[k];
      }
      __syncthreads();
    }
    if (ORDER == 2 && io.sums && threadIdx.x < 3) io.sums[(size_t)threadIdx.x * nsrc + s] = acc;
  }
}

// Blocks [0, BIG_BLOCKS): the big cells (listed by the candidate kernel); the rest: one lane per pair.  One launch, so that the
// few long-running big-cell blocks start first and run beside the others instead of after them.
template <int ORDER>
__global__ __launch_bounds__(256) void k_compact(int nsrc, FgPairSpace ps, FgCompactIo io)
{
  __shared__ double lds[CP_LDS_DOUBLES];
  if ((int)blockIdx.x < BIG_BLOCKS) { d_compact_big<ORDER>(nsrc, ps, io, lds); return; }
  int *sh_d = (int *)lds;                                            // [CP_SPAN]
  double (*sh_v)[CP_SPAN] = (double (*)[CP_SPAN])(lds + CP_SPAN / 2 + 8);   // [3][CP_SPAN]
  const int tid = threadIdx.x;
  const int p0 = ((int)blockIdx.x - BIG_BLOCKS) * 256;
  const unsigned r = (unsigned)p0 / (unsigned)ps.regcap;          // block-uniform
  const unsigned rfill = ps.fill[r * FG_FILL_STRIDE];
  const unsigned live_end = r * (unsigned)ps.regcap + (rfill < (unsigned)ps.regcap ? rfill : (unsigned)ps.regcap);   // end of the region's pairs
  if ((unsigned)p0 >= live_end) return;
  const CpPair c = d_cp_load<ORDER>(ps, io, p0 + tid, p0, live_end, false);
  CpPair h; h.mine = false; h.d = -1;
  if (tid < CP_SMALL) { h = d_cp_load<ORDER>(ps, io, p0 + 256 + tid, p0, live_end, true); sh_d[256 + tid] = h.d; }
  sh_d[tid] = c.d;
  __syncthreads();
  const bool first = (unsigned)(p0 + tid) < live_end && p0 + tid == c.beg;
  {
    const long pos = d_cp_place<ORDER>(io, c, p0, sh_d, sh_v);
    if (!io.tmp_rowpos) { const int slot = d_row_slot(io.row_cnt, pos >= 0, c.d, tid & 63); if (pos >= 0) io.x_rowpos[pos] = slot; }
  }
  if (tid < CP_SMALL) {                                    // (whole waves: CP_SMALL is a multiple of 64)
    const long pos = d_cp_place<ORDER>(io, h, p0, sh_d, sh_v);
    if (!io.tmp_rowpos) { const int slot = d_row_slot(io.row_cnt, pos >= 0, h.d, tid & 63); if (pos >= 0) io.x_rowpos[pos] = slot; }
  }
  if (ORDER != 2 || !io.sums) return;
  __syncthreads();
  if (!first || !c.mine || c.x0 + c.na > io.xcap) return;
  double sa = 0, sl = 0, st = 0;
  const int lb = c.beg - p0;
  for (int k = 0; k < c.na; k++) { sa += sh_v[0][lb + k]; sl += sh_v[1][lb + k]; st += sh_v[2][lb + k]; }
  io.sums[c.s] = sa; io.sums[nsrc + c.s] = sl; io.sums[2 * (size_t)nsrc + c.s] = st;
}

// ---------------------------------------------------------------------------------------
// order-2 centroid pass
// ---------------------------------------------------------------------------------------
// cen[0][s], cen[1][s] = centroid lon/lat of source cell s (conserve_interp.c:327-348)
__global__ __launch_bounds__(256) void k_centroids(int nsrc, FgCells S, const double *sums, double *cen)
{
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  double a = 0, ca = 1, cl = 0, ct = 0;
  bool whole = false;                                      // the cell's own integrals are needed (covered area off by >= 1e-3)
  // (a source cell a culling search left out -- nv = 0 -- has no exchange cell on this rank: its centroid is never read, though
  // the totals handed over by the other ranks are there)
  if (s < nsrc && (a = sums[s]) > 0 && S.nv[s] > 0) {
    ca = S.area[s];
    if (fabs(a - ca) / ca < 1.e-3) {
      cl = sums[nsrc + s] / a;
      ct = sums[2 * (size_t)nsrc + s] / a;
    } else whole = true;
  }
  // the sine table (3.5 KB into LDS, a barrier) only for blocks that hold such a cell: none in a global remap of similar grids
  if (__syncthreads_or(whole)) {
    d_load_trig_table();
    if (whole) {
      double x[G_MAXV], y[G_MAXV];
      int n = S.nv[s];
      const double *vp = S.verts + (size_t)s * 16;
      for (int k = 0; k < G_MAXV; k++) { x[k] = vp[k]; y[k] = vp[8 + k]; }
      cl = d_poly_ctrlon<1>(x, y, n, S.lon_avg[s]) / ca;
      ct = d_poly_ctrlat<1>(x, y, n) / ca;
    }
  }
  if (s < nsrc) { cen[s] = cl; cen[nsrc + s] = ct; }
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

void fgd_cell_struct2(const FgTileSet &ts, const FgTile *tiles_in, FgTile *tiles_out, int ntiles, int nsrc, int ndst, FgCells S, FgCells D,
                      FgBins b, int *slot_cnt, int order, int *src_idx_f, double *sums, unsigned *err, hipStream_t st,
                      unsigned long long *band_keys, int cull, double dst_tlon)
{
  const int nbS = nblk(nsrc, 256), nbD = nblk(ndst, 256);
  if (nbS + nbD > 0)
    k_cell_struct2<<<nbS + nbD, 256, 0, st>>>(ts, tiles_in, tiles_out, ntiles, nsrc, ndst, nbS, S, D, b, slot_cnt, order, src_idx_f, sums, err,
                                              band_keys, cull, dst_tlon);
}

void fgd_bin_count(int ncells, FgCells c, FgBins b, int *slot_cnt, hipStream_t st)
{
  if (ncells > 0) k_bin_count<<<nblk(ncells, 256), 256, 0, st>>>(ncells, c, b, slot_cnt);
}

void fgd_bin_fill(int ndst, FgCells D, FgBins b, int *slot_fill, const int *slot_start, FgBinEntry *entries, int cap,
                  int nsrc, FgCells S, const double *mask, int *heavy_list, int *heavy_cnt, hipStream_t st)
{
  const int nbD = nblk(ndst, 256), nbS = nblk(nsrc, 256);
  if (nbD + nbS > 0)
    k_bin_fill<<<nbD + nbS, 256, 0, st>>>(ndst, nbD, D, b, slot_fill, slot_start, entries, cap, nsrc, S, mask, heavy_list, heavy_cnt);
}

void fgd_candidates1(int c0, int c1, FgCells S, const double *mask, FgBins b, const int *slot_start, const FgBinEntry *entries, int ecap,
                     FgPairSpace ps, int *pair_beg, int *pair_cnt, const int *heavy_list, const int *heavy_cnt, int *big_list, int *big_cnt,
                     hipStream_t st)
{
  if (c1 <= c0) return;
  const int nbR = nblk((long)(c1 - c0) * CAND_G, 64);
  // (as in fgd_candidates_rect, but up to eight times the waves: a curvilinear target that contains a pole -- a polar tile of a cubed
  // sphere -- lists a fifth of a lat-lon source grid's cells; 0.25 deg -> C384 tile 3: tile 2.8 -> 2.07 ms; 8192 waves: 2.20)
  const int H = min(8 * HEAVY_BLOCKS, max(64, nblk(c1 - c0, 8)));
  k_candidates1<<<nbR + H, 64, 0, st>>>(c0, c1, H, S, mask, b, slot_start, entries, ecap, ps, pair_beg, pair_cnt, heavy_list, heavy_cnt, big_list, big_cnt);
}

void fgd_clip_quad(int order, FgPairSpace ps, FgCells S, const double *mask, FgCells D,
                   double *tmp_area, double *tmp_clon, double *tmp_clat, int *nacc, int *defer_list, int *defer_cnt,
                   unsigned long long *stats, unsigned *err, hipStream_t st, const FgRect *rect, int *row_cnt, int *tmp_rowpos)
{
  const long np = fgd_pairs_total(ps);
  if (np <= 0) return;
  const FgRect R = rect ? *rect : FgRect{};
  const int g = nblk(np, CLIP_THREADS);
#define FG_LAUNCH_QUAD(O, RC) k_clip_quad<O, RC><<<g, CLIP_THREADS, 0, st>>>(ps, S, mask, D, R, tmp_area, tmp_clon, tmp_clat, nacc, defer_list, defer_cnt, stats, err, row_cnt, tmp_rowpos)
  if (order == 2) { if (rect) FG_LAUNCH_QUAD(2, true); else FG_LAUNCH_QUAD(2, false); }
  else            { if (rect) FG_LAUNCH_QUAD(1, true); else FG_LAUNCH_QUAD(1, false); }
#undef FG_LAUNCH_QUAD
}

void fgd_clip_general(int order, FgPairSpace ps, FgCells S, const double *mask, FgCells D,
                      double *tmp_area, double *tmp_clon, double *tmp_clat, int *nacc, int *defer_list, int *defer_cnt,
                      unsigned long long *stats, unsigned *err, hipStream_t st, const FgRect *rect, int *row_cnt, int *tmp_rowpos)
{
  const long np = fgd_pairs_total(ps);
  if (np <= 0) return;
  int grid = nblk(np, GEN_THREADS); if (grid > 1024) grid = 1024;
  const FgRect R = rect ? *rect : FgRect{};
#define FG_LAUNCH_GEN(O, RC) k_clip_general<O, RC><<<grid, GEN_THREADS, 0, st>>>(defer_list, defer_cnt, ps.src, ps.dst, S, mask, D, R, tmp_area, tmp_clon, tmp_clat, nacc, stats, err, row_cnt, tmp_rowpos)
  if (order == 2) { if (rect) FG_LAUNCH_GEN(2, true); else FG_LAUNCH_GEN(2, false); }
  else            { if (rect) FG_LAUNCH_GEN(1, true); else FG_LAUNCH_GEN(1, false); }
#undef FG_LAUNCH_GEN
}

// ---- rectilinear destination grid
void fgd_rect_tables(const double *lon, const double *lat, int nx, int ny, double *hdr, double *lat_ax, double *lon_ax, double *col, double *row,
                     unsigned *bad, unsigned *err, hipStream_t st, double dst_tlon)
{
  const long np = (long)(nx + 1) * (ny + 1);
  const int g = (int)std::min<long>(1024, std::max<long>((np + 255) / 256, (std::max(nx, ny) + 1 + 255) / 256));
  k_rect_tables<<<std::max(g, 1), 256, 0, st>>>(lon, lat, nx, ny, hdr, lat_ax, lon_ax, col, row, bad, err, dst_tlon);
}
void fgd_cell_struct2r(const FgTileSet &ts, const FgTile *tiles_in, FgTile *tiles_out, int ntiles, int nsrc, int ndst, FgCells S, double *area_out,
                       FgRect R, const double *mask, int order, int *src_idx_f, double *sums, unsigned *err, hipStream_t st,
                       unsigned long long *band_keys, int cull, int *heavy_list, int *heavy_cnt)
{
  const int nbS = nblk(nsrc, 256), nbD = nblk(ndst, 256);
  if (nbS + nbD > 0)
    k_cell_struct2r<<<nbS + nbD, 256, 0, st>>>(ts, tiles_in, tiles_out, ntiles, nsrc, ndst, nbS, S, area_out, R, mask, order, src_idx_f, sums, err,
                                               band_keys, cull, heavy_list, heavy_cnt);
}
void fgd_polylist_records(FgPolyList P, FgCells S, int *src_idx_f, double *sums, const FgRect *rect, int *heavy_list, int *heavy_cnt,
                          unsigned *err, hipStream_t st)
{
  if (P.npoly > 0)
    k_polylist_records<<<nblk(P.npoly, 256), 256, 0, st>>>(P, S, src_idx_f, sums, rect ? *rect : FgRect{}, rect ? 1 : 0, heavy_list, heavy_cnt, err);
}
void fgd_rect_materialize(int ndst, FgRect R, FgCells D, hipStream_t st)
{
  if (ndst > 0) k_rect_materialize<<<nblk(ndst, 256), 256, 0, st>>>(ndst, R, D);
}
void fgd_candidates_rect(int nsrc, FgCells S, const double *mask, FgRect R, FgPairSpace ps, int *pair_beg, int *pair_cnt,
                         const int *heavy_list, const int *heavy_cnt, int *big_list, int *big_cnt, hipStream_t st)
{
  if (nsrc <= 0) return;
  const int nbR = nblk(nsrc, 64);
  // waves for the listed cells: on a coarse source grid over a fine target EVERY cell is listed (C48 -> 0.25 deg: 13 824 cells; with
  // nsrc / 64 = 216 waves the candidates took 0.24 ms, with nsrc / 8 0.064 -- dealing a cell's (row, column) pairs to all 64 lanes
  // through LDS instead of a lane per column made no difference on top of that)
  const int H = min(HEAVY_BLOCKS, max(64, nblk(nsrc, 8)));
  k_candidates_rect<<<nbR + H, 64, 0, st>>>(nsrc, H, S, mask, R, ps, pair_beg, pair_cnt, heavy_list, heavy_cnt, big_list, big_cnt);
}

void fgd_compact(int order, int nsrc, FgPairSpace ps, const FgCompactIo &io, hipStream_t st)
{
  if (nsrc <= 0) return;
  const long np = fgd_pairs_total(ps);
  if (order == 2) k_compact<2><<<BIG_BLOCKS + nblk(np, 256), 256, 0, st>>>(nsrc, ps, io);
  else            k_compact<1><<<BIG_BLOCKS + nblk(np, 256), 256, 0, st>>>(nsrc, ps, io);
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

// total[3][nsrc] += this plan's exchange cells, cell by cell in exchange-cell order, CONTINUING from the value already there:
// conserve_interp.c:203-221 adds the gathered exchange cells of all ranks (and of one output tile after the other) one by one
// "for the purpose of bitwise reproducing"; a running total handed from plan to plan (and from rank to rank for the cells cut by a
// band boundary) performs the same additions in the same order.  cells == null: all source cells.
__global__ __launch_bounds__(256) void k_accumulate_cell_sums(int n, const int *cells, int nsrc, const int *xoff, const double *xa,
                                                               const double *c1, const double *c2, double *total)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = cells ? cells[i] : i;
  if (s < 0 || s >= nsrc) return;
  const int b = xoff[s], e = xoff[s + 1];
  if (b >= e) return;
  double a0 = total[s], a1 = total[(size_t)nsrc + s], a2 = total[2 * (size_t)nsrc + s];
  for (int q = b; q < e; q++) { a0 += xa[q]; a1 += c1[q]; a2 += c2[q]; }
  total[s] = a0; total[(size_t)nsrc + s] = a1; total[2 * (size_t)nsrc + s] = a2;
}
void fgd_accumulate_cell_sums(int n, const int *cells, int nsrc, const int *xoff, const double *xa, const double *c1, const double *c2,
                              double *total, hipStream_t st)
{
  if (n > 0) k_accumulate_cell_sums<<<(n + 255) / 256, 256, 0, st>>>(n, cells, nsrc, xoff, xa, c1, c2, total);
}
