// xgrid_device.h -- structs and launcher prototypes shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>

// one grid tile as seen by the kernels (device pointers)
struct FgTile {
  const double *lon;   // [(ny+1)*(nx+1)] corner longitudes, radians
  const double *lat;
  int nx, ny;
  int cell_off;        // index of this tile's first cell in the flattened cell numbering
};

// great-circle path: corners as unit vectors (latlon2xyz, mosaic_util.c:212-222), same indexing as FgTile
struct FgTileXyz {
  const double *x, *y, *z;
  int nx, ny;
  int cell_off;
};
#define G_ERRBIT_GC_CONVEX1 16u    // "grid box 1 is not convex" (create_xgrid.c:1575)
#define G_ERRBIT_GC_CONVEX2 32u
#define G_ERRBIT_GC_CLIP    64u    // one of the clip's fatal checks; code in err[1]

// per-cell records, SoA scalars + one 128-byte vertex record per cell
// (the quantities of create_xgrid.c:991-1016 plus the cell area of :66-88)
struct FgCells {
  double *lat_min, *lat_max, *lon_min, *lon_max, *lon_avg, *area;
  int *nv;
  double *verts;       // [ncells][16]: lon[8] then lat[8], after fix_lon
};

// uniform bins over latitude x (longitude mod 2pi)
struct FgBins {
  int nblat, nblon;
  double inv_wlat, inv_wlon;
};

// one destination cell as stored in the bin table (bin order): the five numbers the
// bounding-box rejects need, so the candidate scan streams records instead of chasing indices
struct FgBinEntry {
  double lat_min, lat_max, lon_min, lon_max, lon_avg;
  int d;       // destination cell index
  int row0;    // first bin row of the cell (de-duplicates wide cells)
};

enum {
  FG_STAT_PAIRS = 0,      // candidate pairs after the bounding-box tests
  FG_STAT_NONEMPTY = 1,   // pairs whose clip is non-empty (= nxgrid + FG_STAT_BELOW)
  FG_STAT_NXGRID = 2,
  FG_STAT_BORDERLINE = 3, // |xarea/min_area - 1e-6| < 1e-15
  FG_STAT_BINS = 4,
  FG_STAT_BIN_ENTRIES = 5,
  FG_STAT_DEFERRED = 6,   // pairs handled by the general (non quad x quad) kernel
  FG_STAT_HEAVY = 7,      // source cells whose candidate scan got a whole wave
  FG_STAT_BELOW = 8,      // non-empty clips rejected by the 1e-6 area ratio
  FG_STAT_EXACT = 9,      // 1 if the plan was built by the exactly sized (three-readback) search, 0 by the single-sync one
  FG_NSTATS = 10
};

// ---- candidate pairs: one pair list in FG_NREG regions of `regcap` entries.  A wave of the candidate kernel appends the
// pairs of its source cells to one region with a single atomic on that region's fill counter (a counter per 128-byte
// line: appends to different regions do not serialise), so the list is written in ONE pass -- no count pass, no scan.
// A region may overflow (fill > regcap): writes beyond it are dropped, the host sees it at its one readback and repeats
// the search with regions sized by the counters.  regcap is a multiple of FG_REG_ALIGN, so a block of a pair kernel lies in one region.
#define FG_REG_ALIGN 512
#define FG_NREG 64
#define FG_FILL_STRIDE 32            // unsigned words per region counter (128 B)
struct FgPairSpace {
  int *src, *dst;                    // [nreg * regcap]: source / destination cell of pair p; dst = -1 once a clip rejected it
  unsigned *fill;                    // fill[r * FG_FILL_STRIDE] = pairs appended to region r (true count, may exceed regcap)
  int regcap, nreg;
};
#ifdef __HIPCC__
__device__ __forceinline__ bool d_pair_live(const FgPairSpace &ps, int p)
{
  const unsigned first = blockIdx.x * blockDim.x;                 // block-uniform (scalar) region lookup
  const unsigned r = first / (unsigned)ps.regcap;
  const unsigned f = ps.fill[r * FG_FILL_STRIDE];
  return (unsigned)p - r * (unsigned)ps.regcap < (f < (unsigned)ps.regcap ? f : (unsigned)ps.regcap);
}
#endif
static inline long fgd_pairs_total(const FgPairSpace &ps) { return (long)ps.regcap * ps.nreg; }
#ifdef __HIPCC__
// doubles as unsigned keys with the same order (atomicMax over latitudes: the band of a culling search)
__device__ __forceinline__ unsigned long long d_ord_key(double v)
{
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double d_ord_val(unsigned long long k)
{
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}
#endif

// source tiles + the destination tile as a kernel argument (no descriptor upload before the first kernel)
#define FG_TILESET_MAX 8
struct FgTileSet { FgTile t[FG_TILESET_MAX]; int n; };

// The source cells are searched in up to FG_MAX_CHUNKS chunks so that the VALU-bound clip of one chunk runs (on a second
// stream) beside the latency-bound candidate scan of the next and the compaction of the previous one.
#define FG_MAX_CHUNKS 8
// device-side counters of one search (plan.hip reads them back once)
struct FgCounters {
  unsigned long long total[4];     // [0] bin-table entries  [1] candidate pairs  [2] (unused)  [3] largest region fill
  unsigned long long rows_total;   // total of the destination-row scan (= nxgrid)
  unsigned long long band_keys[2]; // latitude range of the destination cells as ordered keys (source-cell culling)
  unsigned long long xtot[FG_MAX_CHUNKS];   // running nxgrid after chunk k of the source cells (the last one is nxgrid)
  unsigned err[4];
  int heavy_cnt;
  unsigned rect_bad;               // rectilinear path: != 0 = the destination grid failed the check (k_rect_tables)
  int defer_cnt[FG_MAX_CHUNKS], big_cnt[FG_MAX_CHUNKS];
  int gc_list2_cnt[FG_MAX_CHUNKS];           // great-circle path: pairs k_gc_walk handed to the one-kernel clip
  unsigned long long stats[FG_NSTATS];
};
#define G_ERRBIT_LOOKBACK 128u     // a single-pass scan waited too long for its predecessor tile (never observed)

// single-pass exclusive scan (decoupled look-back): n inputs -> n+1 prefixes (out[n] = total, also 64-bit in *total_dev).
// status: zeroed words, one per tile of 2048 inputs (fgd_scan_tiles); ticket: zeroed word
long fgd_scan_tiles(long n);
// base_dev (may be null): a device value added to every prefix and to the total (chunked scans)
void fgd_exclusive_scan1(const int *in, long n, int *out, unsigned long long *status, unsigned *ticket,
                         unsigned long long *total_dev, unsigned *err, hipStream_t st, const unsigned long long *base_dev = nullptr);

void fgd_exclusive_scan2(const int *in_a, long n_a, int *out_a, unsigned long long *status_a, unsigned *ticket_a, unsigned long long *total_a,
                         const int *in_b, long n_b, int *out_b, unsigned long long *status_b, unsigned *ticket_b, unsigned long long *total_b,
                         unsigned *err, hipStream_t st);

// per-cell records of the source tiles and of the destination tile in ONE launch; also counts the destination cells into
// their bins (slot_cnt), fills src_idx_f, zeroes sums[3][nsrc] (may be null) and stores the tile descriptors at tiles_out
void fgd_cell_struct2(const FgTileSet &ts, const FgTile *tiles_in, FgTile *tiles_out, int ntiles, int nsrc, int ndst, FgCells S, FgCells D,
                      FgBins b, int *slot_cnt, int order, int *src_idx_f, double *sums, unsigned *err, hipStream_t st,
                      unsigned long long *band_keys = nullptr, int cull = 0, double dst_tlon = 3.14159265358979323846);
void fgd_cell_struct(const FgTile *tiles_dev, int ntiles, int ncells, FgCells c, unsigned *err, hipStream_t st);
// band_keys[0] / [1] = ordered keys of max / ~min of lat[0..n) (the destination grid's corner latitudes); cull = 2 in fgd_cell_struct2
// then means: keys already there, cull the source blocks in the same launch as the destination blocks
void fgd_band_keys(const double *lat, long n, unsigned long long *band_keys, hipStream_t st);
void fgd_bin_count(int ncells, FgCells c, FgBins b, int *slot_cnt, hipStream_t st);
// bin fill + list of the source cells whose candidate scan gets a whole wave
void fgd_bin_fill(int ndst, FgCells D, FgBins b, int *slot_fill, const int *slot_start, FgBinEntry *entries, int cap,
                  int nsrc, FgCells S, const double *mask, int *heavy_list, int *heavy_cnt, hipStream_t st);
// source cells [c0, c1) only (one chunk); ps, pair_beg are the chunk's
void fgd_candidates1(int c0, int c1, FgCells S, const double *mask, FgBins b, const int *slot_start, const FgBinEntry *entries, int ecap,
                     FgPairSpace ps, int *pair_beg, int *pair_cnt, const int *heavy_list, const int *heavy_cnt, int *big_list, int *big_cnt,
                     hipStream_t st);
// A rectilinear destination grid (lon_out a function of the column, lat_out of the row, bit for bit): see k_rect_tables.
// All pointers are device pointers; bad != 0 after the check kernel = the grid is NOT rectilinear (the tables are then junk).
#define RECT_COLW 8
struct FgRect {
  const double *lon_ax;   // [nx+1] raw longitude axis (copy of row 0 of lon_out)
  const double *lat_ax;   // [ny+1] compact copy of lat_out[j][0]
  const double *col;      // [nx][8] per column: the four longitudes after fix_lon (SW, SE, NE, NW), lon_min, lon_max, lon_avg, width
  const double *hdr;      // [8] lon[0], nx / (lon[nx] - lon[0]), lat[0], ny / (lat[ny] - lat[0])
  const double *row;      // [ny+1][4] per latitude-axis value j: sin(lat[j]); and of row j: sin of the mid latitude, sin(dy)/dy of half the
                          // height, 1.0 if the row is flatter than 1e-10 -- what poly_area evaluates on a cell of that row
  const unsigned *bad;
  int nx, ny;
};
void fgd_rect_tables(const double *lon, const double *lat, int nx, int ny, double *hdr, double *lat_ax, double *lon_ax, double *col,
                     double *row, unsigned *bad, unsigned *err, hipStream_t st, double dst_tlon = 3.14159265358979323846);
void fgd_cell_struct2r(const FgTileSet &ts, const FgTile *tiles_in, FgTile *tiles_out, int ntiles, int nsrc, int ndst, FgCells S, double *area_out,
                       FgRect R, const double *mask, int order, int *src_idx_f, double *sums, unsigned *err, hipStream_t st,
                       unsigned long long *band_keys, int cull, int *heavy_list, int *heavy_cnt);
void fgd_rect_materialize(int ndst, FgRect R, FgCells D, hipStream_t st);
// Source "cells" given as a list of polygons (<= 8 vertices each, already in the longitude frame they are to be clipped in):
// fills the source records a search needs (box, vertices, the caller's lon_avg and reference area), the field index, zeroes the
// sums, and -- rect != null -- lists the heavy ones.  make_coupler_mosaic's atmosphere x land cells against the ocean grid.
struct FgPolyList { const int *n; const double *lon, *lat; const double *lon_avg, *area; int npoly; };
void fgd_polylist_records(FgPolyList P, FgCells S, int *src_idx_f, double *sums, const FgRect *rect, int *heavy_list, int *heavy_cnt,
                          unsigned *err, hipStream_t st);
void fgd_candidates_rect(int nsrc, FgCells S, const double *mask, FgRect R, FgPairSpace ps, int *pair_beg, int *pair_cnt,
                         const int *heavy_list, const int *heavy_cnt, int *big_list, int *big_cnt, hipStream_t st);
// rect != null: the destination cells come from the rectilinear tables (D holds areas only)
void fgd_clip_general(int order, FgPairSpace ps, FgCells S, const double *mask, FgCells D,
              double *tmp_area, double *tmp_clon, double *tmp_clat, int *nacc, int *defer_list, int *defer_cnt,
              unsigned long long *stats, unsigned *err, hipStream_t st, const FgRect *rect = nullptr, int *row_cnt = nullptr,
              int *tmp_rowpos = nullptr);
// row_cnt / tmp_rowpos != null: accepted pairs take their destination-row slot in the clip kernel (tmp_rowpos[pair])
void fgd_clip_quad(int order, FgPairSpace ps, FgCells S, const double *mask, FgCells D,
              double *tmp_area, double *tmp_clon, double *tmp_clat, int *nacc, int *defer_list, int *defer_cnt,
              unsigned long long *stats, unsigned *err, hipStream_t st, const FgRect *rect = nullptr, int *row_cnt = nullptr,
              int *tmp_rowpos = nullptr);
// accepted pairs -> exchange cells in canonical order (xoff = scan of the clip kernels' nacc), per-source-cell sums,
// destination-row sizes and slots
struct FgCompactIo {
  const int *pair_beg, *pair_cnt;
  const double *tmp_area, *tmp_clon, *tmp_clat;
  const int *xoff;
  int *x_src, *x_dst;
  double *x_area, *x_c1, *x_c2;
  int *row_cnt, *x_rowpos;
  const int *tmp_rowpos;             // != null: row slots per PAIR from the clip kernels; row_ptr is final and perm is stored here
  const int *row_ptr; int *perm;
  double *sums;                      // [3][nsrc] (order 2, zeroed: cells without exchange cells are not visited) or null
  const int *big_list;               // the chunk's cells with more than CP_SMALL pairs (from the candidate kernel)
  const int *big_cnt;
  const unsigned *fill_all;          // last chunk only: fill counters of ALL regions, summed up into dc->total[1] / [3]; else null
  int nreg_all;
  FgCounters *dc;
  long xcap;                         // entries the x_* arrays hold
};
void fgd_compact(int order, int nsrc, FgPairSpace ps, const FgCompactIo &io, hipStream_t st);
void fgd_centroids(int nsrc, FgCells S, const double *sums, double *cen, hipStream_t st);
void fgd_distances(long nx, int nsrc, const int *x_src, const double *x_area, const double *cen, double *x_c1,
                   double *x_c2, hipStream_t st);

// ---- sweep (apply_kernels.hip) ----
// CSR by destination cell: row_ptr[ndst+1]; per entry: index of the source value in the
// field array (idx_f), in the gradient arrays (idx_g), area, di, dj -- packed per entry.
// one packed record per entry: a lane reads its row's entries as consecutive 16/32-byte records
struct FgCsrEntry1 { int idx_f; int pad; double area; };                       // order 1, 16 B
struct FgCsrEntry2 { int idx_f; int idx_g; double area, di, dj; };            // order 2, 32 B
struct FgCsr {
  int *row_ptr;
  FgCsrEntry1 *e1;
  FgCsrEntry2 *e2;
};
void fgd_csr_count(long nx, const int *x_dst, int *row_cnt, hipStream_t st);
void fgd_csr_fill(long nx, const int *x_dst, const int *row_ptr, int *row_fill, int *perm, hipStream_t st);
// nx_cap threads; the true count is read from *nx_dev (null: nx_cap is the count)
void fgd_csr_fill_pos(long nx_cap, const unsigned long long *nx_dev, const int *x_dst, const int *row_ptr, const int *x_rowpos, int *perm,
                      hipStream_t st);
// rows of perm into ascending exchange-cell order, then the packed CSR records; cen != null (order 2): x_c1/x_c2 still hold the
// centroid integrals and di/dj are formed here (conserve_interp.c:256-257,355-356), else they are taken as they are
void fgd_csr_sortgather(int order, int ndst, long nx, const int *perm, const int *x_src, const double *x_area, const double *x_c1,
                        const double *x_c2, const int *src_idx_f, const double *cen, int nsrc, FgCsr csr, hipStream_t st, int *tmp = nullptr, long ntmp = 0,
                        int long_rows = 0);   // tmp: 2 * ntmp ints of scratch for rows beyond the LDS staging (ntmp > exchange cells), or null
void fgd_src_field_index(int order, const FgTile *tiles_dev, int ntiles, int nsrc, int *src_idx_f, hipStream_t st);
void fgd_apply1(int order, int ndst, FgCsr csr, const double *f, const double *gx, const double *gy, const int *gmask,
                int has_missing, double missing, double *out, double *row_sum, hipStream_t st, long nx = -1);
void fgd_apply_il(int order, int nb, int ndst, FgCsr csr, const double *f, const double *gx, const double *gy, double missing,
                  double *out, double *row_sum, long out_ld, int nb_valid, hipStream_t st, long nx = -1);
void fgd_interleave(int nb_pad, long n, const double *in, long ld, int nb_valid, double *out, hipStream_t st);
void fgd_apply_il_merged(int nb, int ndst, long nx, FgCsr csr, const double *rec, double missing, double *out, double *row_sum, long out_ld,
                         int nb_valid, hipStream_t st);
void fgd_merge3(int nb_pad, long n, const int *src_idx_f, const double *f, long ld_f, const double *gx, const double *gy, long ld_g,
                int nb_valid, double *out, hipStream_t st);
void fgd_interleave3(int nb_pad, int narr, const double *const *in, const long *ld, const long *n, double *const *out, int nb_valid,
                     hipStream_t st);
void fgd_deinterleave(int nb_pad, long n, const double *in, long ld, int nb_valid, double *out, hipStream_t st);
void fgd_apply_frac(int ndst, FgCsr csr, const double *data, double *out, hipStream_t st);
void fgd_reduce_sum(const double *v, long n, double *partial, double *result, hipStream_t st);
void fgd_xgrid_indices(long nx, const int *x_src, const int *x_dst, const FgTile *tiles_dev, int ntiles, int nx_out,
                       int *t_in, int *i_in, int *j_in, int *i_out, int *j_out, hipStream_t st);

// ---- the sweep with every do_scalar_conserve_interp option (conserve_interp.c:507-910), one level per launch
struct FgApplyEx {
  const double *weight;      // [nsrc] grid_in.weight, or null                       (:574 ...)
  const double *cell_area;   // [nsrc] grid_in.cell_area (sum / cell_measures)
  const double *field_area;  // [nsrc] field_in.area, or null = no cell_measures     (:582-588)
  const double *cell_area_out;  // [ndst] grid_out.cell_area, or null = no --target_grid rescale (:842-869)
  const int *gmask;          // [nsrc] grad_mask or null (= all zero)
  const double *xdata;       // monotone: limited exchange-cell values in CSR order, or null
  double area_missing, missing;
  int has_missing, sum;      // sum = cell_methods == CELL_METHODS_SUM
};
#define FG_XERR_AREA_MISSING 1
#define FG_XERR_ABOVE 2
#define FG_XERR_BELOW 4
void fgd_apply_ex(int order, int ndst, FgCsr csr, const double *f, const double *gx, const double *gy, FgApplyEx o,
                  double *out, double *row_sum, int *err, hipStream_t st, long nx = -1);
// monotone limiter (:617-716): per-source-cell neighbourhood bounds, exchange-cell values + their per-cell
// extremes (atomic min/max: order independent), then the limited values
void fgd_mono_bounds(const FgTile *tiles_dev, int ntiles, int nsrc, const int *src_idx_f, const double *f, double missing,
                     double *fbmax, double *fbmin, double *fmax, double *fmin, hipStream_t st);
void fgd_mono_xdata(long nx, FgCsr csr, const double *f, const double *gx, const double *gy, const int *gmask, double missing,
                    double *xdata, double *fmax, double *fmin, hipStream_t st);
void fgd_mono_limit(long nx, FgCsr csr, const double *f, double missing, const double *fbmax, const double *fbmin,
                    const double *fmax, const double *fmin, double *xdata, int *err, hipStream_t st);

// ---- 1-D x 2-D variants (box_kernels.hip): the regular grid given by its 1-D bounds (device pointers)
struct FgBox { const double *lon, *lat; int nx, ny; };
void fgd_clip_box(int order, FgPairSpace ps, FgBox box, FgTile quad, FgCells S, FgCells D,
                  const double *mask_box, const double *mask_quad, double *tmp_area, double *tmp_clon, double *tmp_clat, int *nacc,
                  unsigned long long *stats, unsigned *err, hipStream_t st);
void fgd_box_area_no_adjust(FgBox box, double *area, hipStream_t st);
void fgd_box_cell_boxes(FgBox box, FgCells c, hipStream_t st);
void fgd_clip_single(const double *lon_in, const double *lat_in, int n_in, double ll_lon, double ll_lat, double ur_lon, double ur_lat,
                     double *lon_out, double *lat_out, int *n_out, hipStream_t st);
void fgd_box_ctr(double ll_lon, double ll_lat, double ur_lon, double ur_lat, double clon, double *out, hipStream_t st);
void fgd_grid_area_no_adjust(int nx, int ny, const double *lon, const double *lat, double *area, hipStream_t st);

void fgd_accumulate_cell_sums(int n, const int *cells, int nsrc, const int *xoff, const double *xa, const double *c1, const double *c2,
                              double *total, hipStream_t st);

// ---- great-circle path (gc_kernels.hip)
// band_keys / band_mode: see k_gc_cell_struct (1: fold this launch's latitude ranges into the keys, 2: cull by them)
void fgd_gc_cell_struct(const FgTileXyz *tiles_dev, int ntiles, int ncells, FgCells c, hipStream_t st,
                        unsigned long long *band_keys = nullptr, int band_mode = 0);
void fgd_gc_clip(FgPairSpace ps, FgCells S, const double *mask, FgCells D,
                 double *tmp_area, int *nacc, int *defer_list, int *defer_cnt, unsigned long long *stats, unsigned *err, hipStream_t st);
// buffers of the three-pass clip (k_gc_screen / k_gc_solve / k_gc_walk, gc_kernels.hip)
struct GcSplit {
  unsigned *meta;          // [pairs] need mask (16) | isInside of the source / destination corners (4 + 4) | corners whose isInside
                           //         k_gc_walk has to take from the angle sum (4 + 4)
  int *tbase;              // [pairs] first task of the pair, -1: not a pair of the three passes (rejected, or on the list)
  unsigned *task;          // [FG_NREG][tcap]  pair << 4 | i1 << 2 | i2; a block of k_gc_screen appends to one region
  double *res;             // [FG_NREG][tcap][2]  (u1, +-u2): sign of the second = inbound 2;  (-1, 0) no intersection;  (2, 0) snapped onto a corner
  unsigned tcap;           // tasks per region
  unsigned *ntask;         // ntask[r * FG_FILL_STRIDE]: tasks handed out in region r (may exceed tcap: the pairs beyond it are on the list)
  int *list, *list_cnt;    // pairs for the one-kernel version (k_gc_clip_list): from k_gc_screen and k_gc_solve, upwards from list[0]
  int *list2_cnt;          // ... from k_gc_walk, downwards from list[list_cap - 1]
  long list_cap;
  // the pairs k_gc_walk works on, listed by k_gc_screen region by region (a pair's own region of the pair list, so a region's list
  // cannot outgrow regcap): pairs with at most two edge-pair tasks from the front of the region, the others from its back -- no lane
  // is spent on a pair the screen rejected (a third of the candidates) and a wave sees pairs of one kind
  int *order;              // [nreg * regcap]
  unsigned *ocnt;          // ocnt[r * FG_FILL_STRIDE] front count, [r * FG_FILL_STRIDE + 1] back count
};
void fgd_gc_clip_split(FgPairSpace ps, FgCells S, const double *mask, FgCells D, double *tmp_area, int *nacc, GcSplit g,
                       unsigned long long *stats, unsigned *err, hipStream_t st, hipStream_t st2, hipEvent_t e1, hipEvent_t e2);
#define FG_GC_POLY_CAP 16
void fgd_gc_clip_batch(int n, const double *a, const double *b, double *out, int *n_out, double *area, hipStream_t st);
void fgd_gc_area_batch(int npoly, int stride_pts, const double *xyz, const int *n, double *area, hipStream_t st);

// ---- batched polygon primitives (poly_kernels.hip); polygons are rows of [npoly][FG_POLY_STRIDE]
#define FG_POLY_STRIDE 24
void fgd_poly_clip(int npoly, const double *lon1, const double *lat1, const int *n1, const double *lon2, const double *lat2,
                   const int *n2, double *lon_out, double *lat_out, int *n_out, hipStream_t st);
void fgd_poly_op(int op, int npoly, double *lon, double *lat, int *n, const double *clon, double *result, hipStream_t st);
// the clipped polygons of exchange cells [k0, k0 + n) of a legacy plan: n_out[n], lon / lat [n][maxv] (rect != null: destination
// cells from the rectilinear tables)
void fgd_xgrid_polygons(long n, const int *x_src, const int *x_dst, FgCells S, FgCells D, const FgRect *rect, int maxv,
                        int *n_out, double *lon_out, double *lat_out, hipStream_t st);
// great-circle plans: the two cells of each exchange cell as [n][12] xyz (for fgd_gc_clip_batch), and [n][16][3] -> three [n][maxv]
void fgd_xgrid_gather_gc(long n, const int *x_src, const int *x_dst, FgCells S, FgCells D, double *a, double *b, hipStream_t st);
void fgd_split_xyz(long n, int maxv, const double *xyz, double *x, double *y, double *z, hipStream_t st);
void fgd_sincos_probe(long n, const double *x, double *s, double *c, hipStream_t st);

// ---- order-2 input preparation (c2l_kernels.hip)
void fgd_pack_interior(const void *tiles, int ntiles, long ncells, long F, int nz, const double *src, double *dst, hipStream_t st);
void fgd_halo_gather(long F, int nz, const int *map, double *data, hipStream_t st);
void fgd_grad_c2l(const void *tiles, int ntiles, long ncells, long F, int nz, const double *data, const double *const *geom,
                  double *grad_x, double *grad_y, hipStream_t st);
void fgd_grad_c2l_rec(const void *tiles, int ntiles, long ncells, long F, int nz, int nb_pad, const double *data,
                      const double *const *geom, double *rec, hipStream_t st);
void fgd_c2l_records(const void *tiles, int ntiles, long ncells, int nz, int nb_pad, const double *src, const int *cell_of,
                     const double *const *geom, double *rec, hipStream_t st);
void fgd_grad_mask(const void *tiles, int ntiles, long ncells, long F, int nz, const double *data, double missing, int *mask, hipStream_t st);
size_t fgd_c2l_tile_size(void);
void fgd_c2l_tile_fill(void *dst, int idx, int nx, int ny, long cell_off, long f_off, long dx_off, long dy_off, long ew_off, long es_off);
