// poly_kernels.hip -- batched spherical-polygon primitives on the device, behind the B1 symbols
// clip_2dx2d / poly_area / poly_ctrlon / poly_ctrlat / fix_lon (create_xgrid.h:35-46,
// mosaic_util.h).  One lane per polygon (pair); polygons live in per-lane private arrays --
// these entry points exist for drop-in completeness and for known-answer tests of the clip
// arithmetic on arbitrary (non-quad) polygons; the search kernels in xgrid_kernels.hip are the
// fast path.
#include "xgrid_device.h"
#include "geom.hip.h"

#define PB_IN   12     // max vertices of an input polygon
#define PB_CAP  24     // max vertices of a clipped polygon

// create_xgrid.c:1266-1341 on private arrays.  Returns n_out, -1 on parallel edges, -2 on overflow.
__device__ inline int d_clip_private(const double *lon1, const double *lat1, int n1,
                                     const double *lon2, const double *lat2, int n2,
                                     double *lon_out, double *lat_out)
{
  double cx[PB_CAP], cy[PB_CAP], ex[PB_IN], ey[PB_IN];
  bool wrap = false;
  for (int k = 0; k < n1; k++) { cx[k] = lon1[k]; cy[k] = lat1[k]; if (cx[k] > G_TPI || cx[k] < 0.0) wrap = true; }
  for (int k = 0; k < n2; k++) { ex[k] = lon2[k]; ey[k] = lat2[k]; }
  if (wrap) {
    for (int k = 0; k < n1; k++) cx[k] = d_pimod1(cx[k]);
    for (int k = 0; k < n2; k++) ex[k] = d_pimod1(ex[k]);
  }
  int n_cur = n1;
  double x2_0 = ex[n2 - 1], y2_0 = ey[n2 - 1];
  for (int e = 0; e < n2; e++) {
    double x2_1 = ex[e], y2_1 = ey[e];
    double x1_0 = cx[n_cur - 1], y1_0 = cy[n_cur - 1];
    int inside_last = d_inside_edge(x2_0, y2_0, x2_1, y2_1, x1_0, y1_0);
    int n_new = 0;
    for (int k = 0; k < n_cur; k++) {
      double x1_1 = cx[k], y1_1 = cy[k];
      int inside = d_inside_edge(x2_0, y2_0, x2_1, y2_1, x1_1, y1_1);
      if (inside != inside_last) {
        double dy1 = y1_1 - y1_0;
        double dy2 = y2_1 - y2_0;
        double dx1 = x1_1 - x1_0;
        double dx2 = x2_1 - x2_0;
        double ds1 = y1_0 * x1_1 - y1_1 * x1_0;
        double ds2 = y2_0 * x2_1 - y2_1 * x2_0;
        double determ = dy2 * dx1 - dy1 * dx2;
        if (fabs(determ) < 1.0e-30) return -1;
        if (n_new >= PB_CAP) return -2;
        lon_out[n_new] = (dx2 * ds1 - dx1 * ds2) / determ;
        lat_out[n_new++] = (dy2 * ds1 - dy1 * ds2) / determ;
      }
      if (inside) {
        if (n_new >= PB_CAP) return -2;
        lon_out[n_new] = x1_1; lat_out[n_new++] = y1_1;
      }
      x1_0 = x1_1; y1_0 = y1_1; inside_last = inside;
    }
    n_cur = n_new;
    if (!n_cur) return 0;
    for (int k = 0; k < n_cur; k++) { cx[k] = lon_out[k]; cy[k] = lat_out[k]; }
    x2_0 = x2_1; y2_0 = y2_1;
  }
  return n_cur;
}

// polygons are rows of [npoly][PB_CAP] arrays
__global__ __launch_bounds__(64) void k_poly_clip(int npoly, const double *lon1, const double *lat1, const int *n1,
                                                   const double *lon2, const double *lat2, const int *n2,
                                                   double *lon_out, double *lat_out, int *n_out)
{
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npoly) return;
  double a[PB_IN], b[PB_IN], c[PB_IN], d[PB_IN], xo[PB_CAP], yo[PB_CAP];
  int m1 = n1[p], m2 = n2[p];
  if (m1 < 1 || m1 > PB_IN || m2 < 1 || m2 > PB_IN) { n_out[p] = -2; return; }
  for (int k = 0; k < m1; k++) { a[k] = lon1[(size_t)p * PB_CAP + k]; b[k] = lat1[(size_t)p * PB_CAP + k]; }
  for (int k = 0; k < m2; k++) { c[k] = lon2[(size_t)p * PB_CAP + k]; d[k] = lat2[(size_t)p * PB_CAP + k]; }
  int n = d_clip_private(a, b, m1, c, d, m2, xo, yo);
  n_out[p] = n;
  for (int k = 0; k < PB_CAP; k++) {
    lon_out[(size_t)p * PB_CAP + k] = (k < n) ? xo[k] : 0.0;
    lat_out[(size_t)p * PB_CAP + k] = (k < n) ? yo[k] : 0.0;
  }
}

// op 0: poly_area, 1: poly_ctrlon (clon[p]), 2: poly_ctrlat, 3: fix_lon in place (tlon = clon[p]; n updated)
__global__ __launch_bounds__(64) void k_poly_op(int op, int npoly, double *lon, double *lat, int *n, const double *clon,
                                                 double *result)
{
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npoly) return;
  double x[PB_CAP], y[PB_CAP];
  int m = n[p];
  if (m < 1 || m > PB_CAP || (op == 3 && m > G_FIXCAP - 4)) { if (result) result[p] = 0; if (op == 3) n[p] = -2; return; }
  for (int k = 0; k < m; k++) { x[k] = lon[(size_t)p * PB_CAP + k]; y[k] = lat[(size_t)p * PB_CAP + k]; }
  if (op == 0) result[p] = d_poly_area<1>(x, y, m);
  else if (op == 1) result[p] = d_poly_ctrlon<1>(x, y, m, clon[p]);
  else if (op == 2) result[p] = d_poly_ctrlat<1>(x, y, m);
  else {
    int mm = d_fix_lon(x, y, m, clon[p]);
    n[p] = mm;
    for (int k = 0; k < PB_CAP; k++) {
      lon[(size_t)p * PB_CAP + k] = (k < mm) ? x[k] : 0.0;
      lat[(size_t)p * PB_CAP + k] = (k < mm) ? y[k] : 0.0;
    }
  }
}

void fgd_poly_clip(int npoly, const double *lon1, const double *lat1, const int *n1, const double *lon2, const double *lat2,
                   const int *n2, double *lon_out, double *lat_out, int *n_out, hipStream_t st)
{
  if (npoly > 0) k_poly_clip<<<(npoly + 63) / 64, 64, 0, st>>>(npoly, lon1, lat1, n1, lon2, lat2, n2, lon_out, lat_out, n_out);
}
void fgd_poly_op(int op, int npoly, double *lon, double *lat, int *n, const double *clon, double *result, hipStream_t st)
{
  if (npoly > 0) k_poly_op<<<(npoly + 63) / 64, 64, 0, st>>>(op, npoly, lon, lat, n, clon, result);
}

// sin/cos of an array with the device's latitude trig (parity probe: must equal the host libm bit for bit)
__global__ __launch_bounds__(256) void k_sincos_probe(long n, const double *x, double *s, double *c)
{
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  s[i] = d_sin_lat(x[i]); c[i] = d_cos_lat(x[i]);
}
void fgd_sincos_probe(long n, const double *x, double *s, double *c, hipStream_t st)
{
  if (n > 0) k_sincos_probe<<<(int)((n + 255) / 256), 256, 0, st>>>(n, x, s, c);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The clipped polygon of every exchange cell of a plan (fg_plan_get_polygons): what clip_2dx2d returned inside create_xgrid for the
// pair -- source cell after fix_lon(pi), destination cell after fix_lon(pi) and the +-2pi shift towards the source cell's mean
// longitude (create_xgrid.c:1062-1079), clip_2dx2d's own wrap (:1282-1290).  make_coupler_mosaic keeps these vertices
// (atmxlnd_x / _y, make_coupler_mosaic.c:1560-1577) to clip them against the ocean grid.  One lane per exchange cell.
__global__ __launch_bounds__(64) void k_xgrid_polygons(long n, const int *x_src, const int *x_dst, FgCells S, FgCells D, FgRect R, int rect,
                                                        int maxv, int *n_out, double *lon_out, double *lat_out)
{
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int s = x_src[k], d = x_dst[k];
  double a[PB_IN], b[PB_IN], c[PB_IN], e[PB_IN], xo[PB_CAP], yo[PB_CAP];
  const int n1 = S.nv[s];
  int n2;
  const double *sv = S.verts + (size_t)s * 16;
  for (int q = 0; q < 8; q++) { a[q] = sv[q]; b[q] = sv[8 + q]; }
  double lon_out_avg;
  if (rect) {
    const int j = d / R.nx, i = d - j * R.nx;
    const double *cr = R.col + (size_t)i * RECT_COLW;
    const double ya = R.lat_ax[j], yb = R.lat_ax[j + 1];
    c[0] = cr[0]; c[1] = cr[1]; c[2] = cr[2]; c[3] = cr[3]; lon_out_avg = cr[6];
    e[0] = ya; e[1] = ya; e[2] = yb; e[3] = yb;
    n2 = 4;
  } else {
    const double *dv = D.verts + (size_t)d * 16;
    for (int q = 0; q < 8; q++) { c[q] = dv[q]; e[q] = dv[8 + q]; }
    lon_out_avg = D.lon_avg[d];
    n2 = D.nv[d];
  }
  const double dx = lon_out_avg - S.lon_avg[s];
  const double shift = (dx < -G_PI) ? G_TPI : ((dx > G_PI) ? -G_TPI : 0.0);
  if (shift != 0.0) for (int q = 0; q < n2; q++) c[q] += shift;
  int m = (n1 < 1 || n2 < 1) ? 0 : d_clip_private(a, b, n1, c, e, n2, xo, yo);
  n_out[k] = m;
  for (int q = 0; q < maxv; q++) {
    lon_out[(size_t)k * maxv + q] = (q < m && q < PB_CAP) ? xo[q] : 0.0;
    lat_out[(size_t)k * maxv + q] = (q < m && q < PB_CAP) ? yo[q] : 0.0;
  }
}
void fgd_xgrid_polygons(long n, const int *x_src, const int *x_dst, FgCells S, FgCells D, const FgRect *rect, int maxv,
                        int *n_out, double *lon_out, double *lat_out, hipStream_t st)
{
  if (n > 0) k_xgrid_polygons<<<(unsigned)((n + 63) / 64), 64, 0, st>>>(n, x_src, x_dst, S, D, rect ? *rect : FgRect{}, rect ? 1 : 0, maxv,
                                                                      n_out, lon_out, lat_out);
}
__global__ __launch_bounds__(256) void k_xgrid_gather_gc(long n, const int *x_src, const int *x_dst, FgCells S, FgCells D, double *a, double *b)
{
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * 12) return;
  const long k = t / 12; const int q = (int)(t - k * 12);
  a[t] = S.verts[(size_t)x_src[k] * 16 + q];
  b[t] = D.verts[(size_t)x_dst[k] * 16 + q];
}
void fgd_xgrid_gather_gc(long n, const int *x_src, const int *x_dst, FgCells S, FgCells D, double *a, double *b, hipStream_t st)
{
  if (n > 0) k_xgrid_gather_gc<<<(unsigned)((n * 12 + 255) / 256), 256, 0, st>>>(n, x_src, x_dst, S, D, a, b);
}
__global__ __launch_bounds__(256) void k_split_xyz(long n, int maxv, const double *xyz, double *x, double *y, double *z)
{
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * maxv) return;
  const long k = t / maxv; const int q = (int)(t - k * maxv);
  const bool in = q < FG_GC_POLY_CAP;
  x[t] = in ? xyz[((size_t)k * FG_GC_POLY_CAP + q) * 3] : 0.0;
  y[t] = in ? xyz[((size_t)k * FG_GC_POLY_CAP + q) * 3 + 1] : 0.0;
  z[t] = in ? xyz[((size_t)k * FG_GC_POLY_CAP + q) * 3 + 2] : 0.0;
}
void fgd_split_xyz(long n, int maxv, const double *xyz, double *x, double *y, double *z, hipStream_t st)
{
  if (n > 0) k_split_xyz<<<(unsigned)((n * maxv + 255) / 256), 256, 0, st>>>(n, maxv, xyz, x, y, z);
}
