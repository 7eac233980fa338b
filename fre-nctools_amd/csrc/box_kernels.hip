// box_kernels.hip -- the 1-D x 2-D exchange-grid variants of libfrencutils for gfx950:
//   create_xgrid_1dx2d_order1/2 (create_xgrid.c:208-292, :311-389)  source = regular lon/lat box grid, destination = quads
//   create_xgrid_2dx1d_order1/2 (create_xgrid.c:414-489, :509-591)  source = quads, destination = box grid
// Both loop box-cell-outer / quad-inner, reject on the quad's corner latitudes, re-centre the quad with
// fix_lon(quad, box centre longitude), clip it against the box with `clip` (create_xgrid.c:1159-1258), and keep
// poly_area * mask when area/min(area_in, area_out) > 1e-6.  Here the box grid (expanded to corner arrays, as the
// reference itself does for its area call, :229-236) plays the "source" of the shared bin/candidate/compaction
// pipeline of xgrid_kernels.hip and the quads the "destination"; this file adds the pair kernel.
#include "xgrid_device.h"
#include "geom.hip.h"

#define BOX_CAP 16

// clip (create_xgrid.c:1159-1258): LEFT, RIGHT, BOTTOM, TOP; same expression trees
__device__ int d_clip_box(const double *lon_in, const double *lat_in, int n_in, double ll_lon, double ll_lat, double ur_lon,
                          double ur_lat, double *lon_out, double *lat_out, bool *overflow)
{
  double x_tmp[BOX_CAP], y_tmp[BOX_CAP], x_last, y_last;
  int i_in, i_out, n_out, inside_last, inside;
  x_last = lon_in[n_in - 1]; y_last = lat_in[n_in - 1];
  inside_last = (x_last >= ll_lon);
  for (i_in = 0, i_out = 0; i_in < n_in; i_in++) {
    if ((inside = (lon_in[i_in] >= ll_lon)) != inside_last) {
      if (i_out < BOX_CAP) { x_tmp[i_out] = ll_lon; y_tmp[i_out] = y_last + (ll_lon - x_last) * (lat_in[i_in] - y_last) / (lon_in[i_in] - x_last); } else *overflow = true;
      i_out++;
    }
    if (inside) { if (i_out < BOX_CAP) { x_tmp[i_out] = lon_in[i_in]; y_tmp[i_out] = lat_in[i_in]; } else *overflow = true; i_out++; }
    x_last = lon_in[i_in]; y_last = lat_in[i_in]; inside_last = inside;
  }
  if (*overflow || !(n_out = i_out)) return 0;
  x_last = x_tmp[n_out - 1]; y_last = y_tmp[n_out - 1];
  inside_last = (x_last <= ur_lon);
  for (i_in = 0, i_out = 0; i_in < n_out; i_in++) {
    if ((inside = (x_tmp[i_in] <= ur_lon)) != inside_last) {
      if (i_out < BOX_CAP) { lon_out[i_out] = ur_lon; lat_out[i_out] = y_last + (ur_lon - x_last) * (y_tmp[i_in] - y_last) / (x_tmp[i_in] - x_last); } else *overflow = true;
      i_out++;
    }
    if (inside) { if (i_out < BOX_CAP) { lon_out[i_out] = x_tmp[i_in]; lat_out[i_out] = y_tmp[i_in]; } else *overflow = true; i_out++; }
    x_last = x_tmp[i_in]; y_last = y_tmp[i_in]; inside_last = inside;
  }
  if (*overflow || !(n_out = i_out)) return 0;
  x_last = lon_out[n_out - 1]; y_last = lat_out[n_out - 1];
  inside_last = (y_last >= ll_lat);
  for (i_in = 0, i_out = 0; i_in < n_out; i_in++) {
    if ((inside = (lat_out[i_in] >= ll_lat)) != inside_last) {
      if (i_out < BOX_CAP) { y_tmp[i_out] = ll_lat; x_tmp[i_out] = x_last + (ll_lat - y_last) * (lon_out[i_in] - x_last) / (lat_out[i_in] - y_last); } else *overflow = true;
      i_out++;
    }
    if (inside) { if (i_out < BOX_CAP) { x_tmp[i_out] = lon_out[i_in]; y_tmp[i_out] = lat_out[i_in]; } else *overflow = true; i_out++; }
    x_last = lon_out[i_in]; y_last = lat_out[i_in]; inside_last = inside;
  }
  if (*overflow || !(n_out = i_out)) return 0;
  x_last = x_tmp[n_out - 1]; y_last = y_tmp[n_out - 1];
  inside_last = (y_last <= ur_lat);
  for (i_in = 0, i_out = 0; i_in < n_out; i_in++) {
    if ((inside = (y_tmp[i_in] <= ur_lat)) != inside_last) {
      if (i_out < BOX_CAP) { lat_out[i_out] = ur_lat; lon_out[i_out] = x_last + (ur_lat - y_last) * (x_tmp[i_in] - x_last) / (y_tmp[i_in] - y_last); } else *overflow = true;
      i_out++;
    }
    if (inside) { if (i_out < BOX_CAP) { lon_out[i_out] = x_tmp[i_in]; lat_out[i_out] = y_tmp[i_in]; } else *overflow = true; i_out++; }
    x_last = x_tmp[i_in]; y_last = y_tmp[i_in]; inside_last = inside;
  }
  if (*overflow) return 0;
  return i_out;
}

// one lane per candidate (box cell s, quad cell d)
template <int ORDER>
__global__ __launch_bounds__(128) void k_clip_box(FgPairSpace ps, FgBox box, FgTile quad, FgCells S, FgCells D,
                                                   const double *mask_box, const double *mask_quad, double *tmp_area, double *tmp_clon,
                                                   double *tmp_clat, int *nacc, unsigned long long *stats, unsigned *err)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (!d_pair_live(ps, p)) return;
  int *pair_dst = ps.dst;
  const int s = ps.src[p], d = pair_dst[p];
  const int ib = s % box.nx, jb = s / box.nx, iq = d % quad.nx, jq = d / quad.nx, nxqp = quad.nx + 1;
  double m = 1.0;
  if (mask_box) m = mask_box[s];
  if (mask_quad) { m = mask_quad[d]; if (!(m > 0.5)) { pair_dst[p] = -1; return; } }      // MASK_THRESH on the quad (2dx1d, :452)
  const double ll_lon = box.lon[ib], ll_lat = box.lat[jb], ur_lon = box.lon[ib + 1], ur_lat = box.lat[jb + 1];
  double x_in[BOX_CAP], y_in[BOX_CAP], x_out[BOX_CAP], y_out[BOX_CAP];
  const int n0 = jq * nxqp + iq, n1 = n0 + 1, n3 = n0 + nxqp, n2 = n3 + 1;
  y_in[0] = quad.lat[n0]; y_in[1] = quad.lat[n1]; y_in[2] = quad.lat[n2]; y_in[3] = quad.lat[n3];
  if ((y_in[0] <= ll_lat) && (y_in[1] <= ll_lat) && (y_in[2] <= ll_lat) && (y_in[3] <= ll_lat)) { pair_dst[p] = -1; return; }
  if ((y_in[0] >= ur_lat) && (y_in[1] >= ur_lat) && (y_in[2] >= ur_lat) && (y_in[3] >= ur_lat)) { pair_dst[p] = -1; return; }
  x_in[0] = quad.lon[n0]; x_in[1] = quad.lon[n1]; x_in[2] = quad.lon[n2]; x_in[3] = quad.lon[n3];
  const int n_in = d_fix_lon(x_in, y_in, 4, (ll_lon + ur_lon) / 2);
  if (n_in > BOX_CAP - 4) { atomicOr(err, G_ERRBIT_OVERFLOW); pair_dst[p] = -1; return; }   // cannot happen for 4 corners (<= 8)
  double lon_in_avg = 0;
  if (ORDER == 2) { for (int k = 0; k < n_in; k++) lon_in_avg += x_in[k]; lon_in_avg /= n_in; }
  bool overflow = false;
  const int n_out = d_clip_box(x_in, y_in, n_in, ll_lon, ll_lat, ur_lon, ur_lat, x_out, y_out, &overflow);
  if (overflow) { atomicOr(err, G_ERRBIT_OVERFLOW); pair_dst[p] = -1; return; }
  if (n_out <= 0) { pair_dst[p] = -1; return; }
  const double xarea = d_poly_area<1>(x_out, y_out, n_out) * m;
  const double a1 = S.area[s], a2 = D.area[d];
  const double min_area = (a1 < a2) ? a1 : a2;
  const double ratio = xarea / min_area;
  if (fabs(ratio - 1.e-6) < 1.e-15) atomicAdd(&stats[FG_STAT_BORDERLINE], 1ull);
  if (ratio > 1.e-6) {
    tmp_area[p] = xarea;
    if (ORDER == 2) { tmp_clon[p] = d_poly_ctrlon<1>(x_out, y_out, n_out, lon_in_avg); tmp_clat[p] = d_poly_ctrlat<1>(x_out, y_out, n_out); }
    atomicAdd(&nacc[s], 1);
  } else {
    pair_dst[p] = -1;
    atomicAdd(&stats[FG_STAT_BELOW], 1ull);
  }
}

// poly_area_no_adjust (mosaic_util.c:608-634) of every box cell: the nx == 1 special case of create_xgrid_1dx2d_order1 (:239-242)
__global__ __launch_bounds__(256) void k_box_area_no_adjust(FgBox box, double *area)
{
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= box.nx * box.ny) return;
  const int ib = s % box.nx, jb = s / box.nx;
  const double x[4] = {box.lon[ib], box.lon[ib + 1], box.lon[ib + 1], box.lon[ib]};
  const double y[4] = {box.lat[jb], box.lat[jb], box.lat[jb + 1], box.lat[jb + 1]};
  double a = 0.0;
  for (int i = 0; i < 4; i++) {
    const int ip = (i + 1) % 4;
    const double dx = (x[ip] - x[i]);
    const double lat1 = y[ip], lat2 = y[i];
    if (dx == 0.0) continue;
    if (fabs(lat1 - lat2) < G_SMALL) a -= dx * d_sin_lat(0.5 * (lat1 + lat2));
    else a += dx * (d_cos_lat(lat1) - d_cos_lat(lat2)) / (lat1 - lat2);
  }
  area[s] = a * G_RADIUS * G_RADIUS;
}

// candidate boxes of the box cells straight from the 1-D bounds (k_cell_struct's fix_lon would fold a cell that spans
// the whole circle, e.g. a single-column zonal grid, to zero width); areas stay those of get_grid_area, as in the reference
__global__ __launch_bounds__(256) void k_box_cell_boxes(FgBox box, FgCells c)
{
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= box.nx * box.ny) return;
  const int ib = s % box.nx, jb = s / box.nx;
  c.lat_min[s] = box.lat[jb]; c.lat_max[s] = box.lat[jb + 1];
  c.lon_min[s] = box.lon[ib]; c.lon_max[s] = box.lon[ib + 1];
  c.lon_avg[s] = (box.lon[ib] + box.lon[ib + 1]) / 2;
  c.nv[s] = 4;
}

static inline int box_nblk(long n, int t) { return (int)((n + t - 1) / t); }

void fgd_clip_box(int order, FgPairSpace ps, FgBox box, FgTile quad, FgCells S, FgCells D,
                  const double *mask_box, const double *mask_quad, double *tmp_area, double *tmp_clon, double *tmp_clat, int *nacc,
                  unsigned long long *stats, unsigned *err, hipStream_t st)
{
  const long np = fgd_pairs_total(ps);
  if (np <= 0) return;
  if (order == 2) k_clip_box<2><<<box_nblk(np, 128), 128, 0, st>>>(ps, box, quad, S, D, mask_box, mask_quad, tmp_area, tmp_clon, tmp_clat, nacc, stats, err);
  else k_clip_box<1><<<box_nblk(np, 128), 128, 0, st>>>(ps, box, quad, S, D, mask_box, mask_quad, tmp_area, tmp_clon, tmp_clat, nacc, stats, err);
}

void fgd_box_area_no_adjust(FgBox box, double *area, hipStream_t st)
{
  const int n = box.nx * box.ny;
  if (n > 0) k_box_area_no_adjust<<<box_nblk(n, 256), 256, 0, st>>>(box, area);
}
void fgd_box_cell_boxes(FgBox box, FgCells c, hipStream_t st)
{
  const int n = box.nx * box.ny;
  if (n > 0) k_box_cell_boxes<<<box_nblk(n, 256), 256, 0, st>>>(box, c);
}

// ---- single-call primitives behind the libfrencutils symbols clip / box_ctrlat / box_ctrlon / get_grid_area_no_adjust
__global__ void k_clip_single(const double *lon_in, const double *lat_in, int n_in, double ll_lon, double ll_lat, double ur_lon,
                              double ur_lat, double *lon_out, double *lat_out, int *n_out)
{
  if (threadIdx.x || blockIdx.x) return;
  double x[BOX_CAP], y[BOX_CAP], xo[BOX_CAP], yo[BOX_CAP];
  for (int k = 0; k < n_in; k++) { x[k] = lon_in[k]; y[k] = lat_in[k]; }
  bool overflow = false;
  int n = d_clip_box(x, y, n_in, ll_lon, ll_lat, ur_lon, ur_lat, xo, yo, &overflow);
  if (overflow) n = -2;
  for (int k = 0; k < n; k++) { lon_out[k] = xo[k]; lat_out[k] = yo[k]; }
  *n_out = n;
}

// box_ctrlat (create_xgrid.c:2223-2232) and box_ctrlon (:2238-2284)
__global__ void k_box_ctr(double ll_lon, double ll_lat, double ur_lon, double ur_lat, double clon, double *out)
{
  if (threadIdx.x || blockIdx.x) return;
  {
    double dphi = ur_lon - ll_lon;
    if (dphi > G_PI) dphi = dphi - 2.0 * G_PI;
    if (dphi < -G_PI) dphi = dphi + 2.0 * G_PI;
    double su, cu, sl, cl;                                   // two sincos() calls in the reference's object code
    d_sincos_lat(ur_lat, &su, &cu);
    d_sincos_lat(ll_lat, &sl, &cl);
    const double ctrlat = dphi * (cu + ur_lat * su - (cl + ll_lat * sl));
    out[0] = ctrlat * G_RADIUS * G_RADIUS;
  }
  double ctrlon = 0.0;
  for (int i = 0; i < 2; i++) {
    double phi1, phi2, lat1, lat2;
    if (i == 0) { phi1 = ur_lon; phi2 = ll_lon; lat1 = lat2 = ll_lat; }
    else { phi1 = ll_lon; phi2 = ur_lon; lat1 = lat2 = ur_lat; }
    double dphi = phi1 - phi2;
    double s1, c1;                                           // lat1 == lat2: one sincos() per pass in the reference
    d_sincos_lat(lat1, &s1, &c1);
    const double f1 = 0.5 * (c1 * s1 + lat1);
    const double f2 = 0.5 * (c1 * s1 + lat2);
    if (dphi > G_PI) dphi = dphi - 2.0 * G_PI;
    if (dphi < -G_PI) dphi = dphi + 2.0 * G_PI;
    double dphi1 = phi1 - clon;
    if (dphi1 > G_PI) dphi1 -= 2.0 * G_PI;
    if (dphi1 < -G_PI) dphi1 += 2.0 * G_PI;
    double dphi2 = phi2 - clon;
    if (dphi2 > G_PI) dphi2 -= 2.0 * G_PI;
    if (dphi2 < -G_PI) dphi2 += 2.0 * G_PI;
    if (fabs(dphi2 - dphi1) < G_PI) ctrlon -= dphi * (dphi1 * f1 + dphi2 * f2) / 2.0;
    else {
      const double fac = (dphi1 > 0.0) ? G_PI : -G_PI;
      const double fint = f1 + (f2 - f1) * (fac - dphi1) / fabs(dphi);
      ctrlon -= 0.5 * dphi1 * (dphi1 - fac) * f1 - 0.5 * dphi2 * (dphi2 + fac) * f2 + 0.5 * fac * (dphi1 + dphi2) * fint;
    }
  }
  out[1] = ctrlon * G_RADIUS * G_RADIUS;
}

// get_grid_area_no_adjust (create_xgrid.c:166-187) on corner arrays
__global__ __launch_bounds__(256) void k_grid_area_no_adjust(int nx, int ny, const double *lon, const double *lat, double *area)
{
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nx * ny) return;
  const int i = s % nx, j = s / nx, nxp = nx + 1;
  const int n0 = j * nxp + i, n1 = n0 + 1, n3 = n0 + nxp, n2 = n3 + 1;
  const double x[4] = {lon[n0], lon[n1], lon[n2], lon[n3]}, y[4] = {lat[n0], lat[n1], lat[n2], lat[n3]};
  double a = 0.0;
  for (int k = 0; k < 4; k++) {
    const int kp = (k + 1) % 4;
    const double dx = (x[kp] - x[k]);
    const double lat1 = y[kp], lat2 = y[k];
    if (dx == 0.0) continue;
    if (fabs(lat1 - lat2) < G_SMALL) a -= dx * d_sin_lat(0.5 * (lat1 + lat2));
    else a += dx * (d_cos_lat(lat1) - d_cos_lat(lat2)) / (lat1 - lat2);
  }
  area[s] = a * G_RADIUS * G_RADIUS;
}

void fgd_clip_single(const double *lon_in, const double *lat_in, int n_in, double ll_lon, double ll_lat, double ur_lon, double ur_lat,
                     double *lon_out, double *lat_out, int *n_out, hipStream_t st)
{
  k_clip_single<<<1, 1, 0, st>>>(lon_in, lat_in, n_in, ll_lon, ll_lat, ur_lon, ur_lat, lon_out, lat_out, n_out);
}
void fgd_box_ctr(double ll_lon, double ll_lat, double ur_lon, double ur_lat, double clon, double *out, hipStream_t st)
{
  k_box_ctr<<<1, 1, 0, st>>>(ll_lon, ll_lat, ur_lon, ur_lat, clon, out);
}
void fgd_grid_area_no_adjust(int nx, int ny, const double *lon, const double *lat, double *area, hipStream_t st)
{
  if (nx * ny > 0) k_grid_area_no_adjust<<<box_nblk((long)nx * ny, 256), 256, 0, st>>>(nx, ny, lon, lat, area);
}
