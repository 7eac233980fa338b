/* TEST INFRASTRUCTURE ONLY -- CPU oracle for the 1-D x 2-D exchange-grid variants of libfrencutils:
 *   clip                          tools/libfrencutils/create_xgrid.c:1159-1258 (Sutherland-Hodgman against a lon/lat box)
 *   create_xgrid_1dx2d_order1/2   create_xgrid.c:208-292, :311-389   (source = regular box grid, destination = quads)
 *   create_xgrid_2dx1d_order1/2   create_xgrid.c:414-489, :509-591   (source = quads, destination = regular box grid)
 *   get_grid_area_no_adjust       create_xgrid.c:166-187 ; poly_area_no_adjust  mosaic_util.c:608-634
 *   box_ctrlat / box_ctrlon       create_xgrid.c:2223-2284
 * PINNED: tests/test_oracle_vs_ref.py compares every function bit for bit with oracle/_ref (the reference's own sources).
 * Uses orc_fix_lon / orc_poly_area / orc_poly_ctrlon / orc_poly_ctrlat / orc_get_grid_area of xgrid_oracle.c. */
#include <math.h>
#include <stdlib.h>

#define B_RADIUS 6371000.0
#define B_PI 3.14159265358979323846
#define B_SMALL 1.e-10
#define B_AREA_RATIO_THRESH 1.e-6
#define B_MASK_THRESH 0.5
#define B_MV 50

int orc_fix_lon(double *x, double *y, int n, double tlon);
double orc_poly_area(const double *x, const double *y, int n);
double orc_poly_ctrlat(const double *x, const double *y, int n);
double orc_poly_ctrlon(const double *x, const double *y, int n, double clon);
void orc_get_grid_area(int nx, int ny, const double *lon, const double *lat, double *area);

int orc_clip(const double lon_in[], const double lat_in[], int n_in, double ll_lon, double ll_lat, double ur_lon, double ur_lat,
             double lon_out[], double lat_out[])
{
  double x_tmp[B_MV], y_tmp[B_MV], x_last, y_last;
  int i_in, i_out, n_out, inside_last, inside;
  /* LEFT */
  x_last = lon_in[n_in - 1]; y_last = lat_in[n_in - 1];
  inside_last = (x_last >= ll_lon);
  for (i_in = 0, i_out = 0; i_in < n_in; i_in++) {
    if ((inside = (lon_in[i_in] >= ll_lon)) != inside_last) {
      x_tmp[i_out] = ll_lon;
      y_tmp[i_out++] = y_last + (ll_lon - x_last) * (lat_in[i_in] - y_last) / (lon_in[i_in] - x_last);
    }
    if (inside) { x_tmp[i_out] = lon_in[i_in]; y_tmp[i_out++] = lat_in[i_in]; }
    x_last = lon_in[i_in]; y_last = lat_in[i_in]; inside_last = inside;
  }
  if (!(n_out = i_out)) return 0;
  /* RIGHT */
  x_last = x_tmp[n_out - 1]; y_last = y_tmp[n_out - 1];
  inside_last = (x_last <= ur_lon);
  for (i_in = 0, i_out = 0; i_in < n_out; i_in++) {
    if ((inside = (x_tmp[i_in] <= ur_lon)) != inside_last) {
      lon_out[i_out] = ur_lon;
      lat_out[i_out++] = y_last + (ur_lon - x_last) * (y_tmp[i_in] - y_last) / (x_tmp[i_in] - x_last);
    }
    if (inside) { lon_out[i_out] = x_tmp[i_in]; lat_out[i_out++] = y_tmp[i_in]; }
    x_last = x_tmp[i_in]; y_last = y_tmp[i_in]; inside_last = inside;
  }
  if (!(n_out = i_out)) return 0;
  /* BOTTOM */
  x_last = lon_out[n_out - 1]; y_last = lat_out[n_out - 1];
  inside_last = (y_last >= ll_lat);
  for (i_in = 0, i_out = 0; i_in < n_out; i_in++) {
    if ((inside = (lat_out[i_in] >= ll_lat)) != inside_last) {
      y_tmp[i_out] = ll_lat;
      x_tmp[i_out++] = x_last + (ll_lat - y_last) * (lon_out[i_in] - x_last) / (lat_out[i_in] - y_last);
    }
    if (inside) { x_tmp[i_out] = lon_out[i_in]; y_tmp[i_out++] = lat_out[i_in]; }
    x_last = lon_out[i_in]; y_last = lat_out[i_in]; inside_last = inside;
  }
  if (!(n_out = i_out)) return 0;
  /* TOP */
  x_last = x_tmp[n_out - 1]; y_last = y_tmp[n_out - 1];
  inside_last = (y_last <= ur_lat);
  for (i_in = 0, i_out = 0; i_in < n_out; i_in++) {
    if ((inside = (y_tmp[i_in] <= ur_lat)) != inside_last) {
      lat_out[i_out] = ur_lat;
      lon_out[i_out++] = x_last + (ur_lat - y_last) * (x_tmp[i_in] - x_last) / (y_tmp[i_in] - y_last);
    }
    if (inside) { lon_out[i_out] = x_tmp[i_in]; lat_out[i_out++] = y_tmp[i_in]; }
    x_last = x_tmp[i_in]; y_last = y_tmp[i_in]; inside_last = inside;
  }
  return i_out;
}

double orc_poly_area_no_adjust(const double x[], const double y[], int n)
{
  double area = 0.0;
  for (int i = 0; i < n; i++) {
    int ip = (i + 1) % n;
    double dx = (x[ip] - x[i]);
    double lat1 = y[ip], lat2 = y[i];
    if (dx == 0.0) continue;
    if (fabs(lat1 - lat2) < B_SMALL) area -= dx * sin(0.5 * (lat1 + lat2));
    else area += dx * (cos(lat1) - cos(lat2)) / (lat1 - lat2);
  }
  return area * B_RADIUS * B_RADIUS;
}

void orc_get_grid_area_no_adjust(int nx, int ny, const double *lon, const double *lat, double *area)
{
  const int nxp = nx + 1;
  for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) {
    double x_in[4], y_in[4];
    x_in[0] = lon[j * nxp + i]; x_in[1] = lon[j * nxp + i + 1]; x_in[2] = lon[(j + 1) * nxp + i + 1]; x_in[3] = lon[(j + 1) * nxp + i];
    y_in[0] = lat[j * nxp + i]; y_in[1] = lat[j * nxp + i + 1]; y_in[2] = lat[(j + 1) * nxp + i + 1]; y_in[3] = lat[(j + 1) * nxp + i];
    area[j * nx + i] = orc_poly_area_no_adjust(x_in, y_in, 4);
  }
}

double orc_box_ctrlat(double ll_lon, double ll_lat, double ur_lon, double ur_lat)
{
  double dphi = ur_lon - ll_lon, ctrlat;
  if (dphi > B_PI) dphi = dphi - 2.0 * B_PI;
  if (dphi < -B_PI) dphi = dphi + 2.0 * B_PI;
  ctrlat = dphi * (cos(ur_lat) + ur_lat * sin(ur_lat) - (cos(ll_lat) + ll_lat * sin(ll_lat)));
  return (ctrlat * B_RADIUS * B_RADIUS);
}

double orc_box_ctrlon(double ll_lon, double ll_lat, double ur_lon, double ur_lat, double clon)
{
  double phi1, phi2, dphi, lat1, lat2, dphi1, dphi2, f1, f2, fac, fint, ctrlon = 0.0;
  for (int i = 0; i < 2; i++) {
    if (i == 0) { phi1 = ur_lon; phi2 = ll_lon; lat1 = lat2 = ll_lat; }
    else { phi1 = ll_lon; phi2 = ur_lon; lat1 = lat2 = ur_lat; }
    dphi = phi1 - phi2;
    f1 = 0.5 * (cos(lat1) * sin(lat1) + lat1);
    f2 = 0.5 * (cos(lat2) * sin(lat2) + lat2);
    if (dphi > B_PI) dphi = dphi - 2.0 * B_PI;
    if (dphi < -B_PI) dphi = dphi + 2.0 * B_PI;
    dphi1 = phi1 - clon;
    if (dphi1 > B_PI) dphi1 -= 2.0 * B_PI;
    if (dphi1 < -B_PI) dphi1 += 2.0 * B_PI;
    dphi2 = phi2 - clon;
    if (dphi2 > B_PI) dphi2 -= 2.0 * B_PI;
    if (dphi2 < -B_PI) dphi2 += 2.0 * B_PI;
    if (fabs(dphi2 - dphi1) < B_PI) ctrlon -= dphi * (dphi1 * f1 + dphi2 * f2) / 2.0;
    else {
      if (dphi1 > 0.0) fac = B_PI; else fac = -B_PI;
      fint = f1 + (f2 - f1) * (fac - dphi1) / fabs(dphi);
      ctrlon -= 0.5 * dphi1 * (dphi1 - fac) * f1 - 0.5 * dphi2 * (dphi2 + fac) * f2 + 0.5 * fac * (dphi1 + dphi2) * fint;
    }
  }
  return (ctrlon * B_RADIUS * B_RADIUS);
}

/* One routine for the four variants.  box_is_src != 0: create_xgrid_1dx2d (lon_b/lat_b 1-D bounds of the SOURCE grid,
 * mask on the box cells, loops box-outer); else create_xgrid_2dx1d (box grid is the DESTINATION, mask on the quad cells).
 * Returns nxgrid or -1 if the capacity is exceeded. */
long orc_create_xgrid_box(int box_is_src, int order, int nxb, int nyb, const double *lon_b, const double *lat_b,
                          int nxq, int nyq, const double *lon_q, const double *lat_q, const double *mask, long capacity,
                          int *i_in, int *j_in, int *i_out, int *j_out, double *xgrid_area, double *xgrid_clon, double *xgrid_clat)
{
  const int nxbp = nxb + 1, nxqp = nxq + 1;
  long nxgrid = 0;
  double *area_b = (double *)malloc((size_t)nxb * nyb * sizeof(double)), *area_q = (double *)malloc((size_t)nxq * nyq * sizeof(double));
  double *tmpx = (double *)malloc((size_t)nxbp * (nyb + 1) * sizeof(double)), *tmpy = (double *)malloc((size_t)nxbp * (nyb + 1) * sizeof(double));
  for (int j = 0; j <= nyb; j++) for (int i = 0; i <= nxb; i++) { tmpx[j * nxbp + i] = lon_b[i]; tmpy[j * nxbp + i] = lat_b[j]; }
  if (box_is_src && order == 1 && !(nxb > 1)) orc_get_grid_area_no_adjust(nxb, nyb, tmpx, tmpy, area_b);   /* :239-242 */
  else orc_get_grid_area(nxb, nyb, tmpx, tmpy, area_b);
  orc_get_grid_area(nxq, nyq, lon_q, lat_q, area_q);
  free(tmpx); free(tmpy);
  for (int jb = 0; jb < nyb && nxgrid >= 0; jb++) for (int ib = 0; ib < nxb && nxgrid >= 0; ib++) {
    if (box_is_src && !(mask[jb * nxb + ib] > B_MASK_THRESH)) continue;
    const double ll_lon = lon_b[ib], ll_lat = lat_b[jb], ur_lon = lon_b[ib + 1], ur_lat = lat_b[jb + 1];
    for (int jq = 0; jq < nyq && nxgrid >= 0; jq++) for (int iq = 0; iq < nxq; iq++) {
      if (!box_is_src && !(mask[jq * nxq + iq] > B_MASK_THRESH)) continue;
      double x_in[B_MV], y_in[B_MV], x_out[B_MV], y_out[B_MV];
      y_in[0] = lat_q[jq * nxqp + iq]; y_in[1] = lat_q[jq * nxqp + iq + 1];
      y_in[2] = lat_q[(jq + 1) * nxqp + iq + 1]; y_in[3] = lat_q[(jq + 1) * nxqp + iq];
      if ((y_in[0] <= ll_lat) && (y_in[1] <= ll_lat) && (y_in[2] <= ll_lat) && (y_in[3] <= ll_lat)) continue;
      if ((y_in[0] >= ur_lat) && (y_in[1] >= ur_lat) && (y_in[2] >= ur_lat) && (y_in[3] >= ur_lat)) continue;
      x_in[0] = lon_q[jq * nxqp + iq]; x_in[1] = lon_q[jq * nxqp + iq + 1];
      x_in[2] = lon_q[(jq + 1) * nxqp + iq + 1]; x_in[3] = lon_q[(jq + 1) * nxqp + iq];
      int n_in = orc_fix_lon(x_in, y_in, 4, (ll_lon + ur_lon) / 2), n_out;
      double lon_in_avg = 0;
      if (order == 2) { for (int k = 0; k < n_in; k++) lon_in_avg += x_in[k]; lon_in_avg /= n_in; }   /* avgval_double */
      if ((n_out = orc_clip(x_in, y_in, n_in, ll_lon, ll_lat, ur_lon, ur_lat, x_out, y_out)) > 0) {
        const double m = box_is_src ? mask[jb * nxb + ib] : mask[jq * nxq + iq];
        const double xarea = orc_poly_area(x_out, y_out, n_out) * m;
        const double a1 = area_b[jb * nxb + ib], a2 = area_q[jq * nxq + iq];
        const double min_area = box_is_src ? ((a1 < a2) ? a1 : a2) : ((a2 < a1) ? a2 : a1);   /* min(area_in, area_out) */
        if (xarea / min_area > B_AREA_RATIO_THRESH) {
          if (nxgrid >= capacity) { nxgrid = -1; break; }
          xgrid_area[nxgrid] = xarea;
          if (order == 2) {
            xgrid_clon[nxgrid] = orc_poly_ctrlon(x_out, y_out, n_out, lon_in_avg);
            xgrid_clat[nxgrid] = orc_poly_ctrlat(x_out, y_out, n_out);
          }
          if (box_is_src) { i_in[nxgrid] = ib; j_in[nxgrid] = jb; i_out[nxgrid] = iq; j_out[nxgrid] = jq; }
          else { i_in[nxgrid] = iq; j_in[nxgrid] = jq; i_out[nxgrid] = ib; j_out[nxgrid] = jb; }
          ++nxgrid;
        }
      }
    }
  }
  free(area_b); free(area_q);
  return nxgrid;
}
