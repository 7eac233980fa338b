/*
 * c2l_host.c -- host-side, once-per-tile set-up for the order-2 gradient of cubed-sphere fields
 * (SURVEY.md §8f-1).  Plain C + libm, same operation order as the reference so the results are bit-identical
 * on the same libm (checked against oracle/_ref in tests/test_c2l_cpu.py):
 *
 *   fg_c2l_grid_info      calc_c2l_grid_info  tools/libfrencutils/gradient_c2l.c:368-454 (get_edge :196-311,
 *                         mid_pt_sphere :313-335) with great_circle_distance / spherical_excess_area /
 *                         spherical_angle / unit_vect_latlon of tools/libfrencutils/mosaic_util.c:747,846,800,937
 *   fg_find_contacts      line contacts between tiles in the convention read_mosaic_contact hands to fregrid
 *                         (tools/libfrencutils/read_mosaic.c:655-777: 0-based model indices, i == const for
 *                         west/east edges); found geometrically instead of read from a mosaic file
 *   fg_halo_map           setup_boundary + update_halo (tools/fregrid/fregrid_util.c:2446-2560, :2614-2658) for
 *                         CENTER data with halo 1, folded into one gather index per halo cell
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "fregrid_hip.h"

#define C_RADIUS 6371000.0
#define C_PI 3.14159265358979323846

/* latlon2xyz of the reference is five separate libm calls (no sincos fusion, see grid_gen.c) */
static double __attribute__((noinline)) sin_sep(double x) { return sin(x); }
static double __attribute__((noinline)) cos_sep(double x) { return cos(x); }
static void ll2xyz(double lon, double lat, double *v)
{
  v[0] = cos_sep(lat) * cos_sep(lon);
  v[1] = cos_sep(lat) * sin_sep(lon);
  v[2] = sin_sep(lat);
}
static void xyz2ll(const double *e, double *lon, double *lat)
{
  double xx = e[0], yy = e[1], zz = e[2];
  double dist = sqrt(xx * xx + yy * yy + zz * zz);
  xx /= dist; yy /= dist; zz /= dist;
  if (fabs(xx) + fabs(yy) < 1.e-10) *lon = 0;
  else *lon = atan2(yy, xx);
  *lat = asin(zz);
  if (*lon < 0.) *lon = 2. * C_PI + *lon;
}
static double gc_dist(const double *p1, const double *p2)
{
  double beta = 2. * asin(sqrt(sin((p1[1] - p2[1]) / 2.) * sin((p1[1] - p2[1]) / 2.) +
                               cos(p1[1]) * cos(p2[1]) * (sin((p1[0] - p2[0]) / 2.) * sin((p1[0] - p2[0]) / 2.))));
  return C_RADIUS * beta;
}
static double sph_angle(const double *v1, const double *v2, const double *v3)
{
  double angle;
  double px = v1[1] * v2[2] - v1[2] * v2[1];
  double py = v1[2] * v2[0] - v1[0] * v2[2];
  double pz = v1[0] * v2[1] - v1[1] * v2[0];
  double qx = v1[1] * v3[2] - v1[2] * v3[1];
  double qy = v1[2] * v3[0] - v1[0] * v3[2];
  double qz = v1[0] * v3[1] - v1[1] * v3[0];
  double ddd = (px * px + py * py + pz * pz) * (qx * qx + qy * qy + qz * qz);
  if (ddd <= 0.0) angle = 0.;
  else {
    ddd = (px * qx + py * qy + pz * qz) / sqrt(ddd);
    if (fabs(ddd - 1) < 1.e-30) ddd = 1;
    if (fabs(ddd + 1) < 1.e-30) ddd = -1;
    if (ddd > 1. || ddd < -1.) angle = (ddd < 0.) ? C_PI : 0.;
    else angle = acosl(ddd);
  }
  return angle;
}
static double excess_area(const double *p_ll, const double *p_ul, const double *p_lr, const double *p_ur)
{
  double v1[3], v2[3], v3[3], a1, a2, a3, a4;
  ll2xyz(p_ll[0], p_ll[1], v1); ll2xyz(p_lr[0], p_lr[1], v2); ll2xyz(p_ul[0], p_ul[1], v3); a1 = sph_angle(v1, v2, v3);
  ll2xyz(p_lr[0], p_lr[1], v1); ll2xyz(p_ur[0], p_ur[1], v2); ll2xyz(p_ll[0], p_ll[1], v3); a2 = sph_angle(v1, v2, v3);
  ll2xyz(p_ur[0], p_ur[1], v1); ll2xyz(p_ul[0], p_ul[1], v2); ll2xyz(p_lr[0], p_lr[1], v3); a3 = sph_angle(v1, v2, v3);
  ll2xyz(p_ul[0], p_ul[1], v1); ll2xyz(p_ur[0], p_ur[1], v2); ll2xyz(p_ll[0], p_ll[1], v3); a4 = sph_angle(v1, v2, v3);
  return (a1 + a2 + a3 + a4 - 2. * C_PI) * C_RADIUS * C_RADIUS;
}
static void cross_unit(const double *p1, const double *p2, double *e)
{
  e[0] = p1[1] * p2[2] - p1[2] * p2[1];
  e[1] = p1[2] * p2[0] - p1[0] * p2[2];
  e[2] = p1[0] * p2[1] - p1[1] * p2[0];
  double pdot = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
  pdot = sqrt(pdot);
  for (int k = 0; k < 3; k++) e[k] /= pdot;
}
static void mid_pt(const double *p1, const double *p2, double *pm)
{
  double e1[3], e2[3], e[3];
  ll2xyz(p1[0], p1[1], e1); ll2xyz(p2[0], p2[1], e2);
  e[0] = e1[0] + e2[0]; e[1] = e1[1] + e2[1]; e[2] = e1[2] + e2[2];
  double dd = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
  e[0] /= dd; e[1] /= dd; e[2] /= dd;
  xyz2ll(e, pm, pm + 1);
}

/* xt, yt: T-cell centres with a halo of 1, [(ny+2)][(nx+2)]; xc, yc: corners [(ny+1)][(nx+1)].
 * All four tile edges are treated as cubed-sphere edges (fregrid passes is_true for all, fregrid_util.c:340-345). */
int fg_c2l_grid_info(int nx, int ny, const double *xt, const double *yt, const double *xc, const double *yc,
                     double *dx, double *dy, double *area, double *edge_w, double *edge_e, double *edge_s,
                     double *edge_n, double *en_n, double *en_e, double *vlon, double *vlat)
{
  if (nx < 1 || ny < 1) return FG_ERR_ARG;
  const int nxp = nx + 1, nyp = ny + 1;
  double p1[3], p2[3], p3[3], p4[3];
  for (int j = 0; j < nyp; j++) for (int i = 0; i < nx; i++) {
    p1[0] = xc[j * nxp + i]; p1[1] = yc[j * nxp + i];
    p2[0] = xc[j * nxp + i + 1]; p2[1] = yc[j * nxp + i + 1];
    dx[j * nx + i] = gc_dist(p1, p2);
  }
  for (int j = 0; j < ny; j++) for (int i = 0; i < nxp; i++) {
    p1[0] = xc[j * nxp + i]; p1[1] = yc[j * nxp + i];
    p2[0] = xc[(j + 1) * nxp + i]; p2[1] = yc[(j + 1) * nxp + i];
    dy[j * nxp + i] = gc_dist(p1, p2);
  }
  for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) {
    p1[0] = xc[j * nxp + i]; p1[1] = yc[j * nxp + i];
    p2[0] = xc[(j + 1) * nxp + i]; p2[1] = yc[(j + 1) * nxp + i];
    p3[0] = xc[j * nxp + i + 1]; p3[1] = yc[j * nxp + i + 1];
    p4[0] = xc[(j + 1) * nxp + i + 1]; p4[1] = yc[(j + 1) * nxp + i + 1];
    area[j * nx + i] = excess_area(p1, p2, p3, p4);
  }
  double *x = (double *)malloc(sizeof(double) * nxp * nyp * 3);
  if (!x) return FG_ERR_HIP;
  for (int k = 0; k < nxp * nyp; k++) ll2xyz(xc[k], yc[k], x + 3 * k);
  for (int j = 0; j < nyp; j++) for (int i = 0; i < nx; i++)
    cross_unit(x + 3 * (j * nxp + i), x + 3 * (j * nxp + i + 1), en_n + 3 * (j * nx + i));
  for (int j = 0; j < ny; j++) for (int i = 0; i < nxp; i++)
    cross_unit(x + 3 * ((j + 1) * nxp + i), x + 3 * (j * nxp + i), en_e + 3 * (j * nxp + i));
  free(x);
  for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) {
    double lon = xt[(j + 1) * (nx + 2) + i + 1], lat = yt[(j + 1) * (nx + 2) + i + 1];
    double sin_lon = sin(lon), cos_lon = cos(lon), sin_lat = sin(lat), cos_lat = cos(lat);
    int n = j * nx + i;
    vlon[3 * n] = -sin_lon; vlon[3 * n + 1] = cos_lon; vlon[3 * n + 2] = 0.;
    vlat[3 * n] = -sin_lat * cos_lon; vlat[3 * n + 1] = -sin_lat * sin_lon; vlat[3 * n + 2] = cos_lat;
  }
  /* get_edge with all four edges "on edge": istart = jstart = 1, iend = nx, jend = ny */
  for (int i = 0; i < nxp; i++) { edge_s[i] = 0.5; edge_n[i] = 0.5; }
  for (int j = 0; j < nyp; j++) { edge_w[j] = 0.5; edge_e[j] = 0.5; }
  double *px = (double *)malloc(2 * (nx + 2) * sizeof(double)), *py = (double *)malloc(2 * (ny + 2) * sizeof(double));
  if (!px || !py) { free(px); free(py); return FG_ERR_HIP; }
  const int istart = 1, iend = nx, jstart = 1, jend = ny;
  for (int side = 0; side < 2; side++) {              /* west (i = 0), east (i = nx) */
    int i = side ? nx : 0;
    double *edge = side ? edge_e : edge_w;
    for (int j = jstart; j <= jend; j++) {
      p1[0] = xt[j * (nx + 2) + i]; p1[1] = yt[j * (nx + 2) + i];
      p2[0] = xt[j * (nx + 2) + i + 1]; p2[1] = yt[j * (nx + 2) + i + 1];
      mid_pt(p1, p2, py + 2 * j);
    }
    for (int j = jstart; j < jend; j++) {
      p1[0] = xc[j * nxp + i]; p1[1] = yc[j * nxp + i];
      double d1 = gc_dist(py + 2 * j, p1), d2 = gc_dist(py + 2 * (j + 1), p1);
      edge[j] = d2 / (d1 + d2);
    }
  }
  for (int side = 0; side < 2; side++) {              /* south (j = 0), north (j = ny) */
    int j = side ? ny : 0;
    double *edge = side ? edge_n : edge_s;
    for (int i = istart; i <= iend; i++) {
      p1[0] = xt[j * (nx + 2) + i]; p1[1] = yt[j * (nx + 2) + i];
      p2[0] = xt[(j + 1) * (nx + 2) + i]; p2[1] = yt[(j + 1) * (nx + 2) + i];
      mid_pt(p1, p2, px + 2 * i);
    }
    for (int i = istart; i < iend; i++) {
      p1[0] = xc[j * nxp + i]; p1[1] = yc[j * nxp + i];
      double d1 = gc_dist(px + 2 * i, p1), d2 = gc_dist(px + 2 * (i + 1), p1);
      edge[i] = d2 / (d1 + d2);
    }
  }
  free(px); free(py);
  return 0;
}

/* ------------------------------------------------------------------------------------------------ contacts */
enum { D_WEST = 0, D_EAST = 1, D_SOUTH = 2, D_NORTH = 3 };

/* corner (lon, lat) number k along edge `dir` of a tile (k = 0..n, n = ny for W/E, nx for S/N) */
static void edge_pt(int nx, int ny, const double *lon, const double *lat, int dir, int k, double *v)
{
  int i, j;
  if (dir == D_WEST) { i = 0; j = k; } else if (dir == D_EAST) { i = nx; j = k; }
  else if (dir == D_SOUTH) { i = k; j = 0; } else { i = k; j = ny; }
  ll2xyz(lon[j * (nx + 1) + i], lat[j * (nx + 1) + i], v);
}
static int same_pt(const double *a, const double *b)
{
  return fabs(a[0] - b[0]) + fabs(a[1] - b[1]) + fabs(a[2] - b[2]) < 1.e-9;
}

/* Whole-edge line contacts between different tiles.  Output arrays hold one entry per contact, in the
 * 0-based model-index convention of read_mosaic_contact: an edge at constant i has istart == iend (0 for
 * west, nx-1 for east), the other pair runs 0..n-1 or n-1..0 (reversed orientation). */
int fg_find_contacts(int ntiles, const int *nx, const int *ny, const double *const *lonc, const double *const *latc,
                     int max_contacts, int *tile1, int *tile2, int *istart1, int *iend1, int *jstart1, int *jend1,
                     int *istart2, int *iend2, int *jstart2, int *jend2)
{
  int nc = 0;
  for (int a = 0; a < ntiles; a++)
    for (int da = 0; da < 4; da++) {
      int na = (da <= D_EAST) ? ny[a] : nx[a];
      double a0[3], a1[3];
      edge_pt(nx[a], ny[a], lonc[a], latc[a], da, 0, a0);
      edge_pt(nx[a], ny[a], lonc[a], latc[a], da, na, a1);
      for (int b = a + 1; b < ntiles; b++)
        for (int db = 0; db < 4; db++) {
          int nb = (db <= D_EAST) ? ny[b] : nx[b];
          if (nb != na) continue;
          double b0[3], b1[3];
          edge_pt(nx[b], ny[b], lonc[b], latc[b], db, 0, b0);
          edge_pt(nx[b], ny[b], lonc[b], latc[b], db, nb, b1);
          int fwd = same_pt(a0, b0) && same_pt(a1, b1), rev = same_pt(a0, b1) && same_pt(a1, b0);
          if (!fwd && !rev) continue;
          int ok = 1;                                   /* every interior corner must coincide too */
          for (int k = 1; k < na && ok; k++) {
            double pa[3], pb[3];
            edge_pt(nx[a], ny[a], lonc[a], latc[a], da, k, pa);
            edge_pt(nx[b], ny[b], lonc[b], latc[b], db, fwd ? k : nb - k, pb);
            ok = same_pt(pa, pb);
          }
          if (!ok) continue;
          if (nc >= max_contacts) return FG_ERR_CAPACITY;
          tile1[nc] = a + 1; tile2[nc] = b + 1;
          int *is[2] = {istart1 + nc, istart2 + nc}, *ie[2] = {iend1 + nc, iend2 + nc};
          int *js[2] = {jstart1 + nc, jstart2 + nc}, *je[2] = {jend1 + nc, jend2 + nc};
          int dd[2] = {da, db}, tt[2] = {a, b};
          for (int s = 0; s < 2; s++) {
            int lo = 0, hi = ((dd[s] <= D_EAST) ? ny[tt[s]] : nx[tt[s]]) - 1;
            if (s == 1 && rev) { int t = lo; lo = hi; hi = t; }
            if (dd[s] == D_WEST)       { *is[s] = 0; *ie[s] = 0; *js[s] = lo; *je[s] = hi; }
            else if (dd[s] == D_EAST)  { *is[s] = nx[tt[s]] - 1; *ie[s] = nx[tt[s]] - 1; *js[s] = lo; *je[s] = hi; }
            else if (dd[s] == D_SOUTH) { *js[s] = 0; *je[s] = 0; *is[s] = lo; *ie[s] = hi; }
            else                       { *js[s] = ny[tt[s]] - 1; *je[s] = ny[tt[s]] - 1; *is[s] = lo; *ie[s] = hi; }
          }
          nc++;
        }
    }
  return nc;
}

/* direction of one contact side: get_contact_direction, fregrid_util.c:2420-2442 */
static int side_dir(int istart, int iend, int jstart, int jend)
{
  if (istart == iend && jstart == jend) return -1;
  if (istart != iend && jstart != jend) return -1;
  if (istart == iend) return (istart == 0) ? D_WEST : D_EAST;
  return (jstart == 0) ? D_SOUTH : D_NORTH;
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* For CENTER data with halo 1: map[t][(ny+2)*(nx+2)] = flat index (within the concatenation of the halo'd
 * tiles) of the interior cell whose value a halo cell receives, or -1 (interior cells, halo corners, edges
 * without a contact).  map_off[t] = offset of tile t in that concatenation (ntiles+1 entries).
 * Rotation rules and index ranges: setup_boundary (position CENTER => shift 0) and update_halo. */
int fg_halo_map(int ntiles, const int *nx, const int *ny, int ncontacts, const int *tile1, const int *tile2,
                const int *istart1, const int *iend1, const int *jstart1, const int *jend1,
                const int *istart2, const int *iend2, const int *jstart2, const int *jend2,
                long *map_off, int *map)
{
  const int halo = 1;
  long off = 0;
  for (int t = 0; t < ntiles; t++) { map_off[t] = off; off += (long)(nx[t] + 2) * (ny[t] + 2); }
  map_off[ntiles] = off;
  for (long k = 0; k < off; k++) map[k] = -1;
  /* both sides of every contact: entries l and l+ncontacts as in setup_boundary (:2463-2466) */
  int n2 = 2 * ncontacts;
  int *tile = (int *)malloc(n2 * sizeof(int)), *is = (int *)malloc(n2 * sizeof(int)), *ie = (int *)malloc(n2 * sizeof(int));
  int *js = (int *)malloc(n2 * sizeof(int)), *je = (int *)malloc(n2 * sizeof(int)), *dir = (int *)malloc(n2 * sizeof(int));
  int rc = 0;
  for (int l = 0; l < ncontacts; l++) {
    tile[l] = tile1[l] - 1; is[l] = istart1[l]; ie[l] = iend1[l]; js[l] = jstart1[l]; je[l] = jend1[l];
    tile[l + ncontacts] = tile2[l] - 1; is[l + ncontacts] = istart2[l]; ie[l + ncontacts] = iend2[l];
    js[l + ncontacts] = jstart2[l]; je[l + ncontacts] = jend2[l];
  }
  for (int l = 0; l < n2; l++) {
    dir[l] = side_dir(is[l], ie[l], js[l], je[l]);
    if (dir[l] < 0 || tile[l] < 0 || tile[l] >= ntiles) rc = FG_ERR_ARG;
  }
  for (int l = 0; l < n2 && !rc; l++) {
    int n = tile[l], l2 = (l + ncontacts) % n2, m = tile[l2];
    int nxn = nx[n], nyn = ny[n];
    int is1, ie1, js1, je1, is2, ie2, js2, je2;
    switch (dir[l]) {                                   /* halo strip of tile n */
    case D_WEST:  is1 = 0; ie1 = halo - 1; js1 = imin(js[l], je[l]) + halo; je1 = imax(js[l], je[l]) + halo; break;
    case D_EAST:  is1 = nxn + halo; ie1 = nxn + halo + halo - 1; js1 = imin(js[l], je[l]) + halo; je1 = imax(js[l], je[l]) + halo; break;
    case D_SOUTH: is1 = imin(is[l], ie[l]) + halo; ie1 = imax(is[l], ie[l]) + halo; js1 = 0; je1 = halo - 1; break;
    default:      is1 = imin(is[l], ie[l]) + halo; ie1 = imax(is[l], ie[l]) + halo; js1 = nyn + halo; je1 = nyn + halo + halo - 1; break;
    }
    switch (dir[l2]) {                                  /* interior strip of the neighbour (sizes of tile n, as the reference) */
    case D_WEST:  is2 = halo; ie2 = halo + halo - 1; js2 = imin(js[l2], je[l2]) + halo; je2 = imax(js[l2], je[l2]) + halo; break;
    case D_EAST:  is2 = nxn - halo + 1; ie2 = nxn; js2 = imin(js[l2], je[l2]) + halo; je2 = imax(js[l2], je[l2]) + halo; break;
    case D_SOUTH: is2 = imin(is[l2], ie[l2]) + halo; ie2 = imax(is[l2], ie[l2]) + halo; js2 = halo; je2 = halo + halo - 1; break;
    default:      is2 = imin(is[l2], ie[l2]) + halo; ie2 = imax(is[l2], ie[l2]) + halo; js2 = nyn - halo + 1; je2 = nyn; break;
    }
    int rotate = 0;                                     /* 0, 90, -90, 180: fregrid_util.c:2541-2546 */
    if (dir[l] == D_WEST && dir[l2] == D_NORTH) rotate = 90;
    if (dir[l] == D_EAST && dir[l2] == D_SOUTH) rotate = 90;
    if (dir[l] == D_SOUTH && dir[l2] == D_EAST) rotate = -90;
    if (dir[l] == D_NORTH && dir[l2] == D_WEST) rotate = -90;
    if (dir[l] == D_NORTH && dir[l2] == D_NORTH) rotate = 180;
    int cnt1 = (ie1 - is1 + 1) * (je1 - js1 + 1), cnt2 = (ie2 - is2 + 1) * (je2 - js2 + 1);
    if (cnt1 != cnt2) { rc = FG_ERR_ARG; break; }       /* "size mismatch between the boundary" */
    int nx2 = nx[m] + 2;
    int *buf = (int *)malloc(cnt2 * sizeof(int));
    int q = 0;
    if (rotate == 0)        { for (int j = js2; j <= je2; j++) for (int i = is2; i <= ie2; i++) buf[q++] = j * nx2 + i; }
    else if (rotate == 90)  { for (int i = ie2; i >= is2; i--) for (int j = js2; j <= je2; j++) buf[q++] = j * nx2 + i; }
    else if (rotate == -90) { for (int i = is2; i <= ie2; i++) for (int j = je2; j >= js2; j--) buf[q++] = j * nx2 + i; }
    else                    { for (int j = je2; j >= js2; j--) for (int i = ie2; i >= is2; i--) buf[q++] = j * nx2 + i; }
    q = 0;
    for (int j = js1; j <= je1; j++) for (int i = is1; i <= ie1; i++)
      map[map_off[n] + (long)j * (nxn + 2) + i] = (int)(map_off[m] + buf[q++]);
    free(buf);
  }
  free(tile); free(is); free(ie); free(js); free(je); free(dir);
  return rc;
}

/* ------------------------------------------------------------------------------------------------ latlon2xyz
 * mosaic_util.c:212-222 for the great-circle search: the corner unit vectors must be the reference's bit for bit
 * (the great-circle areas amplify a last-place difference in a coordinate to ~1e-8 relative at C768), so they are
 * taken with the host's libm -- the same five calls per point -- and uploaded; O(vertices), split over threads. */
#include <pthread.h>
#include <unistd.h>
typedef struct { size_t b, e; const double *lon, *lat; double *x, *y, *z; } Ll2Job;
static void *ll2_worker(void *arg)
{
  Ll2Job *j = (Ll2Job *)arg;
  for (size_t n = j->b; n < j->e; n++) {
    j->x[n] = cos_sep(j->lat[n]) * cos_sep(j->lon[n]);
    j->y[n] = cos_sep(j->lat[n]) * sin_sep(j->lon[n]);
    j->z[n] = sin_sep(j->lat[n]);
  }
  return NULL;
}
void fg_latlon2xyz(long size, const double *lon, const double *lat, double *x, double *y, double *z)
{
  if (size <= 0) return;
  long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
  int nt = (int)(size / 32768);
  if (nt > ncpu) nt = (int)ncpu;
  if (nt > 32) nt = 32;
  if (nt < 1) nt = 1;
  Ll2Job jobs[32];
  pthread_t th[32];
  int started[32];
  size_t chunk = ((size_t)size + nt - 1) / nt;
  for (int k = 0; k < nt; k++) {
    size_t b = (size_t)k * chunk, e = b + chunk;
    if (b > (size_t)size) b = (size_t)size;
    if (e > (size_t)size) e = (size_t)size;
    jobs[k].b = b; jobs[k].e = e;
    jobs[k].lon = lon; jobs[k].lat = lat; jobs[k].x = x; jobs[k].y = y; jobs[k].z = z;
    started[k] = 0;
    if (k > 0) started[k] = (pthread_create(&th[k], NULL, ll2_worker, &jobs[k]) == 0);
  }
  ll2_worker(&jobs[0]);
  for (int k = 1; k < nt; k++) { if (started[k]) pthread_join(th[k], NULL); else ll2_worker(&jobs[k]); }
}
