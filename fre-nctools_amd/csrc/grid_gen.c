/*
 * grid_gen.c -- host-side synthetic grid generators (C, libm only).
 *
 * fregrid needs cell-corner longitudes/latitudes in radians.  The reference gets
 * them from make_hgrid supergrid files; on a machine without those files (the
 * benchmark box) we generate the same grids directly:
 *
 *   fg_gnomonic_ed_corners   equal-distance gnomonic cubed sphere, 6 tiles
 *       follows tools/make_hgrid/create_gnomonic_cubic_grid.c:
 *         gnomonic_ed :1465-1536, mirror_latlon :1569-1590, symm_ed :1596-1632,
 *         mirror_grid :1637-1752, rot_3d :1759-1804, tile-edge consistency :345-385,
 *         west shift by pi/18 and clean-up :336-343, radians->degrees :717-720,
 *       and the degrees->radians read-back of tools/fregrid/fregrid_util.c:227-232.
 *   fg_latlon_corners        regular lat-lon target grid
 *       follows tools/fregrid/fregrid_util.c:564-603,645-654 (get_output_grid_by_size).
 *
 * Same operation order as the reference so the corners agree to the last bit
 * with glibc libm (checked against oracle/_ref in tests/test_oracle_vs_ref.py and tests/test_c2l_cpu.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "fregrid_hip.h"

#define FG_PI 3.14159265358979323846
#define FG_R2D (180 / FG_PI)
#define FG_D2R (FG_PI / 180)
#define FG_RADIUS 6371000.0

/* The reference's latlon2xyz (mosaic_util.c:212-222) compiles to five separate libm calls -- its output pointers
 * may alias its inputs, so gcc cannot fuse sin/cos into sincos -- and glibc's sincos can differ from sin/cos in
 * the last bit.  Keep the calls separate here too (noinline wrappers defeat the sincos fusion). */
static double __attribute__((noinline)) sin_sep(double x) { return sin(x); }
static double __attribute__((noinline)) cos_sep(double x) { return cos(x); }
static void ll2xyz(double lon, double lat, double *p)
{
  p[0] = cos_sep(lat) * cos_sep(lon);
  p[1] = cos_sep(lat) * sin_sep(lon);
  p[2] = sin_sep(lat);
}

/* mosaic_util.c:228-252 */
static void xyz2ll(double xx, double yy, double zz, double *lon, double *lat)
{
  double dist = sqrt(xx * xx + yy * yy + zz * zz);
  xx /= dist; yy /= dist; zz /= dist;
  if (fabs(xx) + fabs(yy) < 1.e-10) *lon = 0;
  else *lon = atan2(yy, xx);
  *lat = asin(zz);
  if (*lon < 0.) *lon = 2. * FG_PI + *lon;
}

/* reflect (lon0,lat0) through the plane spanned by the centre and the two mirror points */
static void reflect_ll(double lon1, double lat1, double lon2, double lat2,
                       double lon0, double lat0, double *lon, double *lat)
{
  double p0[3], p1[3], p2[3], pp[3], nb[3], pdot;
  ll2xyz(lon0, lat0, p0);
  ll2xyz(lon1, lat1, p1);
  ll2xyz(lon2, lat2, p2);
  nb[0] = p1[1] * p2[2] - p1[2] * p2[1];
  nb[1] = p1[2] * p2[0] - p1[0] * p2[2];
  nb[2] = p1[0] * p2[1] - p1[1] * p2[0];
  pdot = sqrt(nb[0] * nb[0] + nb[1] * nb[1] + nb[2] * nb[2]);
  for (int k = 0; k < 3; k++) nb[k] = nb[k] / pdot;
  pdot = p0[0] * nb[0] + p0[1] * nb[1] + p0[2] * nb[2];
  for (int k = 0; k < 3; k++) pp[k] = p0[k] - 2 * pdot * nb[k];
  xyz2ll(pp[0], pp[1], pp[2], lon, lat);
}

/* first face, equal distance along the four edges */
static void face_equal_dist(int ni, double *lam, double *the)
{
  int nip = ni + 1;
  double rsq3 = 1. / sqrt(3.);
  double alpha = asin(rsq3);
  double dely = 2. * alpha / ni;

  for (int j = 0; j < nip; j++) {
    lam[j * nip] = 0.75 * FG_PI;
    lam[j * nip + ni] = 1.25 * FG_PI;
    the[j * nip] = -alpha + dely * j;
    the[j * nip + ni] = the[j * nip];
  }
  for (int i = 1; i < ni; i++) {
    reflect_ll(lam[0], the[0], lam[ni * nip + ni], the[ni * nip + ni],
               lam[i * nip], the[i * nip], &lam[i], &the[i]);
    lam[ni * nip + i] = lam[i];
    the[ni * nip + i] = -the[i];
  }
  size_t np = (size_t)nip * nip;
  double *x = (double *)malloc(np * sizeof(double));
  double *y = (double *)malloc(np * sizeof(double));
  double *z = (double *)malloc(np * sizeof(double));
  double p[3];
  int corner[4] = {0, ni, ni * nip, ni * nip + ni};
  for (int c = 0; c < 4; c++) {
    ll2xyz(lam[corner[c]], the[corner[c]], p);
    x[corner[c]] = p[0]; y[corner[c]] = p[1]; z[corner[c]] = p[2];
  }
  /* project the west and south edges onto the cube face x = -1/sqrt(3) */
  for (int j = 1; j < ni; j++) {
    int n = j * nip;
    ll2xyz(lam[n], the[n], p);
    x[n] = p[0];
    y[n] = -p[1] * rsq3 / x[n];
    z[n] = -p[2] * rsq3 / x[n];
  }
  for (int i = 1; i < ni; i++) {
    ll2xyz(lam[i], the[i], p);
    x[i] = p[0];
    y[i] = -p[1] * rsq3 / x[i];
    z[i] = -p[2] * rsq3 / x[i];
  }
  for (size_t k = 0; k < np; k++) x[k] = -rsq3;
  for (int j = 1; j < nip; j++)
    for (int i = 1; i < nip; i++) {
      y[j * nip + i] = y[i];
      z[j * nip + i] = z[j * nip];
    }
  for (size_t k = 0; k < np; k++) xyz2ll(x[k], y[k], z[k], &lam[k], &the[k]);
  free(x); free(y); free(z);
}

/* symmetrise about the face centre lines */
static void face_symmetrise(int ni, double *lam, double *the)
{
  int nip = ni + 1;
  for (int j = 1; j < nip; j++)
    for (int i = 1; i < ni; i++) lam[j * nip + i] = lam[i];
  for (int j = 0; j < nip; j++)
    for (int i = 0; i < ni / 2; i++) {
      int ip = ni - i;
      double avg = 0.5 * (lam[j * nip + i] - lam[j * nip + ip]);
      lam[j * nip + i] = avg + FG_PI;
      lam[j * nip + ip] = FG_PI - avg;
      avg = 0.5 * (the[j * nip + i] + the[j * nip + ip]);
      the[j * nip + i] = avg;
      the[j * nip + ip] = avg;
    }
  for (int j = 0; j < ni / 2; j++) {
    int jp = ni - j;
    for (int i = 1; i < ni; i++) {
      double avg = 0.5 * (lam[j * nip + i] + lam[jp * nip + i]);
      lam[j * nip + i] = avg;
      lam[jp * nip + i] = avg;
      avg = 0.5 * (the[j * nip + i] - the[jp * nip + i]);
      the[j * nip + i] = avg;
      the[jp * nip + i] = -avg;
    }
  }
}

/* rotate a (lon, lat, r) point about a Cartesian axis by `deg` degrees; the
 * spherical<->Cartesian pair uses the generator's own convention (z = -r sin(lat),
 * lat = acos(z/r) - pi/2), create_gnomonic_cubic_grid.c:1811-1835 */
static void rotate_ll(int axis, double lon, double lat, double r, double deg,
                      double *lon2, double *lat2, double *r2)
{
  double x1 = r * cos(lon) * cos(lat);
  double y1 = r * sin(lon) * cos(lat);
  double z1 = -r * sin(lat);
  double angle = deg * FG_D2R;
  double c = cos(angle), s = sin(angle);
  double x2, y2, z2;
  if (axis == 1)      { x2 = x1;               y2 = c * y1 + s * z1;  z2 = -s * y1 + c * z1; }
  else if (axis == 2) { x2 = c * x1 - s * z1;  y2 = y1;               z2 = s * x1 + c * z1; }
  else                { x2 = c * x1 + s * y1;  y2 = -s * x1 + c * y1; z2 = z1; }
  *r2 = sqrt(x2 * x2 + y2 * y2 + z2 * z2);
  if ((fabs(x2) + fabs(y2)) < 1.e-10) *lon2 = 0.;
  else *lon2 = atan2(y2, x2);
  *lat2 = acos(z2 / (*r2)) - FG_PI / 2.;
}

static double sgn1(double v) { return v >= 0 ? 1 : -1; }

/* symmetrise tile 1 about greenwich/equator, then rotate it into the other five tiles */
static void six_tiles(int ni, double *x, double *y)
{
  int nip = ni + 1;
  int half = (int)ceil(nip / 2.);
  for (int j = 0; j < half; j++) {
    int jp = ni - j;
    for (int i = 0; i < half; i++) {
      int ip = ni - i;
      int a = j * nip + i, b = j * nip + ip, c = jp * nip + i, d = jp * nip + ip;
      double x1 = 0.25 * (fabs(x[a]) + fabs(x[b]) + fabs(x[c]) + fabs(x[d]));
      x[a] = x1 * sgn1(x[a]); x[b] = x1 * sgn1(x[b]);
      x[c] = x1 * sgn1(x[c]); x[d] = x1 * sgn1(x[d]);
      double y1 = 0.25 * (fabs(y[a]) + fabs(y[b]) + fabs(y[c]) + fabs(y[d]));
      y[a] = y1 * sgn1(y[a]); y[b] = y1 * sgn1(y[b]);
      y[c] = y1 * sgn1(y[c]); y[d] = y1 * sgn1(y[d]);
      if (nip % 2) {
        if (i == (nip - 1) / 2) { x[a] = 0.0; x[c] = 0.0; }
      }
    }
  }
  int mid = (nip - 1) / 2;
  for (int nt = 1; nt < 6; nt++)
    for (int j = 0; j < nip; j++)
      for (int i = 0; i < nip; i++) {
        double x1 = x[j * nip + i], y1 = y[j * nip + i], z1 = FG_RADIUS;
        double x2 = 0, y2 = 0, z2 = 0;
        switch (nt) {
        case 1:
          rotate_ll(3, x1, y1, z1, -90., &x2, &y2, &z2);
          break;
        case 2:
          rotate_ll(3, x1, y1, z1, -90., &x2, &y2, &z2);
          rotate_ll(1, x2, y2, z2, 90., &x1, &y1, &z1);
          x2 = x1; y2 = y1; z2 = z1;
          if (nip % 2) {
            if ((i == mid) && (i == j)) { x2 = 0; y2 = FG_PI * 0.5; }
            if ((j == mid) && (i < mid)) x2 = 0;
            if ((j == mid) && (i > mid)) x2 = FG_PI;
          }
          break;
        case 3:
          rotate_ll(3, x1, y1, z1, -180., &x2, &y2, &z2);
          rotate_ll(1, x2, y2, z2, 90., &x1, &y1, &z1);
          x2 = x1; y2 = y1; z2 = z1;
          if (nip % 2) { if (j == mid) x2 = FG_PI; }
          break;
        case 4:
          rotate_ll(3, x1, y1, z1, 90., &x2, &y2, &z2);
          rotate_ll(2, x2, y2, z2, 90., &x1, &y1, &z1);
          x2 = x1; y2 = y1; z2 = z1;
          break;
        case 5:
          rotate_ll(2, x1, y1, z1, 90., &x2, &y2, &z2);
          rotate_ll(3, x2, y2, z2, 0., &x1, &y1, &z1);
          x2 = x1; y2 = y1; z2 = z1;
          if (nip % 2) {
            if ((i == mid) && (i == j)) { x2 = 0; y2 = -FG_PI * 0.5; }
            if ((i == mid) && (j > mid)) x2 = 0;
            if ((i == mid) && (j < mid)) x2 = FG_PI;
          }
          break;
        }
        x[nt * nip * nip + j * nip + i] = x2;
        y[nt * nip * nip + j * nip + i] = y2;
      }
}

/* T-cell centres: normalised sum of the four corner unit vectors (create_gnomonic_cubic_grid.c:2008-2048) */
static void cell_centres(int ni, const double *lonc, const double *latc, double *lont, double *latt)
{
  int nip = ni + 1;
  size_t np = (size_t)nip * nip;
  double *xc = (double *)malloc(np * sizeof(double)), *yc = (double *)malloc(np * sizeof(double)), *zc = (double *)malloc(np * sizeof(double));
  for (size_t k = 0; k < np; k++) { double p[3]; ll2xyz(lonc[k], latc[k], p); xc[k] = p[0]; yc[k] = p[1]; zc[k] = p[2]; }
  for (int j = 0; j < ni; j++)
    for (int i = 0; i < ni; i++) {
      int p1 = j * nip + i, p2 = p1 + 1, p3 = (j + 1) * nip + i + 1, p4 = (j + 1) * nip + i;
      double xt = xc[p1] + xc[p2] + xc[p3] + xc[p4];
      double yt = yc[p1] + yc[p2] + yc[p3] + yc[p4];
      double zt = zc[p1] + zc[p2] + zc[p3] + zc[p4];
      double dd = sqrt(pow(xt, 2) + pow(yt, 2) + pow(zt, 2));
      xt /= dd; yt /= dd; zt /= dd;
      xyz2ll(xt, yt, zt, &lont[j * ni + i], &latt[j * ni + i]);
    }
  free(xc); free(yc); free(zc);
}

static int gnomonic_ed_grid(int ni, double shift_fac, int via_degrees, double *lonc, double *latc, double *lont, double *latt);

int fg_gnomonic_ed_corners(int ni, double shift_fac, int via_degrees, double *lonc, double *latc)
{
  return gnomonic_ed_grid(ni, shift_fac, via_degrees, lonc, latc, NULL, NULL);
}

int fg_gnomonic_ed_grid(int ni, double shift_fac, int via_degrees, double *lonc, double *latc, double *lont, double *latt)
{
  return gnomonic_ed_grid(ni, shift_fac, via_degrees, lonc, latc, lont, latt);
}

static int gnomonic_ed_grid(int ni, double shift_fac, int via_degrees, double *lonc, double *latc, double *lont, double *latt)
{
  if (ni < 2) return -1;
  int nip = ni + 1;
  size_t np = (size_t)nip * nip;
  double *lam = (double *)malloc(np * sizeof(double));
  double *the = (double *)malloc(np * sizeof(double));
  face_equal_dist(ni, lam, the);
  face_symmetrise(ni, lam, the);
  double *xc = lonc, *yc = latc;
  for (size_t k = 0; k < np; k++) { xc[k] = lam[k] - FG_PI; yc[k] = the[k]; }
  free(lam); free(the);
  six_tiles(ni, xc, yc);

  for (size_t n = 0; n < 6 * np; n++) {
    if (shift_fac > 1.e-4) xc[n] -= FG_PI / 18.;
    if (xc[n] < 0.) xc[n] += 2. * FG_PI;
    if (fabs(xc[n]) < 1.e-10) xc[n] = 0;
    if (fabs(yc[n]) < 1.e-10) yc[n] = 0;
  }
  /* make shared tile edges bitwise identical */
  size_t T = np;
  for (int j = 0; j < nip; j++) {
    xc[T + j * nip] = xc[j * nip + ni];                     yc[T + j * nip] = yc[j * nip + ni];
    xc[2 * T + j * nip] = xc[ni * nip + ni - j];            yc[2 * T + j * nip] = yc[ni * nip + ni - j];
  }
  for (int i = 0; i < nip; i++) {
    xc[4 * T + ni * nip + i] = xc[(ni - i) * nip];          yc[4 * T + ni * nip + i] = yc[(ni - i) * nip];
    xc[5 * T + ni * nip + i] = xc[i];                       yc[5 * T + ni * nip + i] = yc[i];
    xc[2 * T + i] = xc[T + ni * nip + i];                   yc[2 * T + i] = yc[T + ni * nip + i];
    xc[3 * T + i] = xc[T + (ni - i) * nip + ni];            yc[3 * T + i] = yc[T + (ni - i) * nip + ni];
  }
  for (int j = 0; j < nip; j++) {
    xc[5 * T + j * nip + ni] = xc[T + ni - j];              yc[5 * T + j * nip + ni] = yc[T + ni - j];
    xc[3 * T + j * nip] = xc[2 * T + j * nip + ni];         yc[3 * T + j * nip] = yc[2 * T + j * nip + ni];
    xc[4 * T + j * nip] = xc[2 * T + ni * nip + ni - j];    yc[4 * T + j * nip] = yc[2 * T + ni * nip + ni - j];
  }
  for (int i = 0; i < nip; i++) {
    xc[4 * T + i] = xc[3 * T + ni * nip + i];               yc[4 * T + i] = yc[3 * T + ni * nip + i];
    xc[5 * T + i] = xc[3 * T + (ni - i) * nip + ni];        yc[5 * T + i] = yc[3 * T + (ni - i) * nip + ni];
  }
  for (int j = 0; j < nip; j++) {
    xc[5 * T + j * nip] = xc[4 * T + j * nip + ni];         yc[5 * T + j * nip] = yc[4 * T + j * nip + ni];
  }
  if (lont && latt)                                  /* centres come from the radian corners (:493) */
    for (int t = 0; t < 6; t++) cell_centres(ni, xc + t * np, yc + t * np, lont + (size_t)t * ni * ni, latt + (size_t)t * ni * ni);
  if (via_degrees) {
    /* grid files hold degrees (create_gnomonic_cubic_grid.c:717-720); fregrid converts back
       (fregrid_util.c:230-231,241-242) */
    for (size_t n = 0; n < 6 * np; n++) {
      double xd = xc[n] * FG_R2D, yd = yc[n] * FG_R2D;
      xc[n] = xd * FG_D2R;
      yc[n] = yd * FG_D2R;
    }
    if (lont && latt)
      for (size_t n = 0; n < (size_t)6 * ni * ni; n++) {
        double xd = lont[n] * FG_R2D, yd = latt[n] * FG_R2D;
        lont[n] = xd * FG_D2R;
        latt[n] = yd * FG_D2R;
      }
  }
  return 0;
}

int fg_latlon_corners(int nlon, int nlat, double lonbegin, double lonend, double latbegin,
                      double latend, int center_y, double *lonc, double *latc)
{
  if (nlon < 1 || nlat < 1) return -1;
  double lon_range = lonend - lonbegin, lat_range = latend - latbegin;
  double dlon = lon_range / nlon;
  double dlat = center_y ? lat_range / nlat : lat_range / (nlat - 1);
  for (int j = 0; j <= nlat; j++) {
    double latv = center_y ? (latbegin + j * dlat) * FG_D2R : (latbegin + (j - 0.5) * dlat) * FG_D2R;
    for (int i = 0; i <= nlon; i++) {
      lonc[(size_t)j * (nlon + 1) + i] = (lonbegin + i * dlon) * FG_D2R;
      latc[(size_t)j * (nlon + 1) + i] = latv;
    }
  }
  return 0;
}

/* ---------------------------------------------------------------------------------------------- tripolar grid
 * make_hgrid --grid_type tripolar_grid with two bounds per axis (uniform spacing) and center "none":
 * tools/make_hgrid/create_lonlat_grid.c:360-447 (grid lines, join latitude, Murray bipolar cap) with
 * compute_grid_bound / tp_trans / bp_lam / bp_phi / lat_dist / lon_in_range of tools/libfrencutils/tool_util.c:263-365,
 * :479-536 and the two-point (linear) case of cubic_spline_sp (interp.c:65-69), followed by fregrid's read-back of
 * every second supergrid point (fregrid_util.c:227-232).
 * NOT PINNED to the reference generator: tool_util.c includes <netcdf.h>, which this image lacks, so it cannot be
 * compiled here.  The generator only synthesises test/benchmark inputs; parity of the search on these inputs is
 * established against the oracle, which does not depend on how the grid was made. */
static double tp_lat_dist(double x1, double x2)
{
  double a = fmod(x1 - x2 + 720, 360.), b = fmod(x2 - x1 + 720, 360.);
  return a < b ? a : b;
}
static double tp_lon_in_range(double lon, double lon_strt)
{
  const double SMALL = 1.0e-4;
  double v = lon, lon_end = lon_strt + 360.;
  if (fabs(v - lon_strt) < SMALL) v = lon_strt;
  else if (fabs(v - lon_end) < SMALL) v = lon_strt;
  else {
    while (1) {
      if (v < lon_strt) v = v + 360.;
      else if (v > lon_end) v = v - 360.;
      else break;
    }
  }
  return v;
}
static void tp_trans1(double *lon, double *lat, double lon_ref, double lon_start, double lam0, double bpeq, double bpsp, double bpnp, double rp)
{
  const double SMALL = 1.0e-4;
  double bp_lam = 2. * atan(tan((0.5 * FG_PI - (*lat) * FG_D2R) / 2) / rp) * FG_R2D;
  if (tp_lat_dist(*lon, bpeq) < 90.) bp_lam = -bp_lam;
  double bp_phi = (tp_lat_dist(*lon, bpsp) < 90.) ? (-90 + tp_lat_dist(*lon, bpsp)) : (90 - tp_lat_dist(*lon, bpnp));
  double lamc = bp_lam * FG_D2R, phic = bp_phi * FG_D2R, chic, phis, lams;
  if (fabs(*lat - 90.) < SMALL) {
    if (phic > 0) *lon = tp_lon_in_range(lon_start, lon_ref);
    else *lon = lon_start + 180.;
    chic = acos(cos(lamc) * cos(phic));
    phis = FG_PI * 0.5 - 2 * atan(rp * tan(chic / 2));
    *lat = phis * FG_R2D;
    return;
  }
  if (fabs(lamc) < SMALL && fabs(phic) < SMALL) { *lat = 90.; *lon = lon_ref; }
  else {
    lams = fmod(lam0 + FG_PI + FG_PI / 2 - atan2(sin(lamc), tan(phic)), 2 * FG_PI);
    chic = acos(cos(lamc) * cos(phic));
    phis = FG_PI * 0.5 - 2 * atan(rp * tan(chic / 2));
    *lon = lams * FG_R2D;
    *lon = tp_lon_in_range(*lon, lon_ref);
    *lat = phis * FG_R2D;
  }
}

/* nlon x nlat model cells; bounds in degrees (e.g. -280, 80, -82, 90 as tests/fregrid/latlon:36-37); lat_join 65.
 * lonc/latc [(nlat+1)*(nlon+1)] radians. */
int fg_tripolar_corners(int nlon, int nlat, double xbnd0, double xbnd1, double ybnd0, double ybnd1, double lat_join_in,
                        double *lonc, double *latc)
{
  if (nlon < 2 || nlat < 2) return -1;
  const int nx = 2 * nlon, ny = 2 * nlat, nxp = nx + 1, nyp = ny + 1;       /* supergrid */
  double *xb = (double *)malloc(nxp * sizeof(double)), *yb = (double *)malloc(nyp * sizeof(double));
  double *x = (double *)malloc((size_t)nxp * nyp * sizeof(double)), *y = (double *)malloc((size_t)nxp * nyp * sizeof(double));
  if (!xb || !yb || !x || !y) { free(xb); free(yb); free(x); free(y); return -2; }
  double px = (xbnd1 - xbnd0) / ((nx + 1.0) - 1.0), py = (ybnd1 - ybnd0) / ((ny + 1.0) - 1.0);
  for (int i = 0; i < nxp; i++) xb[i] = px * ((i + 1.0) - 1.0) + xbnd0;
  for (int j = 0; j < nyp; j++) yb[j] = py * ((j + 1.0) - 1.0) + ybnd0;
  for (int j = 0; j < nyp; j++) for (int i = 0; i < nxp; i++) { x[(size_t)j * nxp + i] = xb[i]; y[(size_t)j * nxp + i] = yb[j]; }
  /* nearest_index (mosaic_util.c:81-109) */
  int j_join;
  if (lat_join_in < yb[0]) j_join = 0;
  else if (lat_join_in > yb[nyp - 1]) j_join = nyp - 1;
  else {
    int i = 0; j_join = 0;
    while (i < nyp) { i = i + 1; if (lat_join_in <= yb[i]) { j_join = i; if (yb[i] - lat_join_in > lat_join_in - yb[i - 1]) j_join = i - 1; break; } }
  }
  const double lat_join = yb[j_join], lon_start = xbnd0;
  const double lon_bpeq = lon_start + 90., lon_bpnp = lon_start, lon_bpsp = lon_start + 180.;
  const double lam0 = fmod(lon_bpeq * FG_D2R + 2 * FG_PI, 2 * FG_PI);
  const double rp = tan((0.5 * FG_PI - lat_join * FG_D2R) / 2.);
  for (int j = j_join; j < nyp; j++)
    for (int i = 0; i < nxp; i++) {
      double lon_last = x[(size_t)j * nxp + (i - 1 > 0 ? i - 1 : 0)];
      tp_trans1(&x[(size_t)j * nxp + i], &y[(size_t)j * nxp + i], lon_last, lon_start, lam0, lon_bpeq, lon_bpsp, lon_bpnp, rp);
    }
  for (int j = 0; j <= nlat; j++)
    for (int i = 0; i <= nlon; i++) {
      lonc[(size_t)j * (nlon + 1) + i] = x[(size_t)(2 * j) * nxp + 2 * i] * FG_D2R;
      latc[(size_t)j * (nlon + 1) + i] = y[(size_t)(2 * j) * nxp + 2 * i] * FG_D2R;
    }
  free(xb); free(yb); free(x); free(y);
  return 0;
}
