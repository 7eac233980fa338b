// geom.hip.h -- device-side spherical polygon arithmetic for gfx950.
//
// Every function keeps the FP64 expression trees of the reference (compile with
// -ffp-contract=off): the exchange-cell membership tests (bounding boxes, inside_edge,
// clip intersections, the |dlat| < 1e-10 branches) are pure IEEE add/sub/mul/div and
// therefore bit-identical to the CPU reference; only sin/cos differ in the last ulp
// (ocml vs glibc), which moves areas/centroids by ~1e-16 relative.
//
// Polygons are addressed through a stride template parameter so the same code runs
// on per-lane private arrays (S = 1) and on LDS tiles laid out [vertex][lane] (S = 64,
// bank = 2*lane mod 64 for every vertex index: conflict-free for ds_read_b64).
//
//   d_fix_lon      tools/libfrencutils/mosaic_util.c:667-738
//   d_poly_area    tools/libfrencutils/mosaic_util.c:417-459
//   d_inside_edge  tools/libfrencutils/create_xgrid.c:2342-2350
//   d_pimod        tools/libfrencutils/create_xgrid.c:1343-1349
//   d_poly_ctrlon  tools/libfrencutils/create_xgrid.c:2170-2217
//   d_poly_ctrlat  tools/libfrencutils/create_xgrid.c:2096-2121
#pragma once
#include <hip/hip_runtime.h>

#define G_PI      3.14159265358979323846
#define G_TPI     (2.0 * G_PI)
#define G_HPI     (0.5 * G_PI)
#define G_RADIUS  6371000.0
#define G_SMALL   1.e-10
#define G_POLETOL 1.e-6
#define G_MAXV    8      // MAX_V, create_xgrid.c:627
#define G_FIXCAP  12     // private capacity while fix_lon inserts vertices

#define G_ERRBIT_MAXV     1u
#define G_ERRBIT_PARALLEL 2u
#define G_ERRBIT_OVERFLOW 4u
#define G_ERRBIT_BADLAT   8u

// ---------------------------------------------------------------------------------------
// sin/cos for latitude-like arguments: the host libm's own operation sequence (sincos_glibc.h -- glibc 2.35 do_sin /
// do_cos over the 1/128 table, checked bit for bit against libm on the host by tests/test_sincos_host.py), so that
// poly_area, poly_ctrlon and poly_ctrlat -- whose other operations are plain IEEE add/mul/div in the reference's
// order -- produce the reference's BITS on the device, not just its value to an ulp.  Valid for |x| < 2.426; grids with a
// latitude outside [-pi/2, pi/2] are rejected by k_cell_struct and clip vertices stay inside their parents' boxes.
// (An earlier fdlibm-coefficient kernel differed from libm in the last place for 6 % of sines and 14 % of cosines; the
// cancellation in poly_area turned that into up to 1.3e-10 relative on the thinnest exchange cells at C768.)
// ---------------------------------------------------------------------------------------
#include "sincos_glibc.h"
__device__ __forceinline__ double d_sin_lat(double x) { return fgs_sin(x); }
__device__ __forceinline__ double d_cos_lat(double x) { return fgs_cos(x); }
// libm sincos(): what gcc -O2 emits in the reference for sin(a) and cos(a) of one argument in one expression
__device__ __forceinline__ void d_sincos_lat(double x, double *s, double *c) { fgs_sincos(x, s, c); }

__device__ __forceinline__ bool d_is_pole(double lat) { return fabs(lat) >= G_HPI - G_POLETOL; }

// x,y: private arrays of capacity G_FIXCAP.  Returns the new vertex count, or -1 if the
// capacity would be exceeded (reported as FG_ERR_MAXV by the caller).
__device__ inline int d_fix_lon(double *x, double *y, int n, double tlon)
{
  int nn = n;
  for (int i = 0; i < nn; i++) {
    if (!d_is_pole(y[i])) continue;
    int im = (i + nn - 1) % nn, ip = (i + 1) % nn;
    if (y[im] == y[i] && y[ip] == y[i]) {
      for (int k = i; k < nn - 1; k++) { x[k] = x[k + 1]; y[k] = y[k + 1]; }
      nn--; i--;
    } else if (y[im] != y[i] && y[ip] != y[i]) {
      if (nn + 1 > G_FIXCAP) return -1;
      for (int k = nn - 1; k >= i; k--) { x[k + 1] = x[k]; y[k + 1] = y[k]; }
      nn++; i++;
    }
  }
  for (int i = 0; i < nn; i++) {
    if (!d_is_pole(y[i])) continue;
    int im = (i + nn - 1) % nn, ip = (i + 1) % nn;
    if (y[im] != y[i]) x[i] = x[im];
    if (y[ip] != y[i]) x[i] = x[ip];
  }
  for (int i = 0; i < nn; i++) {
    int im = (i + nn - 1) % nn;
    double dx = x[i] - x[im];
    if (fabs(dx + G_PI) < G_SMALL || fabs(dx - G_PI) < G_SMALL) {
      if (nn + 2 > G_FIXCAP) return -1;
      double xa = x[im], xb = x[i];
      double ypole = G_HPI;
      if (y[i] < 0.0) ypole = -G_HPI;
      for (int k = nn - 1; k >= i; k--) { x[k + 2] = x[k]; y[k + 2] = y[k]; }
      x[i] = xa; y[i] = ypole;
      x[i + 1] = xb; y[i + 1] = ypole;
      nn += 2;
      break;
    }
  }
  if (!nn) return 0;
  double x_sum = x[0];
  for (int i = 1; i < nn; i++) {
    double dx = x[i] - x[i - 1];
    if (dx < -G_PI)     dx = dx + G_TPI;
    else if (dx > G_PI) dx = dx - G_TPI;
    x_sum += (x[i] = x[i - 1] + dx);
  }
  double d = (x_sum / nn) - tlon;
  if (d < -G_PI)     for (int i = 0; i < nn; i++) x[i] += G_TPI;
  else if (d > G_PI) for (int i = 0; i < nn; i++) x[i] -= G_TPI;
  return nn;
}

// fix_lon of a quadrilateral without a pole vertex and without a |dlon| = pi edge -- every cell of a grid but the handful at
// the poles: the reference's first three loops do nothing then, and what is left is the unwrap and the recentring on four
// statically indexed values (the general routine walks 12-element private arrays with dynamic indices, which the compiler turns
// into select chains: it was most of the ~6000 instructions per wave of the cell-record kernel).  Returns 4, or -1 when the cell
// needs the general routine (nothing has been modified then).  Same operations in the same order as d_fix_lon on this path.
__device__ __forceinline__ int d_fix_lon_quad_fast(double *x, const double *y, double tlon)
{
  if (d_is_pole(y[0]) || d_is_pole(y[1]) || d_is_pole(y[2]) || d_is_pole(y[3])) return -1;
  {
    const double d0 = x[0] - x[3], d1 = x[1] - x[0], d2 = x[2] - x[1], d3 = x[3] - x[2];
    if (fabs(d0 + G_PI) < G_SMALL || fabs(d0 - G_PI) < G_SMALL || fabs(d1 + G_PI) < G_SMALL || fabs(d1 - G_PI) < G_SMALL ||
        fabs(d2 + G_PI) < G_SMALL || fabs(d2 - G_PI) < G_SMALL || fabs(d3 + G_PI) < G_SMALL || fabs(d3 - G_PI) < G_SMALL) return -1;
  }
  double x_sum = x[0];
#pragma unroll
  for (int i = 1; i < 4; i++) {
    double dx = x[i] - x[i - 1];
    if (dx < -G_PI)     dx = dx + G_TPI;
    else if (dx > G_PI) dx = dx - G_TPI;
    x_sum += (x[i] = x[i - 1] + dx);
  }
  const double d = (x_sum / 4) - tlon;
  if (d < -G_PI)     { x[0] += G_TPI; x[1] += G_TPI; x[2] += G_TPI; x[3] += G_TPI; }
  else if (d > G_PI) { x[0] -= G_TPI; x[1] -= G_TPI; x[2] -= G_TPI; x[3] -= G_TPI; }
  return 4;
}

__device__ __forceinline__ int d_inside_edge(double x0, double y0, double x1, double y1, double x, double y)
{
  double product = (x - x0) * (y1 - y0) + (x0 - x1) * (y - y0);
  return (product <= 1.e-12) ? 1 : 0;
}

__device__ __forceinline__ double d_pimod1(double v)
{
  if (v < -G_PI)     v += G_TPI;
  else if (v > G_PI) v -= G_TPI;
  return v;
}

// Area of a polygon whose vertex i sits at x[i*S], y[i*S]  (m^2).
template <int S>
__device__ inline double d_poly_area(const double *x, const double *y, int n)
{
  double area = 0.0;
  for (int i = 0; i < n; i++) {
    int ip = (i + 1 == n) ? 0 : i + 1;
    double dx = (x[ip * S] - x[i * S]);
    double lat1 = y[ip * S], lat2 = y[i * S];
    if (dx > G_PI)  dx = dx - 2.0 * G_PI;
    if (dx < -G_PI) dx = dx + 2.0 * G_PI;
    if (fabs(dx + G_PI) < G_SMALL || fabs(dx - G_PI) < G_SMALL) { area += G_PI; continue; }
    if (fabs(lat1 - lat2) < G_SMALL)
      area -= dx * d_sin_lat(0.5 * (lat1 + lat2));
    else {
      double dy = 0.5 * (lat1 - lat2);
      double dat = d_sin_lat(dy) / dy;
      area -= dx * d_sin_lat(0.5 * (lat1 + lat2)) * dat;
    }
  }
  if (area < 0) return -area * G_RADIUS * G_RADIUS;
  return area * G_RADIUS * G_RADIUS;
}

// Which libm entry each term goes through follows the reference's gcc -O2 object code (oracle/_ref, objdump):
// cos(lat1) is a plain cos() hoisted above the branch; the flat branch calls cos(avg_y) and sin(avg_y) separately;
// the general branch calls sincos(avg_y) and sin(hdy).
template <int S>
__device__ inline double d_poly_ctrlat(const double *x, const double *y, int n)
{
  double ctrlat = 0.0;
  for (int i = 0; i < n; i++) {
    int ip = (i + 1 == n) ? 0 : i + 1;
    double dx = (x[ip * S] - x[i * S]);
    double lat1 = y[ip * S], lat2 = y[i * S];
    double dy = lat2 - lat1;
    double hdy = dy * 0.5;
    double avg_y = (lat1 + lat2) * 0.5;
    if (dx == 0.0) continue;
    if (dx > G_PI)   dx = dx - 2.0 * G_PI;
    if (dx <= -G_PI) dx = dx + 2.0 * G_PI;
    const double cl1 = d_cos_lat(lat1);
    if (fabs(hdy) < G_SMALL)
      ctrlat -= dx * (2 * d_cos_lat(avg_y) + lat2 * d_sin_lat(avg_y) - cl1);
    else {
      double sa, ca;
      d_sincos_lat(avg_y, &sa, &ca);
      ctrlat -= dx * ((d_sin_lat(hdy) / hdy) * (2 * ca + lat2 * sa) - cl1);
    }
  }
  return (ctrlat * G_RADIUS * G_RADIUS);
}

template <int S>
__device__ inline double d_poly_ctrlon(const double *x, const double *y, int n, double clon)
{
  double ctrlon = 0.0;
  for (int i = 0; i < n; i++) {
    int ip = (i + 1 == n) ? 0 : i + 1;
    double phi1 = x[ip * S], phi2 = x[i * S];
    double lat1 = y[ip * S], lat2 = y[i * S];
    double dphi = phi1 - phi2;
    if (dphi == 0.0) continue;
    double sl1, cl1, sl2, cl2;                        // cos(lat)*sin(lat) is one sincos() call in the reference's object code
    d_sincos_lat(lat1, &sl1, &cl1);
    d_sincos_lat(lat2, &sl2, &cl2);
    double f1 = 0.5 * (cl1 * sl1 + lat1);
    double f2 = 0.5 * (cl2 * sl2 + lat2);
    if (dphi > G_PI)  dphi = dphi - 2.0 * G_PI;
    if (dphi < -G_PI) dphi = dphi + 2.0 * G_PI;
    double dphi1 = phi1 - clon;
    if (dphi1 > G_PI)  dphi1 -= 2.0 * G_PI;
    if (dphi1 < -G_PI) dphi1 += 2.0 * G_PI;
    double dphi2 = phi2 - clon;
    if (dphi2 > G_PI)  dphi2 -= 2.0 * G_PI;
    if (dphi2 < -G_PI) dphi2 += 2.0 * G_PI;
    if (fabs(dphi2 - dphi1) < G_PI) {
      ctrlon -= dphi * (dphi1 * f1 + dphi2 * f2) / 2.0;
    } else {
      double fac = (dphi1 > 0.0) ? G_PI : -G_PI;
      double fint = f1 + (f2 - f1) * (fac - dphi1) / fabs(dphi);
      ctrlon -= 0.5 * dphi1 * (dphi1 - fac) * f1 - 0.5 * dphi2 * (dphi2 + fac) * f2
                + 0.5 * fac * (dphi1 + dphi2) * fint;
    }
  }
  return (ctrlon * G_RADIUS * G_RADIUS);
}

// Fused area + centroid line integrals over one polygon (order 2): every edge quantity is
// the same function of the same inputs as in the three separate loops above, evaluated once
// (sin/cos of the edge mid-latitude, d_sin_lat(half dlat)/(half dlat), sin/cos of the vertex
// latitudes), so the three results equal the separate evaluations bit for bit.
template <int S>
__device__ inline void d_poly_area_ctr(const double *x, const double *y, int n, double clon,
                                       double *area_out, double *ctrlon_out, double *ctrlat_out)
{
  double area = 0.0, ctrlat = 0.0, ctrlon = 0.0;
  // vertex i is (phi2, lat2); vertex ip is (phi1, lat1).  Per vertex: sincos() for poly_ctrlon's f, plain cos() for
  // poly_ctrlat's cos(lat1) -- the libm entries the reference's object code goes through (see d_poly_ctrlat above).
  double s2, c2;
  d_sincos_lat(y[0], &s2, &c2);
  const double s_first = s2, c_first = c2, cp_first = d_cos_lat(y[0]);
  for (int i = 0; i < n; i++) {
    const bool last = (i + 1 == n);
    int ip = last ? 0 : i + 1;
    double phi1 = x[ip * S], phi2 = x[i * S];
    double lat1 = y[ip * S], lat2 = y[i * S];
    double s1, c1, cp1;
    if (last) { s1 = s_first; c1 = c_first; cp1 = cp_first; } else { d_sincos_lat(lat1, &s1, &c1); cp1 = d_cos_lat(lat1); }
    double dx0 = phi1 - phi2;               // x[ip]-x[i]
    double avg_y = (lat1 + lat2) * 0.5;      // == 0.5*(lat1+lat2)
    const double savg = d_sin_lat(avg_y);    // poly_area: plain sin()
    double dyh = 0.5 * (lat1 - lat2);        // poly_area's dy; ctrlat's hdy == -dyh
    // poly_area tests |lat1-lat2| < 1e-10, poly_ctrlat tests |(lat2-lat1)/2| < 1e-10; the
    // first implies the second, so d_sin_lat(dyh)/dyh is needed exactly when the first fails.
    // ctrlat's d_sin_lat(hdy)/hdy with hdy == -dyh is the same number (sin is odd, negation exact).
    bool flat = fabs(lat1 - lat2) < G_SMALL;
    double dat = flat ? 1.0 : d_sin_lat(dyh) / dyh;

    // ---- poly_area (mosaic_util.c:421-450)
    {
      double dx = dx0;
      if (dx > G_PI)  dx = dx - 2.0 * G_PI;
      if (dx < -G_PI) dx = dx + 2.0 * G_PI;
      if (fabs(dx + G_PI) < G_SMALL || fabs(dx - G_PI) < G_SMALL) area += G_PI;
      else if (flat) area -= dx * savg;
      else area -= dx * savg * dat;
    }
    // ---- poly_ctrlat (create_xgrid.c:2100-2118)
    if (dx0 != 0.0) {
      double dx = dx0;
      if (dx > G_PI)   dx = dx - 2.0 * G_PI;
      if (dx <= -G_PI) dx = dx + 2.0 * G_PI;
      double hdy = (lat2 - lat1) * 0.5;
      if (fabs(hdy) < G_SMALL)
        ctrlat -= dx * (2 * d_cos_lat(avg_y) + lat2 * savg - cp1);          // flat branch: separate cos() and sin()
      else {
        double sa, ca;
        d_sincos_lat(avg_y, &sa, &ca);                                      // general branch: sincos(avg_y)
        const double datc = flat ? d_sin_lat(dyh) / dyh : dat;              // (|hdy| >= 1e-10 > |lat1-lat2| cannot happen)
        ctrlat -= dx * (datc * (2 * ca + lat2 * sa) - cp1);
      }
    }
    // ---- poly_ctrlon (create_xgrid.c:2176-2214)
    if (dx0 != 0.0) {
      double dphi = dx0;
      double f1 = 0.5 * (c1 * s1 + lat1);
      double f2 = 0.5 * (c2 * s2 + lat2);
      if (dphi > G_PI)  dphi = dphi - 2.0 * G_PI;
      if (dphi < -G_PI) dphi = dphi + 2.0 * G_PI;
      double dphi1 = phi1 - clon;
      if (dphi1 > G_PI)  dphi1 -= 2.0 * G_PI;
      if (dphi1 < -G_PI) dphi1 += 2.0 * G_PI;
      double dphi2 = phi2 - clon;
      if (dphi2 > G_PI)  dphi2 -= 2.0 * G_PI;
      if (dphi2 < -G_PI) dphi2 += 2.0 * G_PI;
      if (fabs(dphi2 - dphi1) < G_PI) {
        ctrlon -= dphi * (dphi1 * f1 + dphi2 * f2) / 2.0;
      } else {
        double fac = (dphi1 > 0.0) ? G_PI : -G_PI;
        double fint = f1 + (f2 - f1) * (fac - dphi1) / fabs(dphi);
        ctrlon -= 0.5 * dphi1 * (dphi1 - fac) * f1 - 0.5 * dphi2 * (dphi2 + fac) * f2
                  + 0.5 * fac * (dphi1 + dphi2) * fint;
      }
    }
    s2 = s1; c2 = c1;
  }
  *area_out = (area < 0) ? -area * G_RADIUS * G_RADIUS : area * G_RADIUS * G_RADIUS;
  *ctrlat_out = ctrlat * G_RADIUS * G_RADIUS;
  *ctrlon_out = ctrlon * G_RADIUS * G_RADIUS;
}
