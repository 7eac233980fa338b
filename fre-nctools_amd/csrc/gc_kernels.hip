// gc_kernels.hip -- great-circle exchange-grid search for gfx950 (MI355X).
//
// create_xgrid_great_circle (tools/libfrencutils/create_xgrid.c:1366-1466) clips every (source cell, destination
// cell) pair with clip_2dx2d_great_circle (:1479-1908): cell edges are great-circle arcs, vertices are unit vectors,
// the intersection parameters are solved in x87 extended precision (mosaic_util.c:967-1044) and the exchange-cell
// area is the spherical excess from acosl angles (mosaic_util.c:763-838).  Here:
//
//   k_gc_cell_struct   per cell: the four corners as xyz (clockwise), the great-circle cell area
//                      [get_grid_great_circle_area, create_xgrid.c:98-137], a bounding cap (centre, cos radius) and
//                      the cap's lat/lon box -- the box feeds the SAME binning / candidate kernels as the legacy path
//                      (xgrid_kernels.hip), which therefore emit a superset of the pairs whose caps touch
//   k_gc_clip          one lane per candidate pair: the reference's xyz bounding-box reject (RANGE_CHECK_CRITERIA),
//                      a cap-separation reject that is provably "n_out == 0" (DESIGN.md), then the reference's clip
//                      on small per-lane ordered arrays (the linked Node pool of mosaic_util.c:1046-1541 restated
//                      as arrays), area, and the 1e-6 area-ratio test
//
// Bit-faithfulness: every double operation has the reference's expression tree (-ffp-contract=off); the extended
// precision solve runs on the software x87 of fp80.h; acosl is fp80.h's fg_acosl.  Compaction into the canonical
// order (source cell, destination index) is the legacy path's k_compact (xgrid_kernels.hip).
#include "xgrid_device.h"
#include "fp80.h"
#include <algorithm>

#define GC_EPSLN8 (1.e-8)
#define GC_EPSLN10 (1.e-10)
#define GC_EPSLN30 (1.e-30)
#define GC_RANGE_CHECK 0.05
#define GC_RADIUS 6371000.0
#define GC_PI 3.14159265358979323846
#define GC_NCAP 14          // nodes per grid list: 4 corners + at most 8 intersections (convex x convex) + slack
#define GC_ICAP 12          // intersection list
#define GC_PCAP 16          // output polygon

struct GcNode { double x, y, z, u; int intersect, inbound, inside, pad; };
struct GcInter { double x, y, z, u, u_clip; int subj_index, clip_index, inbound, pad; };
struct GcList { int n; GcNode v[GC_NCAP]; };

__device__ __forceinline__ bool gc_same_point(double x1, double y1, double z1, double x2, double y2, double z2)
{
  return !(fabs(x1 - x2) > GC_EPSLN10 || fabs(y1 - y2) > GC_EPSLN10 || fabs(z1 - z2) > GC_EPSLN10);
}

// spherical_angle, mosaic_util.c:799-836 (double branch); EXACT selects the reference-exact acosl
template <bool EXACT>
__device__ double gc_spherical_angle(const double *v1, const double *v2, const double *v3)
{
  double angle, px, py, pz, qx, qy, qz, ddd;
  px = v1[1] * v2[2] - v1[2] * v2[1];
  py = v1[2] * v2[0] - v1[0] * v2[2];
  pz = v1[0] * v2[1] - v1[1] * v2[0];
  qx = v1[1] * v3[2] - v1[2] * v3[1];
  qy = v1[2] * v3[0] - v1[0] * v3[2];
  qz = v1[0] * v3[1] - v1[1] * v3[0];
  ddd = (px * px + py * py + pz * pz) * (qx * qx + qy * qy + qz * qz);
  if (ddd <= 0.0) angle = 0.;
  else {
    ddd = (px * qx + py * qy + pz * qz) / sqrt(ddd);
    if (fabs(ddd - 1) < GC_EPSLN30) ddd = 1;
    if (fabs(ddd + 1) < GC_EPSLN30) ddd = -1;
    if (ddd > 1. || ddd < -1.) {
      if (ddd < 0.) angle = GC_PI;
      else angle = 0.;
    } else
      angle = EXACT ? fg_acosl(ddd) : acos(ddd);
  }
  return angle;
}

// great_circle_area, mosaic_util.c:763-787, on a polygon given as strided xyz
__device__ double gc_area(int n, const double *p, int stride)
{
  double sum = 0.0;
  for (int i = 0; i < n; i++) {
    const double *p0 = p + (size_t)i * stride, *p1 = p + (size_t)((i + 1) % n) * stride, *p2 = p + (size_t)((i + 2) % n) * stride;
    sum += gc_spherical_angle<true>(p1, p2, p0);
  }
  return (sum - (n - 2.) * GC_PI) * GC_RADIUS * GC_RADIUS;
}

// insidePolygon, mosaic_util.c:1546-1589.  The angle sum is only compared with 2*pi to 1e-8, so it is taken with the
// fast acos first and redone with the exact one only if it lands within 1e-12 of the threshold.
__device__ int gc_inside_polygon(const GcNode &node, const GcList &l)
{
  const double pnt0[3] = {node.x, node.y, node.z};
  double anglesum = 0;
  for (int k = 0; k < l.n; k++) {
    const int kn = (k + 1 < l.n) ? k + 1 : 0;
    const double pnt1[3] = {l.v[k].x, l.v[k].y, l.v[k].z}, pnt2[3] = {l.v[kn].x, l.v[kn].y, l.v[kn].z};
    if (gc_same_point(pnt0[0], pnt0[1], pnt0[2], pnt1[0], pnt1[1], pnt1[2])) return 1;
    anglesum += gc_spherical_angle<false>(pnt0, pnt2, pnt1);
  }
  double dev = fabs(anglesum - 2 * GC_PI);
  if (fabs(dev - GC_EPSLN8) < 1.e-12) {
    anglesum = 0;
    for (int k = 0; k < l.n; k++) {
      const int kn = (k + 1 < l.n) ? k + 1 : 0;
      const double pnt1[3] = {l.v[k].x, l.v[k].y, l.v[k].z}, pnt2[3] = {l.v[kn].x, l.v[kn].y, l.v[kn].z};
      anglesum += gc_spherical_angle<true>(pnt0, pnt2, pnt1);
    }
    dev = fabs(anglesum - 2 * GC_PI);
  }
  return dev < GC_EPSLN8;
}

// intersect_tri_with_line + invert_matrix_3x3 + mult (mosaic_util.c:967-1044) on the software x87; the third plane
// point is the origin.  Only t = X[0] is consumed by line_intersect_2D_3D.
// Double-precision screen of one plane/segment solve: when the system is well conditioned
// (det^2 > 1e-6 * |l1-l2|^2 * |pnt1-pnt0|^2, both edges longer than 1e-4 rad) the double solution is within ~1e-10 of
// the extended one, and stays within 1e-6 of it when an endpoint is later snapped onto an intersection (a move of at most
// 1e-10 absolute or 1e-8 of an edge), so a t outside [-1e-5, 1+1e-5] is certainly outside the reference's accepted
// range [-1e-8, 1+1e-8] (create_xgrid.c:1993-2001): line_intersect_2D_3D returns 0 for that edge pair.
__device__ bool gc_screen_out(const double *pnt0, const double *pnt1, const double *l1, const double *l2)
{
  {                                                              // both ends clear of the plane on one side (see gc_half_screen)
    double n[3];
    n[0] = pnt0[1] * pnt1[2] - pnt0[2] * pnt1[1]; n[1] = pnt0[2] * pnt1[0] - pnt0[0] * pnt1[2]; n[2] = pnt0[0] * pnt1[1] - pnt0[1] * pnt1[0];
    const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    const double r1 = n[0] * l1[0] + n[1] * l1[1] + n[2] * l1[2], r2 = n[0] * l2[0] + n[1] * l2[1] + n[2] * l2[2];
    const double m1 = fabs(r1), m2 = fabs(r2);
    if ((r1 > 0) == (r2 > 0) && r1 * r1 > 1.e-12 * nn && r2 * r2 > 1.e-12 * nn && fmin(m1, m2) >= 1.e-5 * fmax(m1, m2)) return true;
  }
  const double d0 = l1[0] - l2[0], d1 = pnt1[0] - pnt0[0], d2 = 0.0 - pnt0[0];
  const double d3 = l1[1] - l2[1], d4 = pnt1[1] - pnt0[1], d5 = 0.0 - pnt0[1];
  const double d6 = l1[2] - l2[2], d7 = pnt1[2] - pnt0[2], d8 = 0.0 - pnt0[2];
  const double e0 = d4 * d8 - d5 * d7, e1 = d3 * d8 - d5 * d6, e2 = d3 * d7 - d4 * d6;
  const double det = d0 * e0 - d1 * e1 + d2 * e2;
  const double n1 = d0 * d0 + d3 * d3 + d6 * d6, n2 = d1 * d1 + d4 * d4 + d7 * d7;
  if (!(n1 > 1.e-8 && n2 > 1.e-8 && det * det > 1.e-6 * n1 * n2)) return false;
  const double v0 = l1[0] - pnt0[0], v1 = l1[1] - pnt0[1], v2 = l1[2] - pnt0[2];
  const double td = (e0 * v0 + (d2 * d7 - d1 * d8) * v1 + (d1 * d5 - d2 * d4) * v2) / det;
  return td < -1.e-5 || td > 1.0 + 1.e-5;
}

__device__ bool gc_tri_line_t(const double *pnt0, const double *pnt1, const double *l1, const double *l2, double *t)
{
  const x80 m0 = x80_from_double(l1[0] - l2[0]), m1 = x80_from_double(pnt1[0] - pnt0[0]), m2 = x80_from_double(0.0 - pnt0[0]);
  const x80 m3 = x80_from_double(l1[1] - l2[1]), m4 = x80_from_double(pnt1[1] - pnt0[1]), m5 = x80_from_double(0.0 - pnt0[1]);
  const x80 m6 = x80_from_double(l1[2] - l2[2]), m7 = x80_from_double(pnt1[2] - pnt0[2]), m8 = x80_from_double(0.0 - pnt0[2]);
  const x80 c0 = x80_sub(x80_mul(m4, m8), x80_mul(m5, m7));
  const x80 c1 = x80_sub(x80_mul(m3, m8), x80_mul(m5, m6));
  const x80 c2 = x80_sub(x80_mul(m3, m7), x80_mul(m4, m6));
  const x80 det = x80_add(x80_sub(x80_mul(m0, c0), x80_mul(m1, c1)), x80_mul(m2, c2));
  if (x80_abs_lt(det, x80_from_double(1.e-15))) return false;
  const x80 deti = x80_div(x80_from_double(1.0), det);
  const x80 inv0 = x80_mul(c0, deti);
  const x80 inv1 = x80_mul(x80_sub(x80_mul(m2, m7), x80_mul(m1, m8)), deti);
  const x80 inv2 = x80_mul(x80_sub(x80_mul(m1, m5), x80_mul(m2, m4)), deti);
  const x80 V0 = x80_from_double(l1[0] - pnt0[0]), V1 = x80_from_double(l1[1] - pnt0[1]), V2 = x80_from_double(l1[2] - pnt0[2]);
  *t = x80_to_double(x80_add(x80_add(x80_mul(inv0, V0), x80_mul(inv1, V1)), x80_mul(inv2, V2)));
  return true;
}

__device__ __forceinline__ void gc_cross(const double *p1, const double *p2, double *e)
{
  e[0] = p1[1] * p2[2] - p1[2] * p2[1];
  e[1] = p1[2] * p2[0] - p1[0] * p2[2];
  e[2] = p1[0] * p2[1] - p1[1] * p2[0];
}
__device__ __forceinline__ double gc_metric(const double *p) { return sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]); }

// line_intersect_2D_3D, create_xgrid.c:1919-2081
__device__ int gc_line_intersect(const double *a1, const double *a2, const double *q1, const double *q2, const double *q3,
                                 double *I, double *u_a, double *u_q, int *inbound)
{
  *inbound = 0;
  if (gc_same_point(a1[0], a1[1], a1[2], q1[0], q1[1], q1[2])) { *u_a = 0; *u_q = 0; I[0] = a1[0]; I[1] = a1[1]; I[2] = a1[2]; return 1; }
  else if (gc_same_point(a1[0], a1[1], a1[2], q2[0], q2[1], q2[2])) { *u_a = 0; *u_q = 1; I[0] = a1[0]; I[1] = a1[1]; I[2] = a1[2]; return 1; }
  else if (gc_same_point(a2[0], a2[1], a2[2], q1[0], q1[1], q1[2])) { *u_a = 1; *u_q = 0; I[0] = a2[0]; I[1] = a2[1]; I[2] = a2[2]; return 1; }
  else if (gc_same_point(a2[0], a2[1], a2[2], q2[0], q2[1], q2[2])) { *u_a = 1; *u_q = 1; I[0] = a2[0]; I[1] = a2[1]; I[2] = a2[2]; return 1; }
  if (!gc_tri_line_t(q1, q2, a1, a2, u_a)) return 0;
  if (fabs(*u_a) < GC_EPSLN8) *u_a = 0;
  if (fabs(*u_a - 1) < GC_EPSLN8) *u_a = 1;
  if ((*u_a < 0) || (*u_a > 1)) return 0;
  if (!gc_tri_line_t(a1, a2, q1, q2, u_q)) return 0;
  if (fabs(*u_q) < GC_EPSLN8) *u_q = 0;
  if (fabs(*u_q - 1) < GC_EPSLN8) *u_q = 1;
  if ((*u_q < 0) || (*u_q > 1)) return 0;
  const double u = *u_a;
  double c1[3], c2[3], c3[3];
  gc_cross(a1, a2, c1);
  gc_cross(q1, q2, c2);
  gc_cross(c1, c2, c3);
  const double coincident = gc_metric(c3);
  if (fabs(coincident) < GC_EPSLN30) return 0;
  I[0] = a1[0] + u * (a2[0] - a1[0]);
  I[1] = a1[1] + u * (a2[1] - a1[1]);
  I[2] = a1[2] + u * (a2[2] - a1[2]);
  const double norm = gc_metric(I);
  I[0] /= norm; I[1] /= norm; I[2] /= norm;
  if (*u_q != 0 && *u_q != 1) {
    const double p1[3] = {a2[0] - a1[0], a2[1] - a1[1], a2[2] - a1[2]};
    const double v1[3] = {q2[0] - q1[0], q2[1] - q1[1], q2[2] - q1[2]};
    const double v2[3] = {q3[0] - q2[0], q3[1] - q2[1], q3[2] - q2[2]};
    gc_cross(v1, v2, c1);
    gc_cross(v1, p1, c2);
    const double sense = c1[0] * c2[0] + c1[1] * c2[1] + c1[2] * c2[2];
    *inbound = 1;
    if (sense > 0) *inbound = 2;
  }
  return 1;
}

// addEnd (mosaic_util.c:1096-1135) on a grid list: append unless a point within 1e-10 is present
__device__ int gc_add_end(GcList &l, double x, double y, double z, int intersect, double u, int inbound, int inside)
{
  for (int k = 0; k < l.n; k++) if (gc_same_point(l.v[k].x, l.v[k].y, l.v[k].z, x, y, z)) return 0;
  if (l.n >= GC_NCAP) return -9;
  GcNode &t = l.v[l.n++];
  t.x = x; t.y = y; t.z = z; t.u = u; t.intersect = intersect; t.inbound = inbound; t.inside = inside; t.pad = 0;
  return 0;
}

// insertIntersect, mosaic_util.c:1313-1397
__device__ int gc_insert_intersect(GcList &l, double x, double y, double z, double u1, double u2, int inbound,
                                   double x2, double y2, double z2)
{
  int k1 = -1;
  for (int k = 0; k < l.n; k++) if (l.v[k].x == x2 && l.v[k].y == y2 && l.v[k].z == z2) { k1 = k; break; }
  if (k1 < 0) return -7;
  double u_cur = u1;
  if (u1 == 1) { u_cur = 0; k1 = (k1 + 1 < l.n) ? k1 + 1 : 0; }
  if (u_cur == 0) {
    GcNode &t = l.v[k1];
    t.intersect = 2; t.inside = 1; t.u = u_cur; t.x = x; t.y = y; t.z = z;
    return 0;
  }
  if (u2 != 0 && u2 != 1) {
    if (inbound == 1) {
      int k2 = (k1 + 1 < l.n) ? k1 + 1 : 0, guard = 0;
      while (l.v[k2].intersect) { k2 = (k2 + 1 < l.n) ? k2 + 1 : 0; if (++guard > 2 * GC_NCAP) return -7; }
      l.v[k2].inside = 0;
    } else if (inbound == 2)
      l.v[k1].inside = 0;
  }
  int k2 = k1 + 1;
  while (k2 < l.n) {
    if (l.v[k2].intersect == 1) { if (l.v[k2].u > u_cur) break; }
    else break;
    k1 = k2; k2++;
  }
  if (l.n >= GC_NCAP) return -9;
  for (int k = l.n; k > k2; k--) l.v[k] = l.v[k - 1];
  l.n++;
  GcNode &t = l.v[k2];
  t.x = x; t.y = y; t.z = z; t.u = u_cur; t.intersect = 1; t.inbound = inbound; t.inside = 1; t.pad = 0;
  return 0;
}

__device__ __forceinline__ int gc_find(const GcList &l, double x, double y, double z)
{
  for (int k = 0; k < l.n; k++) if (l.v[k].x == x && l.v[k].y == y && l.v[k].z == z) return k;
  return -1;
}

// polyList: addNode -> addEnd (dedup within 1e-10)
struct GcPoly { int n; double p[GC_PCAP][3]; };
__device__ int gc_poly_add(GcPoly &pl, double x, double y, double z)
{
  for (int k = 0; k < pl.n; k++) if (gc_same_point(pl.p[k][0], pl.p[k][1], pl.p[k][2], x, y, z)) return 0;
  if (pl.n >= GC_PCAP) return -9;
  pl.p[pl.n][0] = x; pl.p[pl.n][1] = y; pl.p[pl.n][2] = z; pl.n++;
  return 0;
}

// clip_2dx2d_great_circle (create_xgrid.c:1479-1908) after its bounding-box rejects.  a, b: the four corners of the
// two cells, [k*3 + axis].  Returns n_out (vertices in out) or a negative error code (see oracle/gc_oracle.c).
__device__ int gc_clip(const double *a, const double *b, GcPoly &out)
{
  GcList g1, g2;
  GcInter il[GC_ICAP];
  int nil = 0;
  g1.n = g2.n = 0; out.n = 0;
  for (int i = 0; i < 4; i++) if (gc_add_end(g1, a[i * 3], a[i * 3 + 1], a[i * 3 + 2], 0, 0, 0, -1)) return -9;
  for (int i = 0; i < 4; i++) if (gc_add_end(g2, b[i * 3], b[i * 3 + 1], b[i * 3 + 2], 0, 0, 0, -1)) return -9;
  const int npts1 = g1.n, npts2 = g2.n;
  for (int k = 0; k < g1.n; k++) g1.v[k].inside = gc_inside_polygon(g1.v[k], g2);
  for (int k = 0; k < g2.n; k++) g2.v[k].inside = gc_inside_polygon(g2.v[k], g1);

  double pt1[4][3], pt2[4][3];
  for (int i = 0; i < npts1; i++) { pt1[i][0] = g1.v[i].x; pt1[i][1] = g1.v[i].y; pt1[i][2] = g1.v[i].z; }
  for (int i = 0; i < npts2; i++) { pt2[i][0] = g2.v[i].x; pt2[i][1] = g2.v[i].y; pt2[i][2] = g2.v[i].z; }

  for (int i1 = 0; i1 < npts1; i1++) {
    const int i1p = (i1 + 1) % npts1;
    double *p1_0 = pt1[i1], *p1_1 = pt1[i1p];
    for (int i2 = 0; i2 < npts2; i2++) {
      const int i2p = (i2 + 1) % npts2, i2p2 = (i2 + 2) % npts2;
      double *p2_0 = pt2[i2], *p2_1 = pt2[i2p], *p2_2 = pt2[i2p2], I[3], u1, u2;
      int inbound;
      if (!gc_line_intersect(p1_0, p1_1, p2_0, p2_1, p2_2, I, &u1, &u2, &inbound)) continue;
      // addIntersect, mosaic_util.c:1139-1190
      double u1c = u1, u2c = u2;
      int i1c = i1, i2c = i2;
      if (u1c == 1) { u1c = 0; i1c = i1p; }
      if (u2c == 1) { u2c = 0; i2c = i2p; }
      bool dup = false;
      for (int k = 0; k < nil; k++) {
        if (il[k].u == u1c && il[k].subj_index == i1c) { dup = true; break; }
        if (il[k].u_clip == u2c && il[k].clip_index == i2c) { dup = true; break; }
      }
      if (dup) continue;
      if (nil >= GC_ICAP) return -9;
      GcInter &t = il[nil++];
      t.x = I[0]; t.y = I[1]; t.z = I[2]; t.u = u1c; t.u_clip = u2c; t.subj_index = i1c; t.clip_index = i2c; t.inbound = inbound; t.pad = 0;
      int rc;
      if (u1 == 1) rc = gc_insert_intersect(g1, I[0], I[1], I[2], 0.0, u2, inbound, p1_1[0], p1_1[1], p1_1[2]);
      else rc = gc_insert_intersect(g1, I[0], I[1], I[2], u1, u2, inbound, p1_0[0], p1_0[1], p1_0[2]);
      if (rc) return rc;
      if (u1 == 1) { p1_1[0] = I[0]; p1_1[1] = I[1]; p1_1[2] = I[2]; }
      else if (u1 == 0) { p1_0[0] = I[0]; p1_0[1] = I[1]; p1_0[2] = I[2]; }
      if (u2 == 1) rc = gc_insert_intersect(g2, I[0], I[1], I[2], 0.0, u1, 0, p2_1[0], p2_1[1], p2_1[2]);
      else rc = gc_insert_intersect(g2, I[0], I[1], I[2], u2, u1, 0, p2_0[0], p2_0[1], p2_0[2]);
      if (rc) return rc;
      if (u2 == 1) { p2_1[0] = I[0]; p2_1[1] = I[1]; p2_1[2] = I[2]; }
      else if (u2 == 0) { p2_0[0] = I[0]; p2_0[1] = I[1]; p2_0[2] = I[2]; }
    }
  }

  // first inbound intersection (getFirstInbound / setInbound, mosaic_util.c:1448-1535)
  int nintersect = nil, first = -1;
  if (nintersect > 1) for (int k = 0; k < nil; k++) if (il[k].inbound == 2) { first = k; break; }
  if (first < 0 && nintersect > 1) {
    for (int k = 0; k < nil; k++) {
      if (il[k].inbound) continue;
      const int f = gc_find(g1, il[k].x, il[k].y, il[k].z);
      if (f < 0) return -8;
      const GcNode &prev = g1.v[f > 0 ? f - 1 : g1.n - 1], &next = g1.v[f + 1 < g1.n ? f + 1 : 0];
      il[k].inbound = (prev.inside == 0 && next.inside == 1) ? 2 : 1;
    }
    for (int k = 0; k < nil; k++) if (il[k].inbound == 2) { first = k; break; }
  }

  int n_out = 0;
  if (first >= 0) {
    const double fx = il[first].x, fy = il[first].y, fz = il[first].z;
    const int maxiter1 = nintersect;
    if (gc_find(g1, fx, fy, fz) < 0) return -3;
    if (gc_poly_add(out, fx, fy, fz)) return -9;
    nintersect--;
    GcList *curl = &g1;
    int cur_num = 0, iter1 = 0, found1 = 0, found2 = 0;
    double cx = fx, cy = fy, cz = fz;
    while (iter1 < maxiter1) {
      const int k1 = gc_find(*curl, cx, cy, cz);
      if (k1 < 0) return -4;
      int k2 = (k1 + 1 < curl->n) ? k1 + 1 : 0;
      const int maxiter2 = curl->n;
      int iter2 = 0;
      found2 = 0;
      while (iter2 < maxiter2) {
        int t2_is_inter = 0;
        const GcNode &t2 = curl->v[k2];
        if (t2.intersect) {
          if (t2.x == fx && t2.y == fy && t2.z == fz) { found1 = 1; break; }
          const GcNode &t3 = curl->v[(k2 + 1 < curl->n) ? k2 + 1 : 0];
          found2 = 1;
          t2_is_inter = 1;
          if (t3.intersect || (t3.inside == 1)) found2 = 0;
        }
        if (found2) { cx = t2.x; cy = t2.y; cz = t2.z; break; }
        else {
          if (gc_poly_add(out, t2.x, t2.y, t2.z)) return -9;
          if (t2_is_inter) nintersect--;
        }
        k2 = (k2 + 1 < curl->n) ? k2 + 1 : 0;
        iter2++;
      }
      if (found1) break;
      if (!found2) return -4;
      if (cx == fx && cy == fy && cz == fz) { found1 = 1; break; }
      if (gc_poly_add(out, cx, cy, cz)) return -9;
      nintersect--;
      if (cur_num == 0) { curl = &g2; cur_num = 1; }
      else { curl = &g1; cur_num = 0; }
      iter1++;
    }
    if (!found1) return -5;
    if (nintersect > 0) return -6;
    n_out = out.n;
    if (n_out < 3) n_out = 0;
  }
  if (n_out == 0) {                                   // grid1 inside grid2, :1839-1870
    int n1in2 = 0;
    for (int k = 0; k < g1.n; k++) if (g1.v[k].intersect != 1 && g1.v[k].inside == 1) n1in2++;
    if (npts1 == n1in2) {
      n_out = npts1;
      for (int k = 0; k < npts1; k++) { out.p[k][0] = g1.v[k].x; out.p[k][1] = g1.v[k].y; out.p[k][2] = g1.v[k].z; }
    }
    if (n_out > 0) { out.n = n_out; return n_out; }
  }
  if (n_out == 0) {                                   // grid2 inside grid1, :1873-1904
    int n2in1 = 0;
    for (int k = 0; k < g2.n; k++) if (g2.v[k].intersect != 1 && g2.v[k].inside == 1) n2in1++;
    if (npts2 == n2in1) {
      n_out = npts2;
      for (int k = 0; k < npts2; k++) { out.p[k][0] = g2.v[k].x; out.p[k][1] = g2.v[k].y; out.p[k][2] = g2.v[k].z; }
    }
  }
  out.n = n_out;
  return n_out;
}

// ------------------------------------------------------------------------------------------------ compact clip
// The same algorithm with ~0.5 KB of per-lane state instead of 2.6 KB (k_gc_clip was bound by scratch traffic:
// 65 GB per launch at C384, profiles/r01_summary.md).  Coordinates live once in three small tables -- the two vertex
// arrays (which ARE the reference's pt1/pt2 arrays, create_xgrid.c:1589-1598: a vertex node and its pt entry are always
// rewritten together, :1639-1665) and the intersections (each appears in both grid lists with the same coordinates);
// the lists are byte codes packed in one 128-bit register value:
//   bits 0-3 ref (0..3 own vertex, 4..15 intersection ref-4), bits 4-5 intersect (0/1/2), bit 6 isInside.
// Every search stays a search by coordinate VALUE in list order, as in the reference.  The two situations the shared
// tables cannot represent (an anchor search that lands on a node other than the expected vertex while a vertex is being
// rewritten) return GC_FALLBACK and the pair is redone by the array version above (k_gc_clip_slow); they need two
// distinct nodes with bit-identical coordinates and have not been observed.
#define GC_FALLBACK (-10)
#define GC_FI 8             // intersections kept by the compact version (two convex quads cross at most 8 times)
typedef unsigned __int128 gc_u128;
struct GcPacked { gc_u128 bits; int n; };
__device__ __forceinline__ unsigned gcp_get(const GcPacked &l, int k) { return (unsigned)(l.bits >> (8 * k)) & 0xffu; }
__device__ __forceinline__ void gcp_set(GcPacked &l, int k, unsigned v)
{
  l.bits = (l.bits & ~((gc_u128)0xff << (8 * k))) | ((gc_u128)(v & 0xffu) << (8 * k));
}
__device__ __forceinline__ void gcp_insert(GcPacked &l, int k, unsigned v)
{
  const gc_u128 low = (k == 0) ? (gc_u128)0 : (l.bits & (((gc_u128)1 << (8 * k)) - 1));
  const gc_u128 high = (l.bits >> (8 * k)) << (8 * (k + 1));
  l.bits = low | ((gc_u128)(v & 0xffu) << (8 * k)) | high;
  l.n++;
}
#define GCN_REF(c) ((c) & 15u)
#define GCN_INTER(c) (((c) >> 4) & 3u)
#define GCN_INSIDE(c) (((c) >> 6) & 1u)

struct GcTables {
  double vt[2][4][3];        // pt1 / pt2
  double ix[GC_FI][3];       // intersection coordinates
  double iu[GC_FI][2];       // u along the grid-1 edge, u along the grid-2 edge (0 when snapped onto a vertex)
  unsigned imeta[GC_FI];     // subj_index | clip_index << 4 | inbound << 8
};
__device__ __forceinline__ const double *gct_xyz(const GcTables &T, int L, unsigned code)
{
  const unsigned r = GCN_REF(code);
  return (r < 4) ? T.vt[L][r] : T.ix[r - 4];
}
__device__ __forceinline__ int gcp_find(const GcTables &T, int L, const GcPacked &l, double x, double y, double z)
{
  for (int k = 0; k < l.n; k++) {
    const double *q = gct_xyz(T, L, gcp_get(l, k));
    if (q[0] == x && q[1] == y && q[2] == z) return k;
  }
  return -1;
}

// insidePolygon against the (not yet modified) vertices of list L
__device__ int gcf_inside(const double *pnt0, const GcTables &T, int L, int n)
{
  double anglesum = 0;
  for (int k = 0; k < n; k++) {
    const int kn = (k + 1 < n) ? k + 1 : 0;
    const double *pnt1 = T.vt[L][k], *pnt2 = T.vt[L][kn];
    if (gc_same_point(pnt0[0], pnt0[1], pnt0[2], pnt1[0], pnt1[1], pnt1[2])) return 1;
    anglesum += gc_spherical_angle<false>(pnt0, pnt2, pnt1);
  }
  double dev = fabs(anglesum - 2 * GC_PI);
  if (fabs(dev - GC_EPSLN8) < 1.e-12) {
    anglesum = 0;
    for (int k = 0; k < n; k++) {
      const int kn = (k + 1 < n) ? k + 1 : 0;
      anglesum += gc_spherical_angle<true>(pnt0, T.vt[L][kn], T.vt[L][k]);
    }
    dev = fabs(anglesum - 2 * GC_PI);
  }
  return dev < GC_EPSLN8;
}

// insertIntersect (mosaic_util.c:1313-1397) on the packed list L.  expect = vertex index whose pt entry the caller
// rewrites when u_cur == 0; iref = index of the intersection in the tables.
__device__ int gcf_insert(GcTables &T, int L, GcPacked &l, const double *I, double u_cur, double u2, int inbound,
                          const double *anchor, int expect, int iref)
{
  int k1 = gcp_find(T, L, l, anchor[0], anchor[1], anchor[2]);
  if (k1 < 0) return -7;
  if (u_cur == 0) {
    const unsigned c = gcp_get(l, k1);
    if ((int)GCN_REF(c) != expect) return GC_FALLBACK;
    gcp_set(l, k1, (c & 15u) | (2u << 4) | (1u << 6));        // intersect = 2, isInside = 1; coordinates: caller rewrites vt
    return 0;
  }
  if (u2 != 0 && u2 != 1) {
    if (inbound == 1) {
      int k2 = (k1 + 1 < l.n) ? k1 + 1 : 0, guard = 0;
      while (GCN_INTER(gcp_get(l, k2))) { k2 = (k2 + 1 < l.n) ? k2 + 1 : 0; if (++guard > 32) return -7; }
      gcp_set(l, k2, gcp_get(l, k2) & ~(1u << 6));
    } else if (inbound == 2)
      gcp_set(l, k1, gcp_get(l, k1) & ~(1u << 6));
  }
  int k2 = k1 + 1;
  while (k2 < l.n) {
    const unsigned c = gcp_get(l, k2);
    if (GCN_INTER(c) == 1) { if (T.iu[GCN_REF(c) - 4][L] > u_cur) break; }
    else break;
    k2++;
  }
  if (l.n >= 12) return -9;
  gcp_insert(l, k2, (unsigned)(4 + iref) | (1u << 4) | (1u << 6));
  return 0;
}

// output polygon as refs: 0..3 grid-1 vertex, 4..7 grid-2 vertex, 8.. intersection; dedup by samePoint like addEnd
struct GcPolyRefs { unsigned long long bits; int n; };
__device__ __forceinline__ const double *gcpoly_xyz(const GcTables &T, unsigned r)
{
  return (r < 4) ? T.vt[0][r] : (r < 8 ? T.vt[1][r - 4] : T.ix[r - 8]);
}
__device__ int gcpoly_add(const GcTables &T, GcPolyRefs &pl, int L, unsigned code)
{
  const unsigned r = (GCN_REF(code) < 4) ? GCN_REF(code) + 4u * L : GCN_REF(code) + 4u;
  const double *q = gcpoly_xyz(T, r);
  for (int k = 0; k < pl.n; k++) {
    const double *e = gcpoly_xyz(T, (unsigned)(pl.bits >> (4 * k)) & 15u);
    if (gc_same_point(e[0], e[1], e[2], q[0], q[1], q[2])) return 0;
  }
  if (pl.n >= 16) return -9;
  pl.bits |= (unsigned long long)r << (4 * pl.n);
  pl.n++;
  return 0;
}

// Returns n_out and the polygon coordinates in out[n_out][3], GC_FALLBACK, or a negative error code.
__device__ int gc_clip_fast(const double *a, const double *b, double *out)
{
  GcTables T;
  GcPacked g[2];
  int nil = 0;
  // addEnd: de-duplicated corners; pt arrays are filled from the lists (create_xgrid.c:1589-1598)
  for (int L = 0; L < 2; L++) {
    const double *v = L ? b : a;
    int n = 0;
    g[L].bits = 0;
    for (int i = 0; i < 4; i++) {
      bool dup = false;
      for (int m = 0; m < n; m++) if (gc_same_point(T.vt[L][m][0], T.vt[L][m][1], T.vt[L][m][2], v[i * 3], v[i * 3 + 1], v[i * 3 + 2])) dup = true;
      if (dup) continue;
      T.vt[L][n][0] = v[i * 3]; T.vt[L][n][1] = v[i * 3 + 1]; T.vt[L][n][2] = v[i * 3 + 2];
      g[L].bits |= (gc_u128)(unsigned)n << (8 * n);
      n++;
    }
    g[L].n = n;
  }
  const int npts1 = g[0].n, npts2 = g[1].n;
  for (int L = 0; L < 2; L++)
    for (int k = 0; k < g[L].n; k++)
      if (gcf_inside(T.vt[L][k], T, 1 - L, g[1 - L].n)) gcp_set(g[L], k, gcp_get(g[L], k) | (1u << 6));

  // pass 1 (all lanes, double only): edge pairs that certainly do not intersect.  Lanes then loop only over their
  // remaining pairs, in the reference's (i1, i2) order -- a wave runs max-over-lanes of ~4-8 extended solves instead of 16.
  unsigned need = 0;
  for (int i1 = 0; i1 < npts1; i1++) {
    const int i1p = (i1 + 1) % npts1;
    for (int i2 = 0; i2 < npts2; i2++) {
      const int i2p = (i2 + 1) % npts2;
      const bool out = gc_screen_out(T.vt[1][i2], T.vt[1][i2p], T.vt[0][i1], T.vt[0][i1p]) ||
                       gc_screen_out(T.vt[0][i1], T.vt[0][i1p], T.vt[1][i2], T.vt[1][i2p]);
      if (!out) need |= 1u << (i1 * 4 + i2);
    }
  }
  while (need) {
    const int bit = __ffs((int)need) - 1;
    need &= need - 1;
    const int i1 = bit >> 2, i2 = bit & 3;
    {
      const int i1p = (i1 + 1) % npts1;
      const int i2p = (i2 + 1) % npts2, i2p2 = (i2 + 2) % npts2;
      double *p1_0 = T.vt[0][i1], *p1_1 = T.vt[0][i1p], *p2_0 = T.vt[1][i2], *p2_1 = T.vt[1][i2p], *p2_2 = T.vt[1][i2p2];
      double I[3], u1, u2;
      int inbound;
      if (!gc_line_intersect(p1_0, p1_1, p2_0, p2_1, p2_2, I, &u1, &u2, &inbound)) continue;
      double u1c = u1, u2c = u2;
      int i1c = i1, i2c = i2;
      if (u1c == 1) { u1c = 0; i1c = i1p; }
      if (u2c == 1) { u2c = 0; i2c = i2p; }
      bool dup = false;
      for (int k = 0; k < nil; k++) {
        if (T.iu[k][0] == u1c && (int)(T.imeta[k] & 15u) == i1c) { dup = true; break; }
        if (T.iu[k][1] == u2c && (int)((T.imeta[k] >> 4) & 15u) == i2c) { dup = true; break; }
      }
      if (dup) continue;
      if (nil >= GC_FI) return GC_FALLBACK;
      const int iref = nil++;
      T.ix[iref][0] = I[0]; T.ix[iref][1] = I[1]; T.ix[iref][2] = I[2];
      T.iu[iref][0] = u1c; T.iu[iref][1] = u2c;
      T.imeta[iref] = (unsigned)i1c | ((unsigned)i2c << 4) | ((unsigned)inbound << 8);
      int rc;
      if (u1 == 1) rc = gcf_insert(T, 0, g[0], I, 0.0, u2, inbound, p1_1, i1p, iref);
      else rc = gcf_insert(T, 0, g[0], I, u1, u2, inbound, p1_0, i1, iref);
      if (rc) return rc;
      if (u1 == 1) { p1_1[0] = I[0]; p1_1[1] = I[1]; p1_1[2] = I[2]; }
      else if (u1 == 0) { p1_0[0] = I[0]; p1_0[1] = I[1]; p1_0[2] = I[2]; }
      if (u2 == 1) rc = gcf_insert(T, 1, g[1], I, 0.0, u1, 0, p2_1, i2p, iref);
      else rc = gcf_insert(T, 1, g[1], I, u2, u1, 0, p2_0, i2, iref);
      if (rc) return rc;
      if (u2 == 1) { p2_1[0] = I[0]; p2_1[1] = I[1]; p2_1[2] = I[2]; }
      else if (u2 == 0) { p2_0[0] = I[0]; p2_0[1] = I[1]; p2_0[2] = I[2]; }
    }
  }

  int nintersect = nil, first = -1;
  if (nintersect > 1) for (int k = 0; k < nil; k++) if (((T.imeta[k] >> 8) & 3u) == 2) { first = k; break; }
  if (first < 0 && nintersect > 1) {
    for (int k = 0; k < nil; k++) {
      if ((T.imeta[k] >> 8) & 3u) continue;
      const int f = gcp_find(T, 0, g[0], T.ix[k][0], T.ix[k][1], T.ix[k][2]);
      if (f < 0) return -8;
      const unsigned prev = gcp_get(g[0], f > 0 ? f - 1 : g[0].n - 1), next = gcp_get(g[0], f + 1 < g[0].n ? f + 1 : 0);
      T.imeta[k] |= ((GCN_INSIDE(prev) == 0 && GCN_INSIDE(next) == 1) ? 2u : 1u) << 8;
    }
    for (int k = 0; k < nil; k++) if (((T.imeta[k] >> 8) & 3u) == 2) { first = k; break; }
  }

  int n_out = 0;
  GcPolyRefs poly; poly.bits = 0; poly.n = 0;
  if (first >= 0) {
    const double fx = T.ix[first][0], fy = T.ix[first][1], fz = T.ix[first][2];
    const int maxiter1 = nintersect;
    const int kf = gcp_find(T, 0, g[0], fx, fy, fz);
    if (kf < 0) return -3;
    // addNode(polyList, firstIntersect): the coordinates of the intersection-list entry
    poly.bits = (unsigned long long)(8 + first); poly.n = 1;
    nintersect--;
    int L = 0, iter1 = 0, found1 = 0, found2 = 0;
    double cx = fx, cy = fy, cz = fz;
    unsigned ccode = 0;
    while (iter1 < maxiter1) {
      const int k1 = gcp_find(T, L, g[L], cx, cy, cz);
      if (k1 < 0) return -4;
      int k2 = (k1 + 1 < g[L].n) ? k1 + 1 : 0;
      const int maxiter2 = g[L].n;
      int iter2 = 0;
      found2 = 0;
      while (iter2 < maxiter2) {
        int t2_is_inter = 0;
        const unsigned c2 = gcp_get(g[L], k2);
        const double *q2 = gct_xyz(T, L, c2);
        if (GCN_INTER(c2)) {
          if (q2[0] == fx && q2[1] == fy && q2[2] == fz) { found1 = 1; break; }
          const unsigned c3 = gcp_get(g[L], (k2 + 1 < g[L].n) ? k2 + 1 : 0);
          found2 = 1;
          t2_is_inter = 1;
          if (GCN_INTER(c3) || GCN_INSIDE(c3) == 1) found2 = 0;
        }
        if (found2) { cx = q2[0]; cy = q2[1]; cz = q2[2]; ccode = c2; break; }
        else {
          if (gcpoly_add(T, poly, L, c2)) return -9;
          if (t2_is_inter) nintersect--;
        }
        k2 = (k2 + 1 < g[L].n) ? k2 + 1 : 0;
        iter2++;
      }
      if (found1) break;
      if (!found2) return -4;
      if (cx == fx && cy == fy && cz == fz) { found1 = 1; break; }
      if (gcpoly_add(T, poly, L, ccode)) return -9;
      nintersect--;
      L = 1 - L;
      iter1++;
    }
    if (!found1) return -5;
    if (nintersect > 0) return -6;
    n_out = poly.n;
    if (n_out < 3) n_out = 0;
    for (int k = 0; k < n_out; k++) {
      const double *q = gcpoly_xyz(T, (unsigned)(poly.bits >> (4 * k)) & 15u);
      out[k * 3] = q[0]; out[k * 3 + 1] = q[1]; out[k * 3 + 2] = q[2];
    }
  }
  if (n_out == 0) {
    for (int L = 0; L < 2 && n_out == 0; L++) {        // grid1 inside grid2 (:1839-1870), then grid2 inside grid1 (:1873-1904)
      const int npts = L ? npts2 : npts1;
      int nin = 0;
      for (int k = 0; k < g[L].n; k++) { const unsigned c = gcp_get(g[L], k); if (GCN_INTER(c) != 1 && GCN_INSIDE(c) == 1) nin++; }
      if (npts == nin) {
        n_out = npts;
        for (int k = 0; k < npts; k++) {
          const double *q = gct_xyz(T, L, gcp_get(g[L], k));
          out[k * 3] = q[0]; out[k * 3 + 1] = q[1]; out[k * 3 + 2] = q[2];
        }
      }
    }
  }
  return n_out;
}

// ------------------------------------------------------------------------------------------------ cell records
// FgCells reuse: verts[c*16 + 0..11] = corners xyz (clockwise), [12..14] = cap centre, [15] = cos(cap radius), or -2
// when the cell is too large for a useful cap.  lat/lon box = the cap's, so the candidate scan is a superset.
#define GC_CAP_MARGIN 2.e-6
// band_mode 1 (the destination launch of a culling search): the block also folds its cells' latitude ranges into band_keys
// {max key of lat_max, max of ~key of lat_min} -- one pair of atomics per block, and only when they would change the value.
// band_mode 2 (the source launch that follows): a cell whose cap latitude range cannot meet that band -- the candidate scan's
// strict latitude reject (d_box_pass) would drop every one of its pairs -- gets nv = 0, area 0 and nothing else: no record, no
// spherical-excess area (the software-x87 acosl is the expensive part of this kernel), no query, no candidates.
__device__ __forceinline__ bool d_gc_cell_record(const FgTileXyz *tiles, int ntiles, int s, FgCells c, const unsigned long long *cull,
                                                 double *lat_lo, double *lat_hi);
__global__ __launch_bounds__(256) void k_gc_cell_struct(const FgTileXyz *tiles, int ntiles, int ncells, FgCells c,
                                                         unsigned long long *band_keys, int band_mode)
{
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  double lo = 0, hi = 0;
  const bool live = s < ncells && d_gc_cell_record(tiles, ntiles, s, c, band_mode == 2 ? band_keys : nullptr, &lo, &hi);
  if (band_mode != 1) return;
  __shared__ unsigned long long sh_k[2][4];
  unsigned long long kmax = live ? d_ord_key(hi) : 0ull, kmin = live ? ~d_ord_key(lo) : 0ull;
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
// returns false for a culled cell; *lat_lo / *lat_hi = the latitude range of the cell's box
__device__ __forceinline__ bool d_gc_cell_record(const FgTileXyz *tiles, int ntiles, int s, FgCells c, const unsigned long long *cull,
                                                 double *lat_lo, double *lat_hi)
{
  int t = 0;
  while (t + 1 < ntiles && s >= tiles[t + 1].cell_off) t++;
  const FgTileXyz T = tiles[t];
  const int loc = s - T.cell_off, i = loc % T.nx, j = loc / T.nx, nxp = T.nx + 1;
  const int idx[4] = {j * nxp + i, (j + 1) * nxp + i, (j + 1) * nxp + i + 1, j * nxp + i + 1};   // clockwise, create_xgrid.c:1413-1420
  double v[12];
  for (int k = 0; k < 4; k++) { v[k * 3] = T.x[idx[k]]; v[k * 3 + 1] = T.y[idx[k]]; v[k * 3 + 2] = T.z[idx[k]]; }
  double *o = c.verts + (size_t)s * 16;
  for (int k = 0; k < 12; k++) o[k] = v[k];
  // bounding cap
  double cx = v[0] + v[3] + v[6] + v[9], cy = v[1] + v[4] + v[7] + v[10], cz = v[2] + v[5] + v[8] + v[11];
  const double nrm = sqrt(cx * cx + cy * cy + cz * cz);
  bool nocap = !(nrm > 1.e-3);
  double r = 0;
  if (!nocap) {
    cx /= nrm; cy /= nrm; cz /= nrm;
    for (int k = 0; k < 4; k++) {
      double d = cx * v[k * 3] + cy * v[k * 3 + 1] + cz * v[k * 3 + 2];
      d = fmin(1.0, fmax(-1.0, d));
      // acos loses accuracy near 1: use the chord for small separations
      const double ex = v[k * 3] - cx, ey = v[k * 3 + 1] - cy, ez = v[k * 3 + 2] - cz;
      const double chord = sqrt(ex * ex + ey * ey + ez * ez);
      const double ang = (chord < 0.5) ? 2.0 * asin(0.5 * chord) : acos(d);
      r = fmax(r, ang);
    }
    r = r * (1.0 + 1.e-9) + GC_CAP_MARGIN;
    if (r > 0.5) nocap = true;
  }
  const double hpi = 0.5 * GC_PI, tpi = 2.0 * GC_PI;
  const double latc = nocap ? 0.0 : asin(fmin(1.0, fmax(-1.0, cz)));
  const double lat_min = nocap ? -hpi : fmax(-hpi, latc - r), lat_max = nocap ? hpi : fmin(hpi, latc + r);
  *lat_lo = lat_min; *lat_hi = lat_max;
  c.lat_min[s] = lat_min; c.lat_max[s] = lat_max;
  if (cull && cull[0]) {
    const double bmax = d_ord_val(cull[0]), bmin = d_ord_val(~cull[1]);
    if (lat_max <= bmin || lat_min >= bmax) { c.nv[s] = 0; c.area[s] = 0; return false; }
  }
  {
    // cell area on the de-duplicated vertex list (addEnd merges the two pole corners of a lat-lon cap cell)
    double p[12]; int n = 0;
    for (int k = 0; k < 4; k++) {
      bool dup = false;
      for (int m = 0; m < n; m++) if (gc_same_point(p[m * 3], p[m * 3 + 1], p[m * 3 + 2], v[k * 3], v[k * 3 + 1], v[k * 3 + 2])) dup = true;
      if (!dup) { p[n * 3] = v[k * 3]; p[n * 3 + 1] = v[k * 3 + 1]; p[n * 3 + 2] = v[k * 3 + 2]; n++; }
    }
    c.area[s] = gc_area(n, p, 3);
    c.nv[s] = 4;
  }
  if (nocap) {
    o[12] = 0; o[13] = 0; o[14] = 1; o[15] = -2.0;
    c.lon_min[s] = 0.0; c.lon_max[s] = tpi; c.lon_avg[s] = GC_PI;
    return true;
  }
  o[12] = cx; o[13] = cy; o[14] = cz; o[15] = cos(r);
  double lonc = atan2(cy, cx);
  if (lonc < 0) lonc += tpi;
  // longitude: a great-circle arc that stays clear of the pole is monotone in longitude, so the polygon's extent is
  // that of its corners; the angular margin becomes margin / cos(lat) in longitude.  Cells whose cap reaches a pole
  // (or nearly) take the whole circle.
  const double cosmin = cos(fmin(hpi, fabs(latc) + r));
  if (fabs(latc) + r >= hpi - 1.e-6 || cosmin < 0.02) { c.lon_min[s] = 0.0; c.lon_max[s] = tpi; c.lon_avg[s] = GC_PI; }
  else {
    double dmin = 0.0, dmax = 0.0;
    for (int k = 0; k < 4; k++) {
      double dl = atan2(v[k * 3 + 1], v[k * 3]) - lonc;
      if (dl > GC_PI) dl -= tpi;
      if (dl < -GC_PI) dl += tpi;
      if (dl > GC_PI) dl -= tpi;
      dmin = fmin(dmin, dl); dmax = fmax(dmax, dl);
    }
    const double mlon = (GC_CAP_MARGIN + 1.e-9) / cosmin + 1.e-9;
    c.lon_min[s] = lonc + dmin - mlon; c.lon_max[s] = lonc + dmax + mlon; c.lon_avg[s] = lonc;
  }
  return true;
}

// ------------------------------------------------------------------------------------------------ pair kernels
#ifndef GC_WAVES_PER_EU
#define GC_WAVES_PER_EU 4
#endif
// the reference's bounding-box reject, convexity check and the cap reject; true = pair survives
__device__ bool gc_prefilter(const double *a, const double *b, double area1, double area2, unsigned *err)
{
  for (int ax = 0; ax < 3; ax++) {                               // create_xgrid.c:1508-1528
    double mn1 = a[ax], mx1 = a[ax], mn2 = b[ax], mx2 = b[ax];
    for (int k = 1; k < 4; k++) {
      mn1 = fmin(mn1, a[k * 3 + ax]); mx1 = fmax(mx1, a[k * 3 + ax]);
      mn2 = fmin(mn2, b[k * 3 + ax]); mx2 = fmax(mx2, b[k * 3 + ax]);
    }
    if (mn1 >= mx2 + GC_RANGE_CHECK || mn2 >= mx1 + GC_RANGE_CHECK) return false;
  }
  if (area1 <= 0) atomicOr(err, G_ERRBIT_GC_CONVEX1);            // :1575-1578 (fatal in the reference)
  if (area2 <= 0) atomicOr(err, G_ERRBIT_GC_CONVEX2);
  if (area1 <= 0 || area2 <= 0) return false;
  // caps further apart than the sum of their radii (+2e-6 rad each): no vertex inside, no edge crossing
  if (a[15] > -1.5 && b[15] > -1.5) {
    const double dotc = a[12] * b[12] + a[13] * b[13] + a[14] * b[14];
    const double sr1 = sqrt(fmax(0.0, 1.0 - a[15] * a[15])), sr2 = sqrt(fmax(0.0, 1.0 - b[15] * b[15]));
    const double cos_sum = a[15] * b[15] - sr1 * sr2;
    if (dotc < cos_sum - 1.e-12) return false;
  }
  return true;
}

__device__ void gc_finish(int p, int s, int n_out, const double *poly, double m, double area1, double area2, int *pair_dst,
                          double *tmp_area, int *nacc, unsigned long long *stats, unsigned *err)
{
  if (n_out < 0) { atomicOr(err, G_ERRBIT_GC_CLIP); atomicMax((int *)(err + 1), -n_out); pair_dst[p] = -1; return; }
  if (n_out == 0) { pair_dst[p] = -1; return; }
  const double xarea = gc_area(n_out, poly, 3) * m;
  const double min_area = (area1 < area2) ? area1 : area2;
  const double ratio = xarea / min_area;
  if (fabs(ratio - 1.e-6) < 1.e-12) atomicAdd(&stats[FG_STAT_BORDERLINE], 1ull);
  if (ratio > 1.e-6) { tmp_area[p] = xarea; atomicAdd(&nacc[s], 1); }
  else { pair_dst[p] = -1; atomicAdd(&stats[FG_STAT_BELOW], 1ull); }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GC_WAVES_PER_EU, GC_WAVES_PER_EU)))
void k_gc_clip(FgPairSpace ps, FgCells S, const double *mask, FgCells D, double *tmp_area, int *nacc,
               int *defer_list, int *defer_cnt, unsigned long long *stats, unsigned *err)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (!d_pair_live(ps, p)) return;
  int *pair_dst = ps.dst;
  const int s = ps.src[p], d = pair_dst[p];
  const double *a = S.verts + (size_t)s * 16, *b = D.verts + (size_t)d * 16;
  const double area1 = S.area[s], area2 = D.area[d];
  if (!gc_prefilter(a, b, area1, area2, err)) { pair_dst[p] = -1; return; }
  double poly[16 * 3];
  const int n_out = gc_clip_fast(a, b, poly);
  if (n_out == GC_FALLBACK) { defer_list[atomicAdd(defer_cnt, 1)] = p; return; }
  gc_finish(p, s, n_out, poly, mask ? mask[s] : 1.0, area1, area2, pair_dst, tmp_area, nacc, stats, err);
}

// pairs the compact version handed back: the array version (prefilter already passed)
__global__ __launch_bounds__(64) void k_gc_clip_slow(const int *defer_list, const int *defer_cnt, const int *pair_src, int *pair_dst,
                                                      FgCells S, const double *mask, FgCells D, double *tmp_area, int *nacc,
                                                      unsigned long long *stats, unsigned *err)
{
  const int nd = *defer_cnt;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nd; q += gridDim.x * blockDim.x) {
    const int p = defer_list[q];
    const int s = pair_src[p], d = pair_dst[p];
    const double *a = S.verts + (size_t)s * 16, *b = D.verts + (size_t)d * 16;
    GcPoly out;
    const int n_out = gc_clip(a, b, out);
    gc_finish(p, s, n_out, &out.p[0][0], mask ? mask[s] : 1.0, S.area[s], D.area[d], pair_dst, tmp_area, nacc, stats, err);
  }
}

// ------------------------------------------------------------------------------------------------ split clip
// k_gc_clip spends its time in three places (C384 -> 0.25 deg, 9.1 ms: 2.3 ms in the 32 angle sums of the inside tests, 3.8 ms
// in a per-lane loop over the extended-precision solves -- a wave runs max-over-lanes = 5.7 of them for 3.0 on average --
// and 2.3 ms in the walk and the area, all of it behind ~1 KB per lane of dynamically indexed tables in scratch).  The
// ordinary pair -- two plainly convex quads with four distinct corners each and no intersection on or next to a corner --
// goes through three passes instead:
//   k_gc_screen   lane per pair.  Eight edge normals and 32 point-against-plane dot products decide (a) the isInside flag of
//                 each corner whenever the corner is clear of every edge plane of the other cell by sin > 1e-6 (the angle sum
//                 of insidePolygon is then 2 pi to ~1e-10, or short of it by > 2e-6; the 1e-8 test cannot go the other way);
//                 a corner closer than that gets the reference's angle sum; (b) the edge pairs that certainly do not
//                 cross (the screen of gc_clip_fast on the same dot products).  One task per remaining edge pair.
//   k_gc_solve    lane per task: line_intersect_2D_3D on the software x87, nothing else; no divergence between pairs.
//   k_gc_walk     lane per pair: the reference's list insertions and walk as index logic on packed lists (no corner is ever
//                 rewritten, so "find by coordinate value" is "find by identity"), output polygon in LDS, area.
// Everything else -- a cell with coincident corners (pole cells), very large or very small cells, an intersection that snaps
// onto a corner (u = 0 or 1: the reference rewrites the corner and later solves see the new coordinates) or lies within 1e-6
// of one, more than 8 intersections, any of the walk's fatal checks -- is put on a list and redone from scratch by the
// one-kernel version above (k_gc_clip_list), which also reports the errors.
#define GC_SIN2_CLEAR 1.e-12
#define GC_WALK_PTS 8      // output polygon of k_gc_walk (convex x convex); a ninth point sends the pair to the list

__device__ __forceinline__ double gc_dot(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// insidePolygon (mosaic_util.c:1546-1589) of the corner pnt0 against the four corners q, both read from the cell records in
// memory; the arithmetic of gcf_inside.  -1: too close to the 1e-8 threshold for the fast acos (the one-kernel clip redoes the pair)
__device__ int gc_inside4(const double *pnt0g, const double *q)
{
  const double pnt0[3] = {pnt0g[0], pnt0g[1], pnt0g[2]};
  double anglesum = 0;
  for (int k = 0; k < 4; k++) {
    const int kn = (k + 1) & 3;
    const double q1[3] = {q[k * 3], q[k * 3 + 1], q[k * 3 + 2]}, q2[3] = {q[kn * 3], q[kn * 3 + 1], q[kn * 3 + 2]};
    if (gc_same_point(pnt0[0], pnt0[1], pnt0[2], q1[0], q1[1], q1[2])) return 1;
    anglesum += gc_spherical_angle<false>(pnt0, q2, q1);
  }
  const double dev = fabs(anglesum - 2 * GC_PI);
  if (fabs(dev - GC_EPSLN8) < 1.e-12) return -1;
  return dev < GC_EPSLN8;
}

// The four corners p against the four edge planes of q.  false: q is not a plainly convex quad.
//   in_bits / out_bits  per corner of p: inside every / outside some half-space of q by more than the clearance
//   miss                bit (edge of the source cell * 4 + edge of the destination cell): edge i of p meets the plane of edge e of
//                       q at a parameter outside [-1e-5, 1 + 1e-5] (t = n.l1 / (n.l1 - n.l2), n = q_e x q_e+1)
template <bool SWAP>
__device__ __forceinline__ bool gc_half_screen(const double *p, const double *q, const double *plen2, const double *qlen2,
                                               unsigned &in_bits, unsigned &out_bits, unsigned &miss)
{
  in_bits = 15u; out_bits = 0; miss = 0;
  double sigma = 1.0;
  bool convex = true;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    double n[3];
    gc_cross(q + e * 3, q + ((e + 1) & 3) * 3, n);
    const double nn = gc_dot(n, n);
    const double c2 = gc_dot(n, q + ((e + 2) & 3) * 3), c3 = gc_dot(n, q + ((e + 3) & 3) * 3);
    if (e == 0) sigma = (c2 > 0) ? 1.0 : -1.0;
    if (!(c2 * sigma > 1.e-14 && c3 * sigma > 1.e-14)) convex = false;
    double raw[4];
#pragma unroll
    for (int k = 0; k < 4; k++) raw[k] = gc_dot(n, p + k * 3);
    unsigned clr = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool clear = raw[k] * raw[k] > GC_SIN2_CLEAR * nn;
      const double sd = raw[k] * sigma;
      if (clear) clr |= 1u << k;
      if (!(clear && sd > 0)) in_bits &= ~(1u << k);
      if (clear && sd < 0) out_bits |= 1u << k;
    }
    // edge i of p against this plane: the parameter of line_intersect_2D_3D is t = r_i / (r_i - r_i+1), outside [0, 1] exactly
    // when both ends lie on one side.  With both ends clear of the plane (sin > 1e-6, far above the 1e-10 a later snap can move
    // an end and the ~1e-15 of the dot products) and |r| ratio >= 1e-5, t is outside [-1e-5, 1 + 1e-5]: certainly no
    // intersection, however parallel the two edges are (meridian edges of a cubed-sphere face against meridians: 4 of the 16
    // edge pairs of most pairs at C384 -> lat-lon, which the determinant-based screen of gc_screen_out let through).
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const double r1 = raw[i], r2 = raw[(i + 1) & 3], m1 = fabs(r1), m2 = fabs(r2);
      const bool same = (r1 > 0) == (r2 > 0);
      if (same && ((clr >> i) & 1u) && ((clr >> ((i + 1) & 3)) & 1u) && fmin(m1, m2) >= 1.e-5 * fmax(m1, m2))
        miss |= 1u << (SWAP ? e * 4 + i : i * 4 + e);
    }
  }
  return convex;
}

// four distinct corners, edges longer than 3e-5 rad; len2[e] = |v_e - v_e+1|^2
__device__ __forceinline__ bool gc_plain_quad(const double *v, double *len2)
{
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int kn = (k + 1) & 3;
    const double ex = v[k * 3] - v[kn * 3], ey = v[k * 3 + 1] - v[kn * 3 + 1], ez = v[k * 3 + 2] - v[kn * 3 + 2];
    len2[k] = ex * ex + ey * ey + ez * ez;
    if (!(len2[k] > 1.e-9)) ok = false;
  }
  if (gc_same_point(v[0], v[1], v[2], v[6], v[7], v[8]) || gc_same_point(v[3], v[4], v[5], v[9], v[10], v[11])) ok = false;   // the diagonals
  return ok;
}

__device__ __forceinline__ void d_gc_order_append(const FgPairSpace &ps, const GcSplit &g, unsigned reg, int lane, int p, bool valid, int cnt);
__global__ __launch_bounds__(64) void k_gc_screen(FgPairSpace ps, FgCells S, FgCells D, GcSplit g, unsigned *err)
{
  const unsigned first = blockIdx.x * 64u, reg = first / (unsigned)ps.regcap;
  if (first - reg * (unsigned)ps.regcap >= ps.fill[reg * FG_FILL_STRIDE]) return;          // block beyond its region's fill
  const int lane = threadIdx.x, p = (int)first + lane;
  const bool live = d_pair_live(ps, p);
  unsigned meta = 0;
  bool defer = false, valid = false;
  if (live) {
    const int s = ps.src[p], d = ps.dst[p];
    double a[16], b[16];
    {
      const double *ga = S.verts + (size_t)s * 16, *gb = D.verts + (size_t)d * 16;
#pragma unroll
      for (int k = 0; k < 16; k++) { a[k] = ga[k]; b[k] = gb[k]; }
    }
    if (!gc_prefilter(a, b, S.area[s], D.area[d], err)) ps.dst[p] = -1;
    else {
      double la[4], lb[4];
      bool plain = gc_plain_quad(a, la);
      plain = gc_plain_quad(b, lb) && plain;
      plain = plain && a[15] >= 0.97 && b[15] >= 0.97;                      // cap radii <= 0.245: every distance in the pair < 1 rad
      unsigned in_a, out_a, miss_a, in_b, out_b, miss_b;
      const bool cvx_b = gc_half_screen<false>(a, b, la, lb, in_a, out_a, miss_a);      // corners of a, edges of b
      const bool cvx_a = gc_half_screen<true>(b, a, lb, la, in_b, out_b, miss_b);
      if (!(plain && cvx_a && cvx_b)) defer = true;
      else {
        // corners near an edge plane: k_gc_walk takes their angle sums
        meta = (~(miss_a | miss_b) & 0xffffu) | (in_a << 16) | (in_b << 20) | ((~(in_a | out_a) & 15u) << 24) | ((~(in_b | out_b) & 15u) << 28);
        valid = true;
      }
    }
  }
  // tasks: one atomic per wave
  int cnt = __popc(meta & 0xffffu), incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
  const int total = __shfl(incl, 63);
  // (a single counter would serialise 100 000 same-address atomics, ~12 ns each: the task space is cut into regions like the pair list)
  const unsigned treg = (blockIdx.x * 2654435761u >> 12) % FG_NREG;
  unsigned wbase = 0;
  if (lane == 0 && total) wbase = atomicAdd(&g.ntask[treg * FG_FILL_STRIDE], (unsigned)total);
  wbase = __shfl(wbase, 0);
  const unsigned local = wbase + (unsigned)(incl - cnt), base = treg * g.tcap + local;
  if (live) {
    // every slot below min(ntask[r], tcap) of a region holds a well-formed task, also those of a pair that did not fit and goes to the list
    unsigned need = meta & 0xffffu, t = local;
    while (need) { const int bit = __ffs((int)need) - 1; need &= need - 1; if (t < g.tcap) g.task[treg * g.tcap + t] = ((unsigned)p << 4) | (unsigned)bit; t++; }
    if (cnt && local + (unsigned)cnt > g.tcap) { defer = true; valid = false; }
    if (defer) g.list[atomicAdd(g.list_cnt, 1)] = p;
    g.meta[p] = meta;
    g.tbase[p] = valid ? (int)base : -1;                   // -1: rejected, or on the list
  }
  d_gc_order_append(ps, g, reg, lane, p, live && valid, cnt);
}
// the second half of k_gc_screen (every lane of the wave arrives here): the pairs for k_gc_walk, in their region's list
__device__ __forceinline__ void d_gc_order_append(const FgPairSpace &ps, const GcSplit &g, unsigned reg, int lane, int p, bool valid, int cnt)
{
  const bool few = valid && cnt <= 2, many = valid && cnt > 2;
  const unsigned long long mf = __ballot(few), mm = __ballot(many);
  unsigned bf = 0, bm = 0;
  if (lane == 0) {
    if (mf) bf = atomicAdd(&g.ocnt[reg * FG_FILL_STRIDE], (unsigned)__popcll(mf));
    if (mm) bm = atomicAdd(&g.ocnt[reg * FG_FILL_STRIDE + 1], (unsigned)__popcll(mm));
  }
  bf = __shfl(bf, 0); bm = __shfl(bm, 0);
  const unsigned long long below = (1ull << lane) - 1ull;
  int *o = g.order + (size_t)reg * ps.regcap;
  if (few) o[bf + __popcll(mf & below)] = p;
  if (many) o[(unsigned)ps.regcap - 1u - (bm + __popcll(mm & below))] = p;
}

__global__ __launch_bounds__(256) void k_gc_solve(FgPairSpace ps, FgCells S, FgCells D, GcSplit g)
{
  const unsigned n = min(g.ntask[blockIdx.y * FG_FILL_STRIDE], g.tcap);      // blockIdx.y = region of the task space
  for (unsigned tl = blockIdx.x * 256u + threadIdx.x; tl < n; tl += gridDim.x * 256u) {
    const unsigned t = blockIdx.y * g.tcap + tl;
    const unsigned w = g.task[t];
    const int p = (int)(w >> 4), i1 = (int)(w >> 2) & 3, i2 = (int)w & 3;
    const double *A = S.verts + (size_t)ps.src[p] * 16, *B = D.verts + (size_t)ps.dst[p] * 16;
    double a1[3], a2[3], q1[3], q2[3], q3[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      a1[c] = A[i1 * 3 + c]; a2[c] = A[((i1 + 1) & 3) * 3 + c];
      q1[c] = B[i2 * 3 + c]; q2[c] = B[((i2 + 1) & 3) * 3 + c]; q3[c] = B[((i2 + 2) & 3) * 3 + c];
    }
    double I[3], u1, u2;
    int inbound;
    double2 r;
    if (!gc_line_intersect(a1, a2, q1, q2, q3, I, &u1, &u2, &inbound)) { r.x = -1.0; r.y = 0.0; }
    else if (u1 == 0 || u1 == 1 || u2 == 0 || u2 == 1) {                 // snapped onto a corner: the pair leaves the three passes
      r.x = 2.0; r.y = 0.0;
      if (atomicExch(&g.tbase[p], -1) >= 0) g.list[atomicAdd(g.list_cnt, 1)] = p;
    }
    else { r.x = u1; r.y = (inbound == 2) ? -u2 : u2; }
    ((double2 *)g.res)[t] = r;
  }
}

// a[k] for a run-time k without an indexed access: bit masks, not a ?: chain -- the optimizer turns the chain back into a table in
// scratch memory (k_gc_walk then moved 2.9 GB per launch through it, profiles/r03_summary.md, and waited on every dependent load)
__device__ __forceinline__ double gcw_sel8(const double *a, int k)
{
  long long v = 0;
#pragma unroll
  for (int m = 0; m < 8; m++) v |= __double_as_longlong(a[m]) & -(long long)(k == m);
  return __longlong_as_double(v);
}
// node lists of k_gc_walk: 4-bit refs in one 64-bit word, the intersect (0 / 1) and isInside flags as bit masks -- the 128-bit
// byte codes of GcPacked cost three times the instructions per access; gl_get returns the same code (ref | intersect << 4 | inside << 6)
struct GcL64 { unsigned long long refs; unsigned inter, inside; int n; };
__device__ __forceinline__ unsigned gl_get(const GcL64 &l, int k)
{
  return ((unsigned)(l.refs >> (4 * k)) & 15u) | (((l.inter >> k) & 1u) << 4) | (((l.inside >> k) & 1u) << 6);
}
__device__ __forceinline__ void gl_insert(GcL64 &l, int k, unsigned ref)       // an intersection node: intersect = 1, isInside = 1
{
  const unsigned long long low = l.refs & ((1ull << (4 * k)) - 1ull);
  l.refs = low | ((unsigned long long)ref << (4 * k)) | ((l.refs >> (4 * k)) << (4 * (k + 1)));
  const unsigned lm = (1u << k) - 1u;
  l.inter = (l.inter & lm) | (1u << k) | ((l.inter >> k) << (k + 1));
  l.inside = (l.inside & lm) | (1u << k) | ((l.inside >> k) << (k + 1));
  l.n++;
}
// insertIntersect (mosaic_util.c:1313-1397) for an intersection strictly inside the edge that starts at corner `vtx`;
// iu[] = the parameters of the intersections along this list's edges
__device__ __forceinline__ int gcw_insert(GcL64 &l, int vtx, double u_cur, int inbound, const double *iu, int iref)
{
  int k1 = -1;
  for (int k = 0; k < l.n; k++) { const unsigned c = gl_get(l, k); if (GCN_INTER(c) == 0 && (int)GCN_REF(c) == vtx) { k1 = k; break; } }
  if (k1 < 0) return -7;
  if (inbound == 1) {
    int k2 = (k1 + 1 < l.n) ? k1 + 1 : 0, guard = 0;
    while ((l.inter >> k2) & 1u) { k2 = (k2 + 1 < l.n) ? k2 + 1 : 0; if (++guard > 32) return -7; }
    l.inside &= ~(1u << k2);
  } else if (inbound == 2)
    l.inside &= ~(1u << k1);
  int k2 = k1 + 1;
  while (k2 < l.n) {
    const unsigned c = gl_get(l, k2);
    if (GCN_INTER(c) == 1) { if (gcw_sel8(iu, (int)GCN_REF(c) - 4) > u_cur) break; }
    else break;
    k2++;
  }
  if (l.n >= 12) return -9;
  gl_insert(l, k2, (unsigned)(4 + iref));
  return 0;
}

#define GCW_P(k, c) lds[((k) * 3 + (c)) * 64 + lane]
// addNode of the output polygon: coordinates of the node `code` of list L, dropped if within 1e-10 of a point already there
__device__ __forceinline__ int gcw_poly_add(double *lds, int lane, int &n, bool dedup, int L, unsigned code, const double *A, const double *B,
                                            const double *iu0, unsigned long long imeta)
{
  double q[3];
  const int r = (int)GCN_REF(code);
  if (r < 4) {
    const double *v = (L ? B : A) + r * 3;
    q[0] = v[0]; q[1] = v[1]; q[2] = v[2];
  } else {
    const int i1 = (int)(imeta >> (6 * (r - 4))) & 3;
    const double u = gcw_sel8(iu0, r - 4);
    const double *a1 = A + i1 * 3, *a2 = A + ((i1 + 1) & 3) * 3;
    q[0] = a1[0] + u * (a2[0] - a1[0]);                       // line_intersect_2D_3D, create_xgrid.c:2050-2056
    q[1] = a1[1] + u * (a2[1] - a1[1]);
    q[2] = a1[2] + u * (a2[2] - a1[2]);
    const double norm = gc_metric(q);
    q[0] /= norm; q[1] /= norm; q[2] /= norm;
  }
  if (dedup)
    for (int k = 0; k < n; k++) if (gc_same_point(GCW_P(k, 0), GCW_P(k, 1), GCW_P(k, 2), q[0], q[1], q[2])) return 0;
  if (n >= GC_WALK_PTS) return -9;
  GCW_P(n, 0) = q[0]; GCW_P(n, 1) = q[1]; GCW_P(n, 2) = q[2];
  n++;
  return 0;
}

// The area of the output polygon -- n_out spherical angles, each through the exact acosl (~1300 instructions) -- is NOT computed by
// the pair's own lane: a wave's 64 pairs hold 0 (rejected by the screen, empty, listed) to 8 vertices, and a per-lane loop ran
// max-over-lanes iterations with 25.7 of 64 lanes live (PMC, profiles/r03_summary.md).  The (pair, vertex) items of the wave are
// dealt round-robin to its lanes through LDS instead: every lane computes angles of whichever pairs have them, the owning lane
// then adds its polygon's angles up in vertex order (great_circle_area's order, mosaic_util.c:763-787: the same sum).
__global__ __launch_bounds__(64) void k_gc_walk(FgPairSpace ps, FgCells S, const double *mask, FgCells D, GcSplit g,
                                                double *tmp_area, int *nacc, unsigned long long *stats, unsigned *err)
{
  __shared__ double lds[GC_WALK_PTS * 3 * 64];
  __shared__ double sh_ang[GC_WALK_PTS * 64];
  __shared__ unsigned short sh_own[GC_WALK_PTS * 64];      // item -> lane | vertex << 6 | n_out << 9
  // slot q of the region's list (k_gc_screen): the pairs with few tasks from the front, the others from the back
  const unsigned first = blockIdx.x * 64u, reg = first / (unsigned)ps.regcap, qa0 = first - reg * (unsigned)ps.regcap;
  const unsigned cnt_f = g.ocnt[reg * FG_FILL_STRIDE], cnt_b = g.ocnt[reg * FG_FILL_STRIDE + 1];
  if (qa0 >= cnt_f && qa0 + 64u <= (unsigned)ps.regcap - cnt_b) return;                  // (block-uniform) no pair in these slots
  const int lane = threadIdx.x;
  const unsigned qa = qa0 + (unsigned)lane;
  int p = -1, d = -1, t0 = -1, s = 0;
  if (qa < cnt_f || qa >= (unsigned)ps.regcap - cnt_b) p = g.order[(size_t)reg * ps.regcap + qa];
  if (p >= 0) { d = ps.dst[p]; if (d >= 0) { t0 = g.tbase[p]; s = ps.src[p]; } }
  const bool live = d >= 0 && t0 >= 0;                     // else: rejected by the screen, or on the list
  int n_out = 0;
  if (live) {
    unsigned meta = g.meta[p];
    const double *A = S.verts + (size_t)s * 16, *B = D.verts + (size_t)d * 16;
    bool bad = false;
    for (unsigned edge = meta >> 24; edge; edge &= edge - 1) {             // corners near an edge plane of the other cell
      const int bit = __ffs((int)edge) - 1;
      const int in = (bit < 4) ? gc_inside4(A + bit * 3, B) : gc_inside4(B + (bit - 4) * 3, A);
      if (in < 0) bad = true;
      if (in > 0) meta |= 1u << (16 + bit);
    }

    GcL64 gl0, gl1;                                          // (two named values: a list indexed by L would live in scratch)
    gl0.refs = gl1.refs = 0x3210ull; gl0.inter = gl1.inter = 0; gl0.n = gl1.n = 4;
    gl0.inside = (meta >> 16) & 15u; gl1.inside = (meta >> 20) & 15u;
    double iu0[8], iu1[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { iu0[k] = -1.0; iu1[k] = -1.0; }
    unsigned long long imeta = 0;                            // per intersection: i1 | i2 << 2 | inbound << 4
    int nil = 0;
    {
      unsigned need = meta & 0xffffu;
      // the first four results (a pair has 3.0 on average) in one round trip; the rest one by one
      const int ntask = __popc(need);
      double2 rpre[4];
#pragma unroll
      for (int i = 0; i < 4; i++) rpre[i] = ((const double2 *)g.res)[t0 + min(i, max(ntask - 1, 0))];
      int ti = 0;
      while (need && !bad) {
        const int bit = __ffs((int)need) - 1;
        need &= need - 1;
        double2 r;
        {
          long long rx = 0, ry = 0;
#pragma unroll
          for (int i = 0; i < 4; i++) { const long long mk = -(long long)(ti == i); rx |= __double_as_longlong(rpre[i].x) & mk; ry |= __double_as_longlong(rpre[i].y) & mk; }
          r.x = __longlong_as_double(rx); r.y = __longlong_as_double(ry);
        }
        if (ti >= 4) r = ((const double2 *)g.res)[t0 + ti];
        ti++;
        if (r.x < 0) continue;
        const double u1 = r.x, u2 = fabs(r.y);
        if (u1 > 1.5 || u1 < 1.e-6 || u1 > 1.0 - 1.e-6 || u2 < 1.e-6 || u2 > 1.0 - 1.e-6) { bad = true; break; }
        const int inbound = (r.y < 0) ? 2 : 1, i1 = bit >> 2, i2 = bit & 3;
        bool dup = false;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int m = (int)(imeta >> (6 * k));
          if (k < nil && ((iu0[k] == u1 && (m & 3) == i1) || (iu1[k] == u2 && ((m >> 2) & 3) == i2))) dup = true;
        }
        if (dup) continue;
        if (nil >= 8) { bad = true; break; }
        const int iref = nil++;
#pragma unroll
        for (int k = 0; k < 8; k++) { if (k == iref) { iu0[k] = u1; iu1[k] = u2; } }
        imeta |= (unsigned long long)(i1 | (i2 << 2) | (inbound << 4)) << (6 * iref);
        if (gcw_insert(gl0, i1, u1, inbound, iu0, iref) || gcw_insert(gl1, i2, u2, 0, iu1, iref)) bad = true;
      }
    }

    if (!bad) {
      int nintersect = nil, firstx = -1;
      if (nintersect > 1) for (int k = 0; k < nil; k++) if (((imeta >> (6 * k + 4)) & 3u) == 2u) { firstx = k; break; }
      if (firstx >= 0) {
        const int maxiter1 = nintersect;
        int np = 0;
        gcw_poly_add(lds, lane, np, false, 0, (unsigned)(4 + firstx) | (1u << 4), A, B, iu0, imeta);
        nintersect--;
        int L = 0, iter1 = 0, found1 = 0, found2 = 0, cur = firstx;
        unsigned ccode = 0;
        while (iter1 < maxiter1 && !bad) {
          const GcL64 lw = L ? gl1 : gl0;                 // the walk only reads the lists
          int k1 = -1;
          for (int k = 0; k < lw.n; k++) { const unsigned c = gl_get(lw, k); if (GCN_INTER(c) == 1 && (int)GCN_REF(c) - 4 == cur) { k1 = k; break; } }
          if (k1 < 0) { bad = true; break; }
          int k2 = (k1 + 1 < lw.n) ? k1 + 1 : 0;
          const int maxiter2 = lw.n;
          int iter2 = 0;
          found2 = 0;
          while (iter2 < maxiter2) {
            int t2_is_inter = 0;
            const unsigned c2 = gl_get(lw, k2);
            if (GCN_INTER(c2)) {
              if ((int)GCN_REF(c2) - 4 == firstx) { found1 = 1; break; }
              const unsigned c3 = gl_get(lw, (k2 + 1 < lw.n) ? k2 + 1 : 0);
              found2 = 1;
              t2_is_inter = 1;
              if (GCN_INTER(c3) || GCN_INSIDE(c3) == 1) found2 = 0;
            }
            if (found2) { cur = (int)GCN_REF(c2) - 4; ccode = c2; break; }
            if (gcw_poly_add(lds, lane, np, true, L, c2, A, B, iu0, imeta)) { bad = true; break; }
            if (t2_is_inter) nintersect--;
            k2 = (k2 + 1 < lw.n) ? k2 + 1 : 0;
            iter2++;
          }
          if (bad || found1) break;
          if (!found2) { bad = true; break; }
          if (gcw_poly_add(lds, lane, np, true, L, ccode, A, B, iu0, imeta)) { bad = true; break; }
          nintersect--;
          L = 1 - L;
          iter1++;
        }
        if (!found1 || nintersect > 0) bad = true;
        n_out = (np < 3) ? 0 : np;
      }
      if (!bad && n_out == 0) {
#pragma unroll
        for (int L = 0; L < 2; L++) {                        // grid1 inside grid2 (:1839-1870), then grid2 inside grid1 (:1873-1904)
          const GcL64 lw = L ? gl1 : gl0;
          int nin = 0;
          for (int k = 0; k < lw.n; k++) { const unsigned c = gl_get(lw, k); if (GCN_INTER(c) != 1 && GCN_INSIDE(c) == 1) nin++; }
          if (n_out == 0 && nin == 4) {
            int np = 0;
            for (int k = 0; k < 4; k++) gcw_poly_add(lds, lane, np, false, L, gl_get(lw, k), A, B, iu0, imeta);
            n_out = 4;
          }
        }
      }
    }
    if (bad) { g.list[g.list_cap - 1 - atomicAdd(g.list2_cnt, 1)] = p; n_out = 0; }
    else if (n_out == 0) ps.dst[p] = -1;
  }
  // --- the wave's (pair, vertex) items, dealt to its lanes
  const unsigned incl = (unsigned)n_out + 0u;
  unsigned pre = incl;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(pre, o, 64); if (lane >= o) pre += v; }
  const int T = (int)__shfl(pre, 63), off = (int)pre - n_out;
  for (int i = 0; i < n_out; i++) sh_own[off + i] = (unsigned short)(lane | (i << 6) | (n_out << 9));
  __syncthreads();
  for (int t = lane; t < T; t += 64) {
    const unsigned w = sh_own[t];
    const int l = (int)(w & 63u), i = (int)(w >> 6) & 7, n = (int)(w >> 9);
    const int i1 = (i + 1 < n) ? i + 1 : 0, i2 = (i1 + 1 < n) ? i1 + 1 : 0;
#define GCW_PL(k, c) lds[((k) * 3 + (c)) * 64 + l]
    const double p0[3] = {GCW_PL(i, 0), GCW_PL(i, 1), GCW_PL(i, 2)}, p1[3] = {GCW_PL(i1, 0), GCW_PL(i1, 1), GCW_PL(i1, 2)},
                 p2[3] = {GCW_PL(i2, 0), GCW_PL(i2, 1), GCW_PL(i2, 2)};
#undef GCW_PL
#if defined(FG_EXP) && FG_EXP == 3
    sh_ang[t] = p0[0] + p1[1] + p2[2];               // (timing experiment: no angles)
#else
    sh_ang[t] = gc_spherical_angle<true>(p1, p2, p0);
#endif
  }
  __syncthreads();
  if (n_out == 0) return;
  double sum = 0.0;                                        // great_circle_area, mosaic_util.c:763-787
  for (int i = 0; i < n_out; i++) sum += sh_ang[off + i];
  const double area1 = S.area[s], area2 = D.area[d];
  const double xarea = (sum - (n_out - 2.) * GC_PI) * GC_RADIUS * GC_RADIUS * (mask ? mask[s] : 1.0);
  const double min_area = (area1 < area2) ? area1 : area2;
  const double ratio = xarea / min_area;
  if (fabs(ratio - 1.e-6) < 1.e-12) atomicAdd(&stats[FG_STAT_BORDERLINE], 1ull);
  if (ratio > 1.e-6) { tmp_area[p] = xarea; atomicAdd(&nacc[s], 1); }
  else { ps.dst[p] = -1; atomicAdd(&stats[FG_STAT_BELOW], 1ull); }
}

// the listed pairs: the one-kernel version (prefilter passed in k_gc_screen), entries list[q * stride]
__global__ __launch_bounds__(64) void k_gc_clip_list(const int *list, int stride, const int *list_cnt, FgPairSpace ps, FgCells S, const double *mask,
                                                     FgCells D, double *tmp_area, int *nacc, unsigned long long *stats, unsigned *err)
{
  const int nl = *list_cnt;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nl; q += gridDim.x * blockDim.x) {
    const int p = list[(long)q * stride];
    const int s = ps.src[p], d = ps.dst[p];
    const double *a = S.verts + (size_t)s * 16, *b = D.verts + (size_t)d * 16;
    GcPoly out;
    int n_out = gc_clip_fast(a, b, &out.p[0][0]);
    if (n_out == GC_FALLBACK) n_out = gc_clip(a, b, out);
    gc_finish(p, s, n_out, &out.p[0][0], mask ? mask[s] : 1.0, S.area[s], D.area[d], ps.dst, tmp_area, nacc, stats, err);
  }
}

static inline int gc_nblk(long n, int t) { return (int)((n + t - 1) / t); }

void fgd_gc_cell_struct(const FgTileXyz *tiles_dev, int ntiles, int ncells, FgCells c, hipStream_t st, unsigned long long *band_keys, int band_mode)
{
  if (ncells > 0) k_gc_cell_struct<<<gc_nblk(ncells, 256), 256, 0, st>>>(tiles_dev, ntiles, ncells, c, band_keys, band_keys ? band_mode : 0);
}

void fgd_gc_clip(FgPairSpace ps, FgCells S, const double *mask, FgCells D,
                 double *tmp_area, int *nacc, int *defer_list, int *defer_cnt, unsigned long long *stats, unsigned *err, hipStream_t st)
{
  const long np = fgd_pairs_total(ps);
  if (np <= 0) return;
  k_gc_clip<<<gc_nblk(np, 64), 64, 0, st>>>(ps, S, mask, D, tmp_area, nacc, defer_list, defer_cnt, stats, err);
  k_gc_clip_slow<<<64, 64, 0, st>>>(defer_list, defer_cnt, ps.src, ps.dst, S, mask, D, tmp_area, nacc, stats, err);
}

// the three-pass version; g.list doubles as the defer list of the one-kernel version when the split cannot be used.
// st2 / e1 / e2 (may be null): the listed pairs -- a few latency-bound waves, 0.55 ms at C384 -> 0.25 deg -- run on st2 beside
// k_gc_walk; st has waited for them when this returns.
void fgd_gc_clip_split(FgPairSpace ps, FgCells S, const double *mask, FgCells D, double *tmp_area, int *nacc, GcSplit g,
                       unsigned long long *stats, unsigned *err, hipStream_t st, hipStream_t st2, hipEvent_t e1, hipEvent_t e2)
{
  const long np = fgd_pairs_total(ps);
  if (np <= 0) return;
  if (np >= (1L << 28) || !g.task) {                      // a task word holds a 28-bit pair index
    fgd_gc_clip(ps, S, mask, D, tmp_area, nacc, g.list, g.list_cnt, stats, err, st);
    return;
  }
  const bool two = st2 && e1 && e2;
  k_gc_screen<<<gc_nblk(np, 64), 64, 0, st>>>(ps, S, D, g, err);
  k_gc_solve<<<dim3((unsigned)std::min<long>(gc_nblk(g.tcap, 256), 256), FG_NREG), 256, 0, st>>>(ps, S, D, g);
  hipStream_t sl = st;
  if (two) { (void)hipEventRecord(e1, st); (void)hipStreamWaitEvent(st2, e1, 0); sl = st2; }
  k_gc_clip_list<<<2048, 64, 0, sl>>>(g.list, 1, g.list_cnt, ps, S, mask, D, tmp_area, nacc, stats, err);
  if (two) (void)hipEventRecord(e2, st2);
  k_gc_walk<<<gc_nblk(np, 64), 64, 0, st>>>(ps, S, mask, D, g, tmp_area, nacc, stats, err);
  k_gc_clip_list<<<256, 64, 0, st>>>(g.list + g.list_cap - 1, -1, g.list2_cnt, ps, S, mask, D, tmp_area, nacc, stats, err);
  if (two) (void)hipStreamWaitEvent(st, e2, 0);
}

// ------------------------------------------------------------------------------------------------ batch primitives
// clip_2dx2d_great_circle / great_circle_area on arrays of polygons (B1 mirrors and tests): one lane per polygon pair
__global__ __launch_bounds__(64) void k_gc_clip_batch(int n, const double *a, const double *b, double *out, int *n_out, double *area)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const double *pa = a + (size_t)p * 12, *pb = b + (size_t)p * 12;
  bool far = false;
  for (int ax = 0; ax < 3 && !far; ax++) {
    double mn1 = pa[ax], mx1 = pa[ax], mn2 = pb[ax], mx2 = pb[ax];
    for (int k = 1; k < 4; k++) {
      mn1 = fmin(mn1, pa[k * 3 + ax]); mx1 = fmax(mx1, pa[k * 3 + ax]);
      mn2 = fmin(mn2, pb[k * 3 + ax]); mx2 = fmax(mx2, pb[k * 3 + ax]);
    }
    if (mn1 >= mx2 + GC_RANGE_CHECK || mn2 >= mx1 + GC_RANGE_CHECK) far = true;
  }
  GcPoly o; o.n = 0;
  int no = 0;
  if (!far) {
    // gridArea of both lists (create_xgrid.c:1575-1578), on the de-duplicated corners
    for (int which = 0; which < 2 && no == 0; which++) {
      const double *v = which ? pb : pa;
      double q[12]; int n = 0;
      for (int k = 0; k < 4; k++) {
        bool dup = false;
        for (int m = 0; m < n; m++) if (gc_same_point(q[m * 3], q[m * 3 + 1], q[m * 3 + 2], v[k * 3], v[k * 3 + 1], v[k * 3 + 2])) dup = true;
        if (!dup) { q[n * 3] = v[k * 3]; q[n * 3 + 1] = v[k * 3 + 1]; q[n * 3 + 2] = v[k * 3 + 2]; n++; }
      }
      if (gc_area(n, q, 3) <= 0) no = which ? -2 : -1;
    }
    if (no == 0) {
      double q[16 * 3];
      no = gc_clip_fast(pa, pb, q);
      if (no == GC_FALLBACK) no = gc_clip(pa, pb, o);
      else for (int k = 0; k < no; k++) { o.p[k][0] = q[k * 3]; o.p[k][1] = q[k * 3 + 1]; o.p[k][2] = q[k * 3 + 2]; }
    }
  }
  n_out[p] = no;
  for (int k = 0; k < GC_PCAP; k++) for (int ax = 0; ax < 3; ax++) out[((size_t)p * GC_PCAP + k) * 3 + ax] = (k < no) ? o.p[k][ax] : 0.0;
  area[p] = (no > 0) ? gc_area(no, &o.p[0][0], 3) : 0.0;
}

void fgd_gc_clip_batch(int n, const double *a, const double *b, double *out, int *n_out, double *area, hipStream_t st)
{
  if (n > 0) k_gc_clip_batch<<<gc_nblk(n, 64), 64, 0, st>>>(n, a, b, out, n_out, area);
}

// great_circle_area of polygons given as [npoly][stride_pts][3] with n[npoly] vertices each
__global__ __launch_bounds__(64) void k_gc_area_batch(int npoly, int stride_pts, const double *xyz, const int *n, double *area)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npoly) return;
  area[p] = gc_area(n[p], xyz + (size_t)p * stride_pts * 3, 3);
}
void fgd_gc_area_batch(int npoly, int stride_pts, const double *xyz, const int *n, double *area, hipStream_t st)
{
  if (npoly > 0) k_gc_area_batch<<<gc_nblk(npoly, 64), 64, 0, st>>>(npoly, stride_pts, xyz, n, area);
}
