/* TEST INFRASTRUCTURE ONLY -- CPU oracle for the great-circle exchange-grid path.
 *
 * Restates, with small ordered arrays instead of the reference's linked Node pool,
 *   create_xgrid_great_circle      tools/libfrencutils/create_xgrid.c:1366-1466
 *   clip_2dx2d_great_circle        create_xgrid.c:1479-1908
 *   line_intersect_2D_3D           create_xgrid.c:1919-2081
 *   get_grid_great_circle_area     create_xgrid.c:98-137
 *   great_circle_area, spherical_angle, intersect_tri_with_line, invert_matrix_3x3, mult
 *                                  tools/libfrencutils/mosaic_util.c:763-838, :967-1044
 *   addEnd, addIntersect, insertIntersect, setInbound, getFirstInbound, insidePolygon, gridArea, samePoint
 *                                  mosaic_util.c:1088-1589
 * The `long double` parts are native x87 here, as in the reference on x86-64 (mosaic_util.c is compiled without
 * HAVE_LONG_DOUBLE_WIDER: the sources never include config.h, so spherical_angle works in double and calls acosl).
 * PINNED: tests/test_oracle_vs_ref.py compares every function below bit for bit with oracle/_ref (the reference's
 * own create_xgrid.c + mosaic_util.c compiled in place) on random quads, pole cells and C48 -> lat-lon tiles.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 *
 * Fatal errors of the reference (error_handler -> exit) are returned as negative n_out:
 *  -1/-2 grid box 1/2 not convex, -3 firstIntersect not in grid1List, -4 next intersection not found,
 *  -5 did not return to the first intersection, -6 nintersect > 0 after clipping, -7 insertIntersect anchor
 *  missing, -8 setInbound anchor missing, -9 list capacity.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define GC_EPSLN8 (1.e-8)
#define GC_EPSLN10 (1.e-10)
#define GC_EPSLN15 (1.e-15)
#define GC_EPSLN30 (1.e-30)
#define GC_RANGE_CHECK 0.05            /* mosaic_util.h:26 */
#define GC_RADIUS 6371000.0
#define GC_AREA_RATIO_THRESH 1.e-6     /* create_xgrid.c:26 */
#define GC_MASK_THRESH 0.5
#define GC_CAP 40

typedef struct { double x, y, z, u, u_clip; int intersect, inbound, inside, subj_index, clip_index; } GcNode;
typedef struct { int n; GcNode v[GC_CAP]; } GcList;

static int same_point(double x1, double y1, double z1, double x2, double y2, double z2)
{                                                           /* mosaic_util.c:1206-1212 */
  if (fabs(x1 - x2) > GC_EPSLN10 || fabs(y1 - y2) > GC_EPSLN10 || fabs(z1 - z2) > GC_EPSLN10) return 0;
  return 1;
}
static int same_node(const GcNode *a, const GcNode *b) { return a->x == b->x && a->y == b->y && a->z == b->z; }

/* addEnd, mosaic_util.c:1096-1135: append unless a point within 1e-10 is already there */
static int add_end(GcList *l, double x, double y, double z, int intersect, double u, int inbound, int inside)
{
  for (int k = 0; k < l->n; k++) if (same_point(l->v[k].x, l->v[k].y, l->v[k].z, x, y, z)) return 0;
  if (l->n >= GC_CAP) return -9;
  GcNode *t = &l->v[l->n++];
  memset(t, 0, sizeof *t);
  t->x = x; t->y = y; t->z = z; t->u = u; t->intersect = intersect; t->inbound = inbound; t->inside = inside;
  return 0;
}

double orc_spherical_angle(const double *v1, const double *v2, const double *v3)
{                                                           /* mosaic_util.c:799-836, double branch */
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
      if (ddd < 0.) angle = M_PI;
      else angle = 0.;
    } else
      angle = acosl(ddd);
  }
  return angle;
}

double orc_great_circle_area(int n, const double *x, const double *y, const double *z)
{                                                           /* mosaic_util.c:763-787 */
  double pnt0[3], pnt1[3], pnt2[3], sum = 0.0;
  for (int i = 0; i < n; i++) {
    pnt0[0] = x[i]; pnt0[1] = y[i]; pnt0[2] = z[i];
    pnt1[0] = x[(i + 1) % n]; pnt1[1] = y[(i + 1) % n]; pnt1[2] = z[(i + 1) % n];
    pnt2[0] = x[(i + 2) % n]; pnt2[1] = y[(i + 2) % n]; pnt2[2] = z[(i + 2) % n];
    sum += orc_spherical_angle(pnt1, pnt2, pnt0);
  }
  return (sum - (n - 2.) * M_PI) * GC_RADIUS * GC_RADIUS;
}

static double list_area(const GcList *l)                    /* gridArea, mosaic_util.c:1399-1419 */
{
  double x[GC_CAP], y[GC_CAP], z[GC_CAP];
  for (int k = 0; k < l->n; k++) { x[k] = l->v[k].x; y[k] = l->v[k].y; z[k] = l->v[k].z; }
  return orc_great_circle_area(l->n, x, y, z);
}

static int inside_polygon(const GcNode *node, const GcList *l)   /* mosaic_util.c:1546-1589 */
{
  double pnt0[3] = {node->x, node->y, node->z}, pnt1[3], pnt2[3], anglesum = 0;
  for (int k = 0; k < l->n; k++) {
    int kn = (k + 1 < l->n) ? k + 1 : 0;
    pnt1[0] = l->v[k].x; pnt1[1] = l->v[k].y; pnt1[2] = l->v[k].z;
    pnt2[0] = l->v[kn].x; pnt2[1] = l->v[kn].y; pnt2[2] = l->v[kn].z;
    if (same_point(pnt0[0], pnt0[1], pnt0[2], pnt1[0], pnt1[1], pnt1[2])) return 1;
    anglesum += orc_spherical_angle(pnt0, pnt2, pnt1);
  }
  return fabs(anglesum - 2 * M_PI) < GC_EPSLN8;
}

/* intersect_tri_with_line + invert_matrix_3x3 + mult (mosaic_util.c:967-1044): only t = X[0] is used by the caller */
static int tri_line_t(const double *pnt0, const double *pnt1, const double *l1, const double *l2, double *t)
{
  long double m[9], inv0, inv1, inv2, V[3];
  const double pnt2[3] = {0.0, 0.0, 0.0};
  m[0] = l1[0] - l2[0]; m[1] = pnt1[0] - pnt0[0]; m[2] = pnt2[0] - pnt0[0];
  m[3] = l1[1] - l2[1]; m[4] = pnt1[1] - pnt0[1]; m[5] = pnt2[1] - pnt0[1];
  m[6] = l1[2] - l2[2]; m[7] = pnt1[2] - pnt0[2]; m[8] = pnt2[2] - pnt0[2];
  const long double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
  if (fabsl(det) < GC_EPSLN15) return 0;
  const long double deti = 1.0 / det;
  inv0 = (m[4] * m[8] - m[5] * m[7]) * deti;
  inv1 = (m[2] * m[7] - m[1] * m[8]) * deti;
  inv2 = (m[1] * m[5] - m[2] * m[4]) * deti;
  V[0] = l1[0] - pnt0[0]; V[1] = l1[1] - pnt0[1]; V[2] = l1[2] - pnt0[2];
  *t = inv0 * V[0] + inv1 * V[1] + inv2 * V[2];
  return 1;
}

static void cross3(const double *p1, const double *p2, double *e)
{
  e[0] = p1[1] * p2[2] - p1[2] * p2[1];
  e[1] = p1[2] * p2[0] - p1[0] * p2[2];
  e[2] = p1[0] * p2[1] - p1[1] * p2[0];
}
static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double metric3(const double *p) { return sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]); }

int orc_line_intersect_2D_3D(const double *a1, const double *a2, const double *q1, const double *q2, const double *q3,
                             double *intersect, double *u_a, double *u_q, int *inbound)
{                                                           /* create_xgrid.c:1919-2081 */
  double u, p1[3], v1[3], v2[3], c1[3], c2[3], c3[3], coincident, sense, norm;
  *inbound = 0;
  if (same_point(a1[0], a1[1], a1[2], q1[0], q1[1], q1[2])) { *u_a = 0; *u_q = 0; memcpy(intersect, a1, 24); return 1; }
  else if (same_point(a1[0], a1[1], a1[2], q2[0], q2[1], q2[2])) { *u_a = 0; *u_q = 1; memcpy(intersect, a1, 24); return 1; }
  else if (same_point(a2[0], a2[1], a2[2], q1[0], q1[1], q1[2])) { *u_a = 1; *u_q = 0; memcpy(intersect, a2, 24); return 1; }
  else if (same_point(a2[0], a2[1], a2[2], q2[0], q2[1], q2[2])) { *u_a = 1; *u_q = 1; memcpy(intersect, a2, 24); return 1; }
  if (!tri_line_t(q1, q2, a1, a2, u_a)) return 0;
  if (fabs(*u_a) < GC_EPSLN8) *u_a = 0;
  if (fabs(*u_a - 1) < GC_EPSLN8) *u_a = 1;
  if ((*u_a < 0) || (*u_a > 1)) return 0;
  if (!tri_line_t(a1, a2, q1, q2, u_q)) return 0;
  if (fabs(*u_q) < GC_EPSLN8) *u_q = 0;
  if (fabs(*u_q - 1) < GC_EPSLN8) *u_q = 1;
  if ((*u_q < 0) || (*u_q > 1)) return 0;
  u = *u_a;
  cross3(a1, a2, c1);
  cross3(q1, q2, c2);
  cross3(c1, c2, c3);
  coincident = metric3(c3);
  if (fabs(coincident) < GC_EPSLN30) return 0;
  intersect[0] = a1[0] + u * (a2[0] - a1[0]);
  intersect[1] = a1[1] + u * (a2[1] - a1[1]);
  intersect[2] = a1[2] + u * (a2[2] - a1[2]);
  norm = metric3(intersect);
  for (int i = 0; i < 3; i++) intersect[i] /= norm;
  if (*u_q != 0 && *u_q != 1) {
    p1[0] = a2[0] - a1[0]; p1[1] = a2[1] - a1[1]; p1[2] = a2[2] - a1[2];
    v1[0] = q2[0] - q1[0]; v1[1] = q2[1] - q1[1]; v1[2] = q2[2] - q1[2];
    v2[0] = q3[0] - q2[0]; v2[1] = q3[1] - q2[1]; v2[2] = q3[2] - q2[2];
    cross3(v1, v2, c1);
    cross3(v1, p1, c2);
    sense = dot3(c1, c2);
    *inbound = 1;
    if (sense > 0) *inbound = 2;
  }
  return 1;
}

/* addIntersect, mosaic_util.c:1139-1190 */
static int add_intersect(GcList *l, double x, double y, double z, double u1, double u2, int inbound, int is1, int ie1, int is2, int ie2)
{
  double u1_cur = u1, u2_cur = u2;
  int i1_cur = is1, i2_cur = is2;
  if (u1_cur == 1) { u1_cur = 0; i1_cur = ie1; }
  if (u2_cur == 1) { u2_cur = 0; i2_cur = ie2; }
  for (int k = 0; k < l->n; k++) {
    if (l->v[k].u == u1_cur && l->v[k].subj_index == i1_cur) return 0;
    if (l->v[k].u_clip == u2_cur && l->v[k].clip_index == i2_cur) return 0;
  }
  if (l->n >= GC_CAP) return -9;
  GcNode *t = &l->v[l->n++];
  memset(t, 0, sizeof *t);
  t->x = x; t->y = y; t->z = z; t->intersect = 1; t->inbound = inbound; t->inside = 0;
  t->u = u1_cur; t->subj_index = i1_cur; t->u_clip = u2_cur; t->clip_index = i2_cur;
  return 1;
}

/* insertIntersect, mosaic_util.c:1313-1397 */
static int insert_intersect(GcList *l, double x, double y, double z, double u1, double u2, int inbound, double x2, double y2, double z2)
{
  int k1 = -1;
  for (int k = 0; k < l->n; k++) if (l->v[k].x == x2 && l->v[k].y == y2 && l->v[k].z == z2) { k1 = k; break; }
  if (k1 < 0) return -7;
  double u_cur = u1;
  if (u1 == 1) { u_cur = 0; k1 = (k1 + 1 < l->n) ? k1 + 1 : 0; }
  if (u_cur == 0) {
    GcNode *t = &l->v[k1];
    t->intersect = 2; t->inside = 1; t->u = u_cur; t->x = x; t->y = y; t->z = z;
    return 0;
  }
  if (u2 != 0 && u2 != 1) {
    if (inbound == 1) {                       /* goes outside: the next non-intersection vertex is outside */
      int k2 = (k1 + 1 < l->n) ? k1 + 1 : 0, guard = 0;
      while (l->v[k2].intersect) { k2 = (k2 + 1 < l->n) ? k2 + 1 : 0; if (++guard > 2 * GC_CAP) return -7; }
      l->v[k2].inside = 0;
    } else if (inbound == 2)
      l->v[k1].inside = 0;
  }
  int k2 = k1 + 1;                            /* no wrap: Next == NULL ends the scan */
  while (k2 < l->n) {
    if (l->v[k2].intersect == 1) { if (l->v[k2].u > u_cur) break; }
    else break;
    k1 = k2; k2++;
  }
  if (l->n >= GC_CAP) return -9;
  for (int k = l->n; k > k2; k--) l->v[k] = l->v[k - 1];
  l->n++;
  GcNode *t = &l->v[k2];
  memset(t, 0, sizeof *t);
  t->x = x; t->y = y; t->z = z; t->u = u_cur; t->intersect = 1; t->inbound = inbound; t->inside = 1;
  return 0;
}

/* setInbound, mosaic_util.c:1497-1535 */
static int set_inbound(GcList *inter, const GcList *l)
{
  for (int k = 0; k < inter->n; k++) {
    if (inter->v[k].inbound) continue;
    int f = -1;
    for (int j = 0; j < l->n; j++) if (same_node(&l->v[j], &inter->v[k])) { f = j; break; }
    if (f < 0) return -8;
    const GcNode *prev = &l->v[f > 0 ? f - 1 : l->n - 1], *next = &l->v[f + 1 < l->n ? f + 1 : 0];
    if (prev->inside == 0 && next->inside == 1) inter->v[k].inbound = 2;
    else inter->v[k].inbound = 1;
  }
  return 0;
}

static int first_inbound(const GcList *l, GcNode *out)      /* getFirstInbound, mosaic_util.c:1448-1463 */
{
  for (int k = 0; k < l->n; k++) if (l->v[k].inbound == 2) { *out = l->v[k]; return 1; }
  return 0;
}
static int find_node(const GcList *l, const GcNode *nd)     /* getNode, mosaic_util.c:1233-1250 */
{
  for (int k = 0; k < l->n; k++) if (same_node(&l->v[k], nd)) return k;
  return -1;
}
static int add_node(GcList *l, const GcNode *nd) { return add_end(l, nd->x, nd->y, nd->z, nd->intersect, nd->u, nd->inbound, nd->inside); }

static double minv(int n, const double *d) { double m = d[0]; for (int k = 1; k < n; k++) if (d[k] < m) m = d[k]; return m; }
static double maxv(int n, const double *d) { double m = d[0]; for (int k = 1; k < n; k++) if (d[k] > m) m = d[k]; return m; }

int orc_clip_2dx2d_great_circle(const double x1_in[], const double y1_in[], const double z1_in[], int n1_in,
                                const double x2_in[], const double y2_in[], const double z2_in[], int n2_in,
                                double x_out[], double y_out[], double z_out[])
{
  GcList g1, g2, il, poly;
  g1.n = g2.n = il.n = poly.n = 0;
  /* create_xgrid.c:1508-1528 */
  if (minv(n1_in, x1_in) >= maxv(n2_in, x2_in) + GC_RANGE_CHECK) return 0;
  if (minv(n2_in, x2_in) >= maxv(n1_in, x1_in) + GC_RANGE_CHECK) return 0;
  if (minv(n1_in, y1_in) >= maxv(n2_in, y2_in) + GC_RANGE_CHECK) return 0;
  if (minv(n2_in, y2_in) >= maxv(n1_in, y1_in) + GC_RANGE_CHECK) return 0;
  if (minv(n1_in, z1_in) >= maxv(n2_in, z2_in) + GC_RANGE_CHECK) return 0;
  if (minv(n2_in, z2_in) >= maxv(n1_in, z1_in) + GC_RANGE_CHECK) return 0;

  for (int i = 0; i < n1_in; i++) if (add_end(&g1, x1_in[i], y1_in[i], z1_in[i], 0, 0, 0, -1)) return -9;
  for (int i = 0; i < n2_in; i++) if (add_end(&g2, x2_in[i], y2_in[i], z2_in[i], 0, 0, 0, -1)) return -9;
  const int npts1 = g1.n, npts2 = g2.n;
  int n_out = 0;
  for (int k = 0; k < g1.n; k++) g1.v[k].inside = inside_polygon(&g1.v[k], &g2);      /* :1549-1570 */
  for (int k = 0; k < g2.n; k++) g2.v[k].inside = inside_polygon(&g2.v[k], &g1);
  if (list_area(&g1) <= 0) return -1;                                                  /* :1575-1578 */
  if (list_area(&g2) <= 0) return -2;

  double pt1[GC_CAP][3], pt2[GC_CAP][3];
  for (int i = 0; i < npts1; i++) { pt1[i][0] = g1.v[i].x; pt1[i][1] = g1.v[i].y; pt1[i][2] = g1.v[i].z; }
  for (int i = 0; i < npts2; i++) { pt2[i][0] = g2.v[i].x; pt2[i][1] = g2.v[i].y; pt2[i][2] = g2.v[i].z; }

  for (int i1 = 0; i1 < npts1; i1++) {                                                 /* :1608-1671 */
    int i1p = (i1 + 1) % npts1;
    double *p1_0 = pt1[i1], *p1_1 = pt1[i1p];
    for (int i2 = 0; i2 < npts2; i2++) {
      int i2p = (i2 + 1) % npts2, i2p2 = (i2 + 2) % npts2, inbound, rc;
      double *p2_0 = pt2[i2], *p2_1 = pt2[i2p], *p2_2 = pt2[i2p2], I[3], u1, u2;
      if (orc_line_intersect_2D_3D(p1_0, p1_1, p2_0, p2_1, p2_2, I, &u1, &u2, &inbound)) {
        rc = add_intersect(&il, I[0], I[1], I[2], u1, u2, inbound, i1, i1p, i2, i2p);
        if (rc < 0) return rc;
        if (rc) {
          if (u1 == 1) rc = insert_intersect(&g1, I[0], I[1], I[2], 0.0, u2, inbound, p1_1[0], p1_1[1], p1_1[2]);
          else rc = insert_intersect(&g1, I[0], I[1], I[2], u1, u2, inbound, p1_0[0], p1_0[1], p1_0[2]);
          if (rc) return rc;
          if (u1 == 1) { p1_1[0] = I[0]; p1_1[1] = I[1]; p1_1[2] = I[2]; }
          else if (u1 == 0) { p1_0[0] = I[0]; p1_0[1] = I[1]; p1_0[2] = I[2]; }
          if (u2 == 1) rc = insert_intersect(&g2, I[0], I[1], I[2], 0.0, u1, 0, p2_1[0], p2_1[1], p2_1[2]);
          else rc = insert_intersect(&g2, I[0], I[1], I[2], u2, u1, 0, p2_0[0], p2_0[1], p2_0[2]);
          if (rc) return rc;
          if (u2 == 1) { p2_1[0] = I[0]; p2_1[1] = I[1]; p2_1[2] = I[2]; }
          else if (u2 == 0) { p2_0[0] = I[0]; p2_0[1] = I[1]; p2_0[2] = I[2]; }
        }
      }
    }
  }

  int has_inbound = 0, nintersect = il.n;                                              /* :1677-1696 */
  GcNode first, cur;
  memset(&first, 0, sizeof first);
  if (nintersect > 1) has_inbound = first_inbound(&il, &first);
  if (!has_inbound && nintersect > 1) {
    int rc = set_inbound(&il, &g1);
    if (rc) return rc;
    has_inbound = first_inbound(&il, &first);
  }

  if (has_inbound) {                                                                   /* :1701-1836 */
    const int maxiter1 = nintersect;
    if (find_node(&g1, &first) < 0) return -3;
    if (add_node(&poly, &first)) return -9;
    nintersect--;
    GcList *curl = &g1;
    int cur_num = 0, iter1 = 0, found1 = 0, found2 = 0;
    cur = first;
    while (iter1 < maxiter1) {
      int k1 = find_node(curl, &cur);
      if (k1 < 0) return -4;
      int k2 = (k1 + 1 < curl->n) ? k1 + 1 : 0;
      const int maxiter2 = curl->n;
      int iter2 = 0;
      found2 = 0;
      while (iter2 < maxiter2) {
        int t2_is_inter = 0;
        const GcNode *t2 = &curl->v[k2];
        if (t2->intersect) {
          if (same_node(t2, &first)) { found1 = 1; break; }
          const GcNode *t3 = &curl->v[(k2 + 1 < curl->n) ? k2 + 1 : 0];
          found2 = 1;
          t2_is_inter = 1;
          if (t3->intersect || (t3->inside == 1)) found2 = 0;
        }
        if (found2) { cur = *t2; break; }
        else {
          if (add_node(&poly, t2)) return -9;
          if (t2_is_inter) nintersect--;
        }
        k2 = (k2 + 1 < curl->n) ? k2 + 1 : 0;
        iter2++;
      }
      if (found1) break;
      if (!found2) return -4;
      if (same_node(&cur, &first)) { found1 = 1; break; }
      if (add_node(&poly, &cur)) return -9;
      nintersect--;
      if (cur_num == 0) { curl = &g2; cur_num = 1; }
      else { curl = &g1; cur_num = 0; }
      iter1++;
    }
    if (!found1) return -5;
    if (nintersect > 0) return -6;
    for (int k = 0; k < poly.n; k++) { x_out[n_out] = poly.v[k].x; y_out[n_out] = poly.v[k].y; z_out[n_out] = poly.v[k].z; n_out++; }
    if (n_out < 3) n_out = 0;
  }

  if (n_out == 0) {                                                                    /* :1839-1870: grid1 inside grid2 */
    int n1in2 = 0;
    for (int k = 0; k < g1.n; k++) if (g1.v[k].intersect != 1 && g1.v[k].inside == 1) n1in2++;
    if (npts1 == n1in2) {
      n_out = npts1;                           /* the reference walks the whole list but reports npts1 vertices */
      for (int k = 0; k < npts1; k++) { x_out[k] = g1.v[k].x; y_out[k] = g1.v[k].y; z_out[k] = g1.v[k].z; }
    }
    if (n_out > 0) return n_out;
  }
  if (n_out == 0) {                                                                    /* :1873-1904: grid2 inside grid1 */
    int n2in1 = 0;
    for (int k = 0; k < g2.n; k++) if (g2.v[k].intersect != 1 && g2.v[k].inside == 1) n2in1++;
    if (npts2 == n2in1) {
      n_out = npts2;
      for (int k = 0; k < npts2; k++) { x_out[k] = g2.v[k].x; y_out[k] = g2.v[k].y; z_out[k] = g2.v[k].z; }
    }
  }
  return n_out;
}

void orc_latlon2xyz(int size, const double *lon, const double *lat, double *x, double *y, double *z)
{                                                           /* mosaic_util.c:212-222 */
  for (int n = 0; n < size; n++) {
    x[n] = cos(lat[n]) * cos(lon[n]);
    y[n] = cos(lat[n]) * sin(lon[n]);
    z[n] = sin(lat[n]);
  }
}

static void cell_xyz(const double *x, const double *y, const double *z, int nxp, int i, int j, double *cx, double *cy, double *cz)
{                                                           /* clockwise: (j,i) (j+1,i) (j+1,i+1) (j,i+1), create_xgrid.c:1413-1420 */
  const int n0 = j * nxp + i, n1 = (j + 1) * nxp + i, n2 = (j + 1) * nxp + i + 1, n3 = j * nxp + i + 1;
  cx[0] = x[n0]; cy[0] = y[n0]; cz[0] = z[n0];
  cx[1] = x[n1]; cy[1] = y[n1]; cz[1] = z[n1];
  cx[2] = x[n2]; cy[2] = y[n2]; cz[2] = z[n2];
  cx[3] = x[n3]; cy[3] = y[n3]; cz[3] = z[n3];
}

void orc_get_grid_great_circle_area(int nx, int ny, const double *lon, const double *lat, double *area)
{                                                           /* create_xgrid.c:98-137 */
  const int nxp = nx + 1, nyp = ny + 1;
  double *x = (double *)malloc((size_t)nxp * nyp * sizeof(double)), *y = (double *)malloc((size_t)nxp * nyp * sizeof(double)),
         *z = (double *)malloc((size_t)nxp * nyp * sizeof(double));
  orc_latlon2xyz(nxp * nyp, lon, lat, x, y, z);
  for (int j = 0; j < ny; j++)
    for (int i = 0; i < nx; i++) {
      double cx[4], cy[4], cz[4];
      GcList g; g.n = 0;
      cell_xyz(x, y, z, nxp, i, j, cx, cy, cz);
      for (int k = 0; k < 4; k++) add_end(&g, cx[k], cy[k], cz[k], 0, 0, 0, -1);
      area[j * nx + i] = list_area(&g);
    }
  free(x); free(y); free(z);
}

/* create_xgrid_great_circle restricted to source rows [j1_begin, j1_end) (the whole call is rows [0, ny1)), with a
 * capacity instead of MAXXGRID.  Returns nxgrid, or -(100 + code) if the clip hit one of the reference's fatal errors,
 * or -99 on capacity. */
long orc_create_xgrid_great_circle_rows(int nx1, int ny1, int nx2, int ny2, const double *lon_in, const double *lat_in,
                                        const double *lon_out, const double *lat_out, const double *mask_in,
                                        int j1_begin, int j1_end, long capacity,
                                        int *i_in, int *j_in, int *i_out, int *j_out,
                                        double *xgrid_area, double *xgrid_clon, double *xgrid_clat)
{
  const int nx1p = nx1 + 1, nx2p = nx2 + 1, ny1p = ny1 + 1, ny2p = ny2 + 1;
  long nxgrid = 0;
  double *x1 = (double *)malloc((size_t)nx1p * ny1p * 3 * sizeof(double)), *y1 = x1 + (size_t)nx1p * ny1p, *z1 = y1 + (size_t)nx1p * ny1p;
  double *x2 = (double *)malloc((size_t)nx2p * ny2p * 3 * sizeof(double)), *y2 = x2 + (size_t)nx2p * ny2p, *z2 = y2 + (size_t)nx2p * ny2p;
  double *area1 = (double *)malloc((size_t)nx1 * ny1 * sizeof(double)), *area2 = (double *)malloc((size_t)nx2 * ny2 * sizeof(double));
  orc_latlon2xyz(nx1p * ny1p, lon_in, lat_in, x1, y1, z1);
  orc_latlon2xyz(nx2p * ny2p, lon_out, lat_out, x2, y2, z2);
  orc_get_grid_great_circle_area(nx1, ny1, lon_in, lat_in, area1);
  orc_get_grid_great_circle_area(nx2, ny2, lon_out, lat_out, area2);
  /* per-destination-cell boxes so the brute-force scan only pays six comparisons for far pairs -- the same six the
   * reference's clip starts with (create_xgrid.c:1508-1528), hoisted */
  double *bb = (double *)malloc((size_t)nx2 * ny2 * 6 * sizeof(double));
  for (int j2 = 0; j2 < ny2; j2++) for (int i2 = 0; i2 < nx2; i2++) {
    double cx[4], cy[4], cz[4], *b = bb + ((size_t)j2 * nx2 + i2) * 6;
    cell_xyz(x2, y2, z2, nx2p, i2, j2, cx, cy, cz);
    b[0] = minv(4, cx); b[1] = maxv(4, cx); b[2] = minv(4, cy); b[3] = maxv(4, cy); b[4] = minv(4, cz); b[5] = maxv(4, cz);
  }
  long rc_err = 0;
  for (int j1 = j1_begin; j1 < j1_end && !rc_err; j1++) for (int i1 = 0; i1 < nx1 && !rc_err; i1++) if (mask_in[j1 * nx1 + i1] > GC_MASK_THRESH) {
    double x1_in[4], y1_in[4], z1_in[4];
    cell_xyz(x1, y1, z1, nx1p, i1, j1, x1_in, y1_in, z1_in);
    const double a0 = minv(4, x1_in), a1 = maxv(4, x1_in), a2 = minv(4, y1_in), a3 = maxv(4, y1_in), a4 = minv(4, z1_in), a5 = maxv(4, z1_in);
    for (int j2 = 0; j2 < ny2 && !rc_err; j2++) for (int i2 = 0; i2 < nx2; i2++) {
      const double *b = bb + ((size_t)j2 * nx2 + i2) * 6;
      if (a0 >= b[1] + GC_RANGE_CHECK || b[0] >= a1 + GC_RANGE_CHECK || a2 >= b[3] + GC_RANGE_CHECK || b[2] >= a3 + GC_RANGE_CHECK ||
          a4 >= b[5] + GC_RANGE_CHECK || b[4] >= a5 + GC_RANGE_CHECK) continue;
      double x2_in[4], y2_in[4], z2_in[4], x_out[GC_CAP], y_out[GC_CAP], z_out[GC_CAP];
      cell_xyz(x2, y2, z2, nx2p, i2, j2, x2_in, y2_in, z2_in);
      int n_out = orc_clip_2dx2d_great_circle(x1_in, y1_in, z1_in, 4, x2_in, y2_in, z2_in, 4, x_out, y_out, z_out);
      if (n_out < 0) { rc_err = -(100 - n_out); break; }
      if (n_out > 0) {
        double xarea = orc_great_circle_area(n_out, x_out, y_out, z_out) * mask_in[j1 * nx1 + i1];
        double min_area = area1[j1 * nx1 + i1] < area2[j2 * nx2 + i2] ? area1[j1 * nx1 + i1] : area2[j2 * nx2 + i2];
        if (xarea / min_area > GC_AREA_RATIO_THRESH) {
          if (nxgrid >= capacity) { rc_err = -99; break; }
          xgrid_area[nxgrid] = xarea;
          if (xgrid_clon) xgrid_clon[nxgrid] = 0;
          if (xgrid_clat) xgrid_clat[nxgrid] = 0;
          i_in[nxgrid] = i1; j_in[nxgrid] = j1; i_out[nxgrid] = i2; j_out[nxgrid] = j2;
          ++nxgrid;
        }
      }
    }
  }
  free(bb); free(area1); free(area2); free(x1); free(x2);
  return rc_err ? rc_err : nxgrid;
}
