/*
 * c2l_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE) for the order-2 input preparation
 * (SURVEY.md §8f-1): the cubed-sphere gradient and the halo update that feeds it.
 *
 *   orc_grad_c2l      grad_c2l + a2b_ord2, tools/libfrencutils/gradient_c2l.c:58-118,124-195.
 *                     PINNED: bit-identical to the compiled reference (oracle/_ref), tests/test_c2l_cpu.py.
 *   orc_update_halo   get_contact_direction + setup_boundary (CENTER, halo 1) + update_halo,
 *                     tools/fregrid/fregrid_util.c:2420-2442, :2446-2560, :2614-2658, kept in the reference's
 *                     two-step form (Bound table, then buffer copy with rotation) so that it checks the
 *                     product's folded gather map independently.
 *                     PARITY UNPINNED against executable reference code: fregrid_util.c includes <netcdf.h>, which
 *                     this image lacks, so it cannot be compiled here; the restatement is cross-checked by a
 *                     geometric property instead (halo cell centres must continue the neighbour tile's centres).
 */
#include <stdlib.h>
#include <string.h>

#define ORC_RADIUS 6371000.0

static void a2b_ord2(int nx, int ny, const double *qin, const double *edge_w, const double *edge_e,
                     const double *edge_s, const double *edge_n, double *qout)
{
  const int nxp = nx + 1, nyp = ny + 1, w = nx + 2;
  const double r3 = 1. / 3.;
  double *q1 = (double *)malloc((nx + 2) * sizeof(double)), *q2 = (double *)malloc((ny + 2) * sizeof(double));
  const int istart = 1, iend = nx, jstart = 1, jend = ny;          /* all four edges are cube edges */
  for (int j = jstart; j < jend; j++)
    for (int i = istart; i < iend; i++)
      qout[j * nxp + i] = 0.25 * (qin[j * w + i] + qin[j * w + i + 1] + qin[(j + 1) * w + i] + qin[(j + 1) * w + i + 1]);
  qout[0] = r3 * (qin[1 * w + 1] + qin[1 * w] + qin[1]);
  qout[nx] = r3 * (qin[1 * w + nx] + qin[nx] + qin[1 * w + nxp]);
  qout[ny * nxp + nx] = r3 * (qin[ny * w + nx] + qin[ny * w + nxp] + qin[nyp * w + nx]);
  qout[ny * nxp] = r3 * (qin[ny * w + 1] + qin[ny * w] + qin[nyp * w + 1]);
  for (int j = jstart; j <= jend; j++) q2[j] = 0.5 * (qin[j * w] + qin[j * w + 1]);
  for (int j = jstart; j < jend; j++) qout[j * nxp] = edge_w[j] * q2[j] + (1 - edge_w[j]) * q2[j + 1];
  for (int j = jstart; j <= jend; j++) q2[j] = 0.5 * (qin[j * w + nx] + qin[j * w + nxp]);
  for (int j = jstart; j < jend; j++) qout[j * nxp + nx] = edge_e[j] * q2[j] + (1 - edge_e[j]) * q2[j + 1];
  for (int i = istart; i <= iend; i++) q1[i] = 0.5 * (qin[i] + qin[w + i]);
  for (int i = istart; i < iend; i++) qout[i] = edge_s[i] * q1[i] + (1 - edge_s[i]) * q1[i + 1];
  for (int i = istart; i <= iend; i++) q1[i] = 0.5 * (qin[ny * w + i] + qin[nyp * w + i]);
  for (int i = istart; i < iend; i++) qout[ny * nxp + i] = edge_n[i] * q1[i] + (1 - edge_n[i]) * q1[i + 1];
  free(q1); free(q2);
}

void orc_grad_c2l(int nx, int ny, const double *pin, const double *dx, const double *dy, const double *area,
                  const double *edge_w, const double *edge_e, const double *edge_s, const double *edge_n,
                  const double *en_n, const double *en_e, const double *vlon, const double *vlat,
                  double *grad_x, double *grad_y)
{
  const int nxp = nx + 1, nyp = ny + 1;
  double *pb = (double *)malloc(sizeof(double) * nxp * nyp);
  double *pdx = (double *)malloc(sizeof(double) * 3 * nx * nyp);
  double *pdy = (double *)malloc(sizeof(double) * 3 * nxp * ny);
  a2b_ord2(nx, ny, pin, edge_w, edge_e, edge_s, edge_n, pb);
  for (int j = 0; j < nyp; j++) for (int i = 0; i < nx; i++) {
    int m0 = j * nx + i, m1 = j * nxp + i;
    for (int n = 0; n < 3; n++) pdx[3 * m0 + n] = 0.5 * (pb[m1] + pb[m1 + 1]) * dx[m0] * en_n[3 * m0 + n];
  }
  for (int j = 0; j < ny; j++) for (int i = 0; i < nxp; i++) {
    int m0 = j * nxp + i;
    for (int n = 0; n < 3; n++) pdy[3 * m0 + n] = 0.5 * (pb[m0] + pb[m0 + nxp]) * dy[m0] * en_e[3 * m0 + n];
  }
  for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) {
    int m0 = j * nx + i;
    double g3[3];
    for (int n = 0; n < 3; n++)
      g3[n] = pdx[3 * ((j + 1) * nx + i) + n] - pdx[3 * m0 + n] - pdy[3 * (j * nxp + i) + n] + pdy[3 * (j * nxp + i + 1) + n];
    grad_x[m0] = (vlon[3 * m0] * g3[0] + vlon[3 * m0 + 1] * g3[1] + vlon[3 * m0 + 2] * g3[2]) / area[m0];
    grad_x[m0] *= ORC_RADIUS;
    grad_y[m0] = (vlat[3 * m0] * g3[0] + vlat[3 * m0 + 1] * g3[1] + vlat[3 * m0 + 2] * g3[2]) / area[m0];
    grad_y[m0] *= ORC_RADIUS;
  }
  free(pb); free(pdx); free(pdy);
}

/* ---- halo update, reference form -------------------------------------------------------------------------- */
enum { O_WEST = 6, O_SOUTH = 7, O_EAST = 4, O_NORTH = 5 };    /* globals.h:38-43 */
enum { R_ZERO = 0, R_NINETY = 90, R_MINUS_NINETY = -90, R_180 = 180 };
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* data[t]: halo'd arrays [nz][(ny+2)][(nx+2)] per tile, interiors filled; halos are overwritten in place.
 * Contacts in read_mosaic_contact's convention (tile numbers 1-based).  Returns 0, or -1 on an inconsistent contact. */
int orc_update_halo(int ntiles, const int *nx, const int *ny, int ncontacts, const int *tile1, const int *tile2,
                    const int *istart1, const int *iend1, const int *jstart1, const int *jend1,
                    const int *istart2, const int *iend2, const int *jstart2, const int *jend2,
                    int nz, double **data)
{
  const int halo = 1, shift = 0;
  const int n2 = 2 * ncontacts;
  int *tile = (int *)malloc(n2 * sizeof(int)), *is = (int *)malloc(n2 * sizeof(int)), *ie = (int *)malloc(n2 * sizeof(int));
  int *js = (int *)malloc(n2 * sizeof(int)), *je = (int *)malloc(n2 * sizeof(int)), *dir = (int *)malloc(n2 * sizeof(int));
  for (int l = 0; l < ncontacts; l++) {
    tile[l] = tile1[l] - 1; is[l] = istart1[l]; ie[l] = iend1[l]; js[l] = jstart1[l]; je[l] = jend1[l];
    tile[l + ncontacts] = tile2[l] - 1; is[l + ncontacts] = istart2[l]; ie[l + ncontacts] = iend2[l];
    js[l + ncontacts] = jstart2[l]; je[l + ncontacts] = jend2[l];
  }
  int rc = 0;
  for (int n = 0; n < n2; n++) {                                    /* get_contact_direction */
    if ((is[n] == ie[n] && js[n] == je[n]) || (is[n] != ie[n] && js[n] != je[n])) { rc = -1; break; }
    if (is[n] == ie[n]) dir[n] = (is[n] == 0) ? O_WEST : O_EAST;
    else dir[n] = (js[n] == 0) ? O_SOUTH : O_NORTH;
  }
  for (int n = 0; n < ntiles && !rc; n++) {
    for (int l = 0; l < n2 && !rc; l++) {
      if (tile[l] != n) continue;
      int is1 = 0, ie1 = 0, js1 = 0, je1 = 0, is2 = 0, ie2 = 0, js2 = 0, je2 = 0;
      const int nxn = nx[n], nyn = ny[n];
      switch (dir[l]) {
      case O_WEST:  is1 = 0; ie1 = halo - 1; js1 = imin(js[l], je[l]) + halo; je1 = imax(js[l], je[l]) + halo + shift; break;
      case O_EAST:  is1 = nxn + shift + halo; ie1 = nxn + shift + halo + halo - 1; js1 = imin(js[l], je[l]) + halo; je1 = imax(js[l], je[l]) + halo + shift; break;
      case O_SOUTH: is1 = imin(is[l], ie[l]) + halo; ie1 = imax(is[l], ie[l]) + halo + shift; js1 = 0; je1 = halo - 1; break;
      case O_NORTH: is1 = imin(is[l], ie[l]) + halo; ie1 = imax(is[l], ie[l]) + halo + shift; js1 = nyn + shift + halo; je1 = nyn + shift + halo + halo - 1; break;
      }
      int l2 = (l + ncontacts) % n2, t2 = tile[l2];
      switch (dir[l2]) {
      case O_WEST:  is2 = halo + shift; ie2 = halo + shift + halo - 1; js2 = imin(js[l2], je[l2]) + halo; je2 = imax(js[l2], je[l2]) + halo + shift; break;
      case O_EAST:  is2 = nxn - halo + 1; ie2 = nxn; js2 = imin(js[l2], je[l2]) + halo; je2 = imax(js[l2], je[l2]) + halo + shift; break;
      case O_SOUTH: is2 = imin(is[l2], ie[l2]) + halo; ie2 = imax(is[l2], ie[l2]) + halo + shift; js2 = halo + shift; je2 = halo + shift + halo - 1; break;
      case O_NORTH: is2 = imin(is[l2], ie[l2]) + halo; ie2 = imax(is[l2], ie[l2]) + halo + shift; js2 = nyn - halo + 1; je2 = nyn; break;
      }
      int rotate = R_ZERO;
      if (dir[l] == O_WEST && dir[l2] == O_NORTH) rotate = R_NINETY;
      if (dir[l] == O_EAST && dir[l2] == O_SOUTH) rotate = R_NINETY;
      if (dir[l] == O_SOUTH && dir[l2] == O_EAST) rotate = R_MINUS_NINETY;
      if (dir[l] == O_NORTH && dir[l2] == O_WEST) rotate = R_MINUS_NINETY;
      if (dir[l] == O_NORTH && dir[l2] == O_NORTH) rotate = R_180;
      if ((ie2 - is2 + 1) * (je2 - js2 + 1) != (ie1 - is1 + 1) * (je1 - js1 + 1)) { rc = -1; break; }
      /* update_halo for this bound */
      const int nxa = nxn + 2, nya = nyn + 2, nxb = nx[t2] + 2, nyb = ny[t2] + 2;
      const long size1 = (long)nxa * nya, size2 = (long)nxb * nyb;
      long bufsize = (long)nz * (ie2 - is2 + 1) * (je2 - js2 + 1), q = 0;
      double *buffer = (double *)malloc(bufsize * sizeof(double));
      const double *src = data[t2];
      switch (rotate) {
      case R_ZERO:         for (int k = 0; k < nz; k++) for (int j = js2; j <= je2; j++) for (int i = is2; i <= ie2; i++) buffer[q++] = src[k * size2 + (long)j * nxb + i]; break;
      case R_NINETY:       for (int k = 0; k < nz; k++) for (int i = ie2; i >= is2; i--) for (int j = js2; j <= je2; j++) buffer[q++] = src[k * size2 + (long)j * nxb + i]; break;
      case R_MINUS_NINETY: for (int k = 0; k < nz; k++) for (int i = is2; i <= ie2; i++) for (int j = je2; j >= js2; j--) buffer[q++] = src[k * size2 + (long)j * nxb + i]; break;
      case R_180:          for (int k = 0; k < nz; k++) for (int j = je2; j >= js2; j--) for (int i = ie2; i >= is2; i--) buffer[q++] = src[k * size2 + (long)j * nxb + i]; break;
      }
      q = 0;
      for (int k = 0; k < nz; k++) for (int j = js1; j <= je1; j++) for (int i = is1; i <= ie1; i++)
        data[n][k * size1 + (long)j * nxa + i] = buffer[q++];
      free(buffer);
    }
  }
  free(tile); free(is); free(ie); free(js); free(je); free(dir);
  return rc;
}

/* fregrid_util.c:2203-2216 */
void orc_grad_mask(int nx, int ny, const double *data, double missing, int *mask)
{
  const int w = nx + 2;
  for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) {
    int ii = i + 1, ip1 = ii + 1, im1 = ii - 1, jj = j + 1, jp1 = jj + 1, jm1 = jj - 1;
    int m = 0;
    if (data[jm1 * w + im1] == missing || data[jm1 * w + ii] == missing || data[jm1 * w + ip1] == missing ||
        data[jj * w + im1] == missing || data[jj * w + ip1] == missing || data[jp1 * w + im1] == missing ||
        data[jp1 * w + ii] == missing || data[jp1 * w + ip1] == missing) m = 1;
    mask[j * nx + i] = m;
  }
}
