/* b2_driver.c -- runs integration/conserve_interp_hip.c (the C99 replacement for tools/fregrid/conserve_interp.c) the way
 * fregrid.c does: fills the reference's OWN Grid_config / Interp_config / Field_config structs (tools/libfrencutils/globals.h),
 * calls setup_conserve_interp and do_scalar_conserve_interp (conserve_interp.h:24-32), and dumps what they return.
 * Built by oracle/Makefile into oracle/_ref/b2_driver (it needs the reference's headers and its mpp.c / mpp_domain.c, compiled
 * where they lie; single process, no MPI); tests/test_gpu_b2_driver.py runs it on the GPU and compares every array with the
 * Python mirror of the same two functions, bit for bit.
 *
 * usage: b2_driver ni nlon nlat out.bin
 *   scenario 1: conserve_order2, nz = 2, plain branch          (halo'd data, grad_x, grad_y)
 *   scenario 2: conserve_order1, nz = 1, has_missing = 1       (every 7th source cell missing)
 *   scenario 3: scenario 1 with WRITE | CHECK_CONSERVE         (remap file <out.bin>.remap.nc; the area check :450-490 prints)
 *   scenario 4: scenario 1 with READ of that file              (fg_remap_read -> fg_plan_set_xgrid instead of a search)
 *   scenario 5: conserve_order2, nz = 1, MONOTONIC             (fg_plan_apply_ex: the limiter, conserve_interp.c:617-742)
 *   scenario 6: conserve_order1, nz = 1, TARGET, cell_methods = sum, grid_in[].weight     (the remaining options of :561-616, :815-870)
 *   scenario 7: conserve_order1 | GREAT_CIRCLE, nz = 1         (fg_plan_create_great_circle, conserve_interp.c:164-168)
 *   scenario 8: conserve_order2, nz = 1, has_missing = 1, grad_mask set on every 5th diagonal   (conserve_interp.c:743-813 with missing data)
 *   out.bin per scenario: int nxgrid; int t_in,i_in,j_in,i_out,j_out [nxgrid]; double area[nxgrid]; (order 2: double di, dj [nxgrid]);
 *                         double field_out[nz * nlon * nlat]
 * Input fields are index formulas that a test can restate exactly. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "globals.h"
#include "conserve_interp.h"
#include "mpp.h"
#include "mpp_domain.h"
void get_grid_area(const int *nlon, const int *nlat, const double *lon, const double *lat, double *area);   /* create_xgrid.h:38 */
#include "fregrid_hip.h"

static void *xcalloc(size_t n, size_t sz) { void *p = calloc(n ? n : 1, sz); if (!p) { fprintf(stderr, "out of memory\n"); exit(2); } return p; }
static void wr(const void *p, size_t sz, size_t n, FILE *f) { if (fwrite(p, sz, n, f) != n) { fprintf(stderr, "short write\n"); exit(2); } }

static double f_data(int t, int k, int j, int i) { return ((t * 7 + k * 3 + j * 5 + i * 11) % 17) * 0.25 + 1.0; }
static double f_gx(int t, int k, int j, int i) { return ((t + k + i * 3 + j) % 5) * 0.125 - 0.25; }
static double f_gy(int t, int k, int j, int i) { return ((t * 2 + k + i + j * 2) % 7) * 0.0625 - 0.1875; }

int main(int argc, char **argv)
{
  int ni, nlon, nlat, t, k, j, i, sc;
  FILE *f;
  mpp_init(&argc, &argv);
  mpp_domain_init();                               /* fregrid.c:414-415 */
  if (argc != 5) { fprintf(stderr, "usage: b2_driver ni nlon nlat out.bin\n"); return 2; }
  ni = atoi(argv[1]); nlon = atoi(argv[2]); nlat = atoi(argv[3]);
  {
    const size_t npt = (size_t)(ni + 1) * (ni + 1), npo = (size_t)(nlon + 1) * (nlat + 1);
    double *lonc = (double *)xcalloc(6 * npt, sizeof(double)), *latc = (double *)xcalloc(6 * npt, sizeof(double));
    double *lono = (double *)xcalloc(npo, sizeof(double)), *lato = (double *)xcalloc(npo, sizeof(double));
    Grid_config grid_in[6], grid_out[1];
    if (fg_gnomonic_ed_corners(ni, 18.0, 1, lonc, latc) || fg_latlon_corners(nlon, nlat, 0.0, 360.0, -90.0, 90.0, 1, lono, lato)) {
      fprintf(stderr, "grid generation failed: %s\n", fg_last_error()); return 3;
    }
    memset(grid_in, 0, sizeof grid_in); memset(grid_out, 0, sizeof grid_out);
    for (t = 0; t < 6; t++) {
      grid_in[t].nx = grid_in[t].nxc = ni; grid_in[t].ny = grid_in[t].nyc = ni;
      grid_in[t].iec = ni - 1; grid_in[t].jec = ni - 1;
      grid_in[t].lonc = lonc + t * npt; grid_in[t].latc = latc + t * npt;
    }
    grid_out[0].nx = grid_out[0].nxc = nlon; grid_out[0].ny = grid_out[0].nyc = nlat;
    grid_out[0].iec = nlon - 1; grid_out[0].jec = nlat - 1;
    grid_out[0].lonc = lono; grid_out[0].latc = lato;

    f = fopen(argv[4], "wb");
    if (!f) { perror(argv[4]); return 2; }
    for (sc = 1; sc <= 8; sc++) {
      const int order = (sc == 2 || sc == 6 || sc == 7) ? 1 : 2, nz = (sc == 2 || sc >= 5) ? 1 : 2, halo = (order == 2) ? 1 : 0;
      unsigned int opcode = (order == 2) ? CONSERVE_ORDER2 : CONSERVE_ORDER1;
      if (sc == 3) opcode |= WRITE | CHECK_CONSERVE;
      if (sc == 4) opcode |= READ;
      if (sc == 5) opcode |= MONOTONIC;
      if (sc == 6) opcode |= TARGET;
      if (sc == 7) opcode |= GREAT_CIRCLE;
      const size_t nd = (size_t)(ni + 2 * halo) * (ni + 2 * halo), nc = (size_t)ni * ni;
      Interp_config interp[1];
      Field_config field_in[6], field_out[1];
      Var_config var;
      int nx;
      memset(interp, 0, sizeof interp); memset(field_in, 0, sizeof field_in); memset(field_out, 0, sizeof field_out); memset(&var, 0, sizeof var);
      if (sc >= 3) { snprintf(interp[0].remap_file, STRING, "%s.remap.nc", argv[4]); interp[0].file_exist = (sc == 4); }
      if (sc == 3 || sc == 6) {                      /* get_input_output_cell_area, fregrid_util.c:363-408: the area check reads them */
        for (t = 0; t < 6; t++) if (!grid_in[t].cell_area) {
          grid_in[t].cell_area = (double *)xcalloc((size_t)ni * ni, sizeof(double));
          get_grid_area(&ni, &ni, grid_in[t].lonc, grid_in[t].latc, grid_in[t].cell_area);
        }
        if (!grid_out[0].cell_area) {
          grid_out[0].cell_area = (double *)xcalloc((size_t)nlon * nlat, sizeof(double));
          get_grid_area(&nlon, &nlat, lono, lato, grid_out[0].cell_area);
        }
      }
      setup_conserve_interp(6, grid_in, 1, grid_out, interp, opcode);
      var.interp_method = order; var.has_missing = (sc == 2 || sc == 8); var.missing = -1.e10;
      var.cell_methods = (sc == 6) ? CELL_METHODS_SUM : CELL_METHODS_MEAN;
      for (t = 0; t < 6; t++) {
        grid_in[t].weight_exist = (sc == 6);
        if (sc == 6 && !grid_in[t].weight) {
          grid_in[t].weight = (double *)xcalloc((size_t)ni * ni, sizeof(double));
          for (j = 0; j < ni; j++) for (i = 0; i < ni; i++) grid_in[t].weight[(size_t)j * ni + i] = 0.5 + ((t + i + 2 * j) % 4) * 0.125;
        }
      }
      for (t = 0; t < 6; t++) {
        field_in[t].var = &var;
        field_in[t].data = (double *)xcalloc((size_t)nz * nd, sizeof(double));
        for (k = 0; k < nz; k++) for (j = 0; j < ni + 2 * halo; j++) for (i = 0; i < ni + 2 * halo; i++) {
          double v = f_data(t, k, j, i);
          if ((sc == 2 || sc == 8) && (t + j * ni + i) % 7 == 0) v = var.missing;
          field_in[t].data[(size_t)k * nd + (size_t)j * (ni + 2 * halo) + i] = v;
        }
        if (order == 2) {
          field_in[t].grad_x = (double *)xcalloc((size_t)nz * nc, sizeof(double));
          field_in[t].grad_y = (double *)xcalloc((size_t)nz * nc, sizeof(double));
          field_in[t].grad_mask = (int *)xcalloc(nc, sizeof(int));
          for (k = 0; k < nz; k++) for (j = 0; j < ni; j++) for (i = 0; i < ni; i++) {
            field_in[t].grad_x[(size_t)k * nc + (size_t)j * ni + i] = f_gx(t, k, j, i);
            field_in[t].grad_y[(size_t)k * nc + (size_t)j * ni + i] = f_gy(t, k, j, i);
            if (sc == 8) field_in[t].grad_mask[(size_t)j * ni + i] = ((i + j) % 5 == 0);
          }
        }
      }
      field_out[0].var = &var;
      field_out[0].data = (double *)xcalloc((size_t)nz * nlon * nlat, sizeof(double));
      do_scalar_conserve_interp(interp, 0, 6, grid_in, 1, grid_out, field_in, field_out, opcode, nz);
      nx = (int)interp[0].nxgrid;
      wr(&nx, sizeof(int), 1, f);
      wr(interp[0].t_in, sizeof(int), nx, f); wr(interp[0].i_in, sizeof(int), nx, f); wr(interp[0].j_in, sizeof(int), nx, f);
      wr(interp[0].i_out, sizeof(int), nx, f); wr(interp[0].j_out, sizeof(int), nx, f);
      wr(interp[0].area, sizeof(double), nx, f);
      if (order == 2) { wr(interp[0].di_in, sizeof(double), nx, f); wr(interp[0].dj_in, sizeof(double), nx, f); }
      wr(field_out[0].data, sizeof(double), (size_t)nz * nlon * nlat, f);
      for (t = 0; t < 6; t++) { free(field_in[t].data); free(field_in[t].grad_x); free(field_in[t].grad_y); free(field_in[t].grad_mask); }
      free(field_out[0].data);
    }
    fclose(f);
  }
  printf("b2_driver ok\n");
  return 0;
}
