/* field_io_driver.c -- fregrid's per-level field loop (fregrid.c:1041-1075) over OUR replacement objects:
 *     get_input_data  ->  do_scalar_conserve_interp  ->  write_field_data
 * with integration/field_io_hip.c + integration/conserve_interp_hip.c, the reference's own structs (globals.h) and prototypes
 * (fregrid_util.h, conserve_interp.h), its mpp.c / mpp_domain.c, and tests/capi/mpp_io_fgnc.c for the two mpp_io calls.
 * Built by oracle/Makefile into oracle/_ref/field_io_driver; tests/test_gpu_field_io_driver.py runs it on the GPU and compares the
 * output files with the Python mirror (halo + grad_c2l + sweep on the device, narrowed to the file type) bit for bit.
 *
 * usage: field_io_driver ni nlon nlat nz workdir
 *   writes workdir/in.tile<N>.nc (variables t_f: NC_FLOAT (time, z, y, x), t_s: NC_SHORT with scale_factor / add_offset,
 *   both index formulas), then remaps  t_f with conserve_order2 and t_s with conserve_order1, level by level, into
 *   workdir/out.nc (same types). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "globals.h"
#include "conserve_interp.h"
#include "fregrid_util.h"
#include "mpp.h"
#include "mpp_io.h"
#include "mpp_domain.h"
#include "fregrid_hip.h"

int shim_register_file(fg_ncfile *f);                      /* tests/capi/mpp_io_fgnc.c */
static void *xcalloc(size_t n, size_t sz) { void *p = calloc(n ? n : 1, sz); if (!p) { fprintf(stderr, "out of memory\n"); exit(2); } return p; }
static void chk(int rc, const char *what) { if (rc < 0) { fprintf(stderr, "%s: %s / %s\n", what, fg_nc_last_error(), fg_last_error()); exit(3); } }

static float f_val(int t, int k, int j, int i) { return (float)(270.0 + ((t * 7 + k * 3 + j * 5 + i * 11) % 23) * 0.375 + 0.01 * k); }
static short s_val(int t, int k, int j, int i) { return (short)(((t * 5 + k * 7 + j * 3 + i * 13) % 4001) - 2000); }

int main(int argc, char **argv)
{
  int ni, nlon, nlat, nz, t, k, j, i, v;
  char path[1024];
  mpp_init(&argc, &argv);
  mpp_domain_init();
  if (argc != 6) { fprintf(stderr, "usage: field_io_driver ni nlon nlat nz workdir\n"); return 2; }
  ni = atoi(argv[1]); nlon = atoi(argv[2]); nlat = atoi(argv[3]); nz = atoi(argv[4]);
  {
    const size_t npt = (size_t)(ni + 1) * (ni + 1), nct = (size_t)ni * ni, npo = (size_t)(nlon + 1) * (nlat + 1);
    double *lonc = (double *)xcalloc(6 * npt, 8), *latc = (double *)xcalloc(6 * npt, 8), *lont = (double *)xcalloc(6 * nct, 8), *latt = (double *)xcalloc(6 * nct, 8);
    double *lono = (double *)xcalloc(npo, 8), *lato = (double *)xcalloc(npo, 8);
    Grid_config grid_in[6], grid_out[1];
    Field_config field_in[6], field_out[1];
    Var_config var_in[2], var_out[2];
    Interp_config interp2[1], interp1[1];
    int fid_in[6], fid_out[1];
    fg_ncfile *fo = NULL;
    const double scale = 0.01, offset = 250.0;
    if (fg_gnomonic_ed_grid(ni, 18.0, 1, lonc, latc, lont, latt) || fg_latlon_corners(nlon, nlat, 0.0, 360.0, -90.0, 90.0, 1, lono, lato)) {
      fprintf(stderr, "grid generation failed: %s\n", fg_last_error()); return 3;
    }
    memset(grid_in, 0, sizeof grid_in); memset(grid_out, 0, sizeof grid_out); memset(field_in, 0, sizeof field_in); memset(field_out, 0, sizeof field_out);
    memset(var_in, 0, sizeof var_in); memset(interp2, 0, sizeof interp2); memset(interp1, 0, sizeof interp1);
    /* --- the input files: one per tile, written with fg_nc_*, then re-opened read-only */
    for (t = 0; t < 6; t++) {
      fg_ncfile *f = NULL;
      int dims[4], vf, vs;
      float *bf = (float *)xcalloc((size_t)nz * nct, sizeof(float));
      short *bs = (short *)xcalloc((size_t)nz * nct, sizeof(short));
      long st[4] = {0, 0, 0, 0}, cn[4];
      snprintf(path, sizeof path, "%s/in.tile%d.nc", argv[5], t + 1);
      chk(fg_nc_create(path, 2, &f), "create");
      dims[0] = fg_nc_def_dim(f, "time", 0); dims[1] = fg_nc_def_dim(f, "z", nz); dims[2] = fg_nc_def_dim(f, "y", ni); dims[3] = fg_nc_def_dim(f, "x", ni);
      vf = fg_nc_def_var(f, "t_f", FG_NC_FLOAT, 4, dims); vs = fg_nc_def_var(f, "t_s", FG_NC_SHORT, 4, dims);
      chk(vf, "def_var"); chk(vs, "def_var");
      chk(fg_nc_put_att_double(f, vs, "scale_factor", FG_NC_DOUBLE, 1, &scale), "att"); chk(fg_nc_put_att_double(f, vs, "add_offset", FG_NC_DOUBLE, 1, &offset), "att");
      chk(fg_nc_enddef(f), "enddef");
      for (k = 0; k < nz; k++) for (j = 0; j < ni; j++) for (i = 0; i < ni; i++) {
        bf[(size_t)k * nct + (size_t)j * ni + i] = f_val(t, k, j, i); bs[(size_t)k * nct + (size_t)j * ni + i] = s_val(t, k, j, i);
      }
      cn[0] = 1; cn[1] = nz; cn[2] = ni; cn[3] = ni;
      chk(fg_nc_put_vara(f, vf, st, cn, bf), "put"); chk(fg_nc_put_vara(f, vs, st, cn, bs), "put");
      chk(fg_nc_close(f), "close");
      free(bf); free(bs);
      chk(fg_nc_open(path, &f), "open");
      fid_in[t] = shim_register_file(f);
      grid_in[t].nx = grid_in[t].nxc = ni; grid_in[t].ny = grid_in[t].nyc = ni; grid_in[t].iec = ni - 1; grid_in[t].jec = ni - 1;
      grid_in[t].lonc = lonc + t * npt; grid_in[t].latc = latc + t * npt; grid_in[t].lont = lont + t * nct; grid_in[t].latt = latt + t * nct;
      field_in[t].fid = &fid_in[t]; field_in[t].var = var_in; field_in[t].nvar = 2;
    }
    grid_out[0].nx = grid_out[0].nxc = nlon; grid_out[0].ny = grid_out[0].nyc = nlat; grid_out[0].iec = nlon - 1; grid_out[0].jec = nlat - 1;
    grid_out[0].lonc = lono; grid_out[0].latc = lato;
    /* --- the output file */
    {
      int dims[4];
      snprintf(path, sizeof path, "%s/out.nc", argv[5]);
      chk(fg_nc_create(path, 2, &fo), "create");
      dims[0] = fg_nc_def_dim(fo, "time", 0); dims[1] = fg_nc_def_dim(fo, "z", nz); dims[2] = fg_nc_def_dim(fo, "lat", nlat); dims[3] = fg_nc_def_dim(fo, "lon", nlon);
      chk(fg_nc_def_var(fo, "t_f", FG_NC_FLOAT, 4, dims), "def_var"); chk(fg_nc_def_var(fo, "t_s", FG_NC_SHORT, 4, dims), "def_var");
      chk(fg_nc_enddef(fo), "enddef");
      fid_out[0] = shim_register_file(fo);
      field_out[0].fid = &fid_out[0]; field_out[0].var = var_out; field_out[0].nvar = 2;
    }
    for (v = 0; v < 2; v++) {                               /* get_field_attribute's results for the two variables */
      snprintf(var_in[v].name, STRING, "%s", v ? "t_s" : "t_f");
      var_in[v].vid = v; var_in[v].type = v ? NC_SHORT : NC_FLOAT; var_in[v].ndim = 4; var_in[v].nz = nz; var_in[v].nn = 1;
      var_in[v].kstart = 0; var_in[v].kend = nz - 1; var_in[v].has_taxis = 1; var_in[v].has_zaxis = 1; var_in[v].has_naxis = 0;
      var_in[v].missing = -1.e20; var_in[v].interp_method = v ? CONSERVE_ORDER1 : CONSERVE_ORDER2; var_in[v].do_regrid = 1;
      var_in[v].scale = v ? scale : 0.0; var_in[v].offset = v ? offset : 0.0;
    }
    memcpy(var_out, var_in, sizeof var_in);

    setup_conserve_interp(6, grid_in, 1, grid_out, interp2, CONSERVE_ORDER2);
    setup_conserve_interp(6, grid_in, 1, grid_out, interp1, CONSERVE_ORDER1);
    for (v = 0; v < 2; v++) {
      Interp_config *interp = v ? interp1 : interp2;
      const unsigned int opcode = v ? CONSERVE_ORDER1 : CONSERVE_ORDER2;
      int level_z;
      for (level_z = var_in[v].kstart; level_z <= var_in[v].kend; level_z++) {            /* fregrid.c:1041-1075 */
        get_input_data(6, field_in, grid_in, NULL, v, level_z, 0, 0, 0, 0.0);
        field_out[0].data = (double *)xcalloc((size_t)nlon * nlat, sizeof(double));      /* allocate_field_data */
        do_scalar_conserve_interp(interp, v, 6, grid_in, 1, grid_out, field_in, field_out, opcode, 1);
        write_field_data(1, field_out, grid_out, v, level_z, 0, 0);
        for (t = 0; t < 6; t++) {
          if (opcode & CONSERVE_ORDER2) { free(field_in[t].grad_x); free(field_in[t].grad_y); free(field_in[t].grad_mask); }
          free(field_in[t].data);
          field_in[t].data = field_in[t].grad_x = field_in[t].grad_y = NULL; field_in[t].grad_mask = NULL;
        }
        free(field_out[0].data); field_out[0].data = NULL;
      }
    }
    chk(fg_nc_close(fo), "close");
  }
  printf("field_io_driver ok\n");
  return 0;
}
