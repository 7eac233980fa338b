/*
 * field_io_hip.c -- device-side replacements for the two functions of tools/fregrid/fregrid_util.c that sit either side of
 * do_scalar_conserve_interp in fregrid's field loop (fregrid.c:1041-1062), same prototypes (fregrid_util.h:45,51):
 *
 *   get_input_data    fregrid_util.c:2036-2216   hyperslab read -> widen / scale / offset -> [halo copy, update_halo, grad_c2l,
 *                                                grad_mask for conserve_order2]
 *   write_field_data  fregrid_util.c:2339-2418   offset / scale back -> cast to the file type -> hyperslab write
 *
 * With conserve_interp_hip.c alone every level crosses PCIe as double three times over (halo'd data, grad_x, grad_y: 21 MB per
 * level of C384) after the CPU has done the halo update and grad_c2l.  Here the raw level goes up ONCE in its file type, the
 * conversion, the halo update, grad_c2l and the gradient mask run on the device (fg_dev_widen, fg_c2l_fill_halo,
 * fg_c2l_gradient), do_scalar_conserve_interp finds the device arrays through fregrid_hip_glue.h, and the remapped level comes
 * back narrowed to the file type (fg_dev_narrow).
 *
 * Building it into fregrid without touching the reference's sources: compile fregrid_util.c with
 *     -Dget_input_data=get_input_data_cpu -Dwrite_field_data=write_field_data_cpu
 * and this file with -DFG_HAVE_CPU_FIELD_IO; the cases this file does not serve (--extrapolate, ranks > 1 on the write side, a
 * grid with a halo of its own) are passed to those.  Without that macro they are fatal errors (the test harness).
 *
 * The host arrays of Field_config are still malloc'ed -- fregrid.c frees them after every level (:1067-1075) -- but the
 * input arrays are NOT filled (zeros): the data lives on the device.  do_scalar_conserve_interp's CHECK_CONSERVE branch
 * downloads what it needs.  Plain C99, no HIP headers.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "globals.h"
#include "mpp.h"
#include "mpp_io.h"
#include "mpp_domain.h"
#include "fregrid_util.h"              /* the prototypes of the two functions defined here (:45, :51) */
#include "fregrid_hip.h"
#include "fregrid_hip_glue.h"

#ifdef FG_HAVE_CPU_FIELD_IO
void get_input_data_cpu(int ntiles, Field_config *field, Grid_config *grid, Bound_config *bound,
                        int varid, int level_z, int level_n, int level_t, int extrapolate, double stop_crit);
void write_field_data_cpu(int ntiles, Field_config *field, Grid_config *grid, int varid, int level_z, int level_n, int level_t);
#endif

static int io_device(void)
{
  const char *e = getenv("FREGRID_HIP_DEVICE");
  return e ? atoi(e) : 0;
}
static void io_fatal(const char *where)
{
  char msg[768];
  snprintf(msg, sizeof msg, "%s: %s", where, fg_last_error());
  mpp_error(msg);
}
static int fg_type(nc_type t)
{
  switch (t) {
    case NC_SHORT: return FG_NC_SHORT;
    case NC_INT: return FG_NC_INT;
    case NC_FLOAT: return FG_NC_FLOAT;
    case NC_DOUBLE: return FG_NC_DOUBLE;
    default: return 0;
  }
}

/* ------------------------------------------------------------------------------------------ the registry */
typedef struct { const Field_config *key; FgDevField f; } InSlot;
typedef struct { const Field_config *key; double *d_out; long n; int nz; } OutSlot;
static InSlot *g_in = NULL;   static int g_nin = 0;
static OutSlot *g_out = NULL; static int g_nout = 0;

static void in_release(InSlot *s)
{
  fg_dev_free(s->f.d_data); fg_dev_free(s->f.d_gx); fg_dev_free(s->f.d_gy); fg_dev_free(s->f.d_gm);
  memset(&s->f, 0, sizeof s->f);
}
const FgDevField *fg_glue_input(const Field_config *field, int varid)
{
  int k;
  for (k = 0; k < g_nin; k++) if (g_in[k].key == field && g_in[k].f.d_data && g_in[k].f.varid == varid) return &g_in[k].f;
  return NULL;
}
static FgDevField *in_slot(const Field_config *field)
{
  int k;
  for (k = 0; k < g_nin; k++) if (g_in[k].key == field) { in_release(&g_in[k]); return &g_in[k].f; }
  g_in = (InSlot *)realloc(g_in, (size_t)(g_nin + 1) * sizeof(InSlot));
  if (!g_in) mpp_error("field_io(hip): out of memory");
  memset(&g_in[g_nin], 0, sizeof(InSlot));
  g_in[g_nin].key = field;
  return &g_in[g_nin++].f;
}
int fg_glue_output_put(const Field_config *field_out_n, double *d_out, long n, int nz)
{
  int k;
  for (k = 0; k < g_nout; k++) if (g_out[k].key == field_out_n) break;
  if (k == g_nout) {
    g_out = (OutSlot *)realloc(g_out, (size_t)(g_nout + 1) * sizeof(OutSlot));
    if (!g_out) mpp_error("field_io(hip): out of memory");
    memset(&g_out[g_nout], 0, sizeof(OutSlot));
    g_out[g_nout++].key = field_out_n;
  }
  fg_dev_free(g_out[k].d_out);                       /* a level nobody wrote */
  g_out[k].d_out = d_out; g_out[k].n = n; g_out[k].nz = nz;
  return 1;
}

/* ------------------------------------------------------------------------------------------ gradient preparation object */
static fg_c2l *g_c2l = NULL;
static const Grid_config *g_c2l_grid = NULL;
static int g_c2l_ntiles = 0;

static fg_c2l *c2l_for(int ntiles, const Grid_config *grid, int dev)
{
  if (g_c2l && g_c2l_grid == grid && g_c2l_ntiles == ntiles) return g_c2l;
  if (g_c2l) { fg_c2l_destroy(g_c2l); g_c2l = NULL; }
  {
    /* the contacts of the mosaic, found from the corner coordinates (what read_mosaic_contact hands to fregrid; fregrid turns
     * them into Bound_config, fregrid_util.c:2446-2560 -- fg_c2l folds both steps into one gather map) */
    const int maxc = 4 * ntiles + 4;
    int *nx = (int *)malloc((size_t)ntiles * sizeof(int)), *ny = (int *)malloc((size_t)ntiles * sizeof(int));
    const double **lonc = (const double **)malloc((size_t)ntiles * sizeof(double *)), **latc = (const double **)malloc((size_t)ntiles * sizeof(double *));
    const double **lont = (const double **)malloc((size_t)ntiles * sizeof(double *)), **latt = (const double **)malloc((size_t)ntiles * sizeof(double *));
    int *c = (int *)malloc((size_t)maxc * 10 * sizeof(int));
    int n, nc;
    if (!nx || !ny || !lonc || !latc || !lont || !latt || !c) mpp_error("field_io(hip): out of memory");
    for (n = 0; n < ntiles; n++) {
      if (grid[n].halo != 0) mpp_error("field_io(hip): input grids with a halo of their own are not supported");
      nx[n] = grid[n].nx; ny[n] = grid[n].ny; lonc[n] = grid[n].lonc; latc[n] = grid[n].latc; lont[n] = grid[n].lont; latt[n] = grid[n].latt;
      if (!lont[n] || !latt[n]) mpp_error("field_io(hip): conserve_order2 needs the T-cell centres (grid_in[].lont / latt)");
    }
    nc = fg_find_contacts(ntiles, nx, ny, lonc, latc, maxc, c, c + maxc, c + 2 * maxc, c + 3 * maxc, c + 4 * maxc, c + 5 * maxc,
                          c + 6 * maxc, c + 7 * maxc, c + 8 * maxc, c + 9 * maxc);
    if (nc < 0) io_fatal("get_input_data");
    if (fg_c2l_create(ntiles, nx, ny, lonc, latc, lont, latt, nc, c, c + maxc, c + 2 * maxc, c + 3 * maxc, c + 4 * maxc, c + 5 * maxc,
                      c + 6 * maxc, c + 7 * maxc, c + 8 * maxc, c + 9 * maxc, dev, &g_c2l)) io_fatal("get_input_data");
    free(nx); free(ny); free(lonc); free(latc); free(lont); free(latt); free(c);
  }
  g_c2l_grid = grid; g_c2l_ntiles = ntiles;
  return g_c2l;
}

/* ------------------------------------------------------------------------------------------ get_input_data */
void get_input_data(int ntiles, Field_config *field, Grid_config *grid, Bound_config *bound,
                    int varid, int level_z, int level_n, int level_t, int extrapolate, double stop_crit)
{
  const int dev = io_device();
  const double missing_value = field->var[varid].missing;
  const int interp_method = field->var[varid].interp_method;
  const int halo = (interp_method == CONSERVE_ORDER1) ? 0 : 1;
  const int ftype = fg_type(field->var[varid].type);
  /* mpp_get_var_value_block hands NC_FLOAT data over as double (nc_get_vara_double, mpp_io.c:454) */
  const int uptype = (ftype == FG_NC_FLOAT) ? FG_NC_DOUBLE : ftype;
  const size_t upsz = (uptype == FG_NC_SHORT) ? 2 : (uptype == FG_NC_INT ? 4 : 8);
  int nz = 1, ndim, pos = 0, i, n;
  size_t start[8], nread[8], ncell = 0, fstride = 0, off;
  char *raw, *d_raw;
  FgDevField *slot;
  (void)bound; (void)stop_crit;
  if (extrapolate || (interp_method != CONSERVE_ORDER1 && interp_method != CONSERVE_ORDER2)) {
#ifdef FG_HAVE_CPU_FIELD_IO
    get_input_data_cpu(ntiles, field, grid, bound, varid, level_z, level_n, level_t, extrapolate, stop_crit);
    return;
#else
    mpp_error("field_io(hip): --extrapolate and non-conservative methods are served by the reference's get_input_data");
#endif
  }
  if (!ftype) mpp_error("fregrid_util(get_input_data): field type should be NC_INT, NC_SHORT, NC_FLOAT or NC_DOUBLE");
  if (level_z < 0) nz = field->var[varid].nz;
  ndim = field->var[varid].ndim;
  if (ndim < 2 || ndim > 6) mpp_error("fregrid_util(get_input_data): ndim must be no less than 2");
  for (i = 0; i < ndim; i++) { start[i] = 0; nread[i] = 1; }
  if (field->var[varid].has_taxis) start[pos++] = (size_t)level_t;
  if (field->var[varid].has_naxis) start[pos++] = (size_t)level_n;
  if (field->var[varid].has_zaxis) {
    if (level_z < 0) { nread[pos] = (size_t)field->var[varid].nz; start[pos++] = (size_t)field->var[varid].kstart; }
    else start[pos++] = (size_t)level_z;
  }
  if (ndim != pos + 2) mpp_error("fregrid_util(get_input_data): mimstch between ndim and has_taxis/has_zaxis/has_naxis");
  for (n = 0; n < ntiles; n++) { ncell += (size_t)grid[n].nx * grid[n].ny; fstride += (size_t)(grid[n].nx + 2 * halo) * (grid[n].ny + 2 * halo); }

  /* the level(s) of every tile, tiles back to back per level, in the type mpp_io hands them over: one upload */
  raw = (char *)malloc(ncell * (size_t)nz * upsz + 8);
  d_raw = (char *)fg_dev_alloc(ncell * (size_t)nz * upsz + 8, dev);
  if (!raw || !d_raw) io_fatal("get_input_data");
  off = 0;
  for (n = 0; n < ntiles; n++) {
    const size_t nc = (size_t)grid[n].nx * grid[n].ny;
    char *tmp = (char *)malloc(nc * (size_t)nz * upsz + 8);
    int k;
    nread[pos] = (size_t)grid[n].ny; nread[pos + 1] = (size_t)grid[n].nx;
    mpp_get_var_value_block(*(field[n].fid), field[n].var[varid].vid, start, nread, tmp);
    for (k = 0; k < nz; k++) memcpy(raw + ((size_t)k * ncell + off) * upsz, tmp + (size_t)k * nc * upsz, nc * upsz);
    free(tmp);
    off += nc;
    /* the host arrays fregrid.c frees after the level: allocated, not filled (see the header of this file) */
    field[n].data = (double *)calloc((size_t)(grid[n].nx + 2 * halo) * (grid[n].ny + 2 * halo) * (size_t)nz, sizeof(double));
    if (interp_method == CONSERVE_ORDER2) {
      field[n].grad_x = (double *)calloc(nc * (size_t)nz, sizeof(double));
      field[n].grad_y = (double *)calloc(nc * (size_t)nz, sizeof(double));
      field[n].grad_mask = (int *)calloc(nc * (size_t)nz, sizeof(int));
    }
    if (field[n].var[varid].cell_measures) {                    /* fregrid_util.c:2147-2162, on the host as there */
      size_t start2[4] = {0, 0, 0, 0}, nread2[4] = {1, 1, 1, 1};
      int q = 0;
      if (!field[n].area) field[n].area = (double *)malloc(nc * sizeof(double));
      if (field[n].var[varid].area_has_taxis) start2[q++] = (size_t)level_t;
      if (field[n].var[varid].area_has_naxis) start2[q++] = (size_t)level_n;
      if (field[n].var[varid].area_has_zaxis) start2[q++] = (size_t)level_z;
      nread2[q] = (size_t)grid[n].ny; nread2[q + 1] = (size_t)grid[n].nx;
      mpp_get_var_value_block(field[n].var[varid].area_fid, field[n].var[varid].area_vid, start2, nread2, field[n].area);
    }
  }
  if (fg_dev_upload(d_raw, raw, ncell * (size_t)nz * upsz)) io_fatal("get_input_data");
  free(raw);

  slot = in_slot(field);
  slot->nz = nz; slot->order = halo ? 2 : 1; slot->varid = varid; slot->ncell = (long)ncell; slot->f_stride = (long)fstride;
  {
    double *d_src = (double *)fg_dev_alloc((ncell * (size_t)nz + 1) * sizeof(double), dev);
    if (!d_src) io_fatal("get_input_data");
    /* data[i] = raw[i]; `*= scale`, `+= offset` where != missing_value (:2097-2123) */
    if (fg_dev_widen(uptype, (long)(ncell * (size_t)nz), d_raw, field->var[varid].scale, field->var[varid].offset, missing_value, d_src))
      io_fatal("get_input_data");
    fg_dev_free(d_raw);
    if (!halo) slot->d_data = d_src;
    else {
      fg_c2l *c2l = c2l_for(ntiles, grid, dev);
      slot->d_data = (double *)fg_dev_alloc((fstride * (size_t)nz + 1) * sizeof(double), dev);
      slot->d_gx = (double *)fg_dev_alloc((ncell * (size_t)nz + 1) * sizeof(double), dev);
      slot->d_gy = (double *)fg_dev_alloc((ncell * (size_t)nz + 1) * sizeof(double), dev);
      if (field->var[varid].has_missing) slot->d_gm = (int *)fg_dev_alloc((ncell * (size_t)nz + 1) * sizeof(int), dev);
      if (!slot->d_data || !slot->d_gx || !slot->d_gy || (field->var[varid].has_missing && !slot->d_gm)) io_fatal("get_input_data");
      /* init_halo + copy onto the compute domain + update_halo (:2066-2084, 2127-2145, 2166-2181), then grad_c2l and the
       * gradient mask (:2183-2214) */
      if (fg_c2l_fill_halo(c2l, d_src, slot->d_data, nz) ||
          fg_c2l_gradient(c2l, slot->d_data, nz, field->var[varid].has_missing, missing_value, slot->d_gx, slot->d_gy, slot->d_gm) ||
          fg_c2l_sync(c2l)) io_fatal("get_input_data");
      fg_dev_free(d_src);
    }
  }
}

/* ------------------------------------------------------------------------------------------ write_field_data */
void write_field_data(int ntiles, Field_config *field, Grid_config *grid, int varid, int level_z, int level_n, int level_t)
{
  const double missing_value = field->var[varid].missing;
  const int ndim = field->var[varid].ndim;
  size_t start[8], nwrite[8];
  int nz = 1, pos = 0, i, n;
  if (ndim < 2 || ndim > 6) mpp_error("fregrid_util(write_field_data): bad ndim");
  if (level_z < 0) nz = field->var[varid].nz;
  for (i = 0; i < ndim; i++) { start[i] = 0; nwrite[i] = 1; }
  if (field->var[varid].has_taxis) start[pos++] = (size_t)level_t;
  if (field->var[varid].has_naxis) start[pos++] = (size_t)level_n;
  if (field->var[varid].has_zaxis) { if (level_z < 0) nwrite[pos++] = (size_t)nz; else start[pos++] = (size_t)level_z; }
  if (ndim != pos + 2) mpp_error("fregrid_util(write_field_data): mimstch between ndim and has_taxis/has_zaxis/has_naxis");

  for (n = 0; n < ntiles; n++) {
    const int nx = grid[n].nx, ny = grid[n].ny;
    const size_t cnt = (size_t)nx * ny * (size_t)nz;
    const int ftype = fg_type(field[n].var[varid].type);
    OutSlot *os = NULL;
    int k;
    for (k = 0; k < g_nout; k++) if (g_out[k].key == &field[n] && g_out[k].d_out) os = &g_out[k];
    nwrite[pos] = (size_t)ny; nwrite[pos + 1] = (size_t)nx;
    if (!ftype) mpp_error("fregrid_util(write_field_data): field type should be NC_SHORT, NC_FLOAT or NC_DOUBLE");
    if (!os || mpp_npes() != 1 || os->n * (long)os->nz != (long)cnt) {
      /* no device copy (or a banded run, whose levels mpp_global_field_double_3D has to assemble on the host) */
      if (os) { fg_dev_free(os->d_out); os->d_out = NULL; }
#ifdef FG_HAVE_CPU_FIELD_IO
      if (n == 0) { write_field_data_cpu(ntiles, field, grid, varid, level_z, level_n, level_t); return; }
      mpp_error("field_io(hip): only some output tiles hold a device copy of the level");
#else
      mpp_error("field_io(hip): write_field_data without a device copy of the level is served by the reference's function");
#endif
    }
    {
      /* `-= offset`, `/= scale` where != missing_value, the C cast to the file type (:2376-2406): on the device; the level comes
       * down in the file's type -- NC_FLOAT as float, widened (exactly) for mpp_put_var_value_block's nc_put_vara_double */
      const size_t osz = (ftype == FG_NC_SHORT) ? 2 : (ftype == FG_NC_DOUBLE ? 8 : 4);
      void *d_fin = fg_dev_alloc(cnt * osz + 8, io_device());
      char *h = (char *)malloc(cnt * osz + 8);
      if (!d_fin || !h) io_fatal("write_field_data");
      if (fg_dev_narrow(ftype, (long)cnt, os->d_out, field[n].var[varid].scale, field[n].var[varid].offset, missing_value, d_fin) ||
          fg_dev_download(h, d_fin, cnt * osz)) io_fatal("write_field_data");
      fg_dev_free(d_fin); fg_dev_free(os->d_out); os->d_out = NULL;
      if (ftype == FG_NC_FLOAT) {
        double *w = (double *)malloc(cnt * sizeof(double) + 8);
        size_t q;
        if (!w) mpp_error("field_io(hip): out of memory");
        for (q = 0; q < cnt; q++) w[q] = (double)((const float *)h)[q];
        mpp_put_var_value_block(*(field[n].fid), field[n].var[varid].vid, start, nwrite, w);
        free(w);
      } else
        mpp_put_var_value_block(*(field[n].fid), field[n].var[varid].vid, start, nwrite, h);
      free(h);
    }
  }
}
