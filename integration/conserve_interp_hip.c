/*
 * conserve_interp_hip.c -- drop-in replacement for tools/fregrid/conserve_interp.c (boundary B2):
 * the same two functions, the same arguments (tools/fregrid/conserve_interp.h:24-32), the work done by libfregrid_hip on an
 * MI355X.  Build it INSIDE the reference tree in place of conserve_interp.o and link -lfregrid_hip (INTEGRATION.md section 2;
 * the reference swaps the same object for its OpenACC port, tools/fregrid_gpu/Makefile.am:28-41).  Plain C99, no HIP headers:
 * device memory is handled through fg_dev_alloc / fg_dev_upload / fg_dev_download.
 *
 *   setup_conserve_interp       conserve_interp.c:42-503
 *       READ   (:62-126)   remap file -> Interp_config + a resident plan (fg_remap_read, fg_plan_set_xgrid)
 *       compute(:127-367)  fg_plan_create[_great_circle] per output tile over ALL input tiles at once, per-source-cell sums
 *                          accumulated in the reference's order over output tiles and ranks (:203-221), fg_plan_finalize
 *                          (centroid pass :319-358), Interp_config arrays malloc'ed as the reference leaves them
 *       WRITE  (:368-445)  mpp_gather_field_* to the root PE, fg_remap_write_interp
 *       CHECK_CONSERVE (:450-490) the area check, verbatim semantics
 *   do_scalar_conserve_interp   conserve_interp.c:507-910: every branch (missing, weight field, cell_methods = sum,
 *                          cell_measures, --target_grid, monotone limiter with its MIN / MAX exchange, conservation sums)
 *
 * The plan of output tile n is kept in a file-scope table keyed by the Interp_config pointer, because Interp_config
 * (globals.h:144-158) has no spare member; fregrid keeps one Interp_config array for the whole run.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "constant.h"
#include "globals.h"
#include "conserve_interp.h"
#include "mpp.h"
#include "mpp_domain.h"
#include "fregrid_hip.h"
#include "fregrid_hip_glue.h"
/* device-resident fields handed over by integration/field_io_hip.c; weak: this object also links and runs without it */
const FgDevField *fg_glue_input(const Field_config *field, int varid) __attribute__((weak));
int fg_glue_output_put(const Field_config *field_out_n, double *d_out, long n, int nz) __attribute__((weak));

#define MAXVAL (1.e20)

/* ------------------------------------------------------------------------------------------ plan table */
typedef struct { const Interp_config *key; fg_plan *plan; } PlanSlot;
static PlanSlot *g_slots = NULL;
static int g_nslots = 0;

static fg_plan *plan_of(const Interp_config *ic)
{
  int k;
  for (k = 0; k < g_nslots; k++) if (g_slots[k].key == ic) return g_slots[k].plan;
  return NULL;
}
static void plan_set(const Interp_config *ic, fg_plan *pl)
{
  int k;
  for (k = 0; k < g_nslots; k++)
    if (g_slots[k].key == ic) { if (g_slots[k].plan) fg_plan_destroy(g_slots[k].plan); g_slots[k].plan = pl; return; }
  g_slots = (PlanSlot *)realloc(g_slots, (size_t)(g_nslots + 1) * sizeof(PlanSlot));
  if (!g_slots) mpp_error("conserve_interp(hip): out of memory");
  g_slots[g_nslots].key = ic; g_slots[g_nslots].plan = pl; g_nslots++;
}

static int hip_device(void)
{
  const char *e = getenv("FREGRID_HIP_DEVICE");        /* one process per GPU: the launcher sets it per rank */
  return e ? atoi(e) : 0;
}
static void hip_fatal(const char *where)
{
  char msg[768];
  snprintf(msg, sizeof msg, "%s: %s", where, fg_last_error());
  mpp_error(msg);
}

/* ------------------------------------------------------------------------------------------ setup */
void setup_conserve_interp(int ntiles_in, const Grid_config *grid_in, int ntiles_out,
                           Grid_config *grid_out, Interp_config *interp, unsigned int opcode)
{
  const int order = (opcode & CONSERVE_ORDER2) ? 2 : 1;
  const int dev = hip_device();
  int n, m;
  size_t i;
  int *nx_in = (int *)malloc((size_t)ntiles_in * sizeof(int)), *ny_in = (int *)malloc((size_t)ntiles_in * sizeof(int));
  long ncells_in = 0;
  for (m = 0; m < ntiles_in; m++) { nx_in[m] = grid_in[m].nx; ny_in[m] = grid_in[m].ny; ncells_in += (long)nx_in[m] * ny_in[m]; }

  if (opcode & READ) {                                                           /* conserve_interp.c:62-126 */
    for (n = 0; n < ntiles_out; n++) {
      if (!interp[n].file_exist) continue;
      const long nx = fg_remap_read_size(interp[n].remap_file);                  /* read_mosaic_xgrid_size */
      if (nx < 0) mpp_error((char *)fg_remap_last_error());
      int *t = (int *)malloc((size_t)(nx + 1) * sizeof(int)), *i1 = (int *)malloc((size_t)(nx + 1) * sizeof(int));
      int *j1 = (int *)malloc((size_t)(nx + 1) * sizeof(int)), *i2 = (int *)malloc((size_t)(nx + 1) * sizeof(int));
      int *j2 = (int *)malloc((size_t)(nx + 1) * sizeof(int));
      double *a = (double *)malloc((size_t)(nx + 1) * sizeof(double));
      double *di = (order == 2) ? (double *)malloc((size_t)(nx + 1) * sizeof(double)) : NULL;
      double *dj = (order == 2) ? (double *)malloc((size_t)(nx + 1) * sizeof(double)) : NULL;
      /* 0-based indices, tile1 - 1 and the area * garea rescale (:86) are applied by fg_remap_read */
      if (fg_remap_read(interp[n].remap_file, order, nx, t, i1, j1, i2, j2, a, di, dj)) mpp_error((char *)fg_remap_last_error());
      /* distribute the exchange grid on each pe according to target grid index (:91-97) */
      size_t keep = 0;
      for (i = 0; i < (size_t)nx; i++)
        if (i2[i] <= grid_out[n].iec && i2[i] >= grid_out[n].isc && j2[i] <= grid_out[n].jec && j2[i] >= grid_out[n].jsc) keep++;
      interp[n].nxgrid = keep;
      interp[n].i_in = (int *)malloc((keep + 1) * sizeof(int)); interp[n].j_in = (int *)malloc((keep + 1) * sizeof(int));
      interp[n].i_out = (int *)malloc((keep + 1) * sizeof(int)); interp[n].j_out = (int *)malloc((keep + 1) * sizeof(int));
      interp[n].t_in = (int *)malloc((keep + 1) * sizeof(int)); interp[n].area = (double *)malloc((keep + 1) * sizeof(double));
      if (order == 2) { interp[n].di_in = (double *)malloc((keep + 1) * sizeof(double)); interp[n].dj_in = (double *)malloc((keep + 1) * sizeof(double)); }
      keep = 0;
      for (i = 0; i < (size_t)nx; i++) {
        if (!(i2[i] <= grid_out[n].iec && i2[i] >= grid_out[n].isc && j2[i] <= grid_out[n].jec && j2[i] >= grid_out[n].jsc)) continue;
        interp[n].i_in[keep] = i1[i]; interp[n].j_in[keep] = j1[i]; interp[n].t_in[keep] = t[i];
        interp[n].i_out[keep] = i2[i] - grid_out[n].isc; interp[n].j_out[keep] = j2[i] - grid_out[n].jsc;
        interp[n].area[keep] = a[i];
        if (order == 2) { interp[n].di_in[keep] = di[i]; interp[n].dj_in[keep] = dj[i]; }
        keep++;
      }
      free(t); free(i1); free(j1); free(i2); free(j2); free(a); free(di); free(dj);
      fg_plan *pl = NULL;
      if (fg_plan_create_empty(order, ntiles_in, nx_in, ny_in, grid_out[n].nxc, grid_out[n].nyc, dev, &pl)) hip_fatal("setup_conserve_interp");
      if (fg_plan_set_xgrid(pl, (long)interp[n].nxgrid, interp[n].t_in, interp[n].i_in, interp[n].j_in, interp[n].i_out, interp[n].j_out,
                            interp[n].area, interp[n].di_in, interp[n].dj_in)) hip_fatal("setup_conserve_interp");
      plan_set(&interp[n], pl);
    }
    if (mpp_pe() == mpp_root_pe()) printf("NOTE: Finish reading index and weight for conservative interpolation from file.\n");
  } else {                                                                        /* :127-367 */
    const double **lon = (const double **)malloc((size_t)ntiles_in * sizeof(double *));
    const double **lat = (const double **)malloc((size_t)ntiles_in * sizeof(double *));
    fg_plan **plans = (fg_plan **)calloc((size_t)ntiles_out, sizeof(fg_plan *));
    for (m = 0; m < ntiles_in; m++) { lon[m] = grid_in[m].lonc; lat[m] = grid_in[m].latc; }
    if ((opcode & GREAT_CIRCLE) && order != 1)
      mpp_error("fregrid: when clip_method is 'conserve_great_circle', interp_method must be 'conserve_order1'");   /* fregrid.c:763 */
    /* a rank of fregrid_parallel meets only the source cells near its band: the search skips the others when it builds its
     * per-cell records (the counterpart of the row trim, :169-184; the great-circle search culls by bounding caps) */
    fg_set_search_cull(mpp_npes() > 1);
    /* first order takes no sums from anybody, second order with one output tile on one rank finds them in its own plan: the search
     * then queues its finalize work itself (no host round trip in between); fg_plan_finalize(plan, NULL) below returns at once */
    const int fused = (order == 1 && !(opcode & GREAT_CIRCLE)) || (order == 2 && ntiles_out == 1 && mpp_npes() == 1);
    fg_set_search_finalize(fused);
    for (n = 0; n < ntiles_out; n++) {
      /* this rank's band of output tile n: nxc x nyc cells, corner arrays lonc / latc (get_output_grid_by_size,
       * fregrid_util.c:645-654).  All input tiles are searched in one call; the reference's row trim (:169-184) is an
       * optimisation of its brute-force scan and has no counterpart. */
      long nx;
      if (opcode & GREAT_CIRCLE)
        nx = fg_plan_create_great_circle(ntiles_in, nx_in, ny_in, lon, lat, NULL, grid_out[n].nxc, grid_out[n].nyc,
                                         grid_out[n].lonc, grid_out[n].latc, dev, &plans[n]);
      else
        nx = fg_plan_create(order, ntiles_in, nx_in, ny_in, lon, lat, NULL, grid_out[n].nxc, grid_out[n].nyc,
                            grid_out[n].lonc, grid_out[n].latc, dev, &plans[n]);
      if (nx < 0) hip_fatal("setup_conserve_interp");
    }
    fg_set_search_cull(0);
    fg_set_search_finalize(0);
    if (order == 2 && fused) {
      if (fg_plan_finalize(plans[0], NULL)) hip_fatal("setup_conserve_interp");
    } else if (order == 2) {
      /* per-source-cell (area, clon, clat): :203-221 gathers the exchange cells of every rank and adds them to the accumulators
       * one by one, output tile after output tile, "for the purpose of bitwise reproducing".  One running total handed from
       * plan to plan (fg_plan_accumulate_cell_sums continues from the values it finds) does the same additions in the same
       * order; across ranks only the source cells that have exchange cells on more than one rank need the hand-over. */
      const size_t nsum = 3 * (size_t)ncells_in;
      double *tot = (double *)calloc(nsum, sizeof(double));
      double *d_tot = (double *)fg_dev_alloc(nsum * sizeof(double), dev);
      if (!tot || !d_tot || fg_dev_upload(d_tot, tot, nsum * sizeof(double))) hip_fatal("setup_conserve_interp");
      for (n = 0; n < ntiles_out; n++)
        if (fg_plan_nxgrid(plans[n]) > 0 && fg_plan_accumulate_cell_sums(plans[n], d_tot, NULL, 0)) hip_fatal("setup_conserve_interp");
      if (mpp_npes() > 1) {
        const int npes = mpp_npes(), pe = mpp_pe() - mpp_root_pe();
        int *cnt = (int *)calloc((size_t)ncells_in, sizeof(int)), *sh, nsh = 0, r, c, k;
        if (nsum > 0x7fffffff) mpp_error("setup_conserve_interp(hip): too many source cells for one mpp_sum_double");
        if (!cnt || fg_dev_download(tot, d_tot, nsum * sizeof(double))) hip_fatal("setup_conserve_interp");
        for (i = 0; i < (size_t)ncells_in; i++) cnt[i] = tot[i] != 0;
        mpp_sum_int((int)ncells_in, cnt);
        for (i = 0; i < (size_t)ncells_in; i++) if (cnt[i] > 1) nsh++;
        sh = (int *)malloc(((size_t)nsh + 1) * sizeof(int));
        for (i = 0, nsh = 0; i < (size_t)ncells_in; i++) if (cnt[i] > 1) sh[nsh++] = (int)i;
        if (nsh > 0) {                      /* the shared cells: rank after rank within an output tile, tile after tile */
          double *run = (double *)calloc(3 * (size_t)nsh, sizeof(double));
          int *d_sh = (int *)fg_dev_alloc((size_t)nsh * sizeof(int), dev);
          double *d_run = (double *)fg_dev_alloc(3 * (size_t)nsh * sizeof(double), dev);
          double *d_scr = (double *)fg_dev_alloc(nsum * sizeof(double), dev);
          if (!run || !d_sh || !d_run || !d_scr || fg_dev_upload(d_sh, sh, (size_t)nsh * sizeof(int))) hip_fatal("setup_conserve_interp");
          for (n = 0; n < ntiles_out; n++)
            for (r = 0; r < npes; r++) {
              if (r == pe && fg_plan_nxgrid(plans[n]) > 0) {
                if (fg_dev_upload(d_run, run, 3 * (size_t)nsh * sizeof(double))) hip_fatal("setup_conserve_interp");
                for (c = 0; c < 3; c++) if (fg_dev_scatter_f64(d_scr + (size_t)c * ncells_in, d_run + (size_t)c * nsh, d_sh, nsh)) hip_fatal("setup_conserve_interp");
                if (fg_plan_accumulate_cell_sums(plans[n], d_scr, d_sh, nsh)) hip_fatal("setup_conserve_interp");
                for (c = 0; c < 3; c++) if (fg_dev_gather_f64(d_run + (size_t)c * nsh, d_scr + (size_t)c * ncells_in, d_sh, nsh)) hip_fatal("setup_conserve_interp");
                if (fg_dev_download(run, d_run, 3 * (size_t)nsh * sizeof(double))) hip_fatal("setup_conserve_interp");
              } else if (r != pe)
                for (k = 0; k < 3 * nsh; k++) run[k] = 0.0;
              mpp_sum_double(3 * nsh, run);                  /* = a broadcast from rank r: the others contribute zeros */
            }
          for (c = 0; c < 3; c++) for (k = 0; k < nsh; k++) tot[(size_t)c * ncells_in + sh[k]] = 0.0;
          mpp_sum_double((int)nsum, tot);                    /* every other cell is complete on its one rank */
          for (c = 0; c < 3; c++) for (k = 0; k < nsh; k++) tot[(size_t)c * ncells_in + sh[k]] = run[(size_t)c * nsh + k];
          fg_dev_free(d_sh); fg_dev_free(d_run); fg_dev_free(d_scr); free(run);
        } else
          mpp_sum_double((int)nsum, tot);
        if (fg_dev_upload(d_tot, tot, nsum * sizeof(double))) hip_fatal("setup_conserve_interp");
        free(cnt); free(sh);
      }
      for (n = 0; n < ntiles_out; n++) if (fg_plan_finalize(plans[n], d_tot)) hip_fatal("setup_conserve_interp");
      fg_dev_free(d_tot); free(tot);
    } else
      for (n = 0; n < ntiles_out; n++) if (fg_plan_finalize(plans[n], NULL)) hip_fatal("setup_conserve_interp");
    for (n = 0; n < ntiles_out; n++) {
      const size_t nx = (size_t)fg_plan_nxgrid(plans[n]);
      interp[n].nxgrid = nx;
      interp[n].i_in = (int *)malloc((nx + 1) * sizeof(int)); interp[n].j_in = (int *)malloc((nx + 1) * sizeof(int));
      interp[n].i_out = (int *)malloc((nx + 1) * sizeof(int)); interp[n].j_out = (int *)malloc((nx + 1) * sizeof(int));
      interp[n].t_in = (int *)malloc((nx + 1) * sizeof(int)); interp[n].area = (double *)malloc((nx + 1) * sizeof(double));
      if (order == 2) { interp[n].di_in = (double *)malloc((nx + 1) * sizeof(double)); interp[n].dj_in = (double *)malloc((nx + 1) * sizeof(double)); }
      if (fg_plan_get_xgrid(plans[n], interp[n].t_in, interp[n].i_in, interp[n].j_in, interp[n].i_out, interp[n].j_out, interp[n].area,
                            order == 2 ? interp[n].di_in : NULL, order == 2 ? interp[n].dj_in : NULL)) hip_fatal("setup_conserve_interp");
      plan_set(&interp[n], plans[n]);
    }
    free(lon); free(lat); free(plans);

    if (opcode & WRITE) {                                                         /* :368-445 */
      for (n = 0; n < ntiles_out; n++) {
        int nxgrid = (int)interp[n].nxgrid;
        mpp_sum_int(1, &nxgrid);
        if (nxgrid <= 0) continue;
        const int nl = (int)interp[n].nxgrid;
        int *g_t = (int *)malloc((size_t)nxgrid * sizeof(int)), *g_i1 = (int *)malloc((size_t)nxgrid * sizeof(int));
        int *g_j1 = (int *)malloc((size_t)nxgrid * sizeof(int)), *g_i2 = (int *)malloc((size_t)nxgrid * sizeof(int));
        int *g_j2 = (int *)malloc((size_t)nxgrid * sizeof(int));
        int *l_i2 = (int *)malloc((size_t)(nl + 1) * sizeof(int)), *l_j2 = (int *)malloc((size_t)(nl + 1) * sizeof(int));
        double *g_a = (double *)malloc((size_t)nxgrid * sizeof(double));
        double *g_di = (order == 2) ? (double *)malloc((size_t)nxgrid * sizeof(double)) : NULL;
        double *g_dj = (order == 2) ? (double *)malloc((size_t)nxgrid * sizeof(double)) : NULL;
        /* global output indices before the gather (:407,:416), 0-based: fg_remap_write_interp adds the file's +1 */
        for (i = 0; i < (size_t)nl; i++) { l_i2[i] = interp[n].i_out[i] + grid_out[n].isc; l_j2[i] = interp[n].j_out[i] + grid_out[n].jsc; }
        mpp_gather_field_int(nl, interp[n].t_in, g_t);
        mpp_gather_field_int(nl, interp[n].i_in, g_i1);
        mpp_gather_field_int(nl, interp[n].j_in, g_j1);
        mpp_gather_field_int(nl, l_i2, g_i2);
        mpp_gather_field_int(nl, l_j2, g_j2);
        mpp_gather_field_double(nl, interp[n].area, g_a);
        if (order == 2) { mpp_gather_field_double(nl, interp[n].di_in, g_di); mpp_gather_field_double(nl, interp[n].dj_in, g_dj); }
        if (mpp_pe() == mpp_root_pe())
          if (fg_remap_write_interp(interp[n].remap_file, order, nxgrid, g_t, g_i1, g_j1, g_i2, g_j2, g_a, g_di, g_dj, 0, 0))
            mpp_error((char *)fg_remap_last_error());
        free(g_t); free(g_i1); free(g_j1); free(g_i2); free(g_j2); free(l_i2); free(l_j2); free(g_a); free(g_di); free(g_dj);
      }
    }
    if (mpp_pe() == mpp_root_pe()) printf("NOTE: done calculating index and weight for conservative interpolation\n");
  }

  /* check the input area match exchange grid area (:450-490) */
  if (opcode & CHECK_CONSERVE) {
    const int nx1 = grid_out[0].nxc, ny1 = grid_out[0].nyc;
    double *area2 = (double *)malloc((size_t)nx1 * ny1 * sizeof(double));
    for (n = 0; n < ntiles_out; n++) {
      int ii, ix, jx, max_i = 0, max_j = 0;
      double max_ratio = 0, ratio_change;
      for (ii = 0; ii < nx1 * ny1; ii++) area2[ii] = 0;
      for (i = 0; i < interp[n].nxgrid; i++) {
        ii = interp[n].j_out[i] * nx1 + interp[n].i_out[i];
        area2[ii] += interp[n].area[i];
      }
      for (jx = 0; jx < ny1; jx++) for (ix = 0; ix < nx1; ix++) {
        ii = jx * nx1 + ix;
        ratio_change = fabs(grid_out[n].cell_area[ii] - area2[ii]) / grid_out[n].cell_area[ii];
        if (ratio_change > max_ratio) { max_ratio = ratio_change; max_i = ix; max_j = jx; }
        if (ratio_change > 1.e-4)
          printf("(i,j)=(%d,%d), change = %g, area1=%g, area2=%g\n", ix, jx, ratio_change, grid_out[n].cell_area[ii], area2[ii]);
      }
      ii = max_j * nx1 + max_i;
      printf("The maximum ratio change at (%d,%d) = %g, area1=%g, area2=%g\n", max_i, max_j, max_ratio, grid_out[n].cell_area[ii], area2[ii]);
    }
    free(area2);
  }
  free(nx_in); free(ny_in);
}

/* ------------------------------------------------------------------------------------------ sweep */
/* tiles back to back: per level `per_tile_elems(m)` doubles of tile m */
static double *stage_tiles(int ntiles, int nz, const size_t *elems, double *const *src, int dev, size_t *total_out)
{
  size_t tot = 0, off = 0;
  int m, k;
  for (m = 0; m < ntiles; m++) tot += elems[m];
  double *d = (double *)fg_dev_alloc((tot * (size_t)nz + 1) * sizeof(double), dev);
  if (!d) hip_fatal("do_scalar_conserve_interp");
  for (k = 0; k < nz; k++) {
    off = 0;
    for (m = 0; m < ntiles; m++) {
      if (fg_dev_upload(d + (size_t)k * tot + off, src[m] + (size_t)k * elems[m], elems[m] * sizeof(double))) hip_fatal("do_scalar_conserve_interp");
      off += elems[m];
    }
  }
  if (total_out) *total_out = tot;
  return d;
}

void do_scalar_conserve_interp(Interp_config *interp, int varid, int ntiles_in, const Grid_config *grid_in,
                               int ntiles_out, const Grid_config *grid_out, const Field_config *field_in,
                               Field_config *field_out, unsigned int opcode, int nz)
{
  const int dev = hip_device();
  const int interp_method = field_in->var[varid].interp_method;
  const int order = (interp_method == CONSERVE_ORDER2) ? 2 : 1;
  const int halo = (order == 2) ? 1 : 0;
  const int monotonic = (order == 2) ? (int)(opcode & MONOTONIC) : 0;               /* :525-531 */
  const double area_missing = field_in->var[varid].area_missing;
  const int has_missing = field_in->var[varid].has_missing;
  const int weight_exist = grid_in[0].weight_exist;
  const int cell_measures = field_in->var[varid].cell_measures;
  const int cell_methods = field_in->var[varid].cell_methods;
  int target_grid = (int)(opcode & TARGET);
  double missing = -MAXVAL, gsum_out = 0;
  int m, n;
  if (field_in->var[varid].use_volume) target_grid = 0;                             /* :536 */
  if (has_missing) missing = field_in->var[varid].missing;
  if (nz > 1 && has_missing) mpp_error("conserve_interp: has_missing should be false when nz > 1");
  if (nz > 1 && cell_measures) mpp_error("conserve_interp: cell_measures should be false when nz > 1");
  if (nz > 1 && cell_methods == CELL_METHODS_SUM) mpp_error("conserve_interp: cell_methods should not be sum when nz > 1");

  /* --- the source side goes up once per call, tiles back to back */
  size_t *e_data = (size_t *)malloc((size_t)ntiles_in * sizeof(size_t)), *e_cell = (size_t *)malloc((size_t)ntiles_in * sizeof(size_t));
  double **p = (double **)malloc((size_t)ntiles_in * sizeof(double *));
  size_t ncell = 0;
  for (n = 0; n < ntiles_in; n++) {
    e_cell[n] = (size_t)grid_in[n].nx * grid_in[n].ny;
    e_data[n] = (size_t)(grid_in[n].nx + 2 * halo) * (grid_in[n].ny + 2 * halo);
    ncell += e_cell[n];
  }
  /* get_input_data of field_io_hip.c leaves the level(s) on the device: halo filled, gradients and gradient mask made there */
  const FgDevField *devf = fg_glue_input ? fg_glue_input(field_in, varid) : NULL;
  if (devf && (devf->nz != nz || devf->order != order || devf->ncell != (long)ncell)) devf = NULL;
  for (n = 0; n < ntiles_in; n++) p[n] = field_in[n].data;
  double *d_data = devf ? devf->d_data : stage_tiles(ntiles_in, nz, e_data, p, dev, NULL);
  double *d_gx = NULL, *d_gy = NULL, *d_w = NULL, *d_fa = NULL, *d_ca = NULL;
  int *d_gm = NULL;
  if (devf) {
    d_gx = devf->d_gx; d_gy = devf->d_gy; d_gm = devf->d_gm;
    if (opcode & CHECK_CONSERVE) {                       /* the flux sum below reads the host arrays */
      size_t off = 0;
      int k;
      for (n = 0; n < ntiles_in; n++) {
        for (k = 0; k < nz; k++)
          if (fg_dev_download(field_in[n].data + (size_t)k * e_data[n], d_data + (size_t)k * devf->f_stride + off, e_data[n] * sizeof(double))) hip_fatal("do_scalar_conserve_interp");
        off += e_data[n];
      }
    }
  } else if (order == 2) {
    for (n = 0; n < ntiles_in; n++) p[n] = field_in[n].grad_x;
    d_gx = stage_tiles(ntiles_in, nz, e_cell, p, dev, NULL);
    for (n = 0; n < ntiles_in; n++) p[n] = field_in[n].grad_y;
    d_gy = stage_tiles(ntiles_in, nz, e_cell, p, dev, NULL);
    if (has_missing) {                                                              /* grad_mask, fregrid_util.c:2203-2216 */
      size_t off = 0;
      d_gm = (int *)fg_dev_alloc((ncell + 1) * sizeof(int), dev);
      if (!d_gm) hip_fatal("do_scalar_conserve_interp");
      for (n = 0; n < ntiles_in; n++) { if (fg_dev_upload(d_gm + off, field_in[n].grad_mask, e_cell[n] * sizeof(int))) hip_fatal("do_scalar_conserve_interp"); off += e_cell[n]; }
    }
  }
  const int extended = weight_exist || cell_measures || cell_methods == CELL_METHODS_SUM || target_grid || monotonic;
  if (weight_exist) { for (n = 0; n < ntiles_in; n++) p[n] = grid_in[n].weight; d_w = stage_tiles(ntiles_in, 1, e_cell, p, dev, NULL); }
  if (cell_measures) { for (n = 0; n < ntiles_in; n++) p[n] = field_in[n].area; d_fa = stage_tiles(ntiles_in, 1, e_cell, p, dev, NULL); }
  if (cell_measures || cell_methods == CELL_METHODS_SUM) { for (n = 0; n < ntiles_in; n++) p[n] = grid_in[n].cell_area; d_ca = stage_tiles(ntiles_in, 1, e_cell, p, dev, NULL); }

  for (m = 0; m < ntiles_out; m++) {
    fg_plan *pl = plan_of(&interp[m]);
    if (!pl) mpp_error("do_scalar_conserve_interp(hip): setup_conserve_interp has not built a plan for this Interp_config");
    const size_t nout = (size_t)grid_out[m].nxc * grid_out[m].nyc;
    double *d_out = (double *)fg_dev_alloc((nout * (size_t)nz + 1) * sizeof(double), dev);
    double *d_cao = NULL, g = 0;
    double *gp = (opcode & CHECK_CONSERVE) ? &g : NULL;
    if (!d_out) hip_fatal("do_scalar_conserve_interp");
    if (!extended) {
      if (fg_plan_apply(pl, d_data, d_gx, d_gy, d_gm, has_missing, missing, nz, d_out, gp)) hip_fatal("do_scalar_conserve_interp");
    } else {
      fg_apply_opts o;
      memset(&o, 0, sizeof o);
      o.has_missing = has_missing; o.missing = missing; o.weight = d_w; o.cell_methods_sum = (cell_methods == CELL_METHODS_SUM);
      o.field_area = d_fa; o.area_missing = area_missing; o.cell_area_in = d_ca; o.monotonic = monotonic;
      if (target_grid) {
        d_cao = (double *)fg_dev_alloc((nout + 1) * sizeof(double), dev);
        if (!d_cao || fg_dev_upload(d_cao, grid_out[m].cell_area, nout * sizeof(double))) hip_fatal("do_scalar_conserve_interp");
        o.cell_area_out = d_cao;
      }
      if (monotonic && mpp_npes() > 1) {
        /* the limiter needs the extremes over ALL exchange cells of a source cell: mpp_min_double / mpp_max_double, :672-677 */
        double *d_min, *d_max;
        double *h_min = (double *)malloc(ncell * sizeof(double)), *h_max = (double *)malloc(ncell * sizeof(double));
        if (fg_plan_mono_begin(pl, &o, d_data, d_gx, d_gy, d_gm) || fg_plan_mono_minmax_dev(pl, &d_min, &d_max) || fg_plan_sync(pl)) hip_fatal("do_scalar_conserve_interp");
        if (fg_dev_download(h_min, d_min, ncell * sizeof(double)) || fg_dev_download(h_max, d_max, ncell * sizeof(double))) hip_fatal("do_scalar_conserve_interp");
        mpp_min_double((int)ncell, h_min); mpp_max_double((int)ncell, h_max);
        if (fg_dev_upload(d_min, h_min, ncell * sizeof(double)) || fg_dev_upload(d_max, h_max, ncell * sizeof(double))) hip_fatal("do_scalar_conserve_interp");
        free(h_min); free(h_max);
        if (fg_plan_mono_end(pl, &o, d_data, d_out, gp)) hip_fatal("do_scalar_conserve_interp");
      } else if (fg_plan_apply_ex(pl, &o, d_data, d_gx, d_gy, d_gm, nz, d_out, gp)) hip_fatal("do_scalar_conserve_interp");
    }
    if (fg_plan_sync(pl) || fg_dev_download(field_out[m].data, d_out, nout * (size_t)nz * sizeof(double))) hip_fatal("do_scalar_conserve_interp");
    gsum_out += g;
    /* write_field_data of field_io_hip.c narrows the level on the device and downloads it in the file's type */
    if (!(fg_glue_output_put && fg_glue_output_put(&field_out[m], d_out, (long)nout, nz))) fg_dev_free(d_out);
    fg_dev_free(d_cao);
  }
  if (!devf) { fg_dev_free(d_data); fg_dev_free(d_gx); fg_dev_free(d_gy); fg_dev_free(d_gm); }
  fg_dev_free(d_w); fg_dev_free(d_fa); fg_dev_free(d_ca);
  free(e_data); free(e_cell); free(p);

  /* conservation check if needed (:874-907) */
  if (opcode & CHECK_CONSERVE) {
    double gsum_in = 0, dd;
    int i, j, k;
    for (n = 0; n < ntiles_in; n++) {
      const int nx1 = grid_in[n].nx, ny1 = grid_in[n].ny;
      if (cell_measures) {
        for (j = 0; j < ny1; j++) for (i = 0; i < nx1; i++) {
          dd = field_in[n].data[(j + halo) * (nx1 + 2 * halo) + i + halo];
          if (dd != missing) gsum_in += dd * field_in[n].area[j * nx1 + i];
        }
      } else if (cell_methods == CELL_METHODS_SUM) {
        for (j = 0; j < ny1; j++) for (i = 0; i < nx1; i++) {
          dd = field_in[n].data[(j + halo) * (nx1 + 2 * halo) + i + halo];
          if (dd != missing) gsum_in += dd;
        }
      } else {
        for (k = 0; k < nz; k++) for (j = 0; j < ny1; j++) for (i = 0; i < nx1; i++) {
          dd = field_in[n].data[k * (nx1 + 2 * halo) * (ny1 + 2 * halo) + (j + halo) * (nx1 + 2 * halo) + i + halo];
          if (dd != missing) gsum_in += dd * grid_in[n].cell_area[j * nx1 + i];
        }
      }
    }
    mpp_sum_double(1, &gsum_out);
    if (mpp_pe() == mpp_root_pe())
      printf("the flux(data*area) sum of %s: input = %g, output = %g, diff = %g. \n", field_in->var[varid].name, gsum_in, gsum_out, gsum_out - gsum_in);
  }
}
