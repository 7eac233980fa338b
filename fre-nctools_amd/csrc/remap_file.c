/*
 * remap_file.c -- the conservative remap file ("--remap_file") without libnetcdf.
 *
 * The remap file is the path's only persistent state: written after the exchange-grid search
 * (tools/fregrid/conserve_interp.c:368-445) and read back by the READ branch (:62-126 through
 * tools/libfrencutils/read_mosaic.c:352-558).  This image has no netCDF library, so the classic netCDF
 * container is written and parsed directly (the format is public: magic "CDF" + version, big-endian header
 * with dimension / attribute / variable lists, then the variable data, each padded to 4 bytes):
 *
 *   fg_remap_write       writes version 2 (64-bit offset), the layout the reference produces when its input
 *                        files are 64-bit-offset files (mpp_io.c:150-175):
 *                          dims   string=255, ncells, two=2
 *                          vars   tile1(ncells) int                "tile_number_in_mosaic1"
 *                                 tile1_cell(ncells,two) int       "parent_cell_indices_in_mosaic1"   (1-based i,j)
 *                                 tile2_cell(ncells,two) int       "parent_cell_indices_in_mosaic2"
 *                                 xgrid_area(ncells) double        "exchange_grid_area", units "m2"
 *                                 tile1_distance(ncells,two) double (order 2) "distance_from_parent1_cell_centroid"
 *   fg_remap_read_size / fg_remap_read   accept versions 1, 2 and 5 (CDF-5), any variable order, fixed-size variables,
 *                        and apply the reference's conversions: 1-based -> 0-based, tile1 - 1, and the
 *                        area / garea * garea round trip of read_mosaic.c:432 + conserve_interp.c:86.
 *   netCDF-4 (HDF5) files are rejected with a clear message.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fregrid_hip.h"

#define NC_DIMENSION 0x0A
#define NC_VARIABLE  0x0B
#define NC_ATTRIBUTE 0x0C
#define T_CHAR 2
#define T_INT 4
#define T_DOUBLE 6
#define STRING_LEN 255                     /* constant.h:24 */
#define RF_RADIUS 6371000.0
#define RF_PI 3.14159265358979323846

static char g_rf_err[256];
const char *fg_remap_last_error(void) { return g_rf_err; }
static int rf_fail(int code, const char *msg) { snprintf(g_rf_err, sizeof g_rf_err, "%s", msg); return code; }

/* ---------------------------------------------------------------- big-endian buffer writer */
typedef struct { unsigned char *p; size_t n, cap; } Buf;
static int b_need(Buf *b, size_t k)
{
  if (b->n + k <= b->cap) return 0;
  size_t c = b->cap ? b->cap * 2 : 1024;
  while (c < b->n + k) c *= 2;
  unsigned char *q = (unsigned char *)realloc(b->p, c);
  if (!q) return -1;
  b->p = q; b->cap = c;
  return 0;
}
static void b_u32(Buf *b, uint32_t v) { if (b_need(b, 4)) return; b->p[b->n++] = v >> 24; b->p[b->n++] = v >> 16; b->p[b->n++] = v >> 8; b->p[b->n++] = v; }
static void b_u64(Buf *b, uint64_t v) { b_u32(b, (uint32_t)(v >> 32)); b_u32(b, (uint32_t)v); }
static void b_name(Buf *b, const char *s)
{
  size_t l = strlen(s), pad = (4 - l % 4) % 4;
  b_u32(b, (uint32_t)l);
  if (b_need(b, l + pad)) return;
  memcpy(b->p + b->n, s, l); b->n += l;
  for (size_t k = 0; k < pad; k++) b->p[b->n++] = 0;
}
static void b_text_att(Buf *b, const char *name, const char *val)
{
  b_name(b, name);
  b_u32(b, T_CHAR);
  b_name(b, val);          /* nelems + chars + padding has the same shape as a name */
}

typedef struct { const char *name; int ndims; int dimid[2]; int natt; const char *att[4]; int type; uint64_t vsize, begin; } VarDef;

static size_t var_header_size(const VarDef *v)
{
  Buf t = {0};
  b_name(&t, v->name); b_u32(&t, v->ndims);
  for (int d = 0; d < v->ndims; d++) b_u32(&t, v->dimid[d]);
  if (v->natt) { b_u32(&t, NC_ATTRIBUTE); b_u32(&t, v->natt); for (int a = 0; a < v->natt; a++) b_text_att(&t, v->att[2 * a], v->att[2 * a + 1]); }
  else { b_u32(&t, 0); b_u32(&t, 0); }
  b_u32(&t, v->type); b_u32(&t, 0); b_u64(&t, 0);
  size_t n = t.n; free(t.p);
  return n;
}

int fg_remap_write(const char *path, int order, long ncells, const int *tile1, const int *tile1_cell,
                   const int *tile2_cell, const double *xgrid_area, const double *tile1_distance)
{
  if (!path || ncells < 0 || (ncells > 0 && (!tile1 || !tile1_cell || !tile2_cell || !xgrid_area))) return rf_fail(FG_ERR_ARG, "fg_remap_write: null argument");
  if (order == 2 && ncells > 0 && !tile1_distance) return rf_fail(FG_ERR_ARG, "fg_remap_write: order 2 needs tile1_distance");
  if (ncells > 0x7fffffffL / 16) return rf_fail(FG_ERR_ARG, "fg_remap_write: too many cells for a classic-format variable");
  VarDef v[5] = {
    {"tile1", 1, {1, 0}, 1, {"standard_name", "tile_number_in_mosaic1"}, T_INT, 0, 0},
    {"tile1_cell", 2, {1, 2}, 1, {"standard_name", "parent_cell_indices_in_mosaic1"}, T_INT, 0, 0},
    {"tile2_cell", 2, {1, 2}, 1, {"standard_name", "parent_cell_indices_in_mosaic2"}, T_INT, 0, 0},
    {"xgrid_area", 1, {1, 0}, 2, {"standard_name", "exchange_grid_area", "units", "m2"}, T_DOUBLE, 0, 0},
    {"tile1_distance", 2, {1, 2}, 1, {"standard_name", "distance_from_parent1_cell_centroid"}, T_DOUBLE, 0, 0},
  };
  const int nvar = (order == 2) ? 5 : 4;
  v[0].vsize = 4ull * ncells; v[1].vsize = 8ull * ncells; v[2].vsize = 8ull * ncells; v[3].vsize = 8ull * ncells; v[4].vsize = 16ull * ncells;
  Buf h = {0};
  /* header: magic, numrecs, dims */
  if (b_need(&h, 4)) return rf_fail(FG_ERR_HIP, "out of memory");
  memcpy(h.p, "CDF\x02", 4); h.n = 4;
  b_u32(&h, 0);
  b_u32(&h, NC_DIMENSION); b_u32(&h, 3);
  b_name(&h, "string"); b_u32(&h, STRING_LEN);
  b_name(&h, "ncells"); b_u32(&h, (uint32_t)ncells);
  b_name(&h, "two"); b_u32(&h, 2);
  b_u32(&h, 0); b_u32(&h, 0);                       /* no global attributes */
  size_t hdr = h.n + 8;
  for (int k = 0; k < nvar; k++) hdr += var_header_size(&v[k]);
  uint64_t off = hdr;
  for (int k = 0; k < nvar; k++) { v[k].begin = off; off += (v[k].vsize + 3) / 4 * 4; }
  b_u32(&h, NC_VARIABLE); b_u32(&h, nvar);
  for (int k = 0; k < nvar; k++) {
    b_name(&h, v[k].name); b_u32(&h, v[k].ndims);
    for (int d = 0; d < v[k].ndims; d++) b_u32(&h, v[k].dimid[d]);
    b_u32(&h, NC_ATTRIBUTE); b_u32(&h, v[k].natt);
    for (int a = 0; a < v[k].natt; a++) b_text_att(&h, v[k].att[2 * a], v[k].att[2 * a + 1]);
    b_u32(&h, v[k].type); b_u32(&h, (uint32_t)((v[k].vsize + 3) / 4 * 4)); b_u64(&h, v[k].begin);
  }
  if (h.n != hdr) { free(h.p); return rf_fail(FG_ERR_STATE, "fg_remap_write: internal header size mismatch"); }
  FILE *f = fopen(path, "wb");
  if (!f) { free(h.p); return rf_fail(FG_ERR_ARG, "fg_remap_write: cannot open file for writing"); }
  int ok = fwrite(h.p, 1, h.n, f) == h.n;
  free(h.p);
  /* data, big-endian */
  Buf d = {0};
  for (int k = 0; k < nvar && ok; k++) {
    d.n = 0;
    const int *iv = (k == 0) ? tile1 : (k == 1) ? tile1_cell : tile2_cell;
    long cnt = (k == 0 || k == 3) ? ncells : 2 * ncells;
    /* one reservation per variable, then whole-word byte swaps (element by element through b_u32 this ran at 1.2 GB/s) */
    if (b_need(&d, (size_t)cnt * (k < 3 ? 4 : 8) + 8)) { ok = 0; break; }
    if (k < 3) {
      uint32_t *o = (uint32_t *)d.p;
      for (long q = 0; q < cnt; q++) o[q] = __builtin_bswap32((uint32_t)iv[q]);
      d.n = (size_t)cnt * 4;
    } else {
      const double *dv = (k == 3) ? xgrid_area : tile1_distance;
      uint64_t *o = (uint64_t *)d.p;
      for (long q = 0; q < cnt; q++) { uint64_t u; memcpy(&u, &dv[q], 8); o[q] = __builtin_bswap64(u); }
      d.n = (size_t)cnt * 8;
    }
    ok = fwrite(d.p, 1, d.n, f) == d.n;
  }
  free(d.p);
  ok = (fclose(f) == 0) && ok;
  return ok ? 0 : rf_fail(FG_ERR_ARG, "fg_remap_write: write failed");
}

/* conserve_interp.c:404-439: 0-based interp arrays -> the 1-based file variables (isc/jsc: start of the output
 * compute domain, 0 for a whole tile) */
int fg_remap_write_interp(const char *path, int order, long n, const int *t_in, const int *i_in, const int *j_in,
                          const int *i_out, const int *j_out, const double *area, const double *di_in,
                          const double *dj_in, int isc, int jsc)
{
  int *t1 = (int *)malloc(sizeof(int) * (n + 1)), *c1 = (int *)malloc(sizeof(int) * (2 * n + 1)), *c2 = (int *)malloc(sizeof(int) * (2 * n + 1));
  double *dist = (order == 2) ? (double *)malloc(sizeof(double) * (2 * n + 1)) : NULL;
  if (!t1 || !c1 || !c2 || (order == 2 && !dist)) { free(t1); free(c1); free(c2); free(dist); return rf_fail(FG_ERR_HIP, "out of memory"); }
  for (long k = 0; k < n; k++) {
    t1[k] = t_in[k] + 1;
    c1[2 * k] = i_in[k] + 1; c1[2 * k + 1] = j_in[k] + 1;
    c2[2 * k] = i_out[k] + isc + 1; c2[2 * k + 1] = j_out[k] + jsc + 1;
    if (order == 2) { dist[2 * k] = di_in[k]; dist[2 * k + 1] = dj_in[k]; }
  }
  int rc = fg_remap_write(path, order, n, t1, c1, c2, area, dist);
  free(t1); free(c1); free(c2); free(dist);
  return rc;
}

/* ---------------------------------------------------------------- reader */
typedef struct { const unsigned char *p; size_t n, pos; int v; int bad; } Rd;
static uint64_t r_u32(Rd *r) { if (r->pos + 4 > r->n) { r->bad = 1; return 0; } const unsigned char *q = r->p + r->pos; r->pos += 4; return ((uint64_t)q[0] << 24) | (q[1] << 16) | (q[2] << 8) | q[3]; }
static uint64_t r_u64(Rd *r) { uint64_t a = r_u32(r); return (a << 32) | r_u32(r); }
static uint64_t r_size(Rd *r) { return r->v == 5 ? r_u64(r) : r_u32(r); }      /* counts/lengths: 8 bytes in CDF-5 */
static void r_name(Rd *r, char *out, size_t cap)
{
  uint64_t l = r_size(r);
  if (r->bad || r->pos + l > r->n) { r->bad = 1; out[0] = 0; return; }
  size_t c = l < cap - 1 ? l : cap - 1;
  memcpy(out, r->p + r->pos, c); out[c] = 0;
  r->pos += (l + 3) / 4 * 4;
}
static size_t type_size(uint64_t t) { return (t == 1 || t == 2 || t == 7) ? 1 : (t == 3 || t == 8) ? 2 : (t == 4 || t == 5 || t == 9) ? 4 : 8; }
static void r_skip_atts(Rd *r)
{
  uint64_t tag = r_u32(r), n = r_size(r);
  if (tag == 0 && n == 0) return;
  if (tag != NC_ATTRIBUTE) { r->bad = 1; return; }
  char nm[300];
  for (uint64_t a = 0; a < n && !r->bad; a++) {
    r_name(r, nm, sizeof nm);
    uint64_t t = r_u32(r), ne = r_size(r);
    r->pos += (ne * type_size(t) + 3) / 4 * 4;
    if (r->pos > r->n) r->bad = 1;
  }
}

typedef struct { char name[64]; int ndims; long dimlen[4]; uint64_t type, begin; long nelem; } RVar;

static int parse(const unsigned char *buf, size_t n, long *ncells, RVar *vars, int maxv, int *nvars)
{
  if (n >= 8 && !memcmp(buf, "\x89HDF\r\n\x1a\n", 8))
    return rf_fail(FG_ERR_ARG, "remap file is netCDF-4/HDF5; this build reads classic netCDF (CDF-1/2/5) only - convert with nccopy -k cdf2");
  if (n < 8 || memcmp(buf, "CDF", 3) || (buf[3] != 1 && buf[3] != 2 && buf[3] != 5)) return rf_fail(FG_ERR_ARG, "not a classic netCDF file");
  Rd r = {buf, n, 4, buf[3], 0};
  r_size(&r);                                         /* numrecs */
  uint64_t tag = r_u32(&r), nd = r_size(&r);
  if (!(tag == NC_DIMENSION || (tag == 0 && nd == 0))) return rf_fail(FG_ERR_ARG, "corrupt netCDF header (dimensions)");
  long dimlen[64]; char dname[64][64];
  if (nd > 64) return rf_fail(FG_ERR_ARG, "too many dimensions");
  *ncells = -1;
  for (uint64_t d = 0; d < nd; d++) {
    r_name(&r, dname[d], 64); dimlen[d] = (long)r_size(&r);
    if (!strcmp(dname[d], "ncells")) *ncells = dimlen[d];
  }
  r_skip_atts(&r);
  tag = r_u32(&r); uint64_t nv = r_size(&r);
  if (r.bad || !(tag == NC_VARIABLE || (tag == 0 && nv == 0))) return rf_fail(FG_ERR_ARG, "corrupt netCDF header (variables)");
  *nvars = 0;
  for (uint64_t k = 0; k < nv && !r.bad; k++) {
    RVar v; memset(&v, 0, sizeof v);
    r_name(&r, v.name, sizeof v.name);
    v.ndims = (int)r_size(&r);
    v.nelem = 1;
    for (int d = 0; d < v.ndims; d++) { uint64_t id = r_size(&r); long len = (id < nd) ? dimlen[id] : 0; if (d < 4) v.dimlen[d] = len; v.nelem *= len; }
    r_skip_atts(&r);
    v.type = r_u32(&r);
    r_size(&r);                                       /* vsize */
    v.begin = (r.v == 1) ? r_u32(&r) : r_u64(&r);
    if (*nvars < maxv) vars[(*nvars)++] = v;
  }
  if (r.bad) return rf_fail(FG_ERR_ARG, "truncated netCDF header");
  if (*ncells < 0) return rf_fail(FG_ERR_ARG, "remap file has no dimension 'ncells'");
  return 0;
}

static unsigned char *slurp(const char *path, size_t *n)
{
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  unsigned char *b = (unsigned char *)malloc(sz > 0 ? sz : 1);
  if (b && fread(b, 1, sz, f) != (size_t)sz) { free(b); b = NULL; }
  fclose(f);
  *n = (size_t)sz;
  return b;
}

long fg_remap_read_size(const char *path)          /* read_mosaic_xgrid_size, read_mosaic.c:352 */
{
  size_t n; unsigned char *b = slurp(path, &n);
  if (!b) return rf_fail(FG_ERR_ARG, "cannot read remap file");
  long nc; RVar v[32]; int nv;
  int rc = parse(b, n, &nc, v, 32, &nv);
  free(b);
  return rc ? rc : nc;
}

static const RVar *find_var(const RVar *v, int nv, const char *name) { for (int k = 0; k < nv; k++) if (!strcmp(v[k].name, name)) return &v[k]; return NULL; }
static int get_ints(const unsigned char *b, size_t n, const RVar *v, long cnt, int *out)
{
  if (!v || v->type != T_INT || v->nelem != cnt || v->begin + 4ull * cnt > n) return -1;
  const unsigned char *q = b + v->begin;
  for (long k = 0; k < cnt; k++, q += 4) { uint32_t u; memcpy(&u, q, 4); out[k] = (int)__builtin_bswap32(u); }
  return 0;
}
static int get_doubles(const unsigned char *b, size_t n, const RVar *v, long cnt, double *out)
{
  if (!v || v->type != T_DOUBLE || v->nelem != cnt || v->begin + 8ull * cnt > n) return -1;
  const unsigned char *q = b + v->begin;
  for (long k = 0; k < cnt; k++, q += 8) { uint64_t u; memcpy(&u, q, 8); u = __builtin_bswap64(u); memcpy(&out[k], &u, 8); }
  return 0;
}

/* read_mosaic_xgrid_order1/2 + the tile1 read and area rescale of conserve_interp.c:80-90; arrays hold ncells entries
 * (0-based i/j/tile).  di_in/dj_in may be NULL for order 1. */
int fg_remap_read(const char *path, int order, long ncells, int *t_in, int *i_in, int *j_in, int *i_out, int *j_out,
                  double *area, double *di_in, double *dj_in)
{
  size_t n; unsigned char *b = slurp(path, &n);
  if (!b) return rf_fail(FG_ERR_ARG, "cannot read remap file");
  long nc; RVar v[32]; int nv;
  int rc = parse(b, n, &nc, v, 32, &nv);
  if (rc) { free(b); return rc; }
  if (nc != ncells) { free(b); return rf_fail(FG_ERR_ARG, "ncells in the remap file differs from the caller's array size"); }
  int *c = (int *)malloc(sizeof(int) * (2 * nc + 1));
  double *dist = (double *)malloc(sizeof(double) * (2 * nc + 1));
  const double garea = 4 * RF_PI * RF_RADIUS * RF_RADIUS;
  rc = 0;
  if (get_ints(b, n, find_var(v, nv, "tile1"), nc, t_in)) rc = rf_fail(FG_ERR_ARG, "remap file: variable tile1 missing or malformed");
  if (!rc && get_ints(b, n, find_var(v, nv, "tile1_cell"), 2 * nc, c)) rc = rf_fail(FG_ERR_ARG, "remap file: variable tile1_cell missing or malformed");
  if (!rc) for (long k = 0; k < nc; k++) { i_in[k] = c[2 * k] - 1; j_in[k] = c[2 * k + 1] - 1; t_in[k] -= 1; }
  if (!rc && get_ints(b, n, find_var(v, nv, "tile2_cell"), 2 * nc, c)) rc = rf_fail(FG_ERR_ARG, "remap file: variable tile2_cell missing or malformed");
  if (!rc) for (long k = 0; k < nc; k++) { i_out[k] = c[2 * k] - 1; j_out[k] = c[2 * k + 1] - 1; }
  if (!rc && get_doubles(b, n, find_var(v, nv, "xgrid_area"), nc, area)) rc = rf_fail(FG_ERR_ARG, "remap file: variable xgrid_area missing or malformed");
  if (!rc) for (long k = 0; k < nc; k++) { area[k] /= garea; area[k] *= garea; }
  if (!rc && order == 2) {
    if (get_doubles(b, n, find_var(v, nv, "tile1_distance"), 2 * nc, dist)) rc = rf_fail(FG_ERR_ARG, "remap file: variable tile1_distance missing or malformed");
    else for (long k = 0; k < nc; k++) { di_in[k] = dist[2 * k]; dj_in[k] = dist[2 * k + 1]; }
  }
  free(c); free(dist); free(b);
  return rc;
}
