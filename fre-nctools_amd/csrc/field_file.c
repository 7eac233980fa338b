/*
 * field_file.c -- field and grid files without libnetcdf: a minimal reader / writer of the classic
 * netCDF container (CDF-1, CDF-2 "64-bit offset", CDF-5 "64-bit data").
 *
 * fregrid moves its fields through tools/libfrencutils/mpp_io.c, a thin layer over libnetcdf: mpp_get_var_value_block
 * (:443-481, nc_get_vara_<type> of a hyperslab), mpp_put_var_value_block (:1349-), mpp_get_var_att (:487-), mpp_def_dim /
 * mpp_def_var / mpp_def_*_att / mpp_end_def.  get_input_data (tools/fregrid/fregrid_util.c:2036-2165) reads one
 * (time, [n], z-range, y, x) hyperslab per tile, write_field_data (:2339-2418) writes one.  This image has no netCDF
 * library, and the format is public, so the same operations are provided here directly on the file:
 *
 *   fg_nc_open / fg_nc_inq_* / fg_nc_get_att_* / fg_nc_get_vara[_double]     (pread of the hyperslab's contiguous runs)
 *   fg_nc_create / fg_nc_def_dim / fg_nc_def_var / fg_nc_put_att_* / fg_nc_enddef / fg_nc_put_vara[_double] / fg_nc_close
 *
 * Record variables (first dimension unlimited) are supported in both directions.  Data are big-endian in the file and
 * host-endian in memory.  fg_nc_get_vara returns the variable's own type (so that NC_FLOAT / NC_SHORT levels can cross
 * PCIe narrow and be widened on the device: fregrid_util.c:2097-2112 does the widening on the host);
 * fg_nc_get_vara_double converts as nc_get_vara_double does.  netCDF-4 (HDF5) files are rejected with a clear message.
 */
#define _GNU_SOURCE
#define _FILE_OFFSET_BITS 64
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>
#include "fregrid_hip.h"

#define TAG_DIM 0x0A
#define TAG_VAR 0x0B
#define TAG_ATT 0x0C

static __thread char g_nc_err[512];
const char *fg_nc_last_error(void) { return g_nc_err; }
static int nc_fail(int code, const char *fmt, const char *a)
{
  snprintf(g_nc_err, sizeof g_nc_err, fmt, a ? a : "");
  return code;
}

typedef struct { char *name; int type; uint64_t n; unsigned char *val; } NcAtt;       /* val: host-endian, n elements */
typedef struct { char *name; uint64_t len; } NcDim;                                     /* len 0: the record dimension */
typedef struct {
  char *name; int type, ndims; int *dimid; int natt; NcAtt *att;
  uint64_t vsize, begin; int is_rec;
} NcVar;
struct fg_ncfile {
  int fd, version, writable, defining;
  uint64_t numrecs, recsize;
  int ndim, nvar, ngatt, recdim;
  NcDim *dim; NcVar *var; NcAtt *gatt;
  uint64_t numrecs_off;            /* file offset of numrecs in the header */
};

static size_t tsize(int t)
{
  switch (t) {
    case FG_NC_BYTE: case FG_NC_CHAR: case 7 /* ubyte */: return 1;
    case FG_NC_SHORT: case 8 /* ushort */: return 2;
    case FG_NC_INT: case FG_NC_FLOAT: case 9 /* uint */: return 4;
    case FG_NC_DOUBLE: case 10: case 11: return 8;
    default: return 0;
  }
}

static void swap_copy(void *dst, const void *src, size_t n, size_t w)
{
  const unsigned char *s = (const unsigned char *)src;
  unsigned char *d = (unsigned char *)dst;
  if (w == 1) { memmove(d, s, n); return; }
  if (w == 2) { for (size_t i = 0; i < n; i++) { uint16_t v; memcpy(&v, s + 2 * i, 2); v = __builtin_bswap16(v); memcpy(d + 2 * i, &v, 2); } return; }
  if (w == 4) { for (size_t i = 0; i < n; i++) { uint32_t v; memcpy(&v, s + 4 * i, 4); v = __builtin_bswap32(v); memcpy(d + 4 * i, &v, 4); } return; }
  for (size_t i = 0; i < n; i++) { uint64_t v; memcpy(&v, s + 8 * i, 8); v = __builtin_bswap64(v); memcpy(d + 8 * i, &v, 8); }
}

/* ------------------------------------------------------------------------------------------ header parsing */
typedef struct { const unsigned char *p; size_t n, pos; int v, bad; } Rd;
static uint64_t rd_u32(Rd *r) { if (r->pos + 4 > r->n) { r->bad = 1; return 0; } const unsigned char *q = r->p + r->pos; r->pos += 4; return ((uint64_t)q[0] << 24) | ((uint64_t)q[1] << 16) | ((uint64_t)q[2] << 8) | q[3]; }
static uint64_t rd_u64(Rd *r) { uint64_t a = rd_u32(r); return (a << 32) | rd_u32(r); }
static uint64_t rd_size(Rd *r) { return r->v == 5 ? rd_u64(r) : rd_u32(r); }
static char *rd_name(Rd *r)
{
  uint64_t l = rd_size(r);
  if (r->bad || l > 65536 || r->pos + l > r->n) { r->bad = 1; return NULL; }
  char *s = (char *)malloc(l + 1);
  if (!s) { r->bad = 1; return NULL; }
  memcpy(s, r->p + r->pos, l); s[l] = 0;
  r->pos += (l + 3) / 4 * 4;
  return s;
}
static int rd_atts(Rd *r, int *natt, NcAtt **out)
{
  uint64_t tag = rd_u32(r), n = rd_size(r);
  *natt = 0; *out = NULL;
  if (r->bad) return -1;
  if (tag == 0 && n == 0) return 0;
  if (tag != TAG_ATT || n > 100000) { r->bad = 1; return -1; }
  NcAtt *a = (NcAtt *)calloc(n ? n : 1, sizeof(NcAtt));
  if (!a) { r->bad = 1; return -1; }
  for (uint64_t k = 0; k < n && !r->bad; k++) {
    a[k].name = rd_name(r);
    a[k].type = (int)rd_u32(r);
    a[k].n = rd_size(r);
    size_t w = tsize(a[k].type);
    /* (a 64-bit CDF-5 count: compare before multiplying, or w * n wraps and the copy below runs past a small buffer) */
    if (!w || r->pos > r->n || a[k].n > (uint64_t)(r->n - r->pos) / w) { r->bad = 1; break; }
    const size_t bytes = w * (size_t)a[k].n;
    a[k].val = (unsigned char *)malloc(bytes + 1);
    if (!a[k].val) { r->bad = 1; break; }
    swap_copy(a[k].val, r->p + r->pos, a[k].n, w);
    a[k].val[bytes] = 0;
    r->pos += (bytes + 3) / 4 * 4;
  }
  *natt = (int)n; *out = a;
  return r->bad ? -1 : 0;
}

static void free_atts(int n, NcAtt *a) { if (!a) return; for (int k = 0; k < n; k++) { free(a[k].name); free(a[k].val); } free(a); }
static void nc_free(fg_ncfile *f)
{
  if (!f) return;
  for (int k = 0; k < f->ndim; k++) free(f->dim[k].name);
  for (int k = 0; k < f->nvar; k++) { free(f->var[k].name); free(f->var[k].dimid); free_atts(f->var[k].natt, f->var[k].att); }
  free_atts(f->ngatt, f->gatt);
  free(f->dim); free(f->var);
  if (f->fd >= 0) close(f->fd);
  free(f);
}

static void compute_recsize(fg_ncfile *f)
{
  uint64_t rs = 0; int nrec = 0;
  for (int k = 0; k < f->nvar; k++) if (f->var[k].is_rec) { rs += f->var[k].vsize; nrec++; }
  /* a single record variable of a type narrower than 4 bytes is stored without padding between its records */
  if (nrec == 1)
    for (int k = 0; k < f->nvar; k++) if (f->var[k].is_rec) {
      uint64_t n = tsize(f->var[k].type);
      for (int d = 1; d < f->var[k].ndims; d++) n *= f->dim[f->var[k].dimid[d]].len;
      rs = n;
    }
  f->recsize = rs;
}

int fg_nc_open(const char *path, fg_ncfile **out)
{
  if (!path || !out) return nc_fail(FG_ERR_ARG, "fg_nc_open: null argument%s", NULL);
  int fd = open(path, O_RDONLY);
  if (fd < 0) return nc_fail(FG_ERR_IO, "fg_nc_open: cannot open %s", path);
  /* the header is read in growing pieces until it parses */
  size_t cap = 1 << 16;
  fg_ncfile *f = NULL;
  for (;;) {
    unsigned char *buf = (unsigned char *)malloc(cap);
    if (!buf) { close(fd); return nc_fail(FG_ERR_IO, "fg_nc_open: out of memory%s", NULL); }
    ssize_t got = pread(fd, buf, cap, 0);
    if (got < 4) { free(buf); close(fd); return nc_fail(FG_ERR_IO, "fg_nc_open: %s is not a netCDF file", path); }
    if (!memcmp(buf, "\x89HDF", 4)) { free(buf); close(fd); return nc_fail(FG_ERR_IO, "fg_nc_open: %s is a netCDF-4 (HDF5) file; convert it to a classic format (nccopy -k cdf5)", path); }
    if (memcmp(buf, "CDF", 3) || (buf[3] != 1 && buf[3] != 2 && buf[3] != 5)) { free(buf); close(fd); return nc_fail(FG_ERR_IO, "fg_nc_open: %s is not a classic netCDF file", path); }
    Rd r = {buf, (size_t)got, 4, buf[3], 0};
    f = (fg_ncfile *)calloc(1, sizeof *f);
    if (!f) { free(buf); close(fd); return nc_fail(FG_ERR_IO, "fg_nc_open: out of memory%s", NULL); }
    f->fd = -1; f->version = buf[3]; f->recdim = -1; f->numrecs_off = 4;
    f->numrecs = rd_size(&r);
    uint64_t tag = rd_u32(&r), n = rd_size(&r);
    if (!r.bad && !(tag == 0 && n == 0)) {
      if (tag != TAG_DIM || n > 100000) r.bad = 1;
      else {
        f->dim = (NcDim *)calloc(n, sizeof(NcDim)); f->ndim = (int)n;
        for (uint64_t k = 0; k < n && !r.bad; k++) { f->dim[k].name = rd_name(&r); f->dim[k].len = rd_size(&r); if (f->dim[k].len == 0) f->recdim = (int)k; }
      }
    }
    if (!r.bad) rd_atts(&r, &f->ngatt, &f->gatt);
    if (!r.bad) {
      tag = rd_u32(&r); n = rd_size(&r);
      if (!r.bad && !(tag == 0 && n == 0)) {
        if (tag != TAG_VAR || n > 100000) r.bad = 1;
        else {
          f->var = (NcVar *)calloc(n, sizeof(NcVar)); f->nvar = (int)n;
          for (uint64_t k = 0; k < n && !r.bad; k++) {
            NcVar *v = &f->var[k];
            v->name = rd_name(&r);
            v->ndims = (int)rd_size(&r);
            if (r.bad || v->ndims < 0 || v->ndims > 32) { r.bad = 1; break; }
            v->dimid = (int *)calloc(v->ndims ? v->ndims : 1, sizeof(int));
            for (int d = 0; d < v->ndims; d++) { v->dimid[d] = (int)rd_size(&r); if (v->dimid[d] < 0 || v->dimid[d] >= f->ndim) r.bad = 1; }
            rd_atts(&r, &v->natt, &v->att);
            v->type = (int)rd_u32(&r);
            v->vsize = rd_size(&r);
            v->begin = (f->version == 1) ? rd_u32(&r) : rd_u64(&r);
            v->is_rec = v->ndims > 0 && !r.bad && v->dimid[0] == f->recdim;
            if (!tsize(v->type)) r.bad = 1;
          }
        }
      }
    }
    /* ran off the end of what was read AND the file has more: read more and parse again (never past the file's own size) */
    const int truncated = r.bad && (size_t)got == cap && r.pos + 8 > (size_t)got;
    free(buf);
    if (!r.bad) break;
    nc_free(f); f = NULL;
    if (!truncated || cap > ((size_t)1 << 30)) { close(fd); return nc_fail(FG_ERR_IO, "fg_nc_open: malformed header in %s", path); }
    cap *= 4;
  }
  f->fd = fd;
  /* vsize in the header saturates for huge variables; recompute it from the shape (as libnetcdf does) */
  for (int k = 0; k < f->nvar; k++) {
    NcVar *v = &f->var[k];
    uint64_t n = tsize(v->type);
    for (int d = v->is_rec ? 1 : 0; d < v->ndims; d++) n *= f->dim[v->dimid[d]].len;
    v->vsize = (n + 3) / 4 * 4;
  }
  compute_recsize(f);
  *out = f;
  return 0;
}

/* ------------------------------------------------------------------------------------------ inquiry */
int fg_nc_inq_ndims(const fg_ncfile *f) { return f ? f->ndim : FG_ERR_ARG; }
int fg_nc_inq_nvars(const fg_ncfile *f) { return f ? f->nvar : FG_ERR_ARG; }
long fg_nc_inq_numrecs(const fg_ncfile *f) { return f ? (long)f->numrecs : FG_ERR_ARG; }
int fg_nc_inq_dimid(const fg_ncfile *f, const char *name)
{
  if (!f || !name) return -1;
  for (int k = 0; k < f->ndim; k++) if (!strcmp(f->dim[k].name, name)) return k;
  return -1;
}
int fg_nc_inq_dim(const fg_ncfile *f, int dimid, char *name, int cap, long *len)
{
  if (!f || dimid < 0 || dimid >= f->ndim) return nc_fail(FG_ERR_ARG, "fg_nc_inq_dim: bad dimension id%s", NULL);
  if (name && cap > 0) snprintf(name, cap, "%s", f->dim[dimid].name);
  if (len) *len = (dimid == f->recdim) ? (long)f->numrecs : (long)f->dim[dimid].len;
  return 0;
}
int fg_nc_inq_varid(const fg_ncfile *f, const char *name)
{
  if (!f || !name) return -1;
  for (int k = 0; k < f->nvar; k++) if (!strcmp(f->var[k].name, name)) return k;
  return -1;
}
int fg_nc_inq_var(const fg_ncfile *f, int varid, char *name, int cap, int *type, int *ndims, int *dimids, long *shape)
{
  if (!f || varid < 0 || varid >= f->nvar) return nc_fail(FG_ERR_ARG, "fg_nc_inq_var: bad variable id%s", NULL);
  const NcVar *v = &f->var[varid];
  if (name && cap > 0) snprintf(name, cap, "%s", v->name);
  if (type) *type = v->type;
  if (ndims) *ndims = v->ndims;
  for (int d = 0; d < v->ndims; d++) {
    if (dimids) dimids[d] = v->dimid[d];
    if (shape) shape[d] = (v->dimid[d] == f->recdim) ? (long)f->numrecs : (long)f->dim[v->dimid[d]].len;
  }
  return 0;
}
static const NcAtt *find_att(const fg_ncfile *f, int varid, const char *name)
{
  int n = (varid < 0) ? f->ngatt : f->var[varid].natt;
  const NcAtt *a = (varid < 0) ? f->gatt : f->var[varid].att;
  for (int k = 0; k < n; k++) if (!strcmp(a[k].name, name)) return &a[k];
  return NULL;
}
static double att_elem(const NcAtt *a, uint64_t i)
{
  switch (a->type) {
    case FG_NC_BYTE: return ((signed char *)a->val)[i];
    case FG_NC_CHAR: return ((unsigned char *)a->val)[i];
    case FG_NC_SHORT: { int16_t v; memcpy(&v, a->val + 2 * i, 2); return v; }
    case FG_NC_INT: { int32_t v; memcpy(&v, a->val + 4 * i, 4); return v; }
    case FG_NC_FLOAT: { float v; memcpy(&v, a->val + 4 * i, 4); return v; }
    case 7 /* NC_UBYTE */: return ((unsigned char *)a->val)[i];
    case 8 /* NC_USHORT */: { uint16_t v; memcpy(&v, a->val + 2 * i, 2); return v; }
    case 9 /* NC_UINT */: { uint32_t v; memcpy(&v, a->val + 4 * i, 4); return v; }
    case 10 /* NC_INT64 */: { int64_t v; memcpy(&v, a->val + 8 * i, 8); return (double)v; }
    case 11 /* NC_UINT64 */: { uint64_t v; memcpy(&v, a->val + 8 * i, 8); return (double)v; }
    default: { double v; memcpy(&v, a->val + 8 * i, 8); return v; }
  }
}
/* numeric attribute as double(s) (nc_get_att_double, which mpp_get_var_att uses for NC_FLOAT / NC_DOUBLE attributes);
 * returns the number of elements copied, FG_ERR_NOTFOUND if absent */
int fg_nc_get_att_double(const fg_ncfile *f, int varid, const char *name, double *val, int cap)
{
  if (!f || !name || varid >= f->nvar) return nc_fail(FG_ERR_ARG, "fg_nc_get_att_double: bad argument%s", NULL);
  const NcAtt *a = find_att(f, varid, name);
  if (!a) return FG_ERR_NOTFOUND;
  int n = (int)(a->n < (uint64_t)cap ? a->n : (uint64_t)cap);
  for (int i = 0; i < n; i++) val[i] = att_elem(a, i);
  return n;
}
int fg_nc_get_att_text(const fg_ncfile *f, int varid, const char *name, char *buf, int cap)
{
  if (!f || !name || !buf || cap < 1 || varid >= f->nvar) return nc_fail(FG_ERR_ARG, "fg_nc_get_att_text: bad argument%s", NULL);
  const NcAtt *a = find_att(f, varid, name);
  if (!a || a->type != FG_NC_CHAR) return FG_ERR_NOTFOUND;
  size_t n = a->n < (uint64_t)(cap - 1) ? (size_t)a->n : (size_t)(cap - 1);
  memcpy(buf, a->val, n); buf[n] = 0;
  return (int)n;
}

/* ------------------------------------------------------------------------------------------ hyperslab access */
/* Walks the hyperslab as contiguous runs: the trailing dimensions read in full collapse into one run. */
typedef int (*run_fn)(fg_ncfile *f, uint64_t off, size_t nelem, size_t w, unsigned char *mem);
static int run_read(fg_ncfile *f, uint64_t off, size_t nelem, size_t w, unsigned char *mem)
{
  size_t bytes = nelem * w, done = 0;
  while (done < bytes) {
    ssize_t g = pread(f->fd, mem + done, bytes - done, (off_t)(off + done));
    if (g <= 0) return -1;
    done += (size_t)g;
  }
  if (w > 1) swap_copy(mem, mem, nelem, w);
  return 0;
}
static int run_write(fg_ncfile *f, uint64_t off, size_t nelem, size_t w, unsigned char *mem)
{
  size_t bytes = nelem * w, done = 0;
  unsigned char *tmp = mem;
  if (w > 1) { tmp = (unsigned char *)malloc(bytes); if (!tmp) return -1; swap_copy(tmp, mem, nelem, w); }
  int rc = 0;
  while (done < bytes) {
    ssize_t g = pwrite(f->fd, tmp + done, bytes - done, (off_t)(off + done));
    if (g <= 0) { rc = -1; break; }
    done += (size_t)g;
  }
  if (tmp != mem) free(tmp);
  return rc;
}

static int vara(fg_ncfile *f, int varid, const long *start, const long *count, void *mem, run_fn fn, int writing)
{
  if (!f || varid < 0 || varid >= f->nvar || !mem) return nc_fail(FG_ERR_ARG, "fg_nc vara: bad argument%s", NULL);
  NcVar *v = &f->var[varid];
  const size_t w = tsize(v->type);
  const int nd = v->ndims;
  uint64_t shape[32], stride[32];
  long st[32], ct[32];
  for (int d = 0; d < nd; d++) {
    st[d] = start ? start[d] : 0;
    shape[d] = (d == 0 && v->is_rec) ? (writing ? (uint64_t)1 << 62 : f->numrecs) : f->dim[v->dimid[d]].len;
    ct[d] = count ? count[d] : (long)shape[d];
    if (st[d] < 0 || ct[d] < 0 || (uint64_t)st[d] + (uint64_t)ct[d] > shape[d]) return nc_fail(FG_ERR_ARG, "fg_nc vara: hyperslab outside variable %s", v->name);
  }
  uint64_t total = 1;
  for (int d = 0; d < nd; d++) total *= (uint64_t)ct[d];
  if (total == 0) return 0;
  /* element strides; the record dimension advances by the record size */
  uint64_t s = 1;
  for (int d = nd - 1; d >= 0; d--) { stride[d] = s; if (!(d == 0 && v->is_rec)) s *= shape[d]; }
  /* innermost contiguous run */
  int d0 = nd;                                   /* dimensions [d0, nd) are inside the run */
  uint64_t run = 1;
  while (d0 > 0) {
    const int d = d0 - 1;
    if (d == 0 && v->is_rec) break;
    run *= (uint64_t)ct[d]; d0--;
    if ((uint64_t)ct[d] != shape[d]) break;      /* partial dimension: the run ends here */
  }
  long idx[32] = {0};
  unsigned char *m = (unsigned char *)mem;
  for (;;) {
    uint64_t off = v->begin;
    for (int d = 0; d < nd; d++) {
      const uint64_t i = (uint64_t)st[d] + (d < d0 ? (uint64_t)idx[d] : 0);
      if (d == 0 && v->is_rec) off += i * f->recsize;
      else off += i * stride[d] * w;
    }
    if (fn(f, off, (size_t)run, w, m)) return nc_fail(FG_ERR_IO, "fg_nc vara: I/O error on variable %s", v->name);
    m += run * w;
    int d = d0 - 1;
    while (d >= 0) { if (++idx[d] < ct[d]) break; idx[d] = 0; d--; }
    if (d < 0) break;
  }
  if (writing && v->is_rec && (uint64_t)(st[0] + ct[0]) > f->numrecs) f->numrecs = (uint64_t)(st[0] + ct[0]);
  return 0;
}

int fg_nc_get_vara(fg_ncfile *f, int varid, const long *start, const long *count, void *out)
{
  if (f && f->defining) return nc_fail(FG_ERR_STATE, "fg_nc_get_vara: file is in define mode%s", NULL);
  return vara(f, varid, start, count, out, run_read, 0);
}

static uint64_t slab_elems(const fg_ncfile *f, const NcVar *v, const long *count)
{
  uint64_t n = 1;
  for (int d = 0; d < v->ndims; d++) n *= count ? (uint64_t)count[d] : ((d == 0 && v->is_rec) ? f->numrecs : f->dim[v->dimid[d]].len);
  return n;
}

/* nc_get_vara_double: any numeric type widened to double */
int fg_nc_get_vara_double(fg_ncfile *f, int varid, const long *start, const long *count, double *out)
{
  if (!f || varid < 0 || varid >= f->nvar || !out) return nc_fail(FG_ERR_ARG, "fg_nc_get_vara_double: bad argument%s", NULL);
  const NcVar *v = &f->var[varid];
  if (v->type == FG_NC_DOUBLE) return fg_nc_get_vara(f, varid, start, count, out);
  const uint64_t n = slab_elems(f, v, count);
  const size_t w = tsize(v->type);
  /* read into the tail of the output buffer, then widen front to back (no second buffer) */
  unsigned char *raw = (unsigned char *)out + n * (8 - w);
  int rc = fg_nc_get_vara(f, varid, start, count, raw);
  if (rc) return rc;
  for (uint64_t i = 0; i < n; i++) {
    double x;
    switch (v->type) {
      case FG_NC_BYTE: x = ((signed char *)raw)[i]; break;
      case FG_NC_CHAR: x = raw[i]; break;
      case FG_NC_SHORT: { int16_t t; memcpy(&t, raw + 2 * i, 2); x = t; break; }
      case FG_NC_INT: { int32_t t; memcpy(&t, raw + 4 * i, 4); x = t; break; }
      case FG_NC_FLOAT: { float t; memcpy(&t, raw + 4 * i, 4); x = t; break; }
      default: return nc_fail(FG_ERR_ARG, "fg_nc_get_vara_double: unsupported type of variable %s", v->name);
    }
    out[i] = x;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ writing */
typedef struct { unsigned char *p; size_t n, cap; int v; } Wb;
static int wb_need(Wb *b, size_t k)
{
  if (b->n + k <= b->cap) return 0;
  size_t c = b->cap ? b->cap * 2 : 4096;
  while (c < b->n + k) c *= 2;
  unsigned char *q = (unsigned char *)realloc(b->p, c);
  if (!q) return -1;
  b->p = q; b->cap = c;
  return 0;
}
static void wb_u32(Wb *b, uint32_t v) { if (wb_need(b, 4)) return; b->p[b->n++] = v >> 24; b->p[b->n++] = v >> 16; b->p[b->n++] = v >> 8; b->p[b->n++] = v; }
static void wb_u64(Wb *b, uint64_t v) { wb_u32(b, (uint32_t)(v >> 32)); wb_u32(b, (uint32_t)v); }
static void wb_size(Wb *b, uint64_t v) { if (b->v == 5) wb_u64(b, v); else wb_u32(b, (uint32_t)v); }
static void wb_bytes(Wb *b, const void *s, size_t l)
{
  size_t pad = (4 - l % 4) % 4;
  if (wb_need(b, l + pad)) return;
  memcpy(b->p + b->n, s, l); b->n += l;
  for (size_t k = 0; k < pad; k++) b->p[b->n++] = 0;
}
static void wb_name(Wb *b, const char *s) { size_t l = strlen(s); wb_size(b, l); wb_bytes(b, s, l); }
static void wb_atts(Wb *b, int n, const NcAtt *a)
{
  if (!n) { wb_u32(b, 0); wb_size(b, 0); return; }
  wb_u32(b, TAG_ATT); wb_size(b, n);
  for (int k = 0; k < n; k++) {
    wb_name(b, a[k].name); wb_u32(b, a[k].type); wb_size(b, a[k].n);
    const size_t w = tsize(a[k].type), bytes = w * a[k].n;
    unsigned char *t = (unsigned char *)malloc(bytes ? bytes : 1);
    if (!t) return;
    swap_copy(t, a[k].val, a[k].n, w);
    wb_bytes(b, t, bytes);
    free(t);
  }
}

int fg_nc_create(const char *path, int version, fg_ncfile **out)
{
  if (!path || !out || (version != 1 && version != 2 && version != 5)) return nc_fail(FG_ERR_ARG, "fg_nc_create: version must be 1, 2 or 5%s", NULL);
  int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) return nc_fail(FG_ERR_IO, "fg_nc_create: cannot create %s", path);
  fg_ncfile *f = (fg_ncfile *)calloc(1, sizeof *f);
  if (!f) { close(fd); return nc_fail(FG_ERR_IO, "fg_nc_create: out of memory%s", NULL); }
  f->fd = fd; f->version = version; f->writable = 1; f->defining = 1; f->recdim = -1; f->numrecs_off = 4;
  *out = f;
  return 0;
}
int fg_nc_def_dim(fg_ncfile *f, const char *name, long len)
{
  if (!f || !name || len < 0 || !f->defining) return nc_fail(FG_ERR_STATE, "fg_nc_def_dim: not in define mode%s", NULL);
  if (len == 0 && f->recdim >= 0) return nc_fail(FG_ERR_ARG, "fg_nc_def_dim: only one unlimited dimension%s", NULL);
  NcDim *d = (NcDim *)realloc(f->dim, (f->ndim + 1) * sizeof(NcDim));
  if (!d) return nc_fail(FG_ERR_IO, "out of memory%s", NULL);
  f->dim = d; d[f->ndim].name = strdup(name); d[f->ndim].len = (uint64_t)len;
  if (len == 0) f->recdim = f->ndim;
  return f->ndim++;
}
int fg_nc_def_var(fg_ncfile *f, const char *name, int type, int ndims, const int *dimids)
{
  if (!f || !name || !f->defining || ndims < 0 || ndims > 32 || !tsize(type)) return nc_fail(FG_ERR_STATE, "fg_nc_def_var: bad argument or not in define mode%s", NULL);
  for (int d = 0; d < ndims; d++) {
    if (dimids[d] < 0 || dimids[d] >= f->ndim) return nc_fail(FG_ERR_ARG, "fg_nc_def_var: bad dimension id%s", NULL);
    if (d > 0 && dimids[d] == f->recdim) return nc_fail(FG_ERR_ARG, "fg_nc_def_var: the unlimited dimension must come first%s", NULL);
  }
  NcVar *v = (NcVar *)realloc(f->var, (f->nvar + 1) * sizeof(NcVar));
  if (!v) return nc_fail(FG_ERR_IO, "out of memory%s", NULL);
  f->var = v; v += f->nvar;
  memset(v, 0, sizeof *v);
  v->name = strdup(name); v->type = type; v->ndims = ndims;
  v->dimid = (int *)calloc(ndims ? ndims : 1, sizeof(int));
  for (int d = 0; d < ndims; d++) v->dimid[d] = dimids[d];
  v->is_rec = ndims > 0 && dimids[0] == f->recdim;
  return f->nvar++;
}
static int put_att(fg_ncfile *f, int varid, const char *name, int type, uint64_t n, const void *val)
{
  if (!f || !name || !f->defining || varid >= f->nvar) return nc_fail(FG_ERR_STATE, "fg_nc_put_att: bad argument or not in define mode%s", NULL);
  int *pn = (varid < 0) ? &f->ngatt : &f->var[varid].natt;
  NcAtt **pa = (varid < 0) ? &f->gatt : &f->var[varid].att;
  NcAtt *a = (NcAtt *)realloc(*pa, (*pn + 1) * sizeof(NcAtt));
  if (!a) return nc_fail(FG_ERR_IO, "out of memory%s", NULL);
  *pa = a; a += *pn;
  a->name = strdup(name); a->type = type; a->n = n;
  a->val = (unsigned char *)malloc(tsize(type) * n + 1);
  memcpy(a->val, val, tsize(type) * n);
  (*pn)++;
  return 0;
}
int fg_nc_put_att_text(fg_ncfile *f, int varid, const char *name, const char *val)
{
  if (!val) return nc_fail(FG_ERR_ARG, "fg_nc_put_att_text: null value%s", NULL);
  return put_att(f, varid, name, FG_NC_CHAR, strlen(val), val);
}
/* numeric attribute of the given type from doubles (mpp_def_var_att_double and friends) */
int fg_nc_put_att_double(fg_ncfile *f, int varid, const char *name, int type, int n, const double *vals)
{
  if (!vals || n < 1 || n > 1024) return nc_fail(FG_ERR_ARG, "fg_nc_put_att_double: bad value list%s", NULL);
  unsigned char tmp[8 * 1024];
  for (int i = 0; i < n; i++) {
    switch (type) {
      case FG_NC_BYTE: { signed char t = (signed char)vals[i]; memcpy(tmp + i, &t, 1); break; }
      case FG_NC_SHORT: { int16_t t = (int16_t)vals[i]; memcpy(tmp + 2 * i, &t, 2); break; }
      case FG_NC_INT: { int32_t t = (int32_t)vals[i]; memcpy(tmp + 4 * i, &t, 4); break; }
      case FG_NC_FLOAT: { float t = (float)vals[i]; memcpy(tmp + 4 * i, &t, 4); break; }
      case FG_NC_DOUBLE: memcpy(tmp + 8 * i, &vals[i], 8); break;
      default: return nc_fail(FG_ERR_ARG, "fg_nc_put_att_double: numeric type expected%s", NULL);
    }
  }
  return put_att(f, varid, name, type, n, tmp);
}

static size_t header_bytes(fg_ncfile *f, Wb *b)
{
  b->n = 0; b->v = f->version;
  wb_need(b, 4);
  memcpy(b->p, "CDF", 3); b->p[3] = (unsigned char)f->version; b->n = 4;
  wb_size(b, f->numrecs);
  if (f->ndim) { wb_u32(b, TAG_DIM); wb_size(b, f->ndim); for (int k = 0; k < f->ndim; k++) { wb_name(b, f->dim[k].name); wb_size(b, f->dim[k].len); } }
  else { wb_u32(b, 0); wb_size(b, 0); }
  wb_atts(b, f->ngatt, f->gatt);
  if (f->nvar) {
    wb_u32(b, TAG_VAR); wb_size(b, f->nvar);
    for (int k = 0; k < f->nvar; k++) {
      const NcVar *v = &f->var[k];
      wb_name(b, v->name); wb_size(b, v->ndims);
      for (int d = 0; d < v->ndims; d++) wb_size(b, v->dimid[d]);
      wb_atts(b, v->natt, v->att);
      wb_u32(b, v->type);
      wb_size(b, (f->version != 5 && v->vsize > 0xffffffffull) ? 0xffffffffull : v->vsize);
      if (f->version == 1) wb_u32(b, (uint32_t)v->begin); else wb_u64(b, v->begin);
    }
  } else { wb_u32(b, 0); wb_size(b, 0); }
  return b->n;
}

int fg_nc_enddef(fg_ncfile *f)
{
  if (!f || !f->defining) return nc_fail(FG_ERR_STATE, "fg_nc_enddef: not in define mode%s", NULL);
  for (int k = 0; k < f->nvar; k++) {
    NcVar *v = &f->var[k];
    uint64_t n = tsize(v->type);
    for (int d = v->is_rec ? 1 : 0; d < v->ndims; d++) n *= f->dim[v->dimid[d]].len;
    v->vsize = (n + 3) / 4 * 4;
  }
  Wb b = {0};
  const size_t hdr = header_bytes(f, &b);            /* begin fields do not change the header's size */
  uint64_t off = (hdr + 3) / 4 * 4;
  for (int k = 0; k < f->nvar; k++) if (!f->var[k].is_rec) { f->var[k].begin = off; off += f->var[k].vsize; }
  for (int k = 0; k < f->nvar; k++) if (f->var[k].is_rec) { f->var[k].begin = off; off += f->var[k].vsize; }
  compute_recsize(f);
  if (f->version == 1 && off > 0x7fffffffull) { free(b.p); return nc_fail(FG_ERR_ARG, "fg_nc_enddef: file too large for CDF-1, use version 2 or 5%s", NULL); }
  header_bytes(f, &b);
  int ok = pwrite(f->fd, b.p, b.n, 0) == (ssize_t)b.n;
  free(b.p);
  if (!ok) return nc_fail(FG_ERR_IO, "fg_nc_enddef: cannot write the header%s", NULL);
  f->defining = 0;
  return 0;
}

int fg_nc_put_vara(fg_ncfile *f, int varid, const long *start, const long *count, const void *data)
{
  if (!f || !f->writable || f->defining) return nc_fail(FG_ERR_STATE, "fg_nc_put_vara: file is not in data mode%s", NULL);
  return vara(f, varid, start, count, (void *)data, run_write, 1);
}
/* nc_put_vara_double: doubles converted to the variable's type with a C cast (what mpp_put_var_value_block gives an
 * NC_FLOAT variable; write_field_data casts to short / int itself, fregrid_util.c:2395-2406) */
int fg_nc_put_vara_double(fg_ncfile *f, int varid, const long *start, const long *count, const double *data)
{
  if (!f || varid < 0 || varid >= f->nvar || !data) return nc_fail(FG_ERR_ARG, "fg_nc_put_vara_double: bad argument%s", NULL);
  const NcVar *v = &f->var[varid];
  if (v->type == FG_NC_DOUBLE) return fg_nc_put_vara(f, varid, start, count, data);
  if (!count) return nc_fail(FG_ERR_ARG, "fg_nc_put_vara_double: count is required%s", NULL);
  const uint64_t n = slab_elems(f, v, count);
  const size_t w = tsize(v->type);
  unsigned char *tmp = (unsigned char *)malloc((n * w) > 0 ? n * w : 1);
  if (!tmp) return nc_fail(FG_ERR_IO, "out of memory%s", NULL);
  for (uint64_t i = 0; i < n; i++) {
    switch (v->type) {
      case FG_NC_BYTE: { signed char t = (signed char)data[i]; tmp[i] = (unsigned char)t; break; }
      case FG_NC_SHORT: { int16_t t = (int16_t)data[i]; memcpy(tmp + 2 * i, &t, 2); break; }
      case FG_NC_INT: { int32_t t = (int32_t)data[i]; memcpy(tmp + 4 * i, &t, 4); break; }
      case FG_NC_FLOAT: { float t = (float)data[i]; memcpy(tmp + 4 * i, &t, 4); break; }
      default: free(tmp); return nc_fail(FG_ERR_ARG, "fg_nc_put_vara_double: unsupported type of variable %s", v->name);
    }
  }
  int rc = fg_nc_put_vara(f, varid, start, count, tmp);
  free(tmp);
  return rc;
}

int fg_nc_close(fg_ncfile *f)
{
  if (!f) return 0;
  int rc = 0;
  if (f->writable) {
    if (f->defining) rc = fg_nc_enddef(f);
    if (!rc) {                                       /* numrecs, and the file's full length (the last variable may be short) */
      unsigned char b[8]; Wb w = {b, 0, 8, f->version};
      wb_size(&w, f->numrecs);
      if (pwrite(f->fd, b, w.n, (off_t)f->numrecs_off) != (ssize_t)w.n) rc = nc_fail(FG_ERR_IO, "fg_nc_close: cannot update numrecs%s", NULL);
      uint64_t end = 0;
      for (int k = 0; k < f->nvar; k++) {
        const NcVar *v = &f->var[k];
        const uint64_t e = v->is_rec ? v->begin + (f->numrecs ? (f->numrecs - 1) * f->recsize + v->vsize : 0) : v->begin + v->vsize;
        if (e > end) end = e;
      }
      struct stat sb;
      if (!rc && !fstat(f->fd, &sb) && (uint64_t)sb.st_size < end && ftruncate(f->fd, (off_t)end)) rc = nc_fail(FG_ERR_IO, "fg_nc_close: cannot extend the file%s", NULL);
    }
  }
  nc_free(f);
  return rc;
}
