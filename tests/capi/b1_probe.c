/* b1_probe.c -- a plain C99 caller of the library's B1 symbols through the REFERENCE's prototypes
 * (tools/libfrencutils/create_xgrid.h:35-80, mosaic_util.h:95-101, interp.h:32-42), written out below exactly as a
 * libfrencutils client sees them -- this file does not include fregrid_hip.h for them.  tests/test_capi_c.py compiles it with
 *   gcc -std=c99 -Wall -Werror -pedantic
 * (CPU: compile + link only; where /root/reference is present the same prototypes are also checked against the reference's
 * own headers) and, under -m gpu, runs it on inputs it wrote and compares the outputs with the oracle.
 *
 * usage: b1_probe in.bin out.bin
 *   in.bin : int nx1, ny1, nx2, ny2; double lon1[(nx1+1)(ny1+1)], lat1[...], lon2[(nx2+1)(ny2+1)], lat2[...]
 *   out.bin: int n1; int i_in,j_in,i_out,j_out [n1]; double area[n1];                      (create_xgrid_2dx2d_order1)
 *            int n2; int ...[n2]; double area, clon, clat [n2];                              (create_xgrid_2dx2d_order2)
 *            double cell_area1[nx1*ny1];                                                     (get_grid_area)
 *            int ng; int ...[ng]; double area[ng];                                           (create_xgrid_great_circle)
 *            double remapped[nx2*ny2];                                                       (conserve_interp of f = lon-index)
 */
#include <stdio.h>
#include <stdlib.h>

/* create_xgrid.h:35-80 */
int get_maxxgrid(void);
void get_grid_area(const int *nlon, const int *nlat, const double *lon, const double *lat, double *area);
int create_xgrid_2dx2d_order1(const int *nlon_in, const int *nlat_in, const int *nlon_out, const int *nlat_out,
                              const double *lon_in, const double *lat_in, const double *lon_out, const double *lat_out,
                              const double *mask_in, int *i_in, int *j_in, int *i_out, int *j_out, double *xgrid_area);
int create_xgrid_2dx2d_order2(const int *nlon_in, const int *nlat_in, const int *nlon_out, const int *nlat_out,
                              const double *lon_in, const double *lat_in, const double *lon_out, const double *lat_out,
                              const double *mask_in, int *i_in, int *j_in, int *i_out, int *j_out,
                              double *xgrid_area, double *xgrid_clon, double *xgrid_clat);
int create_xgrid_great_circle(const int *nlon_in, const int *nlat_in, const int *nlon_out, const int *nlat_out,
                              const double *lon_in, const double *lat_in, const double *lon_out, const double *lat_out,
                              const double *mask_in, int *i_in, int *j_in, int *i_out, int *j_out,
                              double *xgrid_area, double *xgrid_clon, double *xgrid_clat);
/* interp.h:32-36 */
void conserve_interp(int nx_src, int ny_src, int nx_dst, int ny_dst, const double *x_src,
                     const double *y_src, const double *x_dst, const double *y_dst,
                     const double *mask_src, const double *data_src, double *data_dst);
/* mosaic_util.h */
double poly_area(const double lon[], const double lat[], int n);

static void *xmalloc(size_t n) { void *p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "out of memory\n"); exit(2); } return p; }
static void rd(void *p, size_t sz, size_t n, FILE *f) { if (fread(p, sz, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }
static void wr(const void *p, size_t sz, size_t n, FILE *f) { if (fwrite(p, sz, n, f) != n) { fprintf(stderr, "short write\n"); exit(2); } }

int main(int argc, char **argv)
{
  int dims[4], nx1, ny1, nx2, ny2, n, k;
  size_t np1, np2, cap;
  double *lon1, *lat1, *lon2, *lat2, *mask, *area, *clon, *clat, *carea, *src, *dst;
  int *ii, *ji, *io, *jo;
  FILE *f;
  if (argc == 2 && argv[1][0] == '-') { printf("maxxgrid %d\n", get_maxxgrid()); return 0; }   /* link check without a device */
  if (argc != 3) { fprintf(stderr, "usage: b1_probe in.bin out.bin\n"); return 2; }
  f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  rd(dims, sizeof(int), 4, f);
  nx1 = dims[0]; ny1 = dims[1]; nx2 = dims[2]; ny2 = dims[3];
  np1 = (size_t)(nx1 + 1) * (ny1 + 1); np2 = (size_t)(nx2 + 1) * (ny2 + 1);
  lon1 = xmalloc(np1 * sizeof(double)); lat1 = xmalloc(np1 * sizeof(double));
  lon2 = xmalloc(np2 * sizeof(double)); lat2 = xmalloc(np2 * sizeof(double));
  rd(lon1, sizeof(double), np1, f); rd(lat1, sizeof(double), np1, f); rd(lon2, sizeof(double), np2, f); rd(lat2, sizeof(double), np2, f);
  fclose(f);
  cap = (size_t)get_maxxgrid();                       /* the caller allocates MAXXGRID entries, as the reference's callers do */
  ii = xmalloc(cap * sizeof(int)); ji = xmalloc(cap * sizeof(int)); io = xmalloc(cap * sizeof(int)); jo = xmalloc(cap * sizeof(int));
  area = xmalloc(cap * sizeof(double)); clon = xmalloc(cap * sizeof(double)); clat = xmalloc(cap * sizeof(double));
  mask = xmalloc((size_t)nx1 * ny1 * sizeof(double));
  for (k = 0; k < nx1 * ny1; k++) mask[k] = 1.0;
  f = fopen(argv[2], "wb");
  if (!f) { perror(argv[2]); return 2; }
  n = create_xgrid_2dx2d_order1(&nx1, &ny1, &nx2, &ny2, lon1, lat1, lon2, lat2, mask, ii, ji, io, jo, area);
  wr(&n, sizeof(int), 1, f); wr(ii, sizeof(int), (size_t)n, f); wr(ji, sizeof(int), (size_t)n, f); wr(io, sizeof(int), (size_t)n, f); wr(jo, sizeof(int), (size_t)n, f);
  wr(area, sizeof(double), (size_t)n, f);
  n = create_xgrid_2dx2d_order2(&nx1, &ny1, &nx2, &ny2, lon1, lat1, lon2, lat2, mask, ii, ji, io, jo, area, clon, clat);
  wr(&n, sizeof(int), 1, f); wr(ii, sizeof(int), (size_t)n, f); wr(ji, sizeof(int), (size_t)n, f); wr(io, sizeof(int), (size_t)n, f); wr(jo, sizeof(int), (size_t)n, f);
  wr(area, sizeof(double), (size_t)n, f); wr(clon, sizeof(double), (size_t)n, f); wr(clat, sizeof(double), (size_t)n, f);
  carea = xmalloc((size_t)nx1 * ny1 * sizeof(double));
  get_grid_area(&nx1, &ny1, lon1, lat1, carea);
  wr(carea, sizeof(double), (size_t)nx1 * ny1, f);
  n = create_xgrid_great_circle(&nx1, &ny1, &nx2, &ny2, lon1, lat1, lon2, lat2, mask, ii, ji, io, jo, area, clon, clat);
  wr(&n, sizeof(int), 1, f); wr(ii, sizeof(int), (size_t)n, f); wr(ji, sizeof(int), (size_t)n, f); wr(io, sizeof(int), (size_t)n, f); wr(jo, sizeof(int), (size_t)n, f);
  wr(area, sizeof(double), (size_t)n, f);
  src = xmalloc((size_t)nx1 * ny1 * sizeof(double)); dst = xmalloc((size_t)nx2 * ny2 * sizeof(double));
  for (k = 0; k < nx1 * ny1; k++) src[k] = 1.0 + (k % nx1) + 0.5 * (k / nx1);
  conserve_interp(nx1, ny1, nx2, ny2, lon1, lat1, lon2, lat2, mask, src, dst);
  wr(dst, sizeof(double), (size_t)nx2 * ny2, f);
  fclose(f);
  {
    const double x[4] = {0.1, 0.3, 0.3, 0.1}, y[4] = {0.2, 0.2, 0.5, 0.5};
    printf("poly_area %.17g\n", poly_area(x, y, 4));
  }
  printf("b1_probe ok\n");
  return 0;
}
