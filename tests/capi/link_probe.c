/* link_probe.c -- references B1 symbols and two symbols only the reference's archive has (maxval_double, error_handler: what
 * fregrid itself needs from mosaic_util.o); never calls them.  tests/test_capi_c.py links it against libfregrid_hip.so and a
 * static archive of the reference's create_xgrid.o + mosaic_util.o in both orders and reads the bindings off nm / LD_DEBUG. */
#include <stdio.h>
int create_xgrid_2dx2d_order1(const int *, const int *, const int *, const int *, const double *, const double *, const double *,
                              const double *, const double *, int *, int *, int *, int *, double *);
void get_grid_area(const int *, const int *, const double *, const double *, double *);
double poly_area(const double x[], const double y[], int n);
int fix_lon(double x[], double y[], int n, double tlon);
double maxval_double(int size, const double *data);
void error_handler(const char *msg);
int main(int argc, char **argv)
{
  void *volatile refs[6];
  refs[0] = (void *)create_xgrid_2dx2d_order1; refs[1] = (void *)get_grid_area; refs[2] = (void *)poly_area;
  refs[3] = (void *)fix_lon; refs[4] = (void *)maxval_double; refs[5] = (void *)error_handler;
  (void)argv;
  printf("link_probe %d\n", argc + (refs[0] != 0));
  return 0;
}
