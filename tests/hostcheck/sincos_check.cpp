// Host check of csrc/sincos_glibc.h against the host libm's sin()/cos() (test infrastructure; built by tests/test_sincos_host.py).
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <cstdlib>
#include <cmath>
#include "sincos_glibc.h"
extern "C" long sincos_check(long n, long seed, long *bad_sin, long *bad_cos, double *first_bad)
{
  srand48(seed);
  *bad_sin = *bad_cos = 0; *first_bad = 0;
  for (long i = 0; i < n; i++) {
    double x;
    switch (i % 6) {
      case 0: x = (drand48() * 2 - 1) * 1.5707963267948966; break;        // latitudes
      case 1: x = (drand48() * 2 - 1) * 0.126; break;                      // Taylor branch
      case 2: x = (drand48() * 2 - 1) * 2.42; break;                       // whole supported range
      case 3: x = ldexp(drand48() * 2 - 1, -(int)(drand48() * 40)); break; // tiny (half-differences of latitudes)
      case 4: x = 0.85546875 + ldexp(drand48() - 0.5, -(int)(drand48() * 30)); break;   // branch boundary
      default: x = ((long)(drand48() * 220) - 110) / 128.0 + ldexp(drand48() - 0.5, -(int)(drand48() * 45));   // table nodes
    }
    volatile double xv = x;                       // keep gcc from fusing the two calls below into sincos()
    const double ls = sin(xv), lc = cos(xv);
    if (fgs_sin(x) != ls) { if (!*bad_sin && !*bad_cos) *first_bad = x; (*bad_sin)++; }
    if (fgs_cos(x) != lc) { if (!*bad_sin && !*bad_cos) *first_bad = x; (*bad_cos)++; }
  }
  return *bad_sin + *bad_cos;
}
extern "C" long sincos_check_fused(long n, long seed, double *first_bad)
{
  srand48(seed);
  long bad = 0; *first_bad = 0;
  for (long i = 0; i < n; i++) {
    double x;
    switch (i % 4) {
      case 0: x = (drand48() * 2 - 1) * 1.5707963267948966; break;
      case 1: x = (drand48() * 2 - 1) * 2.42; break;
      case 2: x = ldexp(drand48() * 2 - 1, -(int)(drand48() * 40)); break;
      default: x = ((long)(drand48() * 220) - 110) / 128.0 + ldexp(drand48() - 0.5, -(int)(drand48() * 45));
    }
    double s, c, fs, fc;
    sincos(x, &s, &c);
    fgs_sincos(x, &fs, &fc);
    if (fs != s || fc != c) { if (!bad) *first_bad = x; bad++; }
  }
  return bad;
}
