// sin / cos of latitude-like arguments (|x| < 2.426) with the operation sequence of the host libm the reference links
// against (glibc 2.35, sysdeps/ieee754/dbl-64/s_sin.c: do_sin / do_cos / TAYLOR_SIN over the 1/128-spaced table of
// sincostab.c), so that poly_area / poly_ctrlon / poly_ctrlat evaluate to the reference's BITS on the device.
// libm has two behaviours on an FMA-capable x86-64 host, and the reference's gcc -O2 object code uses both:
//   sin(), cos()   -> the multiarch FMA build of s_sin.c: the same source with gcc's mul+add contraction
//                     (fgs_sin, fgs_cos below: the contracted operations written out as explicit fma)
//   sincos()       -> no contraction (fgs_sincos); gcc turns sin(a) and cos(a) of one argument in one expression into
//                     this call -- poly_ctrlon's f1/f2, poly_ctrlat's 2*cos(avg_y)+lat2*sin(avg_y) in its general branch,
//                     box_ctrlat/ctrlon -- while poly_area_main and the remaining terms call sin()/cos() separately
// (they differ in the last place for 0.07 % of arguments).  geom.hip.h uses each where the compiled reference does.
// Nothing is taken on trust: tests/test_sincos_host.py compiles this header for the host and requires bit-identical
// results to libm's sin(), cos() and sincos() on tens of millions of arguments across the range.
// The library itself is built with -ffp-contract=off: the only fused operations are the explicit fma() calls here.
#pragma once
#include <stdint.h>
#include <math.h>
#include "sincos_table.h"

#ifndef FG_HD
#ifdef __HIPCC__
#define FG_HD __host__ __device__ __forceinline__
#else
#define FG_HD static inline
#endif
#endif

/* table access: a translation unit may redirect it (e.g. to a copy in LDS) by defining FGS_TAB before including this header */
#ifndef FGS_TAB
#define FGS_TAB(k, j) FG_SINCOS_TAB[k][j]
#endif

#define FGS_BIG 0x1.8p45                     /* ulp = 1/128: big + |x| rounds |x| to a table node */
#define FGS_HP0 1.5707963267948966           /* pi/2 high */
#define FGS_HP1 6.123233995736766e-17        /* pi/2 low  */

FG_HD int fgs_index(double u)               /* low word of big + |x|: the table node, 0..110 for every supported argument */
{
  union { double d; uint64_t b; } c; c.d = u;
  const uint32_t k = (uint32_t)c.b;
  return (int)(k > 111u ? 111u : k);        /* out-of-range or NaN input (rejected elsewhere) must not index past the table */
}

#define FGS_SN3 (-1.66666666666664880952546298448555E-01)
#define FGS_SN5 8.33333214285722277379541354343671E-03
#define FGS_CS2 4.99999999999999999999950396842453E-01
#define FGS_CS4 (-4.16666666666664434524222570944589E-02)
#define FGS_CS6 1.38888874007937613028114285595617E-03
#define FGS_S1 (-0x1.5555555555555p-3)
#define FGS_S2 0x1.1111111110ECEp-7
#define FGS_S3 (-0x1.A01A019DB08B8p-13)
#define FGS_S4 0x1.71DE27B9A7ED9p-19
#define FGS_S5 (-0x1.ADDFFC2FCDF59p-26)

/* FMA = true: the contraction pattern gcc applies to s_sin.c in the multiarch FMA build (libm sin/cos);
 * FMA = false: every operation rounded separately (libm sincos) */
template <bool FMA>
FG_HD double fgs_taylor_sin(double xx, double x, double dx)
{
  if (FMA) {
    double p = fma(FGS_S5, xx, FGS_S4); p = fma(p, xx, FGS_S3); p = fma(p, xx, FGS_S2); p = fma(p, xx, FGS_S1);
    const double r = fma(p, x, -(0.5 * dx));
    return x + fma(r, xx, dx);
  }
  const double poly = ((((FGS_S5 * xx + FGS_S4) * xx + FGS_S3) * xx + FGS_S2) * xx) + FGS_S1;
  const double t = ((poly * x - 0.5 * dx) * xx + dx);
  return x + t;
}

/* ZDX: the caller passes dx = 0 (arguments below 0.855): `0 + t`, `x*0 + v`, `fma(m, P, 0)`, `fma(x, 0, v)` and `xr + 0` are the
 * identity on the values that occur (xr >= +0 by construction, a - a = +0; the one sign-of-zero case, t = -0 with x = 0,
 * still sums to +0), so they are dropped -- which also lets the compiler share xx, the polynomials and c between the sine
 * and the cosine of one argument.  Checked bit for bit against libm like everything else in this file. */
template <bool FMA, bool ZDX>
FG_HD double fgs_do_sin(double x, double dx)
{
  const double xold = x;
  if (fabs(x) < 0.126) return fgs_taylor_sin<FMA>(x * x, x, ZDX ? 0.0 : dx);
  if (!ZDX && x <= 0) dx = -dx;
  const double u = FGS_BIG + fabs(x);
  x = fabs(x) - (u - FGS_BIG);
  const double xx = x * x;
  const int k = fgs_index(u);
  const double sn = FGS_TAB(k, 0), ssn = FGS_TAB(k, 1), cs = FGS_TAB(k, 2), ccs = FGS_TAB(k, 3);
  double s, c, cor;
  if (FMA) {
    const double P = fma(xx, FGS_SN5, FGS_SN3), Q = fma(xx, fma(xx, FGS_CS6, FGS_CS4), FGS_CS2);
    s = ZDX ? x + (x * xx) * P : x + fma(x * xx, P, dx);
    c = ZDX ? xx * Q : fma(x, dx, xx * Q);
    cor = fma(cs, s, fma(-sn, c, fma(s, ccs, ssn)));
  } else {
    const double P = FGS_SN3 + xx * FGS_SN5, Q = FGS_CS2 + xx * (FGS_CS4 + xx * FGS_CS6);
    s = ZDX ? x + x * xx * P : x + (dx + x * xx * P);
    c = ZDX ? xx * Q : x * dx + xx * Q;
    cor = (ssn + s * ccs - sn * c) + cs * s;
  }
  return copysign(sn + cor, xold);
}

template <bool FMA, bool ZDX>
FG_HD double fgs_do_cos(double x, double dx)
{
  if (!ZDX && x < 0) dx = -dx;
  const double u = FGS_BIG + fabs(x);
  x = ZDX ? fabs(x) - (u - FGS_BIG) : fabs(x) - (u - FGS_BIG) + dx;
  const double xx = x * x;
  const int k = fgs_index(u);
  const double sn = FGS_TAB(k, 0), ssn = FGS_TAB(k, 1), cs = FGS_TAB(k, 2), ccs = FGS_TAB(k, 3);
  double s, c, cor;
  if (FMA) {
    s = fma(x * xx, fma(xx, FGS_SN5, FGS_SN3), x);
    c = xx * fma(xx, fma(xx, FGS_CS6, FGS_CS4), FGS_CS2);
    cor = fma(-sn, s, fma(-cs, c, fma(-s, ssn, ccs)));
  } else {
    s = x + x * xx * (FGS_SN3 + xx * FGS_SN5);
    c = xx * (FGS_CS2 + xx * (FGS_CS4 + xx * FGS_CS6));
    cor = (ccs - s * ssn - cs * c) - sn * s;
  }
  return cs + cor;
}

/* libm sin() / cos() for |x| < 2.426265 on an FMA-capable host (larger arguments never occur for latitudes) */
FG_HD double fgs_sin(double x)
{
  const double ax = fabs(x);
  if (ax < 0x1p-26) return x;
  if (ax < 0.85546875) return fgs_do_sin<true, true>(x, 0);
  const double t = FGS_HP0 - ax;
  return copysign(fgs_do_cos<true, false>(t, FGS_HP1), x);
}
FG_HD double fgs_cos(double x)
{
  const double ax = fabs(x);
  if (ax < 0x1p-27) return 1.0;
  if (ax < 0.85546875) return fgs_do_cos<true, true>(x, 0);
  const double y = FGS_HP0 - ax;
  const double a = y + FGS_HP1;
  const double da = (y - a) + FGS_HP1;
  return fgs_do_sin<true, false>(a, da);
}
/* libm sincos() (s_sincos.c): uncontracted; beyond 0.855 both results come from the (a, da) reduction */
FG_HD void fgs_sincos(double x, double *sinx, double *cosx)
{
  const double ax = fabs(x);
  if (ax < 0x1p-27) { *sinx = x; *cosx = 1.0; return; }
  if (ax < 0.85546875) { *sinx = fgs_do_sin<false, true>(x, 0); *cosx = fgs_do_cos<false, true>(x, 0); return; }
  const double y = FGS_HP0 - ax;
  const double a = y + FGS_HP1;
  const double da = (y - a) + FGS_HP1;
  *sinx = copysign(fgs_do_cos<false, false>(a, da), x);
  *cosx = fgs_do_sin<false, false>(a, da);
}
