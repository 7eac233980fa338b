// Software x87 extended precision (64-bit significand, round-to-nearest-even) and a double-double atan2, for the
// great-circle clip.  The reference solves its plane/segment intersections in `long double`
// (tools/libfrencutils/mosaic_util.c:967-1044: intersect_tri_with_line, invert_matrix_3x3, mult) and takes its
// angles with acosl (mosaic_util.c:829); on x86-64 both are x87 arithmetic.  gfx950 has no 80-bit format, and
// the exchange-cell areas of the great-circle path are only reproducible if every operation rounds exactly where
// the reference's does, so the handful of extended operations are emulated with integer arithmetic:
//   x80_from_double, x80_add/sub, x80_mul, x80_div, x80_sqrt, x80_to_double, magnitude compare
// each bit-identical to the x87 result for finite, non-denormal-extended values (checked against native
// `long double` on the host by tests/test_fp80_host.py).
// acosl: glibc 2.35 x86-64 evaluates fpatan(fsqrt((1-x)*(1+x)), x) in extended precision (verified bit for bit
// against acosl on 5e6 arguments).  The products/roots are emulated exactly; fpatan (<1 ulp extended, not
// reproducible in general) is replaced by an atan2 accurate to ~1e-31 and rounded once to double, which equals
// the reference's double result except when the x87 error straddles a double rounding boundary (measured rate
// in DESIGN.md).
// The header compiles as C++ for the host (tests) and for the device.
#pragma once
#include <stdint.h>
#include <math.h>

#ifdef __HIPCC__
#define FG_HD __host__ __device__ __forceinline__
#define FG_HDN __host__ __device__
#else
#define FG_HD static inline
#define FG_HDN static inline
#endif

// value = (-1)^s * m * 2^(e-63); m has bit 63 set, or m == 0 for zero
struct x80 { uint64_t m; int32_t e; int32_t s; };

FG_HD int x80_clz64(uint64_t v)
{
#ifdef __HIP_DEVICE_COMPILE__
  return __clzll((long long)v);
#else
  return __builtin_clzll(v);
#endif
}

FG_HD void x80_mul64(uint64_t a, uint64_t b, uint64_t *hi, uint64_t *lo)
{
#ifdef __HIP_DEVICE_COMPILE__
  *hi = __umul64hi(a, b);
  *lo = a * b;
#else
  unsigned __int128 p = (unsigned __int128)a * b;
  *hi = (uint64_t)(p >> 64); *lo = (uint64_t)p;
#endif
}

FG_HD x80 x80_zero(int s) { x80 r; r.m = 0; r.e = 0; r.s = s; return r; }

FG_HD x80 x80_from_double(double d)
{
  union { double d; uint64_t u; } c; c.d = d;
  x80 r;
  r.s = (int)(c.u >> 63);
  int ex = (int)((c.u >> 52) & 0x7ff);
  uint64_t fr = c.u & 0xfffffffffffffULL;
  if (ex == 0) {
    if (fr == 0) { r.m = 0; r.e = 0; return r; }
    int lz = x80_clz64(fr);                       // subnormal: value = fr * 2^-1074
    r.m = fr << lz; r.e = -1074 + 63 - lz;
    return r;
  }
  r.m = (fr | 0x10000000000000ULL) << 11;         // value = (2^52+fr) * 2^(ex-1075)
  r.e = ex - 1023;
  return r;
}

// round (m, rest) to nearest even where rest holds the bits below m (rest_hi = next 64 bits, sticky = anything below)
FG_HD x80 x80_round(int s, int e, uint64_t m, uint64_t rest, int sticky)
{
  const uint64_t half = 0x8000000000000000ULL;
  bool up = (rest > half) || (rest == half && (sticky || (m & 1)));
  if (up) { m++; if (m == 0) { m = half; e++; } }
  x80 r; r.m = m; r.e = e; r.s = s;
  return r;
}

FG_HD x80 x80_mul(x80 a, x80 b)
{
  int s = a.s ^ b.s;
  if (a.m == 0 || b.m == 0) return x80_zero(s);
  uint64_t hi, lo;
  x80_mul64(a.m, b.m, &hi, &lo);
  int e = a.e + b.e + 1;
  if (!(hi >> 63)) { hi = (hi << 1) | (lo >> 63); lo <<= 1; e--; }
  return x80_round(s, e, hi, lo, 0);
}

FG_HD x80 x80_neg(x80 a) { a.s ^= 1; return a; }

FG_HD x80 x80_add(x80 a, x80 b)
{
  if (a.m == 0) { if (b.m == 0) { return x80_zero(a.s & b.s); } return b; }
  if (b.m == 0) return a;
  if (b.e > a.e || (b.e == a.e && b.m > a.m)) { x80 t = a; a = b; b = t; }   // |a| >= |b|
  int d = a.e - b.e;
  uint64_t ah = a.m, al = 0, bh, bl;
  int sticky = 0;
  if (d == 0) { bh = b.m; bl = 0; }
  else if (d < 64) { bh = b.m >> d; bl = b.m << (64 - d); }
  else if (d == 64) { bh = 0; bl = b.m; }
  else if (d < 128) { bh = 0; bl = b.m >> (d - 64); sticky = (b.m << (128 - d)) != 0; }
  else { bh = 0; bl = 0; sticky = 1; }
  int e = a.e;
  if (a.s == b.s) {
    uint64_t sl = al + bl, c0 = sl < al;
    uint64_t sh = ah + bh, c1 = sh < ah;
    sh += c0; c1 |= (sh < c0);
    if (c1) {                                       // carry out: shift right one
      sticky |= (int)(sl & 1);
      sl = (sl >> 1) | (sh << 63);
      sh = (sh >> 1) | 0x8000000000000000ULL;
      e++;
    }
    return x80_round(a.s, e, sh, sl, sticky);
  }
  // |a| - |b| on 128 bits; a nonzero sticky means b was slightly larger than (bh:bl)
  uint64_t sl = al - bl, br = al < bl;
  uint64_t sh = ah - bh - br;
  if (sticky) { if (sl == 0) sh--; sl--; }
  if (sh == 0 && sl == 0 && !sticky) return x80_zero(0);      // exact cancellation: +0 in round-to-nearest
  int lz;
  if (sh) lz = x80_clz64(sh); else lz = 64 + x80_clz64(sl);
  if (lz >= 64) { sh = sl << (lz - 64); sl = 0; }
  else if (lz > 0) { sh = (sh << lz) | (sl >> (64 - lz)); sl <<= lz; }
  e -= lz;
  return x80_round(a.s, e, sh, sl, sticky);
}

FG_HD x80 x80_sub(x80 a, x80 b) { return x80_add(a, x80_neg(b)); }

// signed 128-bit -> double without a runtime-library call (magnitudes here are < 2^80; the low word's rounding is
// irrelevant to the estimates it feeds)
FG_HD double x80_i128_to_double(__int128 v)
{
  const int64_t h = (int64_t)(v >> 64);
  const uint64_t l = (uint64_t)v;
  return (double)h * 18446744073709551616.0 + (double)l;
}

// 128 / 64 -> 64-bit quotient and remainder (u1 < v, v normalised).  A double-precision estimate of the quotient
// (good to ~2^-52 relative) is corrected twice with the exact 128-bit remainder: once by a double estimate of the
// remaining quotient (|.| < 2^13), then by +-1 steps until 0 <= rem < v.  Integer-exact; ~5x fewer instructions on
// gfx950 than two emulated 64/32 divisions.
FG_HD uint64_t x80_divlu(uint64_t u1, uint64_t u0, uint64_t v, uint64_t *r)
{
  typedef unsigned __int128 u128;
  typedef __int128 i128;
  const u128 N = ((u128)u1 << 64) | u0;
  const double vd = (double)v;
  const double qd = ((double)u1 * 18446744073709551616.0 + (double)u0) / vd;
  uint64_t q = (qd >= 18446744073709549568.0) ? 0xFFFFFFFFFFFFF800ULL : (uint64_t)qd;
  i128 rem = (i128)(N - (u128)q * v);                        // |rem| < 2^13 * v (wrapping subtraction, small result)
  const double cd = x80_i128_to_double(rem) / vd;
  const int64_t c = (int64_t)cd;                             // truncation: the +-1 steps below finish the job
  q += (uint64_t)c;
  rem -= (i128)c * (i128)v;
  while (rem < 0) { q--; rem += v; }
  while (rem >= (i128)v) { q++; rem -= v; }
  *r = (uint64_t)rem;
  return q;
}

// a / b, correctly rounded
FG_HDN x80 x80_div(x80 a, x80 b)
{
  int s = a.s ^ b.s;
  if (a.m == 0) return x80_zero(s);
  int e = a.e - b.e;
  uint64_t q, r;
  if (a.m >= b.m) q = x80_divlu(a.m >> 1, (a.m & 1) << 63, b.m, &r);      // a.m * 2^63 / b.m  in [2^63, 2^64)
  else { q = x80_divlu(a.m, 0, b.m, &r); e--; }                           // a.m * 2^64 / b.m  in (2^63, 2^64)
  // remaining fraction r / b.m against one half: compare r with b.m - r (2r may not fit)
  const uint64_t rem2 = b.m - r;
  uint64_t rest; int sticky;
  if (r > rem2) { rest = 0x8000000000000001ULL; sticky = 1; }
  else if (r == rem2) { rest = 0x8000000000000000ULL; sticky = 0; }
  else { rest = (r != 0); sticky = (r != 0); }
  return x80_round(s, e, q, rest, sticky);
}

// sqrt, correctly rounded (bit-serial integer root of the 128-bit radicand)
FG_HDN x80 x80_sqrt(x80 a)
{
  if (a.m == 0) return a;
  // value = m * 2^(e-63); make the exponent of the radicand even: rad = m * 2^k (k = 63 or 64), root has 64 bits
  int ex = a.e;
  uint64_t hi, lo;
  if (ex & 1) { hi = a.m; lo = 0; ex -= 1; }             // rad = m * 2^64 * ... (odd exponent: one more factor of two)
  else { hi = a.m >> 1; lo = a.m << 63; }
  // root of (hi:lo) as a 128-bit integer in [2^126, 2^128): 64-bit result.  Double-precision estimate (53 good
  // bits), one Newton correction from the exact 128-bit remainder, then +-1 steps until root^2 <= rad < (root+1)^2.
  typedef unsigned __int128 u128;
  typedef __int128 i128;
  const u128 rad = ((u128)hi << 64) | lo;
  const double rd = sqrt((double)hi * 18446744073709551616.0 + (double)lo);
  uint64_t root = (rd >= 18446744073709549568.0) ? 0xFFFFFFFFFFFFF800ULL : (uint64_t)rd;
  i128 rem = (i128)(rad - (u128)root * root);                // |rem| < 2^13 * 2 * root (wrapping subtraction, small result)
  const int64_t c = (int64_t)(x80_i128_to_double(rem) / (2.0 * (double)root));
  // (root + c)^2 = root^2 + 2 root c + c^2
  rem -= (i128)c * (i128)(2 * (u128)root) + (i128)c * (i128)c;
  root += (uint64_t)c;
  while (rem < 0) { root--; rem += (i128)(2 * (u128)root + 1); }
  while (rem > (i128)(2 * (u128)root)) { rem -= (i128)(2 * (u128)root + 1); root++; }
  const uint64_t remh = (uint64_t)((u128)rem >> 64), reml = (uint64_t)rem;
  // value = root * 2^((ex - 63 - 63)/2 ...): rad = m*2^(63 or 64) = root^2 + rem; sqrt(value) = root * 2^(ex/2 - 63)
  // remainder vs root decides rounding: exact if rem == 0; above half iff rem > root
  uint64_t rest; int sticky;
  if (remh || reml > root) { rest = 0x8000000000000001ULL; sticky = 1; }
  else { rest = (reml != 0); sticky = (reml != 0); }
  return x80_round(0, ex / 2, root, rest, sticky);
}

FG_HD double x80_to_double(x80 a)
{
  if (a.m == 0) return a.s ? -0.0 : 0.0;
  uint64_t m = a.m >> 11, rest = a.m & 0x7ff;
  int e = a.e;
  if (rest > 0x400 || (rest == 0x400 && (m & 1))) { m++; if (m >> 53) { m >>= 1; e++; } }
  double d = ldexp((double)m, e - 52);
  return a.s ? -d : d;
}

// |a| < |b|
FG_HD bool x80_abs_lt(x80 a, x80 b)
{
  if (a.m == 0) return b.m != 0;
  if (b.m == 0) return false;
  return a.e < b.e || (a.e == b.e && a.m < b.m);
}

// ---------------------------------------------------------------------------------------------------- double-double
struct dd2 { double hi, lo; };

FG_HD dd2 dd_fast_two_sum(double a, double b) { dd2 r; r.hi = a + b; r.lo = b - (r.hi - a); return r; }
FG_HD dd2 dd_two_sum(double a, double b)
{
  dd2 r; r.hi = a + b; double bb = r.hi - a; r.lo = (a - (r.hi - bb)) + (b - bb); return r;
}
FG_HD dd2 dd_two_prod(double a, double b) { dd2 r; r.hi = a * b; r.lo = fma(a, b, -r.hi); return r; }
FG_HD dd2 dd_add(dd2 a, dd2 b)
{
  dd2 s = dd_two_sum(a.hi, b.hi), t = dd_two_sum(a.lo, b.lo);
  s.lo += t.hi; s = dd_fast_two_sum(s.hi, s.lo);
  s.lo += t.lo; return dd_fast_two_sum(s.hi, s.lo);
}
FG_HD dd2 dd_neg(dd2 a) { a.hi = -a.hi; a.lo = -a.lo; return a; }
FG_HD dd2 dd_mul(dd2 a, dd2 b)
{
  dd2 p = dd_two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return dd_fast_two_sum(p.hi, p.lo);
}
FG_HD dd2 dd_mul_d(dd2 a, double b)
{
  dd2 p = dd_two_prod(a.hi, b);
  p.lo += a.lo * b;
  return dd_fast_two_sum(p.hi, p.lo);
}
FG_HD dd2 dd_div(dd2 a, dd2 b)
{
  double q1 = a.hi / b.hi;
  dd2 r = dd_add(a, dd_neg(dd_mul_d(b, q1)));
  double q2 = r.hi / b.hi;
  r = dd_add(r, dd_neg(dd_mul_d(b, q2)));
  double q3 = r.hi / b.hi;
  dd2 q = dd_fast_two_sum(q1, q2);
  return dd_add(q, dd2{q3, 0.0});
}

// x80 (64 significant bits) -> exact double-double
FG_HD dd2 dd_from_x80(x80 a)
{
  dd2 r;
  r.hi = x80_to_double(a);
  x80 rem = x80_sub(a, x80_from_double(r.hi));    // exact: at most 11 significant bits
  r.lo = x80_to_double(rem);
  return r;
}

// table of k*pi/64, k = 0..64: angle, cos, sin as double-double (generated by scripts/gen_atan_table.py)
#include "atan_table.h"

// atan2(y, x) for y >= 0 (angle in [0, pi]), ~1e-31 absolute, returned as a normalised double-double
FG_HDN dd2 dd_atan2_pos(dd2 y, dd2 x)
{
  double t0 = atan2(y.hi, x.hi);
  int k = (int)(t0 * (64.0 / 3.14159265358979323846) + 0.5);
  if (k < 0) k = 0;
  if (k > 64) k = 64;
  const dd2 c = {FG_ATAN_TAB[k][2], FG_ATAN_TAB[k][3]}, s = {FG_ATAN_TAB[k][4], FG_ATAN_TAB[k][5]};
  dd2 num = dd_add(dd_mul(y, c), dd_neg(dd_mul(x, s)));
  dd2 den = dd_add(dd_mul(x, c), dd_mul(y, s));
  dd2 t = dd_div(num, den);                        // |t| <= tan(pi/128 + rounding of t0)
  dd2 t2 = dd_mul(t, t);
  // atan t = t * (1 - t2/3 + t2^2/5 - ...), 13 terms: |t2|^13/27 < 1e-43
  dd2 acc = {FG_INV_ODD[12][0], FG_INV_ODD[12][1]};
  for (int j = 11; j >= 0; j--)
    acc = dd_add(dd2{FG_INV_ODD[j][0], FG_INV_ODD[j][1]}, dd_neg(dd_mul(acc, t2)));
  dd2 at = dd_mul(t, acc);
  return dd_add(dd2{FG_ATAN_TAB[k][0], FG_ATAN_TAB[k][1]}, at);
}

// x80 >= 0 -> exact double-double by integer arithmetic (same value as dd_from_x80: hi = the significand rounded to 53 bits,
// lo = the remaining 11 bits, signed)
FG_HD dd2 dd_from_x80_pos(x80 a)
{
  dd2 r;
  if (a.m == 0) { r.hi = 0.0; r.lo = 0.0; return r; }
  uint64_t mh = a.m >> 11;
  const int64_t rest = (int64_t)(a.m & 0x7ff);
  const bool up = rest > 0x400 || (rest == 0x400 && (mh & 1));
  if (up) mh++;
  r.hi = ldexp((double)mh, a.e - 52);
  r.lo = ldexp((double)(rest - (up ? 2048 : 0)), a.e - 63);
  return r;
}

// The double nearest to atan2(y, x), y >= 0, by a short series first: atan t = t + t (-t2/3 + t2^2 (1/5 - t2 (1/7 - ...))) with the
// -t2/3 term in double-double and the rest (< 7.4e-8, six terms, next one < 2e-27) in double -- a relative error below 1e-22 against
// dd_atan2_pos.  When the sum is further than 1e-20 (relative) from the midpoint between two doubles its rounding is that of
// dd_atan2_pos; otherwise (2e-4 of the calls), or when the result is a power of two, dd_atan2_pos decides.
FG_HDN double dd_atan2_pos_rn(dd2 y, dd2 x)
{
  double t0 = atan2(y.hi, x.hi);
  int k = (int)(t0 * (64.0 / 3.14159265358979323846) + 0.5);
  if (k < 0) k = 0;
  if (k > 64) k = 64;
  const dd2 c = {FG_ATAN_TAB[k][2], FG_ATAN_TAB[k][3]}, s = {FG_ATAN_TAB[k][4], FG_ATAN_TAB[k][5]};
  dd2 num = dd_add(dd_mul(y, c), dd_neg(dd_mul(x, s)));
  dd2 den = dd_add(dd_mul(x, c), dd_mul(y, s));
  dd2 t = dd_div(num, den);
  dd2 t2 = dd_mul(t, t);
  const double z = t2.hi;
  const double tail = z * z * (FG_INV_ODD[2][0] - z * (FG_INV_ODD[3][0] - z * (FG_INV_ODD[4][0] - z * (FG_INV_ODD[5][0] -
                      z * (FG_INV_ODD[6][0] - z * FG_INV_ODD[7][0])))));
  dd2 S = dd_add(dd_neg(dd_mul(t2, dd2{FG_INV_ODD[1][0], FG_INV_ODD[1][1]})), dd2{tail, 0.0});
  dd2 at = dd_add(t, dd_mul(t, S));
  dd2 r = dd_add(dd2{FG_ATAN_TAB[k][0], FG_ATAN_TAB[k][1]}, at);
  union { double d; uint64_t u; } h; h.d = r.hi;
  const uint64_t ex = h.u & 0x7ff0000000000000ULL;
  if ((h.u & 0x000fffffffffffffULL) != 0 && ex > (60ULL << 52)) {
    union { double d; uint64_t u; } hu; hu.u = ex - (53ULL << 52);             // half an ulp of r.hi
    if (hu.d - fabs(r.lo) > 1.e-20 * r.hi) return r.hi;
  }
  return dd_atan2_pos(y, x).hi;
}

// (double)acosl((long double)x) as glibc 2.35 / x86-64 computes it, |x| <= 1
FG_HDN double fg_acosl(double x)
{
  const x80 one = x80_from_double(1.0), X = x80_from_double(x);
  x80 y = x80_sqrt(x80_mul(x80_sub(one, X), x80_add(one, X)));      // fsqrt((1-x)*(1+x)), >= 0
  if (y.m == 0) return (x > 0) ? 0.0 : 3.14159265358979323846;      // fpatan(0, +-1) = 0 / pi (rounded to double)
  return dd_atan2_pos_rn(dd_from_x80_pos(y), dd2{x, 0.0});
}
// the same by the long series only (tests compare the two)
FG_HDN double fg_acosl_long(double x)
{
  const x80 one = x80_from_double(1.0), X = x80_from_double(x);
  x80 y = x80_sqrt(x80_mul(x80_sub(one, X), x80_add(one, X)));
  if (y.m == 0) return (x > 0) ? 0.0 : 3.14159265358979323846;
  dd2 r = dd_atan2_pos(dd_from_x80(y), dd2{x, 0.0});
  return r.hi;
}
