// Host check of csrc/fp80.h against native x87 `long double` (test infrastructure; built by tests/test_fp80_host.py).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include "fp80.h"

static x80 from_ld(long double v)
{
  x80 r; uint64_t m; uint16_t se;
  memcpy(&m, &v, 8); memcpy(&se, (char *)&v + 8, 2);
  r.m = m; r.s = se >> 15; r.e = (int)(se & 0x7fff) - 16383;
  if (m == 0) r.e = 0;
  return r;
}
static bool same(x80 a, long double v)
{
  x80 b = from_ld(v);
  if (a.m == 0 && b.m == 0) return true;      // sign of zero: checked separately where it matters
  return a.m == b.m && a.e == b.e && a.s == b.s;
}
static double rnd_double(int spread)
{
  double m = drand48() * 2 - 1;
  int e = (int)(drand48() * spread) - spread / 2;
  return ldexp(m, e);
}
extern "C" long fp80_check_ops(long n, int spread, long *fails)
{
  srand48(12345);
  long bad = 0;
  for (int k = 0; k < 8; k++) fails[k] = 0;
  for (long i = 0; i < n; i++) {
    // build extended operands that are not plain doubles: products/sums of doubles
    double d1 = rnd_double(spread), d2 = rnd_double(spread), d3 = rnd_double(spread), d4 = rnd_double(spread);
    if (i % 7 == 0) d2 = d1 * (1 + ldexp(drand48(), -(int)(drand48() * 60)));   // near-cancellation
    if (i % 11 == 0) d3 = 0.0;
    volatile long double A = (long double)d1 * d3 + (long double)d2, B = (long double)d4 * d2 - (long double)d1;
    x80 a = x80_add(x80_mul(x80_from_double(d1), x80_from_double(d3)), x80_from_double(d2));
    x80 b = x80_sub(x80_mul(x80_from_double(d4), x80_from_double(d2)), x80_from_double(d1));
    if (!same(a, A)) { fails[0]++; bad++; continue; }
    if (!same(b, B)) { fails[0]++; bad++; continue; }
    volatile long double S = A + B, D = A - B, P = A * B;
    if (!same(x80_add(a, b), S)) { fails[1]++; bad++; }
    if (!same(x80_sub(a, b), D)) { fails[2]++; bad++; }
    if (!same(x80_mul(a, b), P)) { fails[3]++; bad++; }
    if (B != 0) { volatile long double Q = A / B; if (!same(x80_div(a, b), Q)) { fails[4]++; bad++; } }
    if (A != 0) { volatile long double Q = 1.0L / A; if (!same(x80_div(x80_from_double(1.0), a), Q)) { fails[4]++; bad++; } }
    { volatile long double R = sqrtl(fabsl(A)); x80 aa = a; aa.s = 0; if (!same(x80_sqrt(aa), R)) { fails[5]++; bad++; } }
    { volatile double t = (double)A; double u = x80_to_double(a); if (memcmp((const void *)&t, &u, 8) != 0 && !(t == 0 && u == 0)) { fails[6]++; bad++; } }
    { bool l1 = fabsl(A) < fabsl(B), l2 = x80_abs_lt(a, b); if (l1 != l2) { fails[7]++; bad++; } }
  }
  return bad;
}
// the reference's intersection solve (mosaic_util.c:967-1044 semantics) in native long double vs the emulation
extern "C" long fp80_check_acosl(long n, long *ulp1)
{
  srand48(777);
  long bad = 0; *ulp1 = 0;
  for (long i = 0; i < n; i++) {
    double x;
    switch (i % 4) {
      case 0: x = drand48() * 2 - 1; break;
      case 1: x = 1 - ldexp(drand48(), -(int)(drand48() * 45)); break;
      case 2: x = -1 + ldexp(drand48(), -(int)(drand48() * 45)); break;
      default: x = ldexp(drand48() * 2 - 1, -(int)(drand48() * 30)); break;
    }
    double ref = (double)acosl((long double)x), got = fg_acosl(x);
    if (ref != got) { bad++; if (fabs(ref - got) <= 4.5e-16 * fabs(ref)) (*ulp1)++; }
  }
  return bad;
}
// fg_acosl (short series + rounding check) against fg_acosl_long (the 13-term series): must agree on every argument; also
// dd_from_x80_pos against dd_from_x80
extern "C" long fp80_check_acosl_fast(long n)
{
  srand48(4711);
  long bad = 0;
  for (long i = 0; i < n; i++) {
    double x;
    switch (i % 6) {
      case 0: x = drand48() * 2 - 1; break;
      case 1: x = 1 - ldexp(drand48(), -(int)(drand48() * 52)); break;
      case 2: x = -1 + ldexp(drand48(), -(int)(drand48() * 52)); break;
      case 3: x = ldexp(drand48() * 2 - 1, -(int)(drand48() * 60)); break;
      case 4: x = cos((i / 6 % 129) * (3.14159265358979323846 / 128) + (drand48() - 0.5) * 1e-9); break;   // next to the table angles
      default: x = cos(ldexp(1.0, -(int)(i / 6 % 40)) * (1 + drand48())); break;                            // results near powers of two
    }
    if (x > 1) x = 1;
    if (x < -1) x = -1;
    if (fg_acosl(x) != fg_acosl_long(x)) bad++;
    x80 y; y.s = 0; y.e = (int)(lrand48() % 80) - 60;
    y.m = (((uint64_t)lrand48() << 33) ^ ((uint64_t)lrand48() << 11) ^ (uint64_t)lrand48()) | 0x8000000000000000ULL;
    if (i % 5 == 0) y.m |= 0x7ff; if (i % 7 == 0) y.m = (y.m & ~0x7ffULL) | 0x400; if (i % 11 == 0) y.m = 0xffffffffffffffffULL - (i % 3);
    const dd2 a = dd_from_x80(y), b = dd_from_x80_pos(y);
    if (a.hi != b.hi || a.lo != b.lo) bad++;
  }
  return bad;
}
// crafted significands for the estimate-and-correct divide and square root: extreme, exact and just-off-exact cases
static long double to_ld(x80 a)
{
  long double v = 0; uint16_t se = (uint16_t)((a.e + 16383) | (a.s << 15));
  if (a.m == 0) se = (uint16_t)(a.s << 15);
  memcpy(&v, &a.m, 8); memcpy((char *)&v + 8, &se, 2);
  return v;
}
extern "C" long fp80_check_edges(long n, long *fails)
{
  srand48(4242);
  long bad = 0; fails[0] = fails[1] = 0;
  const uint64_t special[] = {0x8000000000000000ULL, 0x8000000000000001ULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFEULL,
                              0xFFFFFFFF00000000ULL, 0x80000000FFFFFFFFULL, 0xC000000000000000ULL, 0xB504F333F9DE6484ULL,
                              0xB504F333F9DE6485ULL, 0xAAAAAAAAAAAAAAABULL, 0xFFFFFFFFFFFFF800ULL, 0xFFFFFFFFFFFFFC00ULL};
  const int nsp = sizeof(special) / sizeof(special[0]);
  for (long i = 0; i < n; i++) {
    x80 a, b;
    a.s = 0; b.s = (int)(i & 1);
    a.e = (int)(lrand48() % 41) - 20; b.e = (int)(lrand48() % 41) - 20;
    uint64_t ra = ((uint64_t)lrand48() << 33) ^ ((uint64_t)lrand48() << 11) ^ (uint64_t)lrand48();
    uint64_t rb = ((uint64_t)lrand48() << 33) ^ ((uint64_t)lrand48() << 11) ^ (uint64_t)lrand48();
    a.m = ra | 0x8000000000000000ULL; b.m = rb | 0x8000000000000000ULL;
    switch (i % 8) {
      case 0: a.m = special[(i / 8) % nsp]; break;
      case 1: b.m = special[(i / 8) % nsp]; break;
      case 2: a.m = special[(i / 8) % nsp]; b.m = special[(i / 96) % nsp]; break;
      case 3: { uint64_t r = (rb >> 32) | 0x80000000ULL; a.m = r * r; if (!(a.m >> 63)) a.m <<= 1; } break;       // squares
      case 4: { uint64_t r = (rb >> 32) | 0x80000000ULL; a.m = r * r + ((i & 8) ? 1 : -1); if (!(a.m >> 63)) a.m <<= 1; } break;
      case 5: { uint64_t q = (ra >> 32) | 0x80000000ULL, d = (rb >> 32) | 0x80000000ULL; a.m = q * d; b.m = d << 32; if (!(a.m >> 63)) a.m <<= 1; } break; // exact quotients
      case 6: a.m = b.m; break;
      default: break;
    }
    volatile long double A = to_ld(a), B = to_ld(b);
    { volatile long double Q = A / B; if (!same(x80_div(a, b), Q)) { fails[0]++; bad++; } }
    { volatile long double R = sqrtl(A); if (!same(x80_sqrt(a), R)) { fails[1]++; bad++; } }
    a.e += 1;
    { volatile long double A2 = to_ld(a); volatile long double R = sqrtl(A2); if (!same(x80_sqrt(a), R)) { fails[1]++; bad++; } }
  }
  return bad;
}
