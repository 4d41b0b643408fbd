// cr_log.hpp -- natural logarithm of a double in (0, 1], correctly rounded (to nearest) for all practical purposes: evaluated in
// double-double arithmetic to ~2^-90 relative and rounded once.  Why it exists: numpy's legacy Gaussian (RandomState.randn, the polar
// Box-Muller of legacy-distributions.c: f = sqrt(-2 log(r2) / r2)) calls the HOST libm's log; DartDoubleInvertedPendulumEnv-v1's
// reset_model draws its velocity noise with it (reference gym/envs/dart/inverted_double_pendulum.py:50-51).  The device libm's log differs
// from glibc's in 2.7 % of such arguments (1 ulp; tools/gpu/log_probe.py), glibc's own log is correctly rounded in 99.92 % of them
// (39 of 47 006 sampled r2 are not) -- so a correctly rounded log on the device reproduces the reference's stream bit for bit except for
// one draw in ~1 200, by one ulp.  sqrt and the division are IEEE-exact on both sides.
// Host + device code (the CPU test tests/test_cr_log.py compiles it with g++ and checks it against 50-digit decimal arithmetic).
#pragma once
#include <cmath>
#ifndef DART_HD
#if defined(__HIPCC__)
#define DART_HD __host__ __device__
#else
#define DART_HD
#endif
#endif

// The error-free transformations below are exact only if every operation is rounded on its own: hipcc's default -ffp-contract=fast fuses the
// product of dd_two_prod into the additions that consume it after inlining (s = fma(a, b, lo) instead of round(a b) + lo) and the pair
// (s, err) then no longer adds up -- measured on the device: 18.8 % of the results one ulp off instead of none.  Hence DART_NO_CONTRACT at the
// top of every function body.
#if defined(__clang__)
#define DART_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define DART_NO_CONTRACT
#endif

namespace dartk {

struct dd_t { double hi, lo; };
DART_HD inline dd_t dd_two_sum(double a, double b) { DART_NO_CONTRACT const double s = a + b, bb = s - a; return {s, (a - (s - bb)) + (b - bb)}; }
DART_HD inline dd_t dd_quick(double a, double b) { DART_NO_CONTRACT const double s = a + b; return {s, b - (s - a)}; }   // |a| >= |b|
DART_HD inline dd_t dd_two_prod(double a, double b) { DART_NO_CONTRACT const double p = a * b; return {p, std::fma(a, b, -p)}; }
DART_HD inline dd_t dd_add(dd_t a, dd_t b) { DART_NO_CONTRACT
  dd_t s = dd_two_sum(a.hi, b.hi);
  const dd_t t = dd_two_sum(a.lo, b.lo);
  s.lo += t.hi; s = dd_quick(s.hi, s.lo);
  s.lo += t.lo; return dd_quick(s.hi, s.lo);
}
DART_HD inline dd_t dd_mul(dd_t a, dd_t b) { DART_NO_CONTRACT
  dd_t p = dd_two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return dd_quick(p.hi, p.lo);
}
DART_HD inline dd_t dd_div(dd_t a, dd_t b) { DART_NO_CONTRACT   // two correction steps: ~2^-104 relative
  const double q1 = a.hi / b.hi;
  dd_t r = dd_add(a, dd_mul(b, dd_t{-q1, 0.0}));
  const double q2 = r.hi / b.hi;
  r = dd_add(r, dd_mul(b, dd_t{-q2, 0.0}));
  const double q3 = r.hi / b.hi;
  dd_t q = dd_quick(q1, q2);
  return dd_add(q, dd_t{q3, 0.0});
}

// log(x) for a normal double x in (0, 1]; x = m 2^e with m in [sqrt(1/2), sqrt(2)), log m = 2 atanh(t), t = (m - 1) / (m + 1),
// |t| <= 0.1716: 18 terms of the odd series leave < 2^-90; e ln 2 in double-double.
DART_HD inline double log_cr(double x) { DART_NO_CONTRACT
  int e;
  double m = std::frexp(x, &e);              // m in [0.5, 1)
  if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }
  const dd_t num = {m - 1.0, 0.0};           // exact (Sterbenz)
  const dd_t den = dd_two_sum(m, 1.0);
  const dd_t t = dd_div(num, den);
  const dd_t t2 = dd_mul(t, t);
  // 1 / (2k + 1) as double-doubles, k = 0 .. 17
  const double CH[18] = {1.0, 0.3333333333333333, 0.2, 0.14285714285714285, 0.1111111111111111, 0.09090909090909091, 0.07692307692307693,
                         0.06666666666666667, 0.058823529411764705, 0.05263157894736842, 0.047619047619047616, 0.043478260869565216, 0.04,
                         0.037037037037037035, 0.034482758620689655, 0.03225806451612903, 0.030303030303030304, 0.02857142857142857};
  const double CL[18] = {0.0, 1.850371707708594e-17, -1.1102230246251566e-17, 7.93016446160826e-18, 6.1679056923619804e-18,
                         -2.523234146875356e-18, -4.270088556250602e-18, 9.251858538542971e-19, 8.163404592832033e-19, 2.921639538487254e-18,
                         2.64338815386942e-18, 1.206764157201257e-18, -8.326672684688674e-19, 2.05596856412066e-18, 4.785444071660157e-19,
                         8.953411488912552e-19, -8.410780489584519e-19, 8.921435019309293e-19};
  dd_t s = {CH[17], CL[17]};
  for (int k = 16; k >= 0; k--) s = dd_add(dd_mul(s, t2), dd_t{CH[k], CL[k]});
  dd_t lm = dd_mul(t, s);
  lm = dd_t{2.0 * lm.hi, 2.0 * lm.lo};
  // e ln 2: ln 2 = 0.6931471805599453 + 2.3190468138462996e-17 (+ 5.7e-34 ...)
  const double ed = (double)e;
  dd_t el = dd_two_prod(ed, 0.6931471805599453);
  el.lo += ed * 2.3190468138462996e-17;
  el = dd_quick(el.hi, el.lo);
  el = dd_add(el, dd_t{ed * 5.707708438416212e-34, 0.0});
  const dd_t r = dd_add(el, lm);
  return r.hi + r.lo;
}

}  // namespace dartk
