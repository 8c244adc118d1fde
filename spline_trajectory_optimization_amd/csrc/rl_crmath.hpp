// rl_crmath.hpp -- correctly rounded heading and normal directions for the reference-order arithmetic mode of the sweep.
//
// The reference samples, per waypoint,  yaw = np.arctan2(y', x')  (models/trajectory.py:250) and intersects the segment
// p +- max_dist * (cos, sin)(yaw +- pi/2) with the boundary rings (models/trajectory.py:87-92, 131-135).  What numpy returns
// for these three functions is the platform libm's value: within 0.55 ulp of the true one on glibc, i.e. usually -- not
// always -- the correctly rounded double.  The platform-INDEPENDENT statement of the same arithmetic is "the correctly
// rounded value", and that is what this header computes, so that the kernel's reference-order mode and the oracle's
// -DORC_LIBM_CR build (libquadmath, 113 bits, then one rounding) return the same bits (tests/test_crmath.py on the host,
// tests/test_hip_parity.py::test_cr_heading_on_the_device on the GPU).
//
// Method (own code; double-double = unevaluated sum hi + lo of two doubles, error-free transformations with fma):
//   theta = atan2(y, x) to ~2^-103:  q = min/max of (|x|, |y|) in [0, 1]; j = rint(64 q); the reduced argument
//           w = (q - j/64) / (1 + q j/64) = (num - c den) / (den + c num) is formed exactly in double-double from the two
//           doubles (|w| <= 1/128); atan w by its Taylor series (the first three coefficients in double-double, the rest
//           in double); + atan(j/64) from a 65-entry double-double table; quadrant by pi/2 - ., pi - . in double-double.
//   yaw   = RN(theta) (one rounding of the double-double).
//   (cos, sin)(theta) = (x, y) / sqrt(x^2 + y^2) in double-double (one Newton step on 1/sqrt from the double value).
//   The reference then forms xL = fl(yaw + fl(pi/2)) and xR = fl(yaw - fl(pi/2)) and takes cos / sin OF THOSE DOUBLES.
//   xL = theta + pi/2 + eta with eta = (yaw - theta) + (fl(pi/2) - pi/2) + (rounding error of the addition), all three
//   known to ~2^-104 (|eta| < 1e-15), so  cos xL = -sin(theta + eta) = -(sin theta + eta cos theta - eta^2/2 sin theta)
//   etc. -- no trigonometric evaluation at all.  Headings exactly on an axis (x' == 0 or y' == 0) are literal constants.
//   Accuracy 2^-100 or better relative to a result of magnitude >= 2^-20, i.e. a wrong rounding needs the true value within
//   2^-100 of the midpoint of two doubles (probability ~2^-46 per call); a heading within 2^-40 rad of an axis, where one
//   component is tiny, keeps ~1 ulp accuracy of that component but may round it the other way.
//
// Plain IEEE double operations only (+ - * / sqrt fma): the same source gives the same bits on the host (tests) and on
// the device.  Everything is compiled with floating-point contraction OFF; the fma calls are explicit.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RL_CR_FN __device__ __forceinline__
#else
#define RL_CR_FN inline
#endif
#if defined(__clang__)
#define RL_CR_STRICT _Pragma("clang fp contract(off)")
#else
#define RL_CR_STRICT   /* host tests: compiled with -ffp-contract=off */
#endif

namespace rl {
namespace cr {

struct dd { double hi, lo; };

RL_CR_FN dd two_sum(double a, double b) {
  RL_CR_STRICT
  const double s = a + b, bb = s - a;
  return dd{s, (a - (s - bb)) + (b - bb)};
}
RL_CR_FN dd fast_two_sum(double a, double b) {   // |a| >= |b| or a == 0
  RL_CR_STRICT
  const double s = a + b;
  return dd{s, b - (s - a)};
}
RL_CR_FN dd two_prod(double a, double b) {
  RL_CR_STRICT
  const double p = a * b;
  return dd{p, fma(a, b, -p)};
}
RL_CR_FN dd dd_add(dd a, dd b) {
  RL_CR_STRICT
  dd s = two_sum(a.hi, b.hi);
  const dd t = two_sum(a.lo, b.lo);
  s = fast_two_sum(s.hi, s.lo + t.hi);
  return fast_two_sum(s.hi, s.lo + t.lo);
}
RL_CR_FN dd dd_add_d(dd a, double b) {
  RL_CR_STRICT
  const dd s = two_sum(a.hi, b);
  return fast_two_sum(s.hi, s.lo + a.lo);
}
RL_CR_FN dd dd_neg(dd a) { return dd{-a.hi, -a.lo}; }
RL_CR_FN dd dd_mul(dd a, dd b) {
  RL_CR_STRICT
  const dd p = two_prod(a.hi, b.hi);
  return fast_two_sum(p.hi, fma(a.hi, b.lo, fma(a.lo, b.hi, p.lo)));
}
RL_CR_FN dd dd_mul_d(dd a, double b) {
  RL_CR_STRICT
  const dd p = two_prod(a.hi, b);
  return fast_two_sum(p.hi, fma(a.lo, b, p.lo));
}
RL_CR_FN dd dd_div(dd a, dd b) {
  RL_CR_STRICT
  const double q1 = a.hi / b.hi;
  const dd p = two_prod(q1, b.hi);
  const dd d = two_sum(a.hi, -p.hi);                       // a.hi - q1 b.hi, exact
  const double r = ((d.hi + (d.lo - p.lo)) + a.lo) - q1 * b.lo;
  return fast_two_sum(q1, r / b.hi);
}

// atan(j/64), j = 0 .. 64, as double-double (tools/gen_crmath_tables.c, libquadmath)
#if defined(__HIPCC__)
__device__
#endif
static const double kAtanTab[65][2] = {
  {0x0p+0, 0x0p+0},  {0x1.fff555bbb729bp-7, -0x1.220c39d4dff5p-61},
  {0x1.ffd55bba97625p-6, -0x1.5ec431444912cp-60},  {0x1.7fb818430da2ap-5, -0x1.86ef8f794f105p-63},
  {0x1.ff55bb72cfdeap-5, -0x1.c934d86d23f1dp-60},  {0x1.3f59f0e7c559dp-4, 0x1.ac4ce285df847p-58},
  {0x1.7ee182602f10fp-4, -0x1.cfb654c0c3d98p-58},  {0x1.be39ebe6f07c3p-4, 0x1.f7b8f29a05987p-58},
  {0x1.fd5ba9aac2f6ep-4, -0x1.cd37686760c17p-59},  {0x1.1e1fafb043727p-3, -0x1.b485914dacf8cp-59},
  {0x1.3d6eee8c6626cp-3, 0x1.61a3b0ce9281bp-57},  {0x1.5c9811e3ec26ap-3, -0x1.054ab2c010f3dp-58},
  {0x1.7b97b4bce5b02p-3, 0x1.347b0b4f881cap-58},  {0x1.9a6a8e96c8626p-3, 0x1.cf601e7b4348ep-59},
  {0x1.b90d7529260a2p-3, 0x1.17b10d2e0e5aap-61},  {0x1.d77d5df205736p-3, 0x1.c648d1534597ep-57},
  {0x1.f5b75f92c80ddp-3, 0x1.8ab6e3cf7afbdp-57},  {0x1.09dc597d86362p-2, 0x1.62e47390cb865p-56},
  {0x1.18bf5a30bf178p-2, 0x1.30ca4748b1bf8p-57},  {0x1.278372057ef46p-2, -0x1.077cdd36dfc81p-56},
  {0x1.362773707ebccp-2, -0x1.963a544b672d8p-57},  {0x1.44aa436c2af0ap-2, -0x1.5d5e43c55b3bap-56},
  {0x1.530ad9951cd4ap-2, -0x1.2566480884082p-57},  {0x1.614840309cfe2p-2, -0x1.a725715711fp-56},
  {0x1.6f61941e4def1p-2, -0x1.c63aae6f6e918p-56},  {0x1.7d5604b63b3f7p-2, 0x1.69c885c2b249ap-56},
  {0x1.8b24d394a1b25p-2, 0x1.b6d0ba3748fa8p-56},  {0x1.98cd5454d6b18p-2, 0x1.9e6c988fd0a77p-56},
  {0x1.a64eec3cc23fdp-2, -0x1.24dec1b50b7ffp-56},  {0x1.b3a911da65c6cp-2, 0x1.ae187b1ca504p-56},
  {0x1.c0db4c94ec9fp-2, -0x1.cc1ce70934c34p-56},  {0x1.cde53432c1351p-2, -0x1.a2cfa4418f1adp-56},
  {0x1.dac670561bb4fp-2, 0x1.a2b7f222f65e2p-56},  {0x1.e77eb7f175a34p-2, 0x1.0e53dc1bf3435p-56},
  {0x1.f40dd0b541418p-2, -0x1.a3992dc382a23p-57},  {0x1.0039c73c1a40cp-1, -0x1.b32c949c9d593p-55},
  {0x1.0657e94db30dp-1, -0x1.d5b495f6349e6p-56},  {0x1.0c6145b5b43dap-1, 0x1.974fa13b5404fp-58},
  {0x1.1255d9bfbd2a9p-1, -0x1.2bdaee1c0ee35p-58},  {0x1.1835a88be7c13p-1, 0x1.c621cec00c301p-55},
  {0x1.1e00babdefeb4p-1, -0x1.928df287a668fp-58},  {0x1.23b71e2cc9e6ap-1, 0x1.c421c9f38224ep-57},
  {0x1.2958e59308e31p-1, -0x1.09e73b0c6c087p-56},  {0x1.2ee628406cbcap-1, 0x1.c5d5e9ff0cf8dp-55},
  {0x1.345f01cce37bbp-1, 0x1.1021137c71102p-55},  {0x1.39c391cd4171ap-1, -0x1.2304331d8bf46p-55},
  {0x1.3f13fb89e96f4p-1, 0x1.ecf8b492644fp-56},  {0x1.445065b795b56p-1, -0x1.f76d0163f79c8p-56},
  {0x1.4978fa3269ee1p-1, 0x1.2419a87f2a458p-56},  {0x1.4e8de5bb6ec04p-1, 0x1.4a33dbeb3796cp-55},
  {0x1.538f57b89061fp-1, -0x1.1bb74abda520cp-55},  {0x1.587d81f732fbbp-1, -0x1.5e5c9d8c5a95p-56},
  {0x1.5d58987169b18p-1, 0x1.0028e4bc5e7cap-57},  {0x1.6220d115d7b8ep-1, -0x1.2b785350ee8c1p-57},
  {0x1.66d663923e087p-1, -0x1.6ea6febe8bbbap-56},  {0x1.6b798920b3d99p-1, -0x1.a80386188c50ep-55},
  {0x1.700a7c5784634p-1, -0x1.8c34d25aadef6p-56},  {0x1.748978fba8e0fp-1, 0x1.7b2a6165884a2p-59},
  {0x1.78f6bbd5d315ep-1, 0x1.406a08980374p-55},  {0x1.7d528289fa093p-1, 0x1.560821e2f3aa9p-55},
  {0x1.819d0b7158a4dp-1, -0x1.bf76229d3b917p-56},  {0x1.85d69576cc2c5p-1, 0x1.6b66e7fc8b8c4p-57},
  {0x1.89ff5ff57f1f8p-1, -0x1.55b9a5e177a1bp-55},  {0x1.8e17aa99cc05ep-1, -0x1.ec182ab042f61p-56},
  {0x1.921fb54442d18p-1, 0x1.1a62633145c07p-55},
};

// atan(num / den), 0 <= num <= den, den > 0 finite, as a double-double
RL_CR_FN dd atan_ratio(double num, double den, const double (*tab)[2]) {
  RL_CR_STRICT
  const dd kThird{0x1.5555555555555p-2, 0x1.5555555555555p-56};
  const dd kFifth{0x1.999999999999ap-3, -0x1.999999999999ap-57};
  const dd kSeventh{0x1.2492492492492p-3, 0x1.2492492492492p-57};
  const double q0 = num / den;
  const int j = (int)rint(q0 * 64.0);
  const double c = (double)j * 0.015625;
  // w = (num - c den) / (den + c num), both formed exactly (c has 7 bits)
  const dd p = two_prod(c, den);
  const dd h = two_sum(num, -p.hi);
  const dd nn = fast_two_sum(h.hi, h.lo - p.lo);
  const dd p2 = two_prod(c, num);
  const dd s = two_sum(den, p2.hi);
  const dd dn = fast_two_sum(s.hi, s.lo + p2.lo);
  const dd w = dd_div(nn, dn);
  // atan w = w - w^3/3 + w^5/5 - w^7/7 + w^9/9 - ... ; |w| <= 2^-7 + : the terms from w^9 on in double
  const dd w2 = dd_mul(w, w);
  const double u = w2.hi;
  const double tail = u * (1.0 / 9.0 + u * (-1.0 / 11.0 + u * (1.0 / 13.0 + u * (-1.0 / 15.0))));
  dd pl = dd_add_d(dd_neg(kSeventh), tail);          // -1/7 + w^2/9 - ...
  pl = dd_add(kFifth, dd_mul(w2, pl));               //  1/5 + w^2 (...)
  pl = dd_add(dd_neg(kThird), dd_mul(w2, pl));       // -1/3 + w^2 (...)
  const dd corr = dd_mul(dd_mul(w, w2), pl);         //  w^3 (...)
  dd a = dd_add(w, corr);
  return dd_add(dd{tab[j][0], tab[j][1]}, a);
}

struct Heading {
  double yaw;        // RN(atan2(y, x))
  double cl, sl;     // RN(cos(fl(yaw + fl(pi/2)))), RN(sin(.))   -- trajectory.py:87-90 with norm = +pi/2
  double cr, sr;     // RN(cos(fl(yaw - fl(pi/2)))), RN(sin(.))   --                     norm = -pi/2
};

// the heading of the tangent (x, y) = (x'(u), y'(u)) and the two normal directions, every output correctly rounded
RL_CR_FN Heading heading(double x, double y, const double (*tab)[2] = kAtanTab) {
  RL_CR_STRICT
  const double kPio2Hi = 0x1.921fb54442d18p+0, kPio2Lo = 0x1.1a62633145c07p-54;
  const double kPiHi = 0x1.921fb54442d18p+1, kPiLo = 0x1.1a62633145c07p-53;
  const double kE1 = 0x1.1a62633145c07p-54;    // cos(fl(pi/2))
  const double kE2 = 0x1.1a62633145c07p-53;    // sin(fl(pi))
  const double kE3 = 0x1.a79394c9e8a0ap-53;    // -cos(fl(fl(pi) + fl(pi/2)))
  Heading o;
  if (!(fabs(x) <= 0x1.fffffffffffffp+1023) || !(fabs(y) <= 0x1.fffffffffffffp+1023)) {   // inf / nan: no heading
    o.yaw = o.cl = o.sl = o.cr = o.sr = NAN;
    return o;
  }
  if (y == 0.0) {            // on the x axis (IEEE atan2: (+-0, x >= +0) -> +-0, (+-0, x <= -0) -> +-pi)
    if (!(copysign(1.0, x) < 0.0)) { o.yaw = y; o.cl = kE1; o.sl = 1.0; o.cr = kE1; o.sr = -1.0; }
    else if (!(copysign(1.0, y) < 0.0)) { o.yaw = kPiHi; o.cl = -kE3; o.sl = -1.0; o.cr = kE1; o.sr = 1.0; }
    else { o.yaw = -kPiHi; o.cl = kE1; o.sl = -1.0; o.cr = -kE3; o.sr = 1.0; }
    return o;
  }
  if (x == 0.0) {            // on the y axis
    if (y > 0.0) { o.yaw = kPio2Hi; o.cl = -1.0; o.sl = kE2; o.cr = 1.0; o.sr = 0.0; }
    else { o.yaw = -kPio2Hi; o.cl = 1.0; o.sl = 0.0; o.cr = -1.0; o.sr = -kE2; }
    return o;
  }
  // scale away from over/underflow of x^2 + y^2 (powers of two: exact)
  {
    const double m = fmax(fabs(x), fabs(y));
    if (m > 0x1p+500) { x *= 0x1p-600; y *= 0x1p-600; }
    else if (m < 0x1p-500) { x *= 0x1p+600; y *= 0x1p+600; }
  }
  const double ax = fabs(x), ay = fabs(y);
  const bool swap = ay > ax;
  dd th = atan_ratio(swap ? ax : ay, swap ? ay : ax, tab);          // in [0, pi/4]
  if (swap) th = dd_add(dd{kPio2Hi, kPio2Lo}, dd_neg(th));           // pi/2 - .
  if (x < 0.0) th = dd_add(dd{kPiHi, kPiLo}, dd_neg(th));            // pi - .
  if (y < 0.0) th = dd_neg(th);
  o.yaw = th.hi + th.lo;
  const double th_rest = (th.hi - o.yaw) + th.lo;                     // theta - yaw (th is normalised: th.hi == yaw)
  // (cos, sin)(theta) = (x, y) / sqrt(x^2 + y^2)
  const dd xx = two_prod(x, x), yy = two_prod(y, y);
  dd s2 = two_sum(xx.hi, yy.hi);
  s2 = fast_two_sum(s2.hi, s2.lo + (xx.lo + yy.lo));
  const double y0 = 1.0 / sqrt(s2.hi);
  const dd m = dd_mul(s2, two_prod(y0, y0));
  const double e = (1.0 - m.hi) - m.lo;                               // 1 - s2 y0^2, ~2^-52
  const dd rinv = fast_two_sum(y0, y0 * (0.5 * e + 0.375 * e * e));
  const dd ct = dd_mul_d(rinv, x), st = dd_mul_d(rinv, y);
  // xL = fl(yaw + fl(pi/2)) = theta + pi/2 + etaL ;  xR = fl(yaw - fl(pi/2)) = theta - pi/2 + etaR
  const dd aL = two_sum(o.yaw, kPio2Hi), aR = two_sum(o.yaw, -kPio2Hi);
  const double etaL = (-th_rest - kPio2Lo) - aL.lo;
  const double etaR = (-th_rest + kPio2Lo) - aR.lo;
  // cos xL = -sin(theta + etaL), sin xL = cos(theta + etaL); cos xR = sin(theta + etaR), sin xR = -cos(theta + etaR)
  const dd sL = dd_add_d(st, etaL * ct.hi - 0.5 * etaL * etaL * st.hi);
  const dd cL = dd_add_d(ct, -(etaL * st.hi) - 0.5 * etaL * etaL * ct.hi);
  const dd sR = dd_add_d(st, etaR * ct.hi - 0.5 * etaR * etaR * st.hi);
  const dd cR = dd_add_d(ct, -(etaR * st.hi) - 0.5 * etaR * etaR * ct.hi);
  o.cl = -(sL.hi + sL.lo); o.sl = cL.hi + cL.lo;
  o.cr = sR.hi + sR.lo;    o.sr = -(cR.hi + cR.lo);
  return o;
}

// ---- two-stage evaluation (round 6) ------------------------------------------------------------------------------------
// heading() above costs ~600 instructions per sample because every intermediate is a full double-double.  Almost always a
// much cheaper evaluation DECIDES the five roundings: heading_fast() computes theta = atan2(y, x) and the four direction
// components to about 2^-72 (absolute; far better near an axis) together with a rigorous bound on its own error, and
// returns true only if, for every output, the two ends of the error interval round to the same double (Ziv's rounding
// test).  Then the outputs ARE the correctly rounded values.  Otherwise (about 1 sample in 10^4; always for tangents on or
// extremely near an axis, non-finite or of extreme magnitude) it returns false and the caller evaluates heading().
//
// Method.  theta: the same reduction w = (num - c den) / (den + c num), c = j/64, with numerator and denominator exact as
// double-doubles; the quotient as w = wh + wl through ONE reciprocal (float seed + a Newton step, relative error rho <=
// 2^-44): wh = fl(nh R), wl = (nh - wh dh + nl - wh dl) R, error <= 2^-88 |w| + 2^-96; atan w = w + corr with
// corr = w^3 P(w^2) in plain double (|corr| <= 2^-22.6, relative error <= 5 ulp: the term that sets the bound);
// theta = K +- (T_j + w + corr) with sloppy double-double additions (errors ~2^-105).
//     |theta_true - (th.hi + th.lo)| <= e_th = 2^-50 |corr| + 2^-85 |w| + 2^-100.
// (cos, sin) theta = (x, y) / sqrt(x^2 + y^2): x^2 + y^2 as a double-double, reciprocal square root y1 from a float seed and
// one Newton step in double, the residual e = 1 - s y1^2 formed to ~2^-104, rinv = y1 + y1 e/2; relative error < 2^-100.
// The four outputs are  sin / cos (theta + eta)  with the same eta as in heading(), |eta| < 2^-51, known to e_th:
//     |V_true - (vh + vl)| <= e_th |other component| + 2^-98 |this component|.
// The seeds are the only place where host and device differ (v_rcp_f32 / v_rsq_f32, 1 ulp, against 1.0f / x and
// 1.0f / sqrtf(x)); the bounds hold for any seed within 2^-22 and tests/crmath_check.cpp perturbs the seeds by that much.
#if defined(RL_CR_SEED_PERTURB)
static double g_seed_perturb = 1.0;   // host tests only: multiplies both seeds (1 +- 2^-22.5)
#define RL_CR_PERTURB(v) ((v) * g_seed_perturb)
#else
#define RL_CR_PERTURB(v) (v)
#endif
// The three parts of heading_fast (direction, angle, roundings) are independent until the end; left alone, the scheduler
// interleaves them and the kernel that inlines it runs out of its 128 registers.  Device only: keep them in source order.
#if defined(__HIP_DEVICE_COMPILE__)
#define RL_CR_SEQ() __builtin_amdgcn_sched_barrier(0)
#define RL_CR_PIN6(a, b, c, d, e, f) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f))
#else
#define RL_CR_SEQ() do { } while (0)
#define RL_CR_PIN6(a, b, c, d, e, f) do { } while (0)
#endif
RL_CR_FN double seed_rcp(double d) {      // 1 / d within 2^-22, d in the float range
#if defined(__HIP_DEVICE_COMPILE__)
  return (double)__builtin_amdgcn_rcpf((float)d);
#else
  return RL_CR_PERTURB((double)(1.0f / (float)d));
#endif
}
RL_CR_FN double seed_rsqrt(double d) {    // 1 / sqrt(d) within 2^-22, d in the float range
#if defined(__HIP_DEVICE_COMPILE__)
  return (double)__builtin_amdgcn_rsqf((float)d);
#else
  return RL_CR_PERTURB((double)(1.0f / sqrtf((float)d)));
#endif
}

// diag (host tests): [0,1] th.hi, th.lo  [2] e_th  [3 + 3 q .. ] vh, vl, E of output q = cl, sl, cr, sr (before the sign)
struct NoHook { RL_CR_FN void operator()() const {} };
// after_table(): called once the look-up of atan(j / 64) has ARRIVED (it travels during the direction part) -- the sweep kernel
// requests its ring stretches there, and they travel during the angle and the roundings
template <typename Hook = NoHook>
RL_CR_FN bool heading_fast(double x, double y, Heading& o, const double (*tab)[2] = kAtanTab, double* diag = nullptr,
                           Hook&& after_table = Hook()) {
  RL_CR_STRICT
  const double kPio2Hi = 0x1.921fb54442d18p+0, kPio2Lo = 0x1.1a62633145c07p-54;
  const double kPiHi = 0x1.921fb54442d18p+1, kPiLo = 0x1.1a62633145c07p-53;
  const double ax = fabs(x), ay = fabs(y);
  const bool swap = ay > ax;
  const double den = swap ? ay : ax, num = swap ? ax : ay;
  // ordinary tangents only: finite, not on an axis, |min / max| >= 2^-40, max component within 2^+-60
  // (no early return: anything else flows through as garbage or NaN and is rejected at the end, without a divergent branch)
  const bool ordinary = den >= 0x1p-60 && den <= 0x1p+60 && num >= den * 0x1p-40;
  // ---- the table entry of theta, requested first
  const float qf = (float)num * (float)seed_rcp(den);
  int j = (int)rintf(qf * 64.0f);
  j = j < 0 ? 0 : (j > 64 ? 64 : j);
  double Tj = tab[j][0], Tjl = tab[j][1];
  RL_CR_SEQ();
  // ---- (cos, sin) theta
  const dd dd2 = two_prod(den, den), nn2 = two_prod(num, num);
  const dd s2 = fast_two_sum(dd2.hi, nn2.hi);
  const double s2l = s2.lo + (dd2.lo + nn2.lo);
  const double y0 = seed_rsqrt(s2.hi);
  const double e0 = fma(-(s2.hi * y0), y0, 1.0);
  const double y1 = fma(y0, e0 * fma(0.375, e0, 0.5), y0);
  const dd yy = two_prod(y1, y1);
  const double mh = s2.hi * yy.hi;
  const double ml = fma(s2.hi, yy.hi, -mh) + fma(s2l, yy.hi, s2.hi * yy.lo);
  const double e = (1.0 - mh) - ml;                          // 1 - s2 y1^2
  const double rl = y1 * (0.5 * e);
  double cth = y1 * x, ctl = fma(rl, x, fma(y1, x, -cth));
  double sth = y1 * y, stl = fma(rl, y, fma(y1, y, -sth));
  RL_CR_SEQ();
  // the table entry has to be IN its registers before the hook runs: the compiler waits for every outstanding vector-memory
  // operation in front of a value that was requested before an LDS-DMA, so a look-up still in flight would wait for the stretches
  // (the direction part is named too: left free, it sinks below the hook and the look-up's flight time is spent waiting)
  RL_CR_PIN6(Tj, Tjl, cth, ctl, sth, stl);
  after_table();
  RL_CR_SEQ();
  // ---- theta
  const double c = (double)j * 0.015625;
  const dd p = two_prod(c, den);
  const dd h = two_sum(num, -p.hi);
  const dd nn = two_sum(h.hi, h.lo - p.lo);                 // num - c den, exact
  const dd p2 = two_prod(c, num);
  const dd sd = fast_two_sum(den, p2.hi);
  const double dh = sd.hi, dl = sd.lo + p2.lo;              // den + c num
  const double r0 = seed_rcp(dh);
  const double R = fma(r0, fma(-dh, r0, 1.0), r0);
  const double wh0 = nn.hi * R;
  const double rr = fma(-wh0, dh, nn.hi);
  const double wl0 = fma(-wh0, dl, rr + nn.lo) * R;
  const dd w = fast_two_sum(wh0, wl0);
  const double u2 = w.hi * w.hi;
  // atan w - w = -(w^3 / 3) (1 - 3/5 w^2 (1 - 5/7 w^2 (1 - 7/9 w^2 (1 - 9/11 w^2)))): every addend is the inline constant 1.0.
  // (In Horner form with the constants 1/9, -1/7, 1/5, -1/3 as addends the kernel that inlines this kept four of them in
  // registers across its loop, spilled them, and reloaded each from scratch in the middle of the chain: 1.2 ms of 14.)
  const double q4 = fma(u2 * (-9.0 / 11.0), 1.0, 1.0);
  const double q3 = fma(u2 * (-7.0 / 9.0), q4, 1.0);
  const double q2 = fma(u2 * (-5.0 / 7.0), q3, 1.0);
  const double q1 = fma(u2 * (-3.0 / 5.0), q2, 1.0);
  const double corr = ((w.hi * u2) * (-1.0 / 3.0)) * q1;
  const dd a0 = fast_two_sum(Tj, w.hi);                     // T_j >= |w| for j >= 1, T_0 = 0
  const dd a1 = fast_two_sum(a0.hi, corr);
  dd th = fast_two_sum(a1.hi, a1.lo + (a0.lo + (Tjl + w.lo)));
  // theta = sign(y) (K + sigma th):  (0, +) | x < 0: (pi, -) | swapped: (pi/2, -) | swapped and x < 0: (pi/2, +)
  const bool xneg = x < 0.0;
  const double Kh = swap ? kPio2Hi : (xneg ? kPiHi : 0.0), Kl = swap ? kPio2Lo : (xneg ? kPiLo : 0.0);
  const bool minus = swap != xneg;
  const dd qd = fast_two_sum(Kh, minus ? -th.hi : th.hi);
  th = fast_two_sum(qd.hi, qd.lo + (Kl + (minus ? -th.lo : th.lo)));
  if (y < 0.0) th = dd_neg(th);
  const double e_th = fma(0x1p-50, fabs(corr), fma(0x1p-85, fabs(w.hi), 0x1p-100));
  const double yaw_m = th.hi + (th.lo - e_th), yaw_p = th.hi + (th.lo + e_th);
  bool ok = yaw_m == yaw_p;
  o.yaw = yaw_m;
  const double delta = (th.hi - yaw_m) + th.lo;             // theta - yaw
  RL_CR_SEQ();
  // ---- the four components:  xL = fl(yaw + fl(pi/2)) = theta + pi/2 + etaL,  xR = fl(yaw - fl(pi/2)) = theta - pi/2 + etaR
  const dd aL = two_sum(yaw_m, kPio2Hi), aR = two_sum(yaw_m, -kPio2Hi);
  const double etaL = (-delta - kPio2Lo) - aL.lo;
  const double etaR = (-delta + kPio2Lo) - aR.lo;
  auto decide = [&](double hi, double lo, double other, double eta, double& out, int slot) {
    const dd v = fast_two_sum(hi, fma(eta, other, lo));     // sin / cos (theta + eta) = this + eta * other - ...
    const double E = fma(e_th, fabs(other), 0x1p-98 * fabs(hi));
    const double r_m = v.hi + (v.lo - E), r_p = v.hi + (v.lo + E);
    out = r_m;
    if (diag) { diag[3 + 3 * slot] = v.hi; diag[4 + 3 * slot] = v.lo; diag[5 + 3 * slot] = E; }
    return r_m == r_p;
  };
  double sL, cL, sR, cR;
  ok = decide(sth, stl, cth, etaL, sL, 0) && ok;            // cos xL = -sin(theta + etaL)
  ok = decide(cth, ctl, sth, -etaL, cL, 1) && ok;           // sin xL =  cos(theta + etaL)
  ok = decide(sth, stl, cth, etaR, sR, 2) && ok;            // cos xR =  sin(theta + etaR)
  ok = decide(cth, ctl, sth, -etaR, cR, 3) && ok;           // sin xR = -cos(theta + etaR)
  o.cl = -sL; o.sl = cL; o.cr = sR; o.sr = -cR;
  if (diag) { diag[0] = th.hi; diag[1] = th.lo; diag[2] = ordinary ? e_th : 0.0; }
  return ok && ordinary;
}

// the product's entry point: the cheap stage, the full one where it cannot decide
RL_CR_FN Heading heading2(double x, double y, const double (*tab)[2] = kAtanTab) {
  Heading o;
  if (!heading_fast(x, y, o, tab)) o = heading(x, y, tab);
  return o;
}

}  // namespace cr
}  // namespace rl
