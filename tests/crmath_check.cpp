// tests/crmath_check.cpp -- host check of csrc/rl_crmath.hpp against libquadmath (compiled and run by tests/test_crmath.py:
//   g++ -O2 -std=c++17 -ffp-contract=off crmath_check.cpp -lquadmath).  Prints, per input family, how many of the five
// outputs of rl::cr::heading differ from the correctly rounded values RN(atan2), RN(cos / sin(fl(yaw +- fl(pi/2)))).
#include <quadmath.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#define RL_CR_SEED_PERTURB 1
#include "../spline_trajectory_optimization_amd/csrc/rl_crmath.hpp"

static uint64_t g_s = 0x9E3779B97F4A7C15ULL;
static double urand() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return (double)(g_s >> 11) * (1.0 / 9007199254740992.0); }
static bool same(double a, double b) { return std::memcmp(&a, &b, 8) == 0 || (a != a && b != b); }

// two-stage accounting (round 6): heading_fast() must be RIGHT whenever it says it is sure, and its error bounds must hold
static long g_fast_calls = 0, g_fast_sure = 0, g_fast_bad = 0, g_two_stage_bad = 0;
static double g_ratio_th = 0.0, g_ratio_out = 0.0;   // max over all calls of (actual error) / (claimed bound)
static long g_hard[3] = {0, 0, 0};                    // outputs within 2^-60 / 2^-66 / 2^-72 (relative) of a rounding boundary
static int g_perturb = 0;
static long g_fam_calls = 0, g_fam_sure = 0;
static void family_done(const char* name) {
  std::printf("%-9s fast stage sure on %.4f %% of %ld\n", name, 100.0 * (double)(g_fast_sure - g_fam_sure) / (double)(g_fast_calls - g_fam_calls), g_fast_calls - g_fam_calls);
  g_fam_calls = g_fast_calls; g_fam_sure = g_fast_sure;
}

static double boundary_distance(__float128 v) {   // distance of v from the nearest midpoint of two doubles, in ulps of RN(v)
  const double d = (double)v;
  const double up = nextafter(d, INFINITY), dn = nextafter(d, -INFINITY);
  const __float128 mu = ((__float128)d + (__float128)up) / 2, ml = ((__float128)d + (__float128)dn) / 2;
  const __float128 a = fabsq(v - mu), b = fabsq(v - ml);
  return (double)((a < b ? a : b) / (__float128)(up - d));
}

static long check(double x, double y, long* bad) {
  const rl::cr::Heading h = rl::cr::heading(x, y);
  const __float128 thq = atan2q((__float128)y, (__float128)x);
  const double yaw = (double)thq;
  const double xl = yaw + M_PI / 2.0, xr = yaw + (-M_PI / 2.0);
  const __float128 refq[5] = {thq, cosq((__float128)xl), sinq((__float128)xl), cosq((__float128)xr), sinq((__float128)xr)};
  double ref[5];
  for (int c = 0; c < 5; ++c) ref[c] = (double)refq[c];
  const double got[5] = {h.yaw, h.cl, h.sl, h.cr, h.sr};
  long n = 0;
  for (int c = 0; c < 5; ++c) if (!same(ref[c], got[c])) { ++bad[c]; ++n;
    if (bad[c] <= 3) std::fprintf(stderr, "  mismatch out %d at x=%a y=%a: got %a want %a\n", c, x, y, got[c], ref[c]); }
  // the cheap stage, with its seeds pushed to the edge of their tolerance in turn
  static const double kPert[3] = {1.0, 1.0 + 0x1.6p-23, 1.0 - 0x1.6p-23};
  rl::cr::g_seed_perturb = kPert[(g_perturb++) % 3];
  rl::cr::Heading f;
  double dg[15] = {0.0};
  const bool sure = rl::cr::heading_fast(x, y, f, rl::cr::kAtanTab, dg);
  ++g_fast_calls;
  if (x != 0.0 && y != 0.0 && std::isfinite(x) && std::isfinite(y))
    for (int c = 0; c < 5; ++c) {
      const double bd = boundary_distance(refq[c]);
      if (bd < 0x1p-7) ++g_hard[0];
      if (bd < 0x1p-13) ++g_hard[1];
      if (bd < 0x1p-19) ++g_hard[2];
    }
  if (sure) {
    ++g_fast_sure;
    const double fg[5] = {f.yaw, f.cl, f.sl, f.cr, f.sr};
    for (int c = 0; c < 5; ++c) if (!same(ref[c], fg[c])) { ++g_fast_bad;
      if (g_fast_bad <= 5) std::fprintf(stderr, "  FAST mismatch out %d at x=%a y=%a: got %a want %a\n", c, x, y, fg[c], ref[c]); }
  }
  if (dg[2] > 0.0 && std::isfinite(dg[2])) {   // the bounds themselves (whether or not the stage was sure)
    const double r = (double)(fabsq(thq - ((__float128)dg[0] + (__float128)dg[1])) / (__float128)dg[2]);
    if (r > g_ratio_th) { g_ratio_th = r; if (r >= 1.0) std::fprintf(stderr, "  th ratio %.3f x=%a y=%a e_th=%a\n", r, x, y, dg[2]); }
    if (same(f.yaw, yaw)) {   // the components are defined through the ROUNDED yaw
      const __float128 tv[4] = {-refq[1], refq[2], refq[3], -refq[4]};
      for (int q = 0; q < 4; ++q) {
        const double ro = (double)(fabsq(tv[q] - ((__float128)dg[3 + 3 * q] + (__float128)dg[4 + 3 * q])) / (__float128)dg[5 + 3 * q]);
        if (ro > g_ratio_out) { g_ratio_out = ro; if (ro >= 1.0) std::fprintf(stderr, "  ratio %.3f slot %d x=%a y=%a E=%a e_th=%a\n", ro, q, x, y, dg[5+3*q], dg[2]); }
      }
    }
  }
  {  // what the kernel calls
    const rl::cr::Heading t = rl::cr::heading2(x, y);
    const double tg[5] = {t.yaw, t.cl, t.sl, t.cr, t.sr};
    for (int c = 0; c < 5; ++c) if (!same(tg[c], sure ? ref[c] : got[c])) ++g_two_stage_bad;
  }
  return n;
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? std::atol(argv[1]) : 1000000;
  long total = 0;
  {  // tangents of a racing line: components up to a few thousand, any direction
    long bad[5] = {0, 0, 0, 0, 0};
    for (long i = 0; i < n; ++i) check((urand() - 0.5) * 12000.0, (urand() - 0.5) * 12000.0, bad);
    std::printf("uniform   n=%ld  mismatches yaw=%ld cl=%ld sl=%ld cr=%ld sr=%ld\n", n, bad[0], bad[1], bad[2], bad[3], bad[4]);
    total += bad[0] + bad[1] + bad[2] + bad[3] + bad[4];
    family_done("uniform");
  }
  {  // any magnitude ratio: one component down to 2^-20 of the other, all quadrants, both orders
    long bad[5] = {0, 0, 0, 0, 0};
    for (long i = 0; i < n; ++i) {
      const double big = ldexp(1.0 + urand(), (int)(urand() * 40) - 20), small = big * ldexp(1.0 + urand(), -(int)(urand() * 21));
      const double sx = urand() < 0.5 ? -1.0 : 1.0, sy = urand() < 0.5 ? -1.0 : 1.0;
      if (urand() < 0.5) check(sx * big, sy * small, bad); else check(sx * small, sy * big, bad);
    }
    std::printf("ratios    n=%ld  mismatches yaw=%ld cl=%ld sl=%ld cr=%ld sr=%ld\n", n, bad[0], bad[1], bad[2], bad[3], bad[4]);
    total += bad[0] + bad[1] + bad[2] + bad[3] + bad[4];
    family_done("ratios");
  }
  {  // near the reduction break points q = (j + 1/2)/64 and near the diagonal
    long bad[5] = {0, 0, 0, 0, 0};
    for (long i = 0; i < n / 4; ++i) {
      const int j = (int)(urand() * 64);
      const double den = 1.0 + urand() * 5000.0, num = den * ((j + 0.5) / 64.0) * (1.0 + (urand() - 0.5) * 1e-12);
      check(den, num, bad); check(-num, den, bad);
      const double d = 1.0 + urand() * 5000.0; check(d, d * (1.0 + (urand() - 0.5) * 1e-13), bad);
    }
    std::printf("breaks    n=%ld  mismatches yaw=%ld cl=%ld sl=%ld cr=%ld sr=%ld\n", 3 * (n / 4), bad[0], bad[1], bad[2], bad[3], bad[4]);
    total += bad[0] + bad[1] + bad[2] + bad[3] + bad[4];
    family_done("breaks");
  }
  {  // exactly on an axis, every sign combination of zero
    long bad[5] = {0, 0, 0, 0, 0};
    const double v[6] = {0.0, -0.0, 1.0, -1.0, 5791.25, -3.5e-7};
    for (double x : v) for (double y : v) if (x == 0.0 || y == 0.0) check(x, y, bad);
    std::printf("axes      mismatches yaw=%ld cl=%ld sl=%ld cr=%ld sr=%ld\n", bad[0], bad[1], bad[2], bad[3], bad[4]);
    total += bad[0] + bad[1] + bad[2] + bad[3] + bad[4];
    family_done("axes");
  }
  long near_bad[5] = {0, 0, 0, 0, 0};
  {  // within 2^-22 .. 2^-60 rad of an axis: the tiny component may round the other way (documented); count only
    for (long i = 0; i < n / 4; ++i) {
      const double big = (1.0 + urand()) * 3000.0, small = big * ldexp(1.0 + urand(), -22 - (int)(urand() * 38));
      const double sx = urand() < 0.5 ? -1.0 : 1.0, sy = urand() < 0.5 ? -1.0 : 1.0;
      if (urand() < 0.5) check(sx * big, sy * small, near_bad); else check(sx * small, sy * big, near_bad);
    }
    std::printf("near-axis n=%ld  mismatches yaw=%ld cl=%ld sl=%ld cr=%ld sr=%ld (reported, not required to be 0)\n", n / 4,
                near_bad[0], near_bad[1], near_bad[2], near_bad[3], near_bad[4]);
    family_done("near-axis");
  }
  std::printf("fast stage: calls=%ld sure=%ld (%.5f %%) wrong-when-sure=%ld two-stage-inconsistent=%ld\n", g_fast_calls, g_fast_sure,
              100.0 * (double)g_fast_sure / (double)g_fast_calls, g_fast_bad, g_two_stage_bad);
  std::printf("fast stage: max actual/claimed error  theta=%.4f components=%.4f\n", g_ratio_th, g_ratio_out);
  std::printf("outputs near a rounding boundary: within 2^-60: %ld  2^-66: %ld  2^-72: %ld\n", g_hard[0], g_hard[1], g_hard[2]);
  total += g_fast_bad + g_two_stage_bad + (g_ratio_th >= 1.0 ? 1 : 0) + (g_ratio_out >= 1.0 ? 1 : 0);
  std::printf("TOTAL %ld\n", total);
  return total == 0 ? 0 : 1;
}
