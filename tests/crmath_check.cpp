// tests/crmath_check.cpp -- host check of csrc/rl_crmath.hpp against libquadmath (compiled and run by tests/test_crmath.py:
//   g++ -O2 -std=c++17 -ffp-contract=off crmath_check.cpp -lquadmath).  Prints, per input family, how many of the five
// outputs of rl::cr::heading differ from the correctly rounded values RN(atan2), RN(cos / sin(fl(yaw +- fl(pi/2)))).
#include <quadmath.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include "../spline_trajectory_optimization_amd/csrc/rl_crmath.hpp"

static uint64_t g_s = 0x9E3779B97F4A7C15ULL;
static double urand() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return (double)(g_s >> 11) * (1.0 / 9007199254740992.0); }
static bool same(double a, double b) { return std::memcmp(&a, &b, 8) == 0 || (a != a && b != b); }

static long check(double x, double y, long* bad) {
  const rl::cr::Heading h = rl::cr::heading(x, y);
  const double yaw = (double)atan2q((__float128)y, (__float128)x);
  const double xl = yaw + M_PI / 2.0, xr = yaw + (-M_PI / 2.0);
  const double ref[5] = {yaw, (double)cosq((__float128)xl), (double)sinq((__float128)xl),
                         (double)cosq((__float128)xr), (double)sinq((__float128)xr)};
  const double got[5] = {h.yaw, h.cl, h.sl, h.cr, h.sr};
  long n = 0;
  for (int c = 0; c < 5; ++c) if (!same(ref[c], got[c])) { ++bad[c]; ++n;
    if (bad[c] <= 3) std::fprintf(stderr, "  mismatch out %d at x=%a y=%a: got %a want %a\n", c, x, y, got[c], ref[c]); }
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
  }
  {  // exactly on an axis, every sign combination of zero
    long bad[5] = {0, 0, 0, 0, 0};
    const double v[6] = {0.0, -0.0, 1.0, -1.0, 5791.25, -3.5e-7};
    for (double x : v) for (double y : v) if (x == 0.0 || y == 0.0) check(x, y, bad);
    std::printf("axes      mismatches yaw=%ld cl=%ld sl=%ld cr=%ld sr=%ld\n", bad[0], bad[1], bad[2], bad[3], bad[4]);
    total += bad[0] + bad[1] + bad[2] + bad[3] + bad[4];
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
  }
  std::printf("TOTAL %ld\n", total);
  return total == 0 ? 0 : 1;
}
