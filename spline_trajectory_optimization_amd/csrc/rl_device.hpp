// rl_device.hpp -- device-side building blocks of the min-curvature hot path (gfx950, wave64).
//
// Everything here is written for the MI355X execution model: 64-lane waves, FP64 VALU, LDS
// broadcast reads for data every lane of a wave walks in lock-step (ring vertices), coalesced
// table reads (sample index = lane index).  No CUDA idioms, no compatibility layer.
#pragma once
#include <hip/hip_runtime.h>

namespace rl {

constexpr int kWave = 64;
constexpr int kMaxK = 5;

// Device view of one track: spline knots + the uniform sample grid u_i = i/N and everything that
// depends only on (t, k, N) -- shared by all instances of a batch, L2-resident.
struct TrackDev {
  int k, n, nt, N;
  const double* t;     // [nt]
  const double* c0;    // [2][n] initial control points (x row, y row)
  const int* ell;      // [N]   knot interval of sample i: t[ell] <= u_i < t[ell+1]
  const double* D;     // [3][k+1][N] de Boor values / 1st / 2nd derivatives of the k+1 non-zero
                       //            basis functions at u_i; entry (a,i) is basis ell[i]-k+a
  const int* sup;      // [n][2] sample range [s0,s1) with t[j] <= u_i < t[j+k+1]
                       //        (the mask of optimizer.py:27-29 / 225-227)
  const double* base;  // [4][N] initial line p0x, p0y and unit left normal n0x, n0y
};

// ---------------------------------------------------------------------------------------------
// de Boor recurrences for the k+1 non-zero basis functions (m-th derivative) at x, interval l.
// Same recurrences as scipy's BSpline evaluation (oracle/mincurv_oracle.c: deboor_d), unrolled
// over a compile-time degree so that h/hh live in registers.
template <int K>
__device__ __forceinline__ void deboor(const double* __restrict__ t, double x, int l, int m,
                                       double (&h)[K + 1]) {
  double hh[K + 1];
  h[0] = 1.0;
#pragma unroll
  for (int j = 1; j <= K; ++j) {
    const bool value_step = (j <= K - m);
#pragma unroll
    for (int q = 0; q < j; ++q) hh[q] = h[q];
    h[0] = 0.0;
#pragma unroll
    for (int n = 1; n <= j; ++n) {
      const double xb = t[l + n], xa = t[l + n - j];
      if (xb == xa) {
        if (value_step) h[n] = 0.0; else h[m] = 0.0;
        continue;
      }
      if (value_step) {
        const double w = hh[n - 1] / (xb - xa);
        h[n - 1] += w * (xb - x);
        h[n] = w * (x - xa);
      } else {
        const double w = (double)j * hh[n - 1] / (xb - xa);
        h[n - 1] -= w;
        h[n] = w;
      }
    }
  }
}

// scipy find_interval (extrapolate=True): largest l in [k, n-1] with t[l] <= x.
__device__ __forceinline__ int find_interval(const double* __restrict__ t, int k, int n, double x) {
  int lo = k, hi = n - 1;
  while (lo < hi) {  // invariant: answer in [lo, hi]
    const int mid = (lo + hi + 1) >> 1;
    if (x >= t[mid]) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// ---------------------------------------------------------------------------------------------
// Curve point and derivatives at sample i from the basis tables and the (LDS-resident) control
// points.  NDER = 0: x,y ; 1: + x',y' ; 2: + x'',y''.   Tables are [3][K+1][N].
template <int K, int NDER>
struct CurvePoint {
  double x, y, dx, dy, d2x, d2y;
};

template <int K, int NDER, typename CPtr>
__device__ __forceinline__ void eval_sample(const TrackDev& tr, CPtr cx, CPtr cy, int i, int l,
                                            CurvePoint<K, NDER>& o) {
  const int N = tr.N;
  const double* __restrict__ D0 = tr.D;
  const double* __restrict__ D1 = tr.D + (size_t)(K + 1) * N;
  const double* __restrict__ D2 = tr.D + (size_t)2 * (K + 1) * N;
  double x = 0, y = 0, dx = 0, dy = 0, d2x = 0, d2y = 0;
#pragma unroll
  for (int a = 0; a <= K; ++a) {
    const double c_x = cx[l - K + a], c_y = cy[l - K + a];
    const double b0 = D0[(size_t)a * N + i];
    x += c_x * b0; y += c_y * b0;
    if (NDER >= 1) { const double b1 = D1[(size_t)a * N + i]; dx += c_x * b1; dy += c_y * b1; }
    if (NDER >= 2) { const double b2 = D2[(size_t)a * N + i]; d2x += c_x * b2; d2y += c_y * b2; }
  }
  o.x = x; o.y = y; o.dx = dx; o.dy = dy; o.d2x = d2x; o.d2y = d2y;
}

// ---------------------------------------------------------------------------------------------
// Closest crossing of the normal segment p + s*d, s in [-1,1], with a closed polyline.
// Semantics of Trajectory.fill_bounds (models/trajectory.py:84-129); arithmetic identical to
// oracle/mincurv_oracle.c: closest_hit.  Returns the signed parameter s of the closest hit,
// 0 when there is none (bound = the waypoint itself, trajectory.py:127).
struct Hit {
  double best;    // |s| of the best hit so far (inf = none)
  double best_s;  // its signed parameter
};

__device__ __forceinline__ double edge_side(double vx, double vy, double dx, double dy) {
  return vx * dy - vy * dx;  // cross(V - p, d)
}

__device__ __forceinline__ void edge_hit(double ax, double ay, double ea, double bx, double by,
                                         double eb, double dx, double dy, Hit& h) {
  if ((ea > 0.0 && eb > 0.0) || (ea < 0.0 && eb < 0.0)) return;  // edge on one side of the line
  const double sx = bx - ax, sy = by - ay;
  const double den = dx * sy - dy * sx;
  if (den == 0.0) return;  // parallel / collinear: not a Point intersection
  const double s = (ax * sy - ay * sx) / den;
  const double as = fabs(s);
  if (as > 1.0) return;  // beyond +-max_dist
  if (as < h.best) { h.best = as; h.best_s = s; }
}

// brute force over all edges; ring vertices as double2 (x,y), any address space
template <typename RingPtr>
__device__ __forceinline__ double search_ring_brute(RingPtr ring, int nr, double px, double py,
                                                    double dx, double dy) {
  Hit h{INFINITY, 0.0};
  double2 v = ring[0];
  double ax = v.x - px, ay = v.y - py;
  double ea = edge_side(ax, ay, dx, dy);
  for (int j = 0; j < nr; ++j) {
    const int j1 = (j + 1 == nr) ? 0 : j + 1;
    v = ring[j1];
    const double bx = v.x - px, by = v.y - py;
    const double eb = edge_side(bx, by, dx, dy);
    edge_hit(ax, ay, ea, bx, by, eb, dx, dy, h);
    ax = bx; ay = by; ea = eb;
  }
  return h.best_s;
}

// Culled search: edges are grouped in chunks of kChunk consecutive edges with a bounding circle
// (cx, cy, r).  A chunk can hold a crossing of the segment only if the infinite line passes
// within r of its centre and the centre's projection is within max_dist + r of p; every chunk that
// passes is scanned exactly like the brute-force loop, in ascending edge order, so the result
// (including tie-breaking) is identical to search_ring_brute.
constexpr int kChunk = 16;

template <typename RingPtr, typename CirclePtr>
__device__ __forceinline__ double search_ring_culled(RingPtr ring, int nr, CirclePtr circ,
                                                     int nchunk, double px, double py, double dx,
                                                     double dy, double dlen) {
  Hit h{INFINITY, 0.0};
  for (int c = 0; c < nchunk; ++c) {
    const double mx = circ[3 * c] - px, my = circ[3 * c + 1] - py, r = circ[3 * c + 2];
    const double rr = r * dlen * (1.0 + 1e-9) + 1e-9;
    if (fabs(mx * dy - my * dx) > rr) continue;
    if (fabs(mx * dx + my * dy) > (dlen + r) * dlen * (1.0 + 1e-9) + 1e-9) continue;
    const int j0 = c * kChunk;
    const int j1e = min(j0 + kChunk, nr);
    double2 v = ring[j0];
    double ax = v.x - px, ay = v.y - py;
    double ea = edge_side(ax, ay, dx, dy);
    for (int j = j0; j < j1e; ++j) {
      const int j1 = (j + 1 == nr) ? 0 : j + 1;
      v = ring[j1];
      const double bx = v.x - px, by = v.y - py;
      const double eb = edge_side(bx, by, dx, dy);
      edge_hit(ax, ay, ea, bx, by, eb, dx, dy, h);
      ax = bx; ay = by; ea = eb;
    }
  }
  return h.best_s;
}

// ---------------------------------------------------------------------------------------------
// wave64 / workgroup reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, kWave));
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, kWave));
  return v;
}

}  // namespace rl
