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
  int edge;       // its edge index; ties in |s| keep the LOWEST edge index, whatever the scan order
};
constexpr int kNoEdge = 0x7fffffff;

__device__ __forceinline__ double edge_side(double vx, double vy, double dx, double dy) {
  return vx * dy - vy * dx;  // cross(V - p, d)
}

__device__ __forceinline__ void edge_hit(double ax, double ay, double ea, double bx, double by,
                                         double eb, double dx, double dy, int edge, Hit& h) {
  if ((ea > 0.0 && eb > 0.0) || (ea < 0.0 && eb < 0.0)) return;  // edge on one side of the line
  const double sx = bx - ax, sy = by - ay;
  const double den = dx * sy - dy * sx;
  if (den == 0.0) return;  // parallel / collinear: not a Point intersection
  const double s = (ax * sy - ay * sx) / den;
  const double as = fabs(s);
  if (as > 1.0) return;  // beyond +-max_dist
  if (as < h.best || (as == h.best && edge < h.edge)) { h.best = as; h.best_s = s; h.edge = edge; }
}

// scan edges j0 .. j0+count-1 (indices modulo nr); addresses are lane-uniform when j0 is
template <typename RingPtr>
__device__ __forceinline__ void scan_edges(RingPtr ring, int nr, int j0, int count, double px,
                                           double py, double dx, double dy, Hit& h) {
  int j = j0;
  double2 v = ring[j];
  double ax = v.x - px, ay = v.y - py;
  double ea = edge_side(ax, ay, dx, dy);
  for (int q = 0; q < count; ++q) {
    const int j1 = (j + 1 == nr) ? 0 : j + 1;
    v = ring[j1];
    const double bx = v.x - px, by = v.y - py;
    const double eb = edge_side(bx, by, dx, dy);
    edge_hit(ax, ay, ea, bx, by, eb, dx, dy, j, h);
    ax = bx; ay = by; ea = eb;
    j = j1;
  }
}

// brute force over all edges; ring vertices as double2 (x,y), any address space
template <typename RingPtr>
__device__ __forceinline__ Hit search_ring_brute(RingPtr ring, int nr, double px, double py,
                                                 double dx, double dy) {
  Hit h{INFINITY, 0.0, kNoEdge};
  scan_edges(ring, nr, 0, nr, px, py, dx, dy, h);
  return h;
}

// Culled search: edges are grouped in chunks of kChunk consecutive edges with a bounding circle
// (cx, cy, r).  A chunk can hold a crossing of the segment only if the infinite line passes
// within r of its centre and the centre's projection is within max_dist + r of p; every chunk that
// passes is scanned exactly like the brute-force loop, so the result (including tie-breaking) is
// identical to search_ring_brute.
constexpr int kChunk = 8;

template <typename RingPtr, typename CirclePtr>
__device__ __forceinline__ Hit search_ring_culled(RingPtr ring, int nr, CirclePtr circ, int nchunk,
                                                  double px, double py, double dx, double dy,
                                                  double dlen) {
  Hit h{INFINITY, 0.0, kNoEdge};
  for (int c = 0; c < nchunk; ++c) {
    const double mx = circ[3 * c] - px, my = circ[3 * c + 1] - py, r = circ[3 * c + 2];
    const double rr = r * dlen * (1.0 + 1e-9) + 1e-9;
    if (fabs(mx * dy - my * dx) > rr) continue;
    if (fabs(mx * dx + my * dy) > (dlen + r) * dlen * (1.0 + 1e-9) + 1e-9) continue;
    const int j0 = c * kChunk;
    scan_edges(ring, nr, j0, min(kChunk, nr - j0), px, py, dx, dy, h);
  }
  return h;
}

// ---------------------------------------------------------------------------------------------
// wave64 reductions
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

// Windowed search, wave-cooperative and exact.  Every lane of the wave calls it (inactive lanes
// pass active = false).  Each lane first scans the 2W+1 edges around its hint (the edge its last
// crossing was on; consecutive lanes -> consecutive LDS addresses).  That gives an upper bound
// d_i on the distance of the closest crossing.  A crossing closer than d_i can only sit on an
// edge whose chunk circle comes within d_i of p_i, so:
//   1. the wave forms ONE disk that contains every lane's disk (p_i, d_i) and tests all chunk
//      circles against it, 64 chunks per instruction (ballot);
//   2. for each chunk that passes, every lane checks its own disk; lanes whose window already
//      covers the chunk skip it;
//   3. only if some lane still needs the chunk does the wave scan its edges (uniform addresses,
//      LDS broadcast), and every active lane folds them into its result -- which is harmless:
//      the result is the lexicographic minimum of (|s|, edge) over all edges seen.
// Result == search_ring_brute, bit for bit.
constexpr int kWin = 12;

template <typename RingPtr, typename CirclePtr>
__device__ __forceinline__ Hit search_ring_windowed(RingPtr ring, int nr, CirclePtr circ,
                                                    int nchunk, bool active, int hint, double px,
                                                    double py, double dx, double dy, double dlen) {
  const int lane = threadIdx.x & (kWave - 1);
  Hit h{INFINITY, 0.0, kNoEdge};
  const bool windowed = active && hint >= 0 && hint < nr;
  int lo = hint - kWin;
  if (lo < 0) lo += nr;
  if (windowed) scan_edges(ring, nr, lo, 2 * kWin + 1, px, py, dx, dy, h);
  double di = active ? ((h.best <= 1.0 ? h.best * dlen : dlen) * (1.0 + 1e-9) + 1e-9) : 0.0;
  const double xmin = wave_min(active ? px : INFINITY), xmax = wave_max(active ? px : -INFINITY);
  const double ymin = wave_min(active ? py : INFINITY), ymax = wave_max(active ? py : -INFINITY);
  const double ccx = 0.5 * (xmin + xmax), ccy = 0.5 * (ymin + ymax);
  const double ex = px - ccx, ey = py - ccy;
  const double rho = wave_max(active ? sqrt(ex * ex + ey * ey) + di : 0.0) * (1.0 + 1e-9);
  for (int c0 = 0; c0 < nchunk; c0 += kWave) {
    const int c = c0 + lane;
    bool pass = false;
    if (c < nchunk) {
      const double mx = circ[3 * c] - ccx, my = circ[3 * c + 1] - ccy;
      const double lim = rho + circ[3 * c + 2];
      pass = mx * mx + my * my <= lim * lim;
    }
    unsigned long long mask = __ballot(pass);
    while (mask) {
      const int cc = c0 + (__ffsll((long long)mask) - 1);
      mask &= mask - 1;
      const int e0 = cc * kChunk;
      const int cnt = min(kChunk, nr - e0);
      bool need = false;
      if (active) {
        bool inside = false;
        if (windowed) {
          int off = e0 - lo;
          if (off < 0) off += nr;
          inside = off + cnt - 1 <= 2 * kWin;
        }
        if (!inside) {
          const double mx = circ[3 * cc] - px, my = circ[3 * cc + 1] - py;
          const double lim = (di + circ[3 * cc + 2]) * (1.0 + 1e-9);
          need = mx * mx + my * my <= lim * lim;
        }
      }
      if (__any(need)) {
        Hit h2 = h;
        scan_edges(ring, nr, e0, cnt, px, py, dx, dy, h2);
        if (active) h = h2;
      }
    }
  }
  return h;
}

}  // namespace rl
