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
  // reference-order mode only (rl_kernels.hpp: StrictRows), built on first use:
  const double* Ds;      // [5k+2][N] unfused de Boor values D0 | D1 | D2 | E1 | E2
  const double* base_s;  // [6][N] p0x, p0y, cos / sin(yaw0 + pi/2), cos / sin(yaw0 - pi/2)
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
    x = fma(c_x, b0, x); y = fma(c_y, b0, y);  // explicit: identical rounding at every call site
    if (NDER >= 1) { const double b1 = D1[(size_t)a * N + i]; dx = fma(c_x, b1, dx); dy = fma(c_y, b1, dy); }
    if (NDER >= 2) { const double b2 = D2[(size_t)a * N + i]; d2x = fma(c_x, b2, d2x); d2y = fma(c_y, b2, d2y); }
  }
  o.x = x; o.y = y; o.dx = dx; o.dy = dy; o.d2x = d2x; o.d2y = d2y;
}

// The same evaluation (same operations in the same order) that also hands back the basis values and second derivatives it
// loaded: the basis value of a given control point at the sample is then a select among registers, not another look-up.
template <int K, typename CPtr>
__device__ __forceinline__ void eval_sample_keep(const TrackDev& tr, CPtr cx, CPtr cy, int i, int l, CurvePoint<K, 2>& o,
                                                 double (&b0v)[K + 1], double (&b2v)[K + 1]) {
  const int N = tr.N;
  const double* __restrict__ D0 = tr.D;
  const double* __restrict__ D1 = tr.D + (size_t)(K + 1) * N;
  const double* __restrict__ D2 = tr.D + (size_t)2 * (K + 1) * N;
  double x = 0, y = 0, dx = 0, dy = 0, d2x = 0, d2y = 0;
#pragma unroll
  for (int a = 0; a <= K; ++a) {
    const double c_x = cx[l - K + a], c_y = cy[l - K + a];
    const double b0 = D0[(size_t)a * N + i], b1 = D1[(size_t)a * N + i], b2 = D2[(size_t)a * N + i];
    b0v[a] = b0; b2v[a] = b2;
    x = fma(c_x, b0, x); y = fma(c_y, b0, y);
    dx = fma(c_x, b1, dx); dy = fma(c_y, b1, dy);
    d2x = fma(c_x, b2, d2x); d2y = fma(c_y, b2, d2y);
  }
  o.x = x; o.y = y; o.dx = dx; o.dy = dy; o.d2x = d2x; o.d2y = d2y;
}

// Left normal of the curve scaled to max_dist: max_dist * (cos, sin)(yaw + pi/2) without the trip
// through atan2/cos/sin (the oracle takes that trip; the two agree to rounding).  One place, so
// that the re-intersection and the constraint assembly see bit-identical direction vectors.
__device__ __forceinline__ void scaled_normal(double dx, double dy, double max_dist, double& nx,
                                              double& ny, double& inv_s2) {
  const double s2 = fma(dx, dx, dy * dy);
  const double rs = rsqrt(s2);
  inv_s2 = rs * rs;
  const double k = max_dist * rs;
  nx = -dy * k;
  ny = dx * k;
}

// ---------------------------------------------------------------------------------------------
// Closest crossing of the normal segment p + s*d, s in [-1,1], with a closed polyline.
// Semantics of Trajectory.fill_bounds (models/trajectory.py:84-129); arithmetic identical to
// A zero the compiler cannot see through, re-made wherever it is called (one scalar move).  Added to a thread or lane index it
// keeps the address arithmetic that hangs on that index INSIDE the loop that uses it: hoisted out of a long loop, such per-lane
// addresses fill the register file and come back from scratch memory, one memory round trip in front of every access.
__device__ __forceinline__ int opaque_zero() {
  int z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  return z;
}

// A workgroup barrier that orders LDS traffic only: __syncthreads() also waits until the wave's GLOBAL stores have completed
// (s_waitcnt vmcnt(0)), a memory round trip in front of the barrier.  Where everything the other waves read next lives in
// LDS -- and the next reader of the global data sits behind a full __syncthreads() anyway -- that wait can ride along there.
// (Written as instructions: with LDS-DMA in the kernel the compiler's own address-space-restricted fence still waits for vmcnt.)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Exit rule of the interior-point iterations of the global QPs (k_global_qp, k_global_qp2, k_global_xy and their CPU twins,
// oracle/mincurv_oracle.c: ipm_done), round 6: ONE test, on the complementarity mu alone -- below `tol_mid` for every
// linearisation before the last, below 1e-10 for the last.
//   * Round 5 left the intermediate linearisations at mu < 1e-5 (1e-7 with both coordinates free) plus residual and "the
//     residual has stopped falling" clauses.  The iterate of an UNDER-CONVERGED interior-point method is an ill-conditioned
//     function of the arithmetic (near the end of a QP the normal matrix has condition 1e12: the two implementations' steps
//     differ in their fourth digit, and only further iterations correct that), so kernel and twin parted by up to 7e-4 m on
//     2-3 % of a batch, and the residual clauses -- comparisons of two nearly equal numbers -- let them leave a QP up to five
//     iterations apart.  Measured on the benchmarked batch (tools/global_full_batch_vs_twin.py, profiles/r06_*): a fixed
//     iteration schedule instead of tolerances made it worse (60 of 1024 beyond 1e-6 m: the cause is the convergence level,
//     not the exit flip); tol_mid = 1e-6 leaves 6 instances beyond 1e-6 m, 1e-7 none (largest 2.5e-7 m, 1024 of 1024 equal
//     iteration counts) for the one-offset formulation; with both coordinates free -- the cost is nearly flat along the line --
//     1e-7 / 1e-8 / 1e-9 leave 22 / 3 / 0 instances beyond 1e-6 m.
//   * mu falls by one to two orders per iteration near the end, so a threshold on it is passed by a wide margin almost always
//     (kernel and twin disagree on 0 resp. 4 of 1024 instances, by one iteration); going on PAST the complementarity drives
//     the residual up with the conditioning and breaks down within two or three iterations (NaN seen), so nothing else is
//     waited for.
constexpr double kIpmTolOneOffset = 1e-7;   // one lateral offset per control point
constexpr double kIpmTolTwoCoords = 1e-9;   // both coordinates free
__device__ __forceinline__ bool ipm_done(double tol_mid, bool last_qp, double mu) { return mu < (last_qp ? 1e-10 : tol_mid); }

// oracle/mincurv_oracle.c: closest_hit.  Returns the signed parameter s of the closest hit,
// 0 when there is none (bound = the waypoint itself, trajectory.py:127).
struct Hit {
  double best;    // |s| of the best hit so far (inf = none)
  double best_s;  // its signed parameter
  int edge;       // its edge index; ties in |s| keep the LOWEST edge index, whatever the scan order
};
constexpr int kNoEdge = 0x7fffffff;

// cross(V - p, d).  Written with an explicit fma so that EVERY inlined copy rounds identically:
// the sign of this value decides on which edge a crossing through a ring vertex is found, and the
// window scan evaluates it twice (candidate pass, exact pass).  Left to the compiler, the
// contraction of a*b - c*d differs between call sites.
// ST (the reference-order mode, RL_ARITH_REFERENCE): two rounded products and a subtraction, like numpy and like
// oracle/mincurv_oracle.c: closest_hit -- with one product exact (fma) a crossing through a sample that sits ON a ring
// leaves another residual than the reference's (DESIGN_HISTORY.md section 5).
__device__ __forceinline__ double cross_unfused(double a, double b, double c, double d) {
#pragma clang fp contract(off)
  return a * b - c * d;
}
template <bool ST = false>
__device__ __forceinline__ double edge_side(double vx, double vy, double dx, double dy) {
  if constexpr (ST) return cross_unfused(vx, dy, vy, dx);
  else return fma(vx, dy, -(vy * dx));
}

// the crossing of the line with an edge whose end points are known NOT to lie strictly on one side of it
template <bool ST = false>
__device__ __forceinline__ void edge_cross(double ax, double ay, double bx, double by, double dx, double dy, int edge, Hit& h) {
  const double sx = bx - ax, sy = by - ay;
  const double den = ST ? cross_unfused(dx, sy, dy, sx) : fma(dx, sy, -(dy * sx));
  if (den == 0.0) return;  // parallel / collinear: not a Point intersection
  const double s = (ST ? cross_unfused(ax, sy, ay, sx) : fma(ax, sy, -(ay * sx))) / den;
  const double as = fabs(s);
  if (as > 1.0) return;  // beyond +-max_dist
  if (as < h.best || (as == h.best && edge < h.edge)) { h.best = as; h.best_s = s; h.edge = edge; }
}

template <bool ST = false>
__device__ __forceinline__ void edge_hit(double ax, double ay, double ea, double bx, double by,
                                         double eb, double dx, double dy, int edge, Hit& h) {
  // edge strictly on one side of the line <=> ea, eb non-zero with equal signs.  One multiply
  // decides it; a zero product with two non-zero factors (underflow) is re-checked exactly.
  const double prod = ea * eb;
  if (prod > 0.0) return;
  if (prod == 0.0 && ea != 0.0 && eb != 0.0 && ((ea > 0.0) == (eb > 0.0))) return;
  const double sx = bx - ax, sy = by - ay;
  const double den = ST ? cross_unfused(dx, sy, dy, sx) : fma(dx, sy, -(dy * sx));
  if (den == 0.0) return;  // parallel / collinear: not a Point intersection
  const double s = (ST ? cross_unfused(ax, sy, ay, sx) : fma(ax, sy, -(ay * sx))) / den;
  const double as = fabs(s);
  if (as > 1.0) return;  // beyond +-max_dist
  if (as < h.best || (as == h.best && edge < h.edge)) { h.best = as; h.best_s = s; h.edge = edge; }
}

// scan edges j0 .. j0+count-1 (indices modulo nr); addresses are lane-uniform when j0 is
template <bool ST = false, typename RingPtr>
__device__ __forceinline__ void scan_edges(RingPtr ring, int nr, int j0, int count, double px,
                                           double py, double dx, double dy, Hit& h) {
  int j = j0;
  double2 v = ring[j];
  double ax = v.x - px, ay = v.y - py;
  double ea = edge_side<ST>(ax, ay, dx, dy);
  for (int q = 0; q < count; ++q) {
    const int j1 = (j + 1 == nr) ? 0 : j + 1;
    v = ring[j1];
    const double bx = v.x - px, by = v.y - py;
    const double eb = edge_side<ST>(bx, by, dx, dy);
    edge_hit<ST>(ax, ay, ea, bx, by, eb, dx, dy, j, h);
    ax = bx; ay = by; ea = eb;
    j = j1;
  }
}

// Same scan with the vertex loads issued BATCH at a time ahead of the arithmetic: with one wave per
// SIMD nothing else hides the LDS latency, so the loads of a batch share one wait.
template <int BATCH, bool ST = false, typename RingPtr>
__device__ __forceinline__ void scan_edges_batched(RingPtr ring, int nr, int j0, int count, double px,
                                                   double py, double dx, double dy, Hit& h) {
  int j = j0;
  double2 v = ring[j];
  double ax = v.x - px, ay = v.y - py;
  double ea = edge_side<ST>(ax, ay, dx, dy);
  int q = 0;
  for (; q + BATCH <= count; q += BATCH) {
    double2 w[BATCH];
    int jj[BATCH];
    int t = j;
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      t = (t + 1 == nr) ? 0 : t + 1;
      jj[u] = t;
      w[u] = ring[t];
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const double bx = w[u].x - px, by = w[u].y - py;
      const double eb = edge_side<ST>(bx, by, dx, dy);
      edge_hit<ST>(ax, ay, ea, bx, by, eb, dx, dy, u == 0 ? j : jj[u - 1], h);
      ax = bx; ay = by; ea = eb;
    }
    j = jj[BATCH - 1];
  }
  if (q < count) {
    // tail: re-derive the state for the plain loop (the current vertex is j)
    for (; q < count; ++q) {
      const int j1 = (j + 1 == nr) ? 0 : j + 1;
      v = ring[j1];
      const double bx = v.x - px, by = v.y - py;
      const double eb = edge_side<ST>(bx, by, dx, dy);
      edge_hit<ST>(ax, ay, ea, bx, by, eb, dx, dy, j, h);
      ax = bx; ay = by; ea = eb;
      j = j1;
    }
  }
}

// Window scan over kWinEdges consecutive edges starting at edge `lo` of a ring stored with its
// first kRingPad vertices repeated after the last one (no index wrap inside the loop).
// Pass 1 touches every vertex once (4 FP64 ops + 1 multiply per edge) and only records, as one bit
// per edge, where the line changes side; pass 2 runs the exact edge test (with its division) on
// the recorded edges -- all lanes work on their own first candidate at the same time, so the
// expensive block runs once or twice per window instead of once per distinct candidate position.
// The window is CHUNK ALIGNED: the three whole chunks around the chunk of the hint (the edge of the last
// crossing).  A crossing found in the middle chunk then has both neighbouring chunks fully scanned, and the
// certificate below reduces to one table look-up (plus one circle test when the crossing moved into an outer
// chunk of the window).
constexpr int kChunk = 8;
constexpr int kWinChunks = 3;
constexpr int kNear = 1;                        // chunks ce-kNear..ce+kNear are checked one by one
constexpr int kWinEdges = kWinChunks * kChunk;  // 24; rings must be longer than twice this
constexpr int kRingPad = kWinEdges + 1;         // vertices repeated behind the ring
constexpr int kWinBatch = 8;

// `lo` addresses the vertices (ring[lo .. lo + kWinEdges]); `lo_edge` numbers the edges (edge q of the window is edge
// lo_edge + q of the ring, modulo nr).  Scanning the ring itself: lo == lo_edge; scanning a staged stretch: lo = offset
// of the window inside the stretch.
template <int BATCH = kWinBatch, bool ST = false, typename RingPtr>
__device__ __forceinline__ void scan_window(RingPtr ring, int nr, int lo, int lo_edge, double px, double py,
                                            double dx, double dy, Hit& h, const double* cmax_p,
                                            unsigned long long* counts = nullptr) {   // counts (diagnostic build): [1] exact passes, [2] pass-2 rounds, [3] scans
  // Pass 1 per vertex: its side value e, then TWO bookkeeping instructions -- the sign bit of e shifted into a 25-bit word
  // (v_alignbit) and a running minimum of |e|.  Edge q is a candidate when the signs of its two vertices differ; a vertex
  // exactly ON the line (e == +-0, seen as a zero minimum) makes every edge of the window a candidate.  That is a superset of
  // the edges on which edge_hit() records a crossing (it needs e_a e_b <= 0 with a sign change or a zero: same-sign pairs whose
  // product merely underflows, and NaNs, never give one), and pass 2 runs the exact test on the candidates.  (Until round 6:
  // `!(e_a * e_b > 0)` per edge -- a multiplication, a comparison, a select and an or.)
  // Round 6, QUICK pass 1: all pass 1 has to deliver is the SIGN of every vertex's side value as the exact arithmetic rounds
  // it.  e' = v_x d_y - v_y d_x - (p_x d_y - p_y d_x), two fma per vertex on the untranslated coordinates (instead of two
  // subtractions and the cross product: 4 instead of 7 vector instructions per vertex with the bookkeeping), differs from the
  // real-number value by at most 4u Mp D + u |e'| (u = 2^-53, Mp >= every coordinate involved, D = |d_x| + |d_y|), and so
  // does the difference of the two products whose sign the exact arithmetic returns (fused or not: the final subtraction
  // cannot change a sign or make a zero).  |e'| > tol = 2^-46 Mp D = 128 u Mp D for every vertex of the window therefore
  // proves: same signs, no exact zero.  tol outside [2^-900, 2^900], a non-finite operand (cmax is +inf for a ring with a
  // non-finite vertex) or a vertex closer to the line than that sends the WAVE through the exact pass -- the untouched far
  // end of a support in the first sweep (a width-form ring's vertex i sits on sample i's first normal), hardly ever later.
  // A NaN vertex inside a finite ring gives no crossing on its edges whichever way its sign bit is read.
  unsigned cand;
  {
    unsigned signs;
    double emin;
    bool sure;
    {
      const double c0 = fma(px, dy, -(py * dx));
      double2 v = ring[lo];
      const double e0 = fma(v.x, dy, fma(-v.y, dx, -c0));
      signs = (unsigned)__double2hiint(e0) >> 31;   // vertex k of the window ends up in bit 24 - k
      emin = fabs(e0);
#pragma unroll
      for (int q0 = 0; q0 < kWinEdges; q0 += BATCH) {
        double2 w[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) w[u] = ring[lo + q0 + u + 1];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
          const double eb = fma(w[u].x, dy, fma(-w[u].y, dx, -c0));
          signs = __builtin_amdgcn_alignbit(signs, (unsigned)__double2hiint(eb), 31);   // (signs << 1) | sign(eb)
          emin = fmin(emin, fabs(eb));
        }
        if constexpr (BATCH != kWinBatch) __builtin_amdgcn_sched_barrier(0);   // staged scan: keep the next batch's LDS reads behind this batch's arithmetic (registers)
      }
      asm volatile("" : "+v"(signs));   // the word is made HERE: left alone, the compiler sinks the 24 shifts below the branch and keeps 25 side values alive for them
      const double tol = 0x1p-46 * ((cmax_p[opaque_zero()] + fabs(px)) + fabs(py)) * (fabs(dx) + fabs(dy));   // read here: a register held across the scan would be spilled
      sure = emin > tol && tol > 0x1p-900 && tol < 0x1p+900;
    }
#ifdef RL_STAMPS
    if (counts) { counts[3] += 1; if (__any(!sure)) counts[1] += 1; }
#endif
    if (__any(!sure)) {   // the exact pass, for the whole wave (a lane that was sure gets the same word again)
      double2 v = ring[lo];
      const double e0 = edge_side<ST>(v.x - px, v.y - py, dx, dy);
      signs = (unsigned)__double2hiint(e0) >> 31;
      emin = fabs(e0);
#pragma unroll
      for (int q0 = 0; q0 < kWinEdges; q0 += BATCH) {
        double2 w[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) w[u] = ring[lo + q0 + u + 1];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
          const double eb = edge_side<ST>(w[u].x - px, w[u].y - py, dx, dy);
          signs = __builtin_amdgcn_alignbit(signs, (unsigned)__double2hiint(eb), 31);
          emin = fmin(emin, fabs(eb));
        }
        if constexpr (BATCH != kWinBatch) __builtin_amdgcn_sched_barrier(0);
      }
      sure = emin != 0.0;   // NaN: "sure" -- its edges give no crossing
    }
    // sign change across edge q <-> bits 24 - q and 23 - q of `signs` differ <-> bit 23 - q of signs ^ (signs >> 1); reversed: bit q
    cand = __builtin_bitreverse32((signs ^ (signs >> 1)) & 0xFFFFFFu) >> 8;
    if (!sure) cand = 0xFFFFFFu;
    // An edge flagged for a CHANGE OF SIGN has its end points on different sides (or one of them is a zero of the other
    // sign bit): edge_hit's two early exits -- product positive; product zero with both factors non-zero and of one sign -- can
    // not fire, so pass 2 goes straight to the crossing, without forming the two side values again.  Only a wave that holds a
    // lane with a vertex exactly on its line (every edge of that lane flagged) runs the full test.
    const bool sides_known = !__any(!sure);
    static_assert(kWinEdges == 24 && kWinEdges % BATCH == 0, "window scan is unrolled in whole batches; the candidate word holds 24 edges");
    while (__any(cand != 0u)) {
#ifdef RL_STAMPS
      if (counts) counts[2] += 1;
#endif
      if (cand != 0u) {
        const int q = __ffs((int)cand) - 1;
        cand &= cand - 1u;
        const int j = lo + q, je = lo_edge + q;
        const double2 va = ring[j], vb = ring[j + 1];
        const double ax = va.x - px, ay = va.y - py, bx = vb.x - px, by = vb.y - py;
        if (sides_known) edge_cross<ST>(ax, ay, bx, by, dx, dy, je >= nr ? je - nr : je, h);
        else edge_hit<ST>(ax, ay, edge_side<ST>(ax, ay, dx, dy), bx, by, edge_side<ST>(bx, by, dx, dy), dx, dy,
                          je >= nr ? je - nr : je, h);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// STAGED windows (round 4).  When the ring vertices live in global memory (the residency of full batches), a wave copies
// the stretch of the ring its 64 lanes are about to scan -- kStage consecutive vertices from a chunk-aligned `base` --
// into a wave-private LDS buffer with two LDS-DMA instructions (global_load_lds_dwordx4: lane t -> slot t, no VGPR
// destination), and every lane whose window lies inside the stretch scans it from LDS.  The windows of 64 consecutive
// samples span 8-9 chunk starts on a width-form ring (vertex i sits on sample i's normal), the stretch holds 9.  Lanes whose
// window does not fit (seam, a hint far away) are served by re-staging from their own window start: ONE scan path, whatever
// the hints.  Measured on the benchmarked batch (same box, interleaved, bit-identical results; tools/sweep_variant.py):
// direct loads 9.66-9.88 ms, staged 9.48-9.75 ms (-1.5 %), vector-memory read instructions per launch 2.04e8 -> 0.96e8.
// Requesting the stretches at the START of a step (before the cost pass, so that the refresh waits for no vertex at all)
// was 2 % SLOWER than direct loads (9.88-10.02 ms): the step is not waiting for these loads (DESIGN_HISTORY.md section 3).
constexpr int kStage = 96;
constexpr int kStageBatch = 8;   // LDS reads in flight per lane (round 6: 4 -> 8 with the quick sign pass's shorter batches, -0.7 %; 12 spills)
static_assert(kStage >= kWinEdges + 1 + 64 && kStage <= 104, "a stretch holds the windows of 64 consecutive chunk starts; two DMA instructions fill it");
// LDS image of a stretch: vertex v of the stretch sits in slot v + (v >> 3) -- one empty 16-byte slot behind every chunk of 8.
// The lanes of a wave scan windows that start 0, 8, 16, ... vertices into the stretch; unpadded, those starts are 128 bytes
// apart and every second one falls on the same banks (measured: SQ_LDS_BANK_CONFLICT 34 % of the LDS cycles); padded, chunk c
// starts 144 c bytes in: 0, 36, 8, 44, 16, ... dwords modulo 64 -- disjoint 4-bank groups.  The DMA writes slot t from lane t
// (lane-linear destination), so the padding is made on the SOURCE side: lane t fetches the vertex whose slot is t.
constexpr int kStageSlots = kStage + kStage / kChunk;   // 108
__device__ __forceinline__ int stage_slot(int v) { return v + (v >> 3); }

__device__ __forceinline__ void stage_issue(const double2* ring, int nr, int base, double2* stg, int lane) {
  // two DMA instructions, neither predicated: slots 0..63 and slots kStageSlots-64..kStageSlots-1 (the twenty slots both cover
  // receive the same vertex twice).  Exactly two vector-memory operations whatever the lanes do: the wait counts of loads
  // issued BEFORE a stretch request stay exact (a conditional request makes the compiler wait for everything).
  static_assert(kStageSlots > kWave && kStageSlots <= 2 * kWave, "two instructions cover the stretch");
#pragma unroll
  for (int t0 = 0; t0 <= kStageSlots - kWave; t0 += kStageSlots - kWave) {
    const int t = t0 + lane;                 // slot
    const int v = t - t / 9;                 // the vertex of slot t (slots 8, 17, 26, ... are the pads: they fetch a neighbour)
    int idx = base + (v < kStage ? v : kStage - 1);
    if (idx >= nr) idx -= nr;                // (base + v) mod nr: base < nr and v < kStage <= nr (the caller checks)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ring + idx),
                                     (__attribute__((address_space(3))) void*)(stg + t0), 16, 0, 0);
  }
}
// a window inside a staged stretch: vertex k of the window (its start is chunk aligned) -> its padded slot
struct StagedWindow {
  const double2* p;   // slot of the window's first vertex
  __device__ __forceinline__ double2 operator[](int k) const { return p[k + (k >> 3)]; }
};
// the stretch has landed and is visible to this wave's LDS reads
__device__ __forceinline__ void stage_wait() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// first edge of the chunk-aligned window around `hint` (an edge index < nr)
__device__ __forceinline__ int window_start(int hint, int nchunk) {
  int cs = hint / kChunk - 1;
  if (cs < 0) cs += nchunk;
  return cs * kChunk;
}

// Request the stretch a wave's next window scan will read, ahead of time (the hints are known before the curve point is):
// returns the base to hand to search_ring_windowed as `staged_base`.  The search waits for the copy (stage_wait) before its
// first read -- also when no lane had a usable hint and the stretch at 0 was fetched for nothing: the stretches share their
// LDS with the cost terms of the next step, no copy may still be on its way when the refresh ends.
__device__ __forceinline__ int stage_prefetch(const double2* ring, int nr, int nchunk, bool active, int hint, double2* stg, int lane) {
  const unsigned long long m = __ballot(active && hint >= 0 && hint < nr);
  // no branch: with no usable hint in the wave the stretch at 0 is fetched for nothing (see stage_issue on the wait counts)
  const int first = m ? __ffsll((long long)m) - 1 : 0;
  const int lo = __builtin_amdgcn_readlane(window_start(hint, nchunk), first);
  const int base = m ? lo : 0;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // nobody still reads the stretch it replaces
  stage_issue(ring, nr, base, stg, lane);
  return base;
}

// brute force over all edges; ring vertices as double2 (x,y), any address space
template <bool ST = false, typename RingPtr>
__device__ __forceinline__ Hit search_ring_brute(RingPtr ring, int nr, double px, double py,
                                                 double dx, double dy) {
  Hit h{INFINITY, 0.0, kNoEdge};
  scan_edges<ST>(ring, nr, 0, nr, px, py, dx, dy, h);
  return h;
}

// Culled search: edges are grouped in chunks of kChunk consecutive edges with a bounding circle
// (cx, cy, r).  A chunk can hold a crossing of the segment only if the infinite line passes
// within r of its centre and the centre's projection is within max_dist + r of p; every chunk that
// passes is scanned exactly like the brute-force loop, so the result (including tie-breaking) is
// identical to search_ring_brute.

template <bool ST = false, typename RingPtr, typename CirclePtr>
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
    scan_edges<ST>(ring, nr, j0, min(kChunk, nr - j0), px, py, dx, dy, h);
  }
  return h;
}

// ---------------------------------------------------------------------------------------------
// wave64 reductions on the DPP path (no LDS crossbar traffic): an inclusive row scan with
// row_shr 1/2/4/8, then row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3 leaves the total
// in lane 63; v_readlane broadcasts it as a wave-uniform (SGPR) value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double identity, double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(__double2loint(identity), lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(__double2hiint(identity), hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
struct OpSum { static __device__ __forceinline__ double id() { return 0.0; }
               static __device__ __forceinline__ double f(double a, double b) { return a + b; } };
struct OpMax { static __device__ __forceinline__ double id() { return -INFINITY; }
               static __device__ __forceinline__ double f(double a, double b) { return fmax(a, b); } };
struct OpMin { static __device__ __forceinline__ double id() { return INFINITY; }
               static __device__ __forceinline__ double f(double a, double b) { return fmin(a, b); } };

template <typename Op>
__device__ __forceinline__ double wave_reduce(double v) {
  const double e = Op::id();
  v = Op::f(v, dpp_mov<0x111, 0xf>(e, v));  // row_shr:1
  v = Op::f(v, dpp_mov<0x112, 0xf>(e, v));  // row_shr:2
  v = Op::f(v, dpp_mov<0x114, 0xf>(e, v));  // row_shr:4
  v = Op::f(v, dpp_mov<0x118, 0xf>(e, v));  // row_shr:8
  v = Op::f(v, dpp_mov<0x142, 0xa>(e, v));  // row_bcast:15 -> rows 1 and 3
  v = Op::f(v, dpp_mov<0x143, 0xc>(e, v));  // row_bcast:31 -> rows 2 and 3
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
// The same reduction as a butterfly through the LDS crossbar (ds_swizzle xor 1, 2, 4, 8, 16, then ds_bpermute across the two halves):
// ONE vector instruction per stage -- the operation -- where the DPP path spends five (a 64-bit DPP move is two moves, each with its
// `old` operand initialised first); the exchanges are LDS-pipe instructions, of which the sweep kernels, bound by VALU issue, have
// plenty to spare.  Every lane ends with the result.  For order-independent operations only (max, min): a butterfly adds in another
// order than the scan above, so the SUMS of the fast arithmetic stay on the DPP path (their order is their result).
template <typename Op>
__device__ __forceinline__ double wave_reduce_bfly(double v) {
#define RL_BFLY(pattern)                                                                                        \
  {                                                                                                             \
    const int lo_ = __builtin_amdgcn_ds_swizzle(__double2loint(v), pattern);                                    \
    const int hi_ = __builtin_amdgcn_ds_swizzle(__double2hiint(v), pattern);                                    \
    v = Op::f(v, __hiloint2double(hi_, lo_));                                                                   \
  }
  RL_BFLY(0x041F) RL_BFLY(0x081F) RL_BFLY(0x101F) RL_BFLY(0x201F) RL_BFLY(0x401F)   // and 0x1f, or 0, xor 1 / 2 / 4 / 8 / 16
#undef RL_BFLY
  const int other = (((int)threadIdx.x ^ 32) & 63) << 2;
  const int lo_ = __builtin_amdgcn_ds_bpermute(other, __double2loint(v));
  const int hi_ = __builtin_amdgcn_ds_bpermute(other, __double2hiint(v));
  return Op::f(v, __hiloint2double(hi_, lo_));
}
__device__ __forceinline__ double wave_max_bfly(double v) { return wave_reduce_bfly<OpMax>(v); }
__device__ __forceinline__ double wave_min_bfly(double v) { return wave_reduce_bfly<OpMin>(v); }
__device__ __forceinline__ double wave_sum(double v) { return wave_reduce<OpSum>(v); }
__device__ __forceinline__ double wave_max(double v) { return wave_reduce<OpMax>(v); }
__device__ __forceinline__ double wave_min(double v) { return wave_reduce<OpMin>(v); }

// Windowed search, wave-cooperative and exact.  Every lane of the wave calls it (inactive lanes
// pass active = false).
//
// Fast path, per lane, no cross-lane traffic:
//   * scan the kWinChunks = 3 whole chunks (24 edges) around the chunk of the hint (the edge of the last crossing): a
//     chunk-ALIGNED window, so the lanes of one chunk read the same addresses; consecutive lanes ->
//     consecutive LDS addresses.  This yields a crossing at distance d_i on an
//     edge of chunk ce.
//   * certificate that nothing closer exists anywhere else on the ring:
//       - chunks further than kNear chunks (in ring order) from ce:  sep[ce] > 2 d_i, where sep[c] is the
//         smallest gap between chunk c's bounding circle and the circle of any chunk outside
//         [c-kNear, c+kNear] = [c-1, c+1] (precomputed once per instance, k_sweep's prologue).  A point q of such a chunk has
//         |q - h_i| >= sep[ce] (h_i = the crossing found, inside circle ce), hence
//         |q - p_i| >= sep[ce] - d_i > d_i.
//       - the chunks of [ce-kNear, ce+kNear] that are not entirely inside the window: their circles must
//         not come within d_i of p_i.
// Slow path (any lane without a certificate): the wave forms ONE disk that contains the disks
// (p_i, d_i) of those lanes, tests all chunk circles against it 64 at a time (ballot), and scans,
// with wave-uniform LDS addresses, every chunk some lane still needs.
// Either way the result is the lexicographic minimum of (|s|, edge) over a set of edges that
// contains every edge with a crossing not farther than the best one == search_ring_brute, bit for bit.
// STAGED: the window is scanned from the wave's LDS stretch `stg` (staged_base = first vertex of what it holds now, -1 =
// nothing yet); `ring` is then the global ring (slow path and re-staging).
template <bool STAGED = false, bool ST = false, typename RingPtr, typename CirclePtr>
__device__ __forceinline__ Hit search_ring_windowed(RingPtr ring, int nr, CirclePtr circ,
                                                    CirclePtr sep, int nchunk, bool active, int hint,
                                                    double px, double py, double dx, double dy,
                                                    double dlen, const double* cmax, bool skip_guard = false,
                                                    unsigned long long* stamps = nullptr,
                                                    double2* stg = nullptr, int staged_base = -1) {
  const int lane = threadIdx.x & (kWave - 1);
  Hit h{INFINITY, 0.0, kNoEdge};
  const bool windowed = active && hint >= 0 && hint < nr;
  const int lo = window_start(hint, nchunk);  // first edge of the window (the lanes of one chunk read the same addresses)
#ifdef RL_STAMPS
  unsigned long long w0 = 0, w1 = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w0) :: "memory");
#endif
  if constexpr (!STAGED) {
    if (windowed) scan_window<kWinBatch, ST>(ring, nr, lo, lo, px, py, dx, dy, h, cmax);
  } else {
    bool todo = windowed;
    int base = staged_base;
    if (base >= 0) stage_wait();   // requested by the caller (stage_prefetch)
    for (;;) {
      if (base >= 0) {
        int off = lo - base;
        if (off < 0) off += nr;
        // (off & 7): a window is served from the stretch only when it starts on a chunk boundary OF THE STRETCH -- the
        // padded slot of its vertex k is then stage_slot(off) + k + (k >> 3) (StagedWindow).  Window starts and stretch
        // bases are multiples of kChunk, so this only fails behind the seam of a ring whose length is not a multiple of
        // kChunk (off = lo - base + nr); such a lane is re-staged from its own window start (off = 0) below.
        if (todo && (off & (kChunk - 1)) == 0 && off + kWinEdges + 1 <= kStage) {
          scan_window<kStageBatch, ST>(StagedWindow{stg + stage_slot(off)}, nr, 0, lo, px, py, dx, dy, h, cmax, stamps);
          todo = false;
        }
      }
      const unsigned long long left = __ballot(todo);
      if (left == 0ull) break;
      base = __builtin_amdgcn_readlane(lo, __ffsll((long long)left) - 1);   // the first unserved lane's window start
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // nobody still reads the stretch it replaces
      stage_issue(reinterpret_cast<const double2*>(ring), nr, base, stg, lane);
      stage_wait();
    }
  }
#ifdef RL_STAMPS
  asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w1) :: "memory");
  if (stamps) stamps[0] += w1 - w0;   // window scan (loads, sign pass, exact tests)
#endif
  if (skip_guard) return h;
  const bool found = h.best <= 1.0;
  const double di = active ? ((found ? h.best * dlen : dlen) * (1.0 + 1e-9) + 1e-9) : 0.0;
  // is chunk c entirely inside this lane's window?
  auto chunk_in_window = [&](int c) {
    const int e0 = c * kChunk;
    const int cnt = min(kChunk, nr - e0);
    int off = e0 - lo;
    if (off < 0) off += nr;
    return off + cnt - 1 <= kWinEdges - 1;
  };
  bool slow = active;
  if (windowed && found && nchunk > 2 * kNear + 1) {
    const int ce = h.edge / kChunk;
    bool ok = sep[ce] > 2.0 * di;
    // The usual case: the crossing is still in the middle chunk of a window that does not run over the ring's seam -- then
    // ce - 1, ce, ce + 1 ARE the window (three whole chunks) and there is no neighbour left to test.  (A crossing moves out of
    // its chunk only when the line moves by more than a chunk's length in one step; the window is re-centred on it next time.)
    const bool centred = ce * kChunk == lo + kChunk && lo + kWinEdges <= nr;
    if (ok && !centred) {
#pragma unroll
      for (int u = -kNear; u <= kNear; ++u) {
        int c = ce + u;
        if (c < 0) c += nchunk;
        if (c >= nchunk) c -= nchunk;
        if (!chunk_in_window(c)) {
          const double mx = circ[3 * c] - px, my = circ[3 * c + 1] - py;
          const double lim = (di + circ[3 * c + 2]) * (1.0 + 1e-9);
          ok = ok && !(mx * mx + my * my <= lim * lim);
        }
      }
    }
    slow = !ok;
  }
  const unsigned long long sm = __ballot(slow);
  if (sm == 0ull) return h;
  // ---- slow path for the lanes in `sm`
  const int first = __ffsll((long long)sm) - 1, last = 63 - __builtin_clzll(sm);
  auto bcast = [](double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src),
                            __builtin_amdgcn_readlane(__double2loint(v), src));
  };
  const double ccx = 0.5 * (bcast(px, first) + bcast(px, last));
  const double ccy = 0.5 * (bcast(py, first) + bcast(py, last));
  const double ex = px - ccx, ey = py - ccy;
  const double rho = wave_max(slow ? sqrt(ex * ex + ey * ey) + di : 0.0) * (1.0 + 1e-9);
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
      if (slow) {
        if (!(windowed && chunk_in_window(cc))) {
          const double mx = circ[3 * cc] - px, my = circ[3 * cc + 1] - py;
          const double lim = (di + circ[3 * cc + 2]) * (1.0 + 1e-9);
          need = mx * mx + my * my <= lim * lim;
        }
      }
      if (__any(need)) {
        Hit h2 = h;
        scan_edges_batched<kChunk, ST>(ring, nr, e0, cnt, px, py, dx, dy, h2);
        if (slow) h = h2;
      }
    }
  }
  return h;
}

}  // namespace rl
