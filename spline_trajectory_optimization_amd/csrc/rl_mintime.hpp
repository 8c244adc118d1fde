// rl_mintime.hpp -- the SOLVE of the min-time double-track NLP (BASELINE config 5):
// a batched primal-dual interior-point SQP-type iteration, one QP subproblem per iteration.
//
// The NLP is the reference's: min_time_optm/min_time_optimizer.py:93-163 (variables, scaling :109-113,
// objective :119-123, node pairing, bounds) with models/double_track.py:10-204 (dynamics, Hermite-Simpson
// defect, tyre ellipses, load transfer, actuator limits).  The reference gives it to IPOPT through
// casadi.Opti (:158-161); IPOPT is a third-party solver absent from this image, so what is built here is
// the same kind of method written for the GPU (CPU twin: oracle/sqp_twin.py, same iteration, independent
// derivatives and linear algebra):
//
//   per node j, 9 scaled unknowns  w_j = (n, xi, omega, beta, v | F, delta, gamma | t) / scale
//       (the abscissa is pinned by an equality, u[1] only enters the objective: both eliminated)
//   7 equalities    Hermite-Simpson defect with node j+1 (6), load-transfer residual          c_j(w_j, w_j+1) = 0
//   17 inequalities tyre ellipses (4), power, v >= 1, force / steer limits, force / steer RATE limits
//       (two rows each), lateral limits, t >= 0                                   g_j(w_j, w_j+1) + s_j = 0, s_j > 0
//
//   every iteration:  k_mt_values   functions; values of the dynamics at the two ends -> midpoint
//                     k_mt_jac_dirs   Jacobians of the three dynamics evaluations of a pair (forward duals)
//                     k_mt_hes_point  the EXACT Hessians of the weighted dynamics at the midpoint and the ends (forward over
//                                   forward duals; the pairs with the lateral offset / heading in closed form)
//                     k_mt_node     chain rule through the Hermite-Simpson midpoint: Jacobian and Hessian of the pair, the
//                                   16 x 16 diagonal / coupling blocks and the right-hand side, in one pass over runs of pairs
//                     k_mt_prepare2 convergence test, barrier update, rhs = rhs0 - mu r1
//                     k_mt_kkt4     the barrier QP  [K  A'; A  -eps I] [dw; dy] = rhs  with
//                                   K = H_cost + Hess + G' S^-1 Z G + delta I: block tridiagonal and CYCLIC
//                                   (closed lap), 16 x 16 blocks (9 unknowns + 7 multipliers per node),
//                                   factorised node by node by four fronts (block LDL' without pivoting; the lap closure
//                                   is a border block row carried along); the count of negative pivots
//                                   (7 N of 16 N) is the inertia test that drives delta
//                     k_mt_dir, k_mt_trial, k_mt_step
//                                   fraction-to-the-boundary rule, backtracking on (infeasibility, barrier
//                                   objective) with a small filter, update of w, s, y, z, delta
//   Four waves per instance in k_mt_kkt4 (two in k_mt_kkt, its cross-check and the kernel of short laps; every 16 x 16 block lives in registers, 4 entries per lane, in the
//   operand layout of v_mfma_f64_16x16x4_f64: the block products run on the matrix cores, the block
//   inverse on cross-lane moves); everything else is node- or row-parallel over the whole batch, and one workgroup per
//   instance only takes decisions.  (k_mt_derivs<1/2>, k_mt_jac_assemble, k_mt_hes_assemble, k_mt_prepare, k_mt_assemble:
//   the cross-check paths RL_MT_HES_SWEEP=1 / RL_MT_UNFUSED=1.)
#pragma once
#include "rl_dtrack.hpp"
#include "rl_device.hpp"

namespace rl {

constexpr int kMtNv = 9, kMtNe = 7, kMtNi = 17, kMtNf = kMtNe + kMtNi, kMtNb = kMtNv + kMtNe;  // 16
constexpr int kMtLoc = 2 * kMtNv;  // 18 local unknowns of a pair: own 9 + next node's 9
constexpr int kMtJacSlices = kMtLoc / 3;
// Hessian slices: one thread differentiates along ONE pair of local unknowns (a <= b) with nested duals of
// (1 + 1)(1 + 1) = 4 doubles per value; the model keeps ~50 values alive, so wider direction groups spill:
// measured on MI355X (MGKT, 256 instances, whole solve) 1 x 1 directions (0.5 KB of spills per lane) 4.7 s,
// 1 x 2 8.4 s, 1 x 3 8.8 s, 3 x 3 (21 slices, 8.5 KB of spills) 13.8 s.
// Structure of the pair functions (mt_pair_emit): the next node's gamma and t (local 16, 17) do not occur; its
// F and delta (14, 15) occur only in the rate rows (u' - u) / t, i.e. together with this node's F, delta, t.
// Those pairs are never evaluated (their entries of `hes` stay at the zero they are initialised with).
__host__ __device__ constexpr bool mt_hes_pair_kept(int a, int b) {   // a <= b
  if (b >= 16) return false;
  if (b >= 14) return a >= 5 && a <= 8;
  return true;
}
constexpr int mt_hes_slices() {
  int c = 0;
  for (int b = 0; b < kMtLoc; ++b)
    for (int a = 0; a <= b; ++a) c += mt_hes_pair_kept(a, b) ? 1 : 0;
  return c;
}
constexpr int kMtHesSlices = mt_hes_slices();   // 113 of the 171 pairs
constexpr double kMtEpsReg = 1e-8;  // dual regularisation of the KKT system
constexpr double kMtThetaGrowth = 2.0, kMtThetaFloor = 1e-5;
constexpr int kMtFilter = 8;   // entries of the filter (a ring: the oldest is overwritten)
constexpr double kMtCostDiag = 2e-4 + 4e-1, kMtCostOff = -2e-1;  // Hessian of 1e-4 |U|^2 + 1e-1 |dU|^2 (:119-123)

struct MtProblem {
  double p[DT_NPARAM];
  int N;
  const double* s; const double* kappa; const double* left; const double* right;  // [N]
  int bounds_per_instance;   // left / right are [B,N] instead of [N]
  double margin, track_length;
  double sw[kMtNv];   // scale of the unknowns: scale_x[1..5], scale_u[0,2,3], 1
  double se[kMtNe];   // row scale of the equalities: 1 / scale_x (6), 1 / scale_u[3]
};

template <typename S> struct MtLift;
template <> struct MtLift<double> { static __device__ __forceinline__ double make(double v) { return v; } };
template <int ND, typename T> struct MtLift<Dual<ND, T>> {
  static __device__ __forceinline__ Dual<ND, T> make(double v) {
    Dual<ND, T> r; r.v = MtLift<T>::make(v);
#pragma unroll
    for (int i = 0; i < ND; ++i) r.d[i] = MtLift<T>::make(0.0);
    return r;
  }
};

// eq (7, row-scaled) and g (17, row-scaled) of the pair (j, j+1) from the scaled unknowns of both nodes.  The
// rows are handed to a SINK as soon as they exist -- sink.eq(c, v) / sink.g(c, v) ADD v to row c; the defect
// rows arrive in two pieces -- so that a caller that only needs a weighted sum of the rows (the Lagrangian of
// the Hessian kernel) never holds the 24 of them at once.
template <typename S, typename Sink>
__device__ __forceinline__ void mt_pair_emit(const MtProblem& P, int j, const S (&wo)[kMtNv], const S (&wn)[kMtNv], Sink& sink) {
  const double* p = P.p;
  const int jn = j + 1 == P.N ? 0 : j + 1;
  S x[6], xn[6], u[4];
  x[0] = MtLift<S>::make(P.s[j]); xn[0] = MtLift<S>::make(P.s[jn]);
#pragma unroll
  for (int c = 0; c < 5; ++c) { x[1 + c] = wo[c] * P.sw[c]; xn[1 + c] = wn[c] * P.sw[c]; }
  u[0] = wo[5] * P.sw[5]; u[1] = MtLift<S>::make(0.0); u[2] = wo[6] * P.sw[6]; u[3] = wo[7] * P.sw[7];
  const S t = wo[8] * P.sw[8];
  const double k = P.kappa[j];
  const double su0 = P.sw[5], su2 = P.sw[6], sx1 = P.sw[0], sx5 = P.sw[4];
  {  // the rows that only need the unknowns themselves
    const S un0 = wn[5] * P.sw[5], un2 = wn[6] * P.sw[6];
    const S v = x[5], delta = u[2];
    const S fd = u[0] * (m_tanh(u[0]) * 0.5 + 0.5);
    sink.g(4, (v * fd - p[DT_PMAX]) / p[DT_PMAX]);
    sink.g(5, (1.0 - v) / sx5);
    sink.g(6, (p[DT_FB_MAX] - u[0]) / su0); sink.g(7, (u[0] - p[DT_FD_MAX]) / su0);
    sink.g(8, (-p[DT_DELTA_MAX] - delta) / su2); sink.g(9, (delta - p[DT_DELTA_MAX]) / su2);
    const S ru = (un0 - u[0]) / t, rd = (un2 - delta) / t;
    sink.g(10, (p[DT_FB_MAX] / p[DT_TB] - ru) / su0); sink.g(11, (ru - p[DT_FD_MAX] / p[DT_TD]) / su0);
    sink.g(12, (-p[DT_DELTA_MAX] / p[DT_TDELTA] - rd) / su2); sink.g(13, (rd - p[DT_DELTA_MAX] / p[DT_TDELTA]) / su2);
    sink.g(14, ((P.right[j] + P.margin) - x[1]) / sx1); sink.g(15, (x[1] - (P.left[j] - P.margin)) / sx1);
    sink.g(16, -t);
  }
  {  // utils/utils.py:10-18
    const S d = xn[2] - x[2];
    S sdd, cdd;
    m_sincos(d, sdd, cdd);
    xn[2] = m_atan2(sdd, cdd) + x[2];
    const S ds = x[0] - xn[0], kk = m_abs(ds) + P.track_length / 2.0;
    xn[0] = xn[0] + (kk - m_fmod(kk, P.track_length)) * dt_sign(m_val(ds));
  }
  S f1[6], f2[6], xm[6];
  {
    DtTyres<S> ty;
    dt_dynamics(p, x, u, k, f1, ty);
    const S delta = u[2], gam = u[3];
    S sd, cd_;
    m_sincos(delta, sd, cd_);
    sink.eq(6, (gam - p[DT_HCOG] / (0.5 * (p[DT_TWF] + p[DT_TWR])) *
                         (ty.fy[2] + ty.fy[3] + (ty.fx[0] + ty.fx[1]) * sd + (ty.fy[0] + ty.fy[1]) * cd_)) * P.se[6]);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const S qx = ty.fx[w] / (p[DT_MU] * ty.fz[w]), qy = ty.fy[w] / (p[DT_MU] * ty.fz[w]);
      sink.g(w, qx * qx + qy * qy - 1.0);
    }
  }
  {
    DtTyres<S> ty2;
    dt_dynamics(p, xn, u, k, f2, ty2);
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    xm[c] = 0.5 * (x[c] + xn[c]) + (t / 8.0) * (f1[c] - f2[c]);
    sink.eq(c, (x[c] + (t / 6.0) * (f1[c] + f2[c]) - xn[c]) * P.se[c]);
  }
  {
    S fm[6];
    DtTyres<S> ty2;
    dt_dynamics(p, xm, u, k, fm, ty2);
#pragma unroll
    for (int c = 0; c < 6; ++c) sink.eq(c, ((t / 6.0) * (4.0 * fm[c])) * P.se[c]);
  }
}

template <typename S>
struct MtRowSink {     // all 24 rows (function values, Jacobian slices)
  S (&eq_)[kMtNe];
  S (&g_)[kMtNi];
  __device__ __forceinline__ void eq(int c, const S& v) { eq_[c] = eq_[c] + v; }
  __device__ __forceinline__ void g(int c, const S& v) { g_[c] = g_[c] + v; }
};

template <typename S>
__device__ __forceinline__ void mt_pair(const MtProblem& P, int j, const S (&wo)[kMtNv], const S (&wn)[kMtNv],
                                        S (&eq)[kMtNe], S (&g)[kMtNi]) {
#pragma unroll
  for (int c = 0; c < kMtNe; ++c) eq[c] = MtLift<S>::make(0.0);
#pragma unroll
  for (int c = 0; c < kMtNi; ++c) g[c] = MtLift<S>::make(0.0);
  MtRowSink<S> sink{eq, g};
  mt_pair_emit<S>(P, j, wo, wn, sink);
}

// ------------------------------------------------------------------------------------------------
struct MtState {       // per-batch device arrays, instance-major
  int B, N;
  double* w;      // [B,N,9]  scaled unknowns
  double* s;      // [B,N,17] slacks
  double* y;      // [B,N,7]  equality multipliers
  double* z;      // [B,N,17] inequality multipliers
  double* fun;    // [B,N,24] eq | g at w
  double* jac;    // [B,N,24,18]
  double* hes;    // [B,N,18,18] Hessian of y.eq + z.g of the pair
  double* dw;     // [B,N,9]
  double* dy;     // [B,N,7]
  double* blk;    // [B,N,3,256] factor blocks of k_mt_kkt, per node: P [256], Q [256], a' [16] (+ padding)
  double* vec;    // [B,N,16] right-hand side / solution scratch
  double* dblk;   // [B,N,256] assembled diagonal blocks D_j (without delta)
  double* eblk;   // [B,N,256] assembled coupling blocks E_j = M[j+1][j]
  double* rhs;    // [B,N,16]  assembled right-hand sides
  double* r1;     // [B,N,16]  k_mt_node: 0..8 the coefficient of mu in the right-hand side, 9 the node's max |r_d|
  double* gc;     // [B,N,64]  the 60 entries of the inequality Jacobian that can be non-zero (kMtGc*), what k_mt_dir reads
  double* hw;     // [B,N,kMtHw] work array of the chain-rule Hessian (k_mt_hes_*)
  double* filt;   // [B,16] filter: (infeasibility, barrier objective) of up to 8 earlier iterates of the current barrier problem
  double* scal;   // [B,16] per-instance scalars: 0 mu, 1 delta, 2 kkt, 3 viol, 4 compl, 5 status (0 run, 1 converged,
                  //        2 failed), 6 iterations, 7 last alpha, 8 theta0, 9 phi0, 10 refactorisations, 11 lap time
  double tol;
  // strategy constants (defaults in rl_mincurv.hip; the RL_MT_* environment switches exist for experiments)
  double d_down, d_up, a_hi, a_lo, mu_fac, mu_pow, mu_kappa, th_filter, dual_cap;
};

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mt_instance(MtProblem& P, int b) {
  if (P.bounds_per_instance) { P.left += (size_t)b * P.N; P.right += (size_t)b * P.N; }
}

// k_mt_derivs<KIND>: one thread per (node, instance, slice).  KIND 0: functions (1 slice), 1: Jacobian (6 slices
// of three directions), 2: Hessian (kMtHesSlices pairs of direction groups) -- three kernels, so that the cheap
// ones do not inherit the register budget of the forward-over-forward one.
template <int KIND>
__global__ void __launch_bounds__(64) k_mt_derivs(MtProblem P, MtState st) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  const int slice = KIND == 0 ? 0 : (KIND == 1 ? 1 + (int)blockIdx.z : 1 + kMtJacSlices + (int)blockIdx.z);
  if (j >= P.N) return;
  if (st.scal[(size_t)b * 16 + 5] != 0.0) return;   // instance finished
  mt_instance(P, b);
  const int N = P.N, jn = j + 1 == N ? 0 : j + 1;
  const double* wo_ = st.w + ((size_t)b * N + j) * kMtNv;
  const double* wn_ = st.w + ((size_t)b * N + jn) * kMtNv;
  const size_t o = (size_t)b * N + j;
  if (KIND == 0) {
    double wo[kMtNv], wn[kMtNv], eq[kMtNe], g[kMtNi];
#pragma unroll
    for (int a = 0; a < kMtNv; ++a) { wo[a] = wo_[a]; wn[a] = wn_[a]; }
    mt_pair<double>(P, j, wo, wn, eq, g);
#pragma unroll
    for (int c = 0; c < kMtNe; ++c) st.fun[o * kMtNf + c] = eq[c];
#pragma unroll
    for (int c = 0; c < kMtNi; ++c) st.fun[o * kMtNf + kMtNe + c] = g[c];
  } else if (KIND == 1) {
    using D = Dual<3>;
    const int v0 = 3 * (slice - 1);
    D wo[kMtNv], wn[kMtNv], eq[kMtNe], g[kMtNi];
#pragma unroll
    for (int a = 0; a < kMtNv; ++a) {
      wo[a].v = wo_[a]; wn[a].v = wn_[a];
#pragma unroll
      for (int i = 0; i < 3; ++i) { wo[a].d[i] = (a == v0 + i) ? 1.0 : 0.0; wn[a].d[i] = (kMtNv + a == v0 + i) ? 1.0 : 0.0; }
    }
    mt_pair<D>(P, j, wo, wn, eq, g);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int c = 0; c < kMtNe; ++c) st.jac[(o * kMtNf + c) * kMtLoc + v0 + i] = eq[c].d[i];
#pragma unroll
      for (int c = 0; c < kMtNi; ++c) st.jac[(o * kMtNf + kMtNe + c) * kMtLoc + v0 + i] = g[c].d[i];
    }
  } else {
    // Hessian of L = y . eq + z . g: slice = the q-th kept pair (ga, gb), ga <= gb, of local unknowns
    constexpr int NA = 1, NB = 1;
    int q = (int)blockIdx.z, ga = 0, gb = 0;
    for (gb = 0; gb < kMtLoc; ++gb) {
      int cnt = 0;
      for (int a = 0; a <= gb; ++a) cnt += mt_hes_pair_kept(a, gb) ? 1 : 0;
      if (q < cnt) {
        for (ga = 0;; ++ga) {
          if (!mt_hes_pair_kept(ga, gb)) continue;
          if (q == 0) break;
          --q;
        }
        break;
      }
      q -= cnt;
    }
    using D1 = Dual<NA>;
    using D2 = Dual<NB, D1>;
    D2 wo[kMtNv], wn[kMtNv];
    auto seed = [&](D2& r, double val, int var) {
      r.v.v = val;
#pragma unroll
      for (int ia = 0; ia < NA; ++ia) r.v.d[ia] = (var == NA * ga + ia) ? 1.0 : 0.0;   // inner directions: group a
#pragma unroll
      for (int ib = 0; ib < NB; ++ib) {
        r.d[ib].v = (var == NB * gb + ib) ? 1.0 : 0.0;                                   // outer directions: group b
#pragma unroll
        for (int ia = 0; ia < NA; ++ia) r.d[ib].d[ia] = 0.0;
      }
    };
#pragma unroll
    for (int a = 0; a < kMtNv; ++a) { seed(wo[a], wo_[a], a); seed(wn[a], wn_[a], kMtNv + a); }
    // the sink keeps only the mixed second derivatives of the Lagrangian
    struct LagrSink {
      const double* y; const double* z;
      double h[NB][NA];
      __device__ __forceinline__ void add(double m, const D2& v) {
#pragma unroll
        for (int ib = 0; ib < NB; ++ib)
#pragma unroll
          for (int ia = 0; ia < NA; ++ia) h[ib][ia] = fma(m, v.d[ib].d[ia], h[ib][ia]);
      }
      __device__ __forceinline__ void eq(int c, const D2& v) { add(y[c], v); }
      __device__ __forceinline__ void g(int c, const D2& v) { add(z[c], v); }
    } sink;
    sink.y = st.y + o * kMtNe; sink.z = st.z + o * kMtNi;
#pragma unroll
    for (int ib = 0; ib < NB; ++ib)
#pragma unroll
      for (int ia = 0; ia < NA; ++ia) sink.h[ib][ia] = 0.0;
    mt_pair_emit<D2>(P, j, wo, wn, sink);
    const auto& h = sink.h;
    // every entry (a <= b) lies in exactly one slice: it writes the entry and its mirror image
    double* H = st.hes + o * kMtLoc * kMtLoc;
#pragma unroll
    for (int ib = 0; ib < NB; ++ib)
#pragma unroll
      for (int ia = 0; ia < NA; ++ia) {
        const int va = NA * ga + ia, vb = NB * gb + ib;
        if (va <= vb) { H[vb * kMtLoc + va] = h[ib][ia]; H[va * kMtLoc + vb] = h[ib][ia]; }
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Hessian of the pair Lagrangian by the CHAIN RULE through the Hermite-Simpson midpoint (the default; the
// forward-over-forward sweep over the whole pair function, k_mt_derivs<2>, stays as the cross-check).
//
// With Y = (n, xi, omega, beta, v), U = (F, delta, gamma), f = the dynamics (6 rows), F = its rows 1..5,
// yh = y * se (the defect multipliers), the only rows with curvature are
//     L = (t/6) yh.f(Y,U) + (t/6) yh.f(Y',U) + (2t/3) yh.f(Ym,U) + own(Y,U) + rate(F,F',delta,delta',t),
//     Ym = (Y + Y')/2 + (t/8) (F(Y,U) - F(Y',U)),
// own = load-transfer row, four tyre ellipses, power row.  Every dynamics evaluation is differentiated on its own
// 8 variables (36 pairs of directions instead of 105 of the 14 that the pair function depends on, one dynamics
// evaluation per pair instead of three):
//   values   f(Y,U), f(Y',U) -> Ym, F1 - F2                                               k_mt_hes_values
//   middle   gradient gm and Hessian Hm of phi = yh.f at (Ym,U)                            k_mt_hes_point<0>
//   ends     Jacobians J1, J2 of f at (Y,U), (Y',U) (and Jm at the midpoint: the Jacobian    k_mt_jac_dirs
//            of the pair rows is assembled from the three by the same chain rule)            k_mt_jac_assemble
//            Hessians H1, H2 of  (t/6) yh.f +- (t^2/12) gm_Y.F (+ own at the first point)   k_mt_hes_point<1>
//   assembly H = H1 (+) H2 + (2t/3) M' Hm M + the products with t + the rate rows, scaled   k_mt_hes_assemble
// with M = d(Ym,U)/d(Y,U,t,Y').  3.1 times fewer instructions than the sweep, and fewer live values per thread.
constexpr int kMtPv = 8;
constexpr int kMtHwXm = 0, kMtHwDf = 5, kMtHwGm = 10, kMtHwHm = 18, kMtHwJ1 = 54, kMtHwJ2 = 102, kMtHwH1 = 150, kMtHwH2 = 186,
              kMtHwJm = 222, kMtHwOw = 270, kMtHwF = 318, kMtHw = 336;   // Jm [6][8], own-row gradients [6][8], f1 f2 fm [3][6]

__device__ __forceinline__ int mt_tri8(int a, int b) {               // index of the pair (a, b) of 8, any order: row-wise upper triangle
  const int lo = a < b ? a : b, hi = a < b ? b : a;
  return lo * kMtPv - lo * (lo - 1) / 2 + (hi - lo);
}

// physical unknowns of the pair; the next state is brought next to this one (utils/utils.py:10-18: locally the identity)
__device__ __forceinline__ void mt_phys(const MtProblem& P, const double* wo, const double* wn, double (&Y)[5], double (&Yn)[5],
                                        double (&U)[3], double& t) {
#pragma unroll
  for (int c = 0; c < 5; ++c) { Y[c] = wo[c] * P.sw[c]; Yn[c] = wn[c] * P.sw[c]; }
  U[0] = wo[5] * P.sw[5]; U[1] = wo[6] * P.sw[6]; U[2] = wo[7] * P.sw[7];
  t = wo[8] * P.sw[8];
  const double d = Yn[1] - Y[1];
  double sd, cd;
  m_sincos(d, sd, cd);
  Yn[1] = m_atan2(sd, cd) + Y[1];
}

template <typename S>
__device__ __forceinline__ void mt_dyn(const MtProblem& P, int j, const S (&Y)[5], const S (&U)[3], S (&f)[6], DtTyres<S>& ty) {
  S x[6], u[4];
  x[0] = MtLift<S>::make(P.s[j]);
#pragma unroll
  for (int c = 0; c < 5; ++c) x[1 + c] = Y[c];
  u[0] = U[0]; u[1] = MtLift<S>::make(0.0); u[2] = U[1]; u[3] = U[2];
  dt_dynamics(P.p, x, u, P.kappa[j], f, ty);
}

// the rows of this node that are curved and depend on (Y, U) only, weighted with their multipliers
template <typename S>
__device__ __forceinline__ S mt_own_rows(const MtProblem& P, const S (&Y)[5], const S (&U)[3], const DtTyres<S>& ty,
                                         const double* y, const double* z) {
  const double* p = P.p;
  S sd, cd_;
  m_sincos(U[1], sd, cd_);
  S L = (U[2] - p[DT_HCOG] / (0.5 * (p[DT_TWF] + p[DT_TWR])) *
                    (ty.fy[2] + ty.fy[3] + (ty.fx[0] + ty.fx[1]) * sd + (ty.fy[0] + ty.fy[1]) * cd_)) * (P.se[6] * y[6]);
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const S qx = ty.fx[w] / (p[DT_MU] * ty.fz[w]), qy = ty.fy[w] / (p[DT_MU] * ty.fz[w]);
    L = L + (qx * qx + qy * qy - 1.0) * z[w];
  }
  const S fd = U[0] * (m_tanh(U[0]) * 0.5 + 0.5);
  L = L + ((Y[4] * fd - p[DT_PMAX]) / p[DT_PMAX]) * z[4];
  return L;
}

// values of the two end evaluations -> midpoint and F1 - F2
__device__ __forceinline__ void mt_hes_values_body(const MtProblem& P, const MtState& st, int b, int j) {
  const int N = P.N;
  const int jn = j + 1 == N ? 0 : j + 1;
  double Y[5], Yn[5], U[3], t;
  mt_phys(P, st.w + ((size_t)b * N + j) * kMtNv, st.w + ((size_t)b * N + jn) * kMtNv, Y, Yn, U, t);
  double f1[6], f2[6];
  {
    DtTyres<double> ty;
    mt_dyn<double>(P, j, Y, U, f1, ty);
    mt_dyn<double>(P, j, Yn, U, f2, ty);
  }
  double* hw = st.hw + ((size_t)b * N + j) * kMtHw;
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    const double df = f1[1 + c] - f2[1 + c];
    hw[kMtHwXm + c] = 0.5 * (Y[c] + Yn[c]) + (t / 8.0) * df;
    hw[kMtHwDf + c] = df;
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) { hw[kMtHwF + c] = f1[c]; hw[kMtHwF + 6 + c] = f2[c]; }
}
// functions eq, g of the pair at the current point -> st.fun
__device__ __forceinline__ void mt_fun_body(MtProblem P, const MtState& st, int b, int j) {
  mt_instance(P, b);
  const int N = P.N, jn = j + 1 == N ? 0 : j + 1;
  const double* wo_ = st.w + ((size_t)b * N + j) * kMtNv;
  const double* wn_ = st.w + ((size_t)b * N + jn) * kMtNv;
  const size_t o = (size_t)b * N + j;
  double wo[kMtNv], wn[kMtNv], eq[kMtNe], g[kMtNi];
#pragma unroll
  for (int a = 0; a < kMtNv; ++a) { wo[a] = wo_[a]; wn[a] = wn_[a]; }
  mt_pair<double>(P, j, wo, wn, eq, g);
#pragma unroll
  for (int c = 0; c < kMtNe; ++c) st.fun[o * kMtNf + c] = eq[c];
#pragma unroll
  for (int c = 0; c < kMtNi; ++c) st.fun[o * kMtNf + kMtNe + c] = g[c];
}
// grid (node blocks, B): values of the two end evaluations -> midpoint and F1 - F2
__global__ void __launch_bounds__(64) k_mt_hes_values(MtProblem P, MtState st) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (j >= P.N || st.scal[(size_t)b * 16 + 5] != 0.0) return;
  mt_hes_values_body(P, st, b, j);
}
// grid (node blocks, B): both of the above in one launch (the first kernel of an iteration of the default path)
__global__ void __launch_bounds__(64) k_mt_values(MtProblem P, MtState st) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (j >= P.N || st.scal[(size_t)b * 16 + 5] != 0.0) return;
  mt_fun_body(P, st, b, j);
  mt_hes_values_body(P, st, b, j);
}

// the curved rows of this node that depend on (Y, U) only, one by one (row order: eq 6, g 0..3, g 4)
template <typename S>
__device__ __forceinline__ void mt_own_rows_each(const MtProblem& P, const S (&Y)[5], const S (&U)[3], const DtTyres<S>& ty,
                                                 S (&r)[6]) {
  const double* p = P.p;
  S sd, cd_;
  m_sincos(U[1], sd, cd_);
  r[0] = (U[2] - p[DT_HCOG] / (0.5 * (p[DT_TWF] + p[DT_TWR])) *
                     (ty.fy[2] + ty.fy[3] + (ty.fx[0] + ty.fx[1]) * sd + (ty.fy[0] + ty.fy[1]) * cd_)) * P.se[6];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const S qx = ty.fx[w] / (p[DT_MU] * ty.fz[w]), qy = ty.fy[w] / (p[DT_MU] * ty.fz[w]);
    r[1 + w] = qx * qx + qy * qy - 1.0;
  }
  const S fd = U[0] * (m_tanh(U[0]) * 0.5 + 0.5);
  r[5] = (Y[4] * fd - p[DT_PMAX]) / p[DT_PMAX];
}

// grid (blocks of 10 nodes, B, 3): Jacobians of f at the two ends and at the midpoint, one direction per thread; the
// first end also differentiates its own curved rows, the midpoint leaves the value of f there.  Lane = (node, direction):
// the directions of a node write consecutive doubles of its rows (instead of lone 8 B words 2.7 KB apart) and read the
// node's unknowns as broadcasts.  Only the six variables of the force and moment balance (omega, beta, v, F, delta, gamma)
// go through the dual arithmetic; the lateral offset and the heading enter through s' = v cos(xi + beta) / (1 - n kappa),
// n' = v sin(xi + beta), xi' = omega - kappa s' only (see k_mt_hes_point): their two columns are closed forms, written by
// the first two lanes of the node.
constexpr int kMtJacNodes = 10, kMtJacDirs = 6;
__global__ void __launch_bounds__(64) k_mt_jac_dirs(MtProblem P, MtState st) {
  const int nl = (int)threadIdx.x / kMtJacDirs, dl = (int)threadIdx.x - kMtJacDirs * nl;
  const int j = blockIdx.x * kMtJacNodes + nl, b = blockIdx.y, N = P.N;
  const int pt = blockIdx.z, d = 2 + dl;
  if (nl >= kMtJacNodes || j >= N || st.scal[(size_t)b * 16 + 5] != 0.0) return;
  const int jn = j + 1 == N ? 0 : j + 1;
  double Y[5], Yn[5], U[3], t;
  mt_phys(P, st.w + ((size_t)b * N + j) * kMtNv, st.w + ((size_t)b * N + jn) * kMtNv, Y, Yn, U, t);
  double* hw = st.hw + ((size_t)b * N + j) * kMtHw;
  using D = Dual<1>;
  D Yd[5], Ud[3], f[6];
  double Yp[5];
#pragma unroll
  for (int c = 0; c < 5; ++c) { Yp[c] = pt == 0 ? Y[c] : (pt == 1 ? Yn[c] : hw[kMtHwXm + c]); Yd[c].v = Yp[c]; Yd[c].d[0] = d == c ? 1.0 : 0.0; }
#pragma unroll
  for (int c = 0; c < 3; ++c) { Ud[c].v = U[c]; Ud[c].d[0] = d == 5 + c ? 1.0 : 0.0; }
  DtTyres<D> ty;
  mt_dyn<D>(P, j, Yd, Ud, f, ty);
  double* J = hw + (pt == 0 ? kMtHwJ1 : (pt == 1 ? kMtHwJ2 : kMtHwJm));
#pragma unroll
  for (int c = 0; c < 6; ++c) J[c * kMtPv + d] = f[c].d[0];
  if (pt == 0) {
    D r[6];
    mt_own_rows_each<D>(P, Yd, Ud, ty, r);
#pragma unroll
    for (int c = 0; c < 6; ++c) hw[kMtHwOw + c * kMtPv + d] = r[c].d[0];
  }
  if (dl < 2) {   // column 0 (n) or 1 (xi) of the six rows, in closed form; the own rows do not depend on either
    const double kap = P.kappa[j], v = Yp[4], D_ = 1.0 / (1.0 - Yp[0] * kap);
    double sn, cs;
    m_sincos(Yp[1] + Yp[3], sn, cs);
    const double d0 = dl == 0 ? v * cs * kap * D_ * D_ : -v * sn * D_;   // d s' / d n, d s' / d xi
    const double d1 = dl == 0 ? 0.0 : v * cs;                            // d n'
    J[0 * kMtPv + dl] = d0; J[1 * kMtPv + dl] = d1; J[2 * kMtPv + dl] = -kap * d0;
#pragma unroll
    for (int c = 3; c < 6; ++c) J[c * kMtPv + dl] = 0.0;
    if (pt == 0) {
#pragma unroll
      for (int c = 0; c < 6; ++c) hw[kMtHwOw + c * kMtPv + dl] = 0.0;
    }
  }
  if (pt == 2 && dl == 0) {
#pragma unroll
    for (int c = 0; c < 6; ++c) hw[kMtHwF + 12 + c] = f[c].v;
  }
}

// grid (N, B), one wave per node: the 24 x 18 Jacobian of the pair rows in the scaled unknowns, from J1, J2, Jm by
// the chain rule (defect rows), the own-row gradients, and the closed forms of the linear and the rate rows
__global__ void __launch_bounds__(64) k_mt_jac_assemble(MtProblem P, MtState st) {
  __shared__ double M[kMtPv][14], JM[6][14];
  const int j = blockIdx.x, b = blockIdx.y, lane = threadIdx.x, N = P.N;
  if (st.scal[(size_t)b * 16 + 5] != 0.0) return;
  const int jn = j + 1 == N ? 0 : j + 1;
  const size_t o = (size_t)b * N + j;
  const double* wo = st.w + o * kMtNv;
  const double* wn = st.w + ((size_t)b * N + jn) * kMtNv;
  const double* hw = st.hw + o * kMtHw;
  const double t = wo[8] * P.sw[8];
  const double* J1 = hw + kMtHwJ1;
  const double* J2 = hw + kMtHwJ2;
  const double* Jm = hw + kMtHwJm;
  for (int e = lane; e < kMtPv * 14; e += 64) {   // M = d(Ym, U) / d(Y, U, t, Y')  (as in k_mt_hes_assemble)
    const int r = e / 14, zi = e - 14 * r;
    double v = 0.0;
    if (r < 5) {
      if (zi < 5) v = (zi == r ? 0.5 : 0.0) + (t / 8.0) * J1[(1 + r) * kMtPv + zi];
      else if (zi < 8) v = (t / 8.0) * (J1[(1 + r) * kMtPv + zi] - J2[(1 + r) * kMtPv + zi]);
      else if (zi == 8) v = hw[kMtHwDf + r] / 8.0;
      else v = (zi - 9 == r ? 0.5 : 0.0) - (t / 8.0) * J2[(1 + r) * kMtPv + (zi - 9)];
    } else {
      v = zi == r ? 1.0 : 0.0;
    }
    M[r][zi] = v;
  }
  __syncthreads();
  for (int e = lane; e < 6 * 14; e += 64) {
    const int c = e / 14, zi = e - 14 * c;
    double v = 0.0;
#pragma unroll
    for (int p = 0; p < kMtPv; ++p) v += Jm[c * kMtPv + p] * M[p][zi];
    JM[c][zi] = v;
  }
  __syncthreads();
  const double ru = (wn[5] - wo[5]) * P.sw[5] / t, rd = (wn[6] - wo[6]) * P.sw[6] / t;
  double* Jout = st.jac + o * kMtNf * kMtLoc;
  for (int e = lane; e < kMtNf * kMtLoc; e += 64) {
    const int row = e / kMtLoc, col = e - kMtLoc * row;
    double v = 0.0;   // derivative with respect to the PHYSICAL unknown of column col
    if (row < 6) {
      const int c = row;
      if (col < 14) {
        const int zi = col;
        double dv = 4.0 * JM[c][zi];
        if (zi < 5) dv += J1[c * kMtPv + zi];
        else if (zi < 8) dv += J1[c * kMtPv + zi] + J2[c * kMtPv + zi];
        else if (zi >= 9) dv += J2[c * kMtPv + (zi - 9)];
        v = (t / 6.0) * dv;
        if (zi == 8) v += (hw[kMtHwF + c] + 4.0 * hw[kMtHwF + 12 + c] + hw[kMtHwF + 6 + c]) / 6.0;
        if (zi < 5 && c == 1 + zi) v += 1.0;
        if (zi >= 9 && c == 1 + (zi - 9)) v -= 1.0;
        v *= P.se[c];
      }
    } else if (row <= 11) {                 // eq 6, g 0..3, g 4: own rows of (Y, U)
      if (col < kMtPv) v = hw[kMtHwOw + (row - 6) * kMtPv + col];
    } else {
      const int gq = row - kMtNe;           // 5 .. 16
      const double su0 = P.sw[5], su2 = P.sw[6];
      if (gq == 5) v = col == 4 ? -1.0 / P.sw[4] : 0.0;
      else if (gq == 6) v = col == 5 ? -1.0 / su0 : 0.0;
      else if (gq == 7) v = col == 5 ? 1.0 / su0 : 0.0;
      else if (gq == 8) v = col == 6 ? -1.0 / su2 : 0.0;
      else if (gq == 9) v = col == 6 ? 1.0 / su2 : 0.0;
      else if (gq == 10 || gq == 11) {      // (c - ru) / su0, (ru - c') / su0
        double dr = col == 5 ? -1.0 / t : (col == 14 ? 1.0 / t : (col == 8 ? -ru / t : 0.0));
        v = (gq == 10 ? -dr : dr) / su0;
      } else if (gq == 12 || gq == 13) {
        double dr = col == 6 ? -1.0 / t : (col == 15 ? 1.0 / t : (col == 8 ? -rd / t : 0.0));
        v = (gq == 12 ? -dr : dr) / su2;
      } else if (gq == 14) v = col == 0 ? -1.0 / P.sw[0] : 0.0;
      else if (gq == 15) v = col == 0 ? 1.0 / P.sw[0] : 0.0;
      else v = col == 8 ? -1.0 : 0.0;       // g 16 = -t
    }
    Jout[e] = v * P.sw[col % kMtNv];
  }
}

// grid (blocks of 3 nodes, B, 1 [MID] or 2 [ENDS]): one pair of directions of one dynamics evaluation per thread.
// Lane = (node, pair): the pairs of a node leave next to each other and its unknowns arrive as broadcasts.
//   Only 21 of the 36 pairs go through the dual arithmetic: the lateral offset n and the heading xi enter the dynamics
// through  s' = v cos(xi + beta) / (1 - n kappa),  n' = v sin(xi + beta),  xi' = omega - kappa s'  only
// (double_track.py:122-129: the force and moment balance knows nothing of where the car is on the road), so the
// weighted sum  phi = m . f  splits into  phi_kin(n, xi, beta, v) + m2 omega + phi_dyn(omega, beta, v, F, delta, gamma):
// the pairs among the six variables of phi_dyn are differentiated as before (phi_kin's share of (beta, v) rides along),
// the 7 non-zero entries with n or xi are closed forms of  A v cos(theta) D + m1 v sin(theta),
// A = m0 - kappa m2, theta = xi + beta, D = 1 / (1 - n kappa),  and the other 8 are zero (written: the work array is not cleared).
constexpr int kMtHesNodes = 3, kMtDynVars = 6, kMtDynPairs = 21;
__host__ __device__ constexpr int mt_hes_blocks(int N) { return (N + kMtHesNodes - 1) / kMtHesNodes; }
template <int ENDS>
__global__ void __launch_bounds__(64, 2) k_mt_hes_point(MtProblem P, MtState st) {
  const int nl = (int)threadIdx.x / kMtDynPairs, q = (int)threadIdx.x - kMtDynPairs * nl;
  const int j = blockIdx.x * kMtHesNodes + nl, b = blockIdx.y, N = P.N;
  const int pt = ENDS ? (int)blockIdx.z : 0;
  if (nl >= kMtHesNodes || j >= N || st.scal[(size_t)b * 16 + 5] != 0.0) return;
  const int jn = j + 1 == N ? 0 : j + 1;
  const size_t o = (size_t)b * N + j;
  double Y[5], Yn[5], U[3], t;
  mt_phys(P, st.w + o * kMtNv, st.w + ((size_t)b * N + jn) * kMtNv, Y, Yn, U, t);
  double* hw = st.hw + o * kMtHw;
  int a = 0, bb = q;   // q-th pair a <= b of the six variables 2 .. 7
  while (bb >= kMtDynVars - a) { bb -= kMtDynVars - a; ++a; }
  bb += a;
  a += 2; bb += 2;
  using D2 = Dual<1, Dual<1>>;
  D2 Yd[5], Ud[3], f[6];
  auto seed = [&](D2& r, double val, int var) {
    r.v.v = val; r.v.d[0] = var == a ? 1.0 : 0.0; r.d[0].v = var == bb ? 1.0 : 0.0; r.d[0].d[0] = 0.0;
  };
  double Yp[5];   // the point of this evaluation
#pragma unroll
  for (int c = 0; c < 5; ++c) { Yp[c] = ENDS ? (pt ? Yn[c] : Y[c]) : hw[kMtHwXm + c]; seed(Yd[c], Yp[c], c); }
#pragma unroll
  for (int c = 0; c < 3; ++c) seed(Ud[c], U[c], 5 + c);
  const double* y = st.y + o * kMtNe;
  DtTyres<D2> ty;
  mt_dyn<D2>(P, j, Yd, Ud, f, ty);
  double m[6];
  m[0] = y[0] * P.se[0] * (ENDS ? t / 6.0 : 1.0);
#pragma unroll
  for (int c = 1; c < 6; ++c) {
    m[c] = y[c] * P.se[c];
    if (ENDS) m[c] = m[c] * (t / 6.0) + (pt ? -1.0 : 1.0) * (t * t / 12.0) * hw[kMtHwGm + c - 1];
  }
  D2 L = f[0] * m[0];
#pragma unroll
  for (int c = 1; c < 6; ++c) L = L + f[c] * m[c];
  double* H = hw + (ENDS ? (pt ? kMtHwH2 : kMtHwH1) : kMtHwHm);
  if (ENDS && pt == 0) L = L + mt_own_rows<D2>(P, Yd, Ud, ty, y, st.z + o * kMtNi);
  H[mt_tri8(a, bb)] = L.d[0].d[0];
  if (!ENDS && a == bb) hw[kMtHwGm + a] = L.v.d[0];
  // the entries with n (row 0: lanes 0 .. 7, column = lane) or xi (row 1: lanes 8 .. 14, column = lane - 7)
  if (q < 15) {
    const int r = q < 8 ? 0 : 1, c = q < 8 ? q : q - 7;
    const double kap = P.kappa[j], A = m[0] - kap * m[2];
    const double v = Yp[4], D = 1.0 / (1.0 - Yp[0] * kap);
    double sn, cs;
    m_sincos(Yp[1] + Yp[3], sn, cs);
    const double dth = -A * v * sn * D + m[1] * v * cs;          // d / d theta
    const double dn = A * v * cs * kap * D * D;                  // d / d n
    double e = 0.0;
    if (r == 0) {
      if (c == 0) e = 2.0 * dn * kap * D;
      else if (c == 1 || c == 3) e = -A * v * sn * kap * D * D;
      else if (c == 4) e = A * cs * kap * D * D;
    } else {
      if (c == 1 || c == 3) e = -A * v * cs * D - m[1] * v * sn;
      else if (c == 4) e = -A * sn * D + m[1] * cs;
    }
    H[mt_tri8(r, c)] = e;
    if (!ENDS && q == 0) hw[kMtHwGm + 0] = dn;
    if (!ENDS && q == 8) hw[kMtHwGm + 1] = dth;
  }
}

// grid (N, B), one wave per node: the 18 x 18 Hessian of the scaled unknowns from the pieces
__global__ void __launch_bounds__(64) k_mt_hes_assemble(MtProblem P, MtState st) {
  __shared__ double M[kMtPv][14], T[kMtPv][14], Hm[kMtPv][kMtPv], g[kMtPv], gpsi[14], gphi[2][kMtPv], gF[2][kMtPv];
  const int j = blockIdx.x, b = blockIdx.y, lane = threadIdx.x, N = P.N;
  if (st.scal[(size_t)b * 16 + 5] != 0.0) return;
  const int jn = j + 1 == N ? 0 : j + 1;
  const size_t o = (size_t)b * N + j;
  const double* wo = st.w + o * kMtNv;
  const double* wn = st.w + ((size_t)b * N + jn) * kMtNv;
  const double* hw = st.hw + o * kMtHw;
  const double* y = st.y + o * kMtNe;
  const double* z = st.z + o * kMtNi;
  const double t = wo[8] * P.sw[8];
  const double* J1 = hw + kMtHwJ1;
  const double* J2 = hw + kMtHwJ2;
  // z = (Y 0..4, U 5..7, t 8, Y' 9..13); variables of the second end evaluation: Y' then U
  auto map2 = [](int zi) { return zi >= 9 ? zi - 9 : (zi >= 5 && zi < 8 ? zi : -1); };
  if (lane < kMtPv) g[lane] = hw[kMtHwGm + lane];
  {
    const int a = lane >> 3, c = lane & 7;
    Hm[a][c] = hw[kMtHwHm + mt_tri8(a, c)];
  }
  if (lane < 2 * kMtPv) {   // gradients of yh.f and of gm_Y.F at the two ends
    const int pt = lane >> 3, a = lane & 7;
    const double* J = pt ? J2 : J1;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) s1 += y[c] * P.se[c] * J[c * kMtPv + a];
#pragma unroll
    for (int r = 0; r < 5; ++r) s2 += hw[kMtHwGm + r] * J[(1 + r) * kMtPv + a];
    gphi[pt][a] = s1; gF[pt][a] = s2;
  }
  for (int e = lane; e < kMtPv * 14; e += 64) {   // M = d(Ym, U) / d(Y, U, t, Y')
    const int r = e / 14, zi = e - 14 * r;
    double v = 0.0;
    if (r < 5) {
      if (zi < 5) v = (zi == r ? 0.5 : 0.0) + (t / 8.0) * J1[(1 + r) * kMtPv + zi];
      else if (zi < 8) v = (t / 8.0) * (J1[(1 + r) * kMtPv + zi] - J2[(1 + r) * kMtPv + zi]);
      else if (zi == 8) v = hw[kMtHwDf + r] / 8.0;
      else v = (zi - 9 == r ? 0.5 : 0.0) - (t / 8.0) * J2[(1 + r) * kMtPv + (zi - 9)];
    } else {
      v = zi == r ? 1.0 : 0.0;   // U rows: z indices 5..7
    }
    M[r][zi] = v;
  }
  __syncthreads();
  for (int e = lane; e < kMtPv * 14; e += 64) {   // T = Hm M
    const int r = e / 14, zi = e - 14 * r;
    double v = 0.0;
#pragma unroll
    for (int p = 0; p < kMtPv; ++p) v += Hm[r][p] * M[p][zi];
    T[r][zi] = v;
  }
  if (lane < 14) {
    double v = 0.0;
#pragma unroll
    for (int p = 0; p < kMtPv; ++p) v += M[p][lane] * g[p];
    gpsi[lane] = v;
  }
  __syncthreads();
  const double su0 = P.sw[5], su2 = P.sw[6];
  const double k1 = (z[11] - z[10]) / su0, k2 = (z[13] - z[12]) / su2;   // rate rows: k (u' - u) / t
  const double dF = (wn[5] - wo[5]) * P.sw[5], dD = (wn[6] - wo[6]) * P.sw[6];
  double* H = st.hes + o * kMtLoc * kMtLoc;
  for (int e = lane; e < kMtLoc * kMtLoc; e += 64) {
    const int la = e / kMtLoc, lb = e - kMtLoc * la;
    double v = 0.0;
    if (la < 14 && lb < 14) {
      double s = 0.0;
#pragma unroll
      for (int p = 0; p < kMtPv; ++p) s += M[p][la] * T[p][lb];
      v = (2.0 * t / 3.0) * s;
      if (la < 8 && lb < 8) v += hw[kMtHwH1 + mt_tri8(la, lb)];
      const int a2 = map2(la), b2 = map2(lb);
      if (a2 >= 0 && b2 >= 0) v += hw[kMtHwH2 + mt_tri8(a2, b2)];
      if ((la == 8) != (lb == 8)) {   // products with t
        const int oth = la == 8 ? lb : la, o2 = map2(oth);
        v += (2.0 / 3.0) * gpsi[oth];
        if (oth < 8) v += gphi[0][oth] / 6.0 + (t / 12.0) * gF[0][oth];
        if (o2 >= 0) v += gphi[1][o2] / 6.0 - (t / 12.0) * gF[1][o2];
      } else if (la == 8 && lb == 8) {
        v += (4.0 / 3.0) * gpsi[8];
      }
    }
    // rate rows k1 (F' - F) / t + k2 (delta' - delta) / t: F 5, delta 6, t 8, F' 14, delta' 15
    const int lo = la < lb ? la : lb, hi = la < lb ? lb : la;
    if (lo == 8 && hi == 8) v += 2.0 * (k1 * dF + k2 * dD) / (t * t * t);
    else if (lo == 5 && hi == 8) v += k1 / (t * t);
    else if (lo == 6 && hi == 8) v += k2 / (t * t);
    else if (lo == 8 && hi == 14) v -= k1 / (t * t);
    else if (lo == 8 && hi == 15) v -= k2 / (t * t);
    H[e] = v * P.sw[la % kMtNv] * P.sw[lb % kMtNv];
  }
}

// ------------------------------------------------------------------------------------------------
// 16 x 16 blocks in REGISTERS, one wave.  Lane l = (i = l & 15, q = l >> 4) holds the four entries
// (i, 4 r + q), r = 0..3, of a block ("A-layout") -- exactly what v_mfma_f64_16x16x4_f64 wants from BOTH
// operands of a product X Y' (the r-th instruction consumes columns 4 r .. 4 r + 3), and its result,
// rows 4 r + q of column i, is the A-layout of (X Y')' = Y X'.  Every product of the elimination has the form
// Y X' (S^-1 symmetric), so the blocks never leave the registers and never change layout.
// Vectors: "i-layout" (lane holds v[i], the same in the four q groups) or "k-layout" (lane holds v[4 r + q]).
__device__ __forceinline__ void mt_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct MtBlk { double v[4]; };

__device__ __forceinline__ double mt_bcast(double v, int l) {  // l wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// A-layout of Y X'
__device__ __forceinline__ MtBlk mt_mul_t(const MtBlk& X, const MtBlk& Y) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X.v[kk], Y.v[kk], acc, 0, 0, 0);
  MtBlk C;
#pragma unroll
  for (int r = 0; r < 4; ++r) C.v[r] = acc[r];
  return C;
}

// M v: M in A-layout, v in k-layout -> i-layout
__device__ __forceinline__ double mt_gemv(const MtBlk& M, const double (&vk)[4]) {
  double p = M.v[0] * vk[0];
#pragma unroll
  for (int r = 1; r < 4; ++r) p = fma(M.v[r], vk[r], p);
  p += __shfl_xor(p, 16, 64);
  p += __shfl_xor(p, 32, 64);
  return p;
}

__device__ __forceinline__ void mt_to_k(double vi, int q, double (&vk)[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) vk[r] = __shfl(vi, 4 * r + q, 64);
}

// In-place inverse of a symmetric 16 x 16 block by Gauss-Jordan elimination WITHOUT pivoting, in registers: per
// pivot k one v_readlane pair (the pivot), five ds_bpermute pairs (column k for the lane's row, row k for the
// lane's four columns) and four lane-local updates.  Returns the number of negative pivots (they are the LDL'
// pivots of the block: their signs, summed over the whole elimination, are the inertia of the KKT matrix), or
// -1 when a pivot is negligible / not finite.  Measured and dropped:
//   * (round 3) issuing the moves of pivot k + 1 before the update of pivot k and advancing the moved values by the
//     update formula, which shortens the dependent chain to 1/p_k -> p_{k+1}: 11 % slower;
//   * (round 4) the cross-lane moves without LDS -- row k by DPP row_newbcast (two v_mov_b32_dpp per value), column k by
//     v_permlane32_swap + v_permlane16_swap (row Q of the wave into all four rows): bit-identical results, one NLP (the latency-bound
//     case the change was meant for) 0.1026 s against 0.1020 s: no gain.  The ten ds_bpermute of a pivot are LDS
//     instructions; their replacement is ~28 VALU instructions (moves that seed the DPP destinations, the swaps' operand
//     copies, wait states), so the VALU stream of a pivot grows by a third and the chain -- set by 1/p and its two Newton
//     steps -- does not shrink.
__device__ __forceinline__ int mt_invert(MtBlk& S, int lane) {
  const int i = lane & 15, q = lane >> 4;
  int neg = 0;
  bool bad = false;
#pragma unroll
  for (int k = 0; k < kMtNb; ++k) {
    const int kr = k >> 2, kq = k & 3;  // column k: register kr of the lanes with q == kq
    const double piv = mt_bcast(S.v[kr], 16 * kq + k);
    if (piv < 0.0) ++neg;
    if (!(fabs(piv) >= 1e-11) || !isfinite(piv)) bad = true;
    double ip = __builtin_amdgcn_rcp(piv);
    ip = fma(fma(-piv, ip, 1.0), ip, ip);
    ip = fma(fma(-piv, ip, 1.0), ip, ip);
    const double colk = __shfl(S.v[kr], 16 * kq + i, 64);
    double rowk[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rowk[r] = __shfl(S.v[r], (lane & 48) + k, 64);
    const bool ik = i == k;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double t = rowk[r] * ip;                 // row k of the result; S_ij -= S_ik t_j elsewhere
      double v = fma(-colk, t, S.v[r]);
      if (r == kr) v = q == kq ? -colk * ip : v;     // column k
      v = ik ? t : v;
      if (r == kr) v = (ik && q == kq) ? ip : v;
      S.v[r] = v;
    }
  }
  return bad ? -1 : neg;
}

// cost gradient of node j (scaled unknowns)
__device__ __forceinline__ double mt_cost_grad(const double* w, int N, int j, int a) {
  if (a == 8) return 1.0;
  if (a < 5) return 0.0;
  const int jp = j == 0 ? N - 1 : j - 1, jn = j + 1 == N ? 0 : j + 1;
  const double u = w[(size_t)j * kMtNv + a];
  return 2e-4 * u + 2e-1 * (2.0 * u - w[(size_t)jn * kMtNv + a] - w[(size_t)jp * kMtNv + a]);
}

// max |r_d|, max constraint violation, max s z, max |s z - mu|, lap time [s] of one instance (one wave)
//   r_d[j][a] = gc + Ao(j)' y_j + An(j-1)' y_{j-1} + Go(j)' z_j + Gn(j-1)' z_{j-1}
__device__ __forceinline__ void mt_residuals(const MtProblem& P, const double* w, const double* sv, const double* yv,
                                             const double* zv, const double* fun, const double* jac, double mu, int lane,
                                             double& kkt, double& viol, double& compl_, double& errmu, double& lap) {
  const int N = P.N;
  kkt = 0.0; viol = 0.0; compl_ = 0.0; errmu = 0.0; lap = 0.0;
  for (int idx = lane; idx < N * kMtNv; idx += 64) {
    const int j = idx / kMtNv, a = idx - j * kMtNv, jp = j == 0 ? N - 1 : j - 1;
    double r = mt_cost_grad(w, N, j, a);
    const double* Jo = jac + (size_t)j * kMtNf * kMtLoc;
    const double* Jp = jac + (size_t)jp * kMtNf * kMtLoc;
    for (int c = 0; c < kMtNe; ++c) r += Jo[c * kMtLoc + a] * yv[j * kMtNe + c] + Jp[c * kMtLoc + kMtNv + a] * yv[jp * kMtNe + c];
    for (int c = 0; c < kMtNi; ++c)
      r += Jo[(kMtNe + c) * kMtLoc + a] * zv[j * kMtNi + c] + Jp[(kMtNe + c) * kMtLoc + kMtNv + a] * zv[jp * kMtNi + c];
    kkt = fmax(kkt, fabs(r));
    if (a == 8) lap += w[idx] * P.sw[8];
  }
  for (int idx = lane; idx < N * kMtNe; idx += 64) {
    const int j = idx / kMtNe, c = idx - j * kMtNe;
    viol = fmax(viol, fabs(fun[j * kMtNf + c]));
  }
  for (int idx = lane; idx < N * kMtNi; idx += 64) {
    const int j = idx / kMtNi, c = idx - j * kMtNi;
    const double s_ = sv[idx], z_ = zv[idx];
    viol = fmax(viol, fabs(fun[j * kMtNf + kMtNe + c] + s_));
    compl_ = fmax(compl_, s_ * z_);
    errmu = fmax(errmu, fabs(s_ * z_ - mu));
  }
  kkt = wave_max(kkt); viol = wave_max(viol); compl_ = wave_max(compl_); errmu = wave_max(errmu); lap = wave_sum(lap);
}

// k_mt_prepare: one wave per instance: residuals r_d, r_c, r_g at the current point -> kkt / viol / compl,
// convergence test, barrier update (everything the assembly of the right-hand side needs to know)
__global__ void __launch_bounds__(64) k_mt_prepare(MtProblem P, MtState st) {
  const int b = blockIdx.x, lane = threadIdx.x, N = P.N;
  double* scal = st.scal + (size_t)b * 16;
  if (scal[5] != 0.0) return;
  mt_instance(P, b);
  double mu = scal[0];
  double kkt, viol, compl_, errmu, lap;
  mt_residuals(P, st.w + (size_t)b * N * kMtNv, st.s + (size_t)b * N * kMtNi, st.y + (size_t)b * N * kMtNe,
               st.z + (size_t)b * N * kMtNi, st.fun + (size_t)b * N * kMtNf, st.jac + (size_t)b * N * kMtNf * kMtLoc,
               mu, lane, kkt, viol, compl_, errmu, lap);
  if (lane == 0) { scal[2] = kkt; scal[3] = viol; scal[4] = compl_; scal[11] = lap; }
  if (fmax(kkt, fmax(viol, compl_)) <= st.tol) {
    if (lane == 0) scal[5] = 1.0;
    return;
  }
  // monotone barrier update once the barrier problem is solved to mu_kappa mu; no lower than compl <= tol needs
  if (fmax(fmax(kkt, viol), errmu) <= st.mu_kappa * mu) {
    const double mu_new = fmax(fmin(st.mu_fac * mu, pow(mu, st.mu_pow)), st.tol / 10.0);
    if (mu_new != mu && lane == 0) scal[14] = 0.0;   // a new barrier problem: its objective is another function, the filter starts empty
    mu = mu_new;
  }
  if (lane == 0) scal[0] = mu;
}

// k_mt_assemble: one wave per (node, instance) -- the blocks of the KKT matrix and the right-hand side, all
// nodes in parallel (this is where the Jacobians / Hessians are read; the sequential elimination then streams
// ready-made 16 x 16 blocks).  Per pair j the 9 x 9 pieces (W = z / s)
//   Do(j) = Hess_oo + Go' W Go -> K[j][j]    Dn(j) = Hess_nn + Gn' W Gn -> K[j+1][j+1]    C(j) = Hess_no + Gn' W Go -> K[j+1][j]
//   D_j = M[j][j]   = [[Do(j) + Dn(j-1) + H_cost (+ delta I, added by the elimination), Ao(j)'], [Ao(j), -eps I]]
//   E_j = M[j+1][j] = [[C(j) + H_cost off-diagonal, An(j)'], [0, 0]]
//   r_j = -[ gc_j + Ao(j)' y_j + An(j-1)' y_{j-1} + Go(j)' zeta_j + Gn(j-1)' zeta_{j-1} ;  eq_j ],
//         zeta = mu / s + W (g + s)
struct MtAsmLds { double Do[81], Dn[81], Ct[81], w[2][kMtNi], zeta[2][kMtNi], Go[kMtNi][kMtLoc], Gp[kMtNi][kMtNv]; };

__global__ void __launch_bounds__(64) k_mt_assemble(MtProblem P, MtState st) {
  __shared__ MtAsmLds L;
  const int j = blockIdx.x, b = blockIdx.y, lane = threadIdx.x, N = P.N;
  const double* scal = st.scal + (size_t)b * 16;
  if (scal[5] != 0.0) return;
  const int jp = j == 0 ? N - 1 : j - 1;
  const double* w = st.w + (size_t)b * N * kMtNv;
  const double* sv = st.s + (size_t)b * N * kMtNi;
  const double* yv = st.y + (size_t)b * N * kMtNe;
  const double* zv = st.z + (size_t)b * N * kMtNi;
  const double* fun = st.fun + (size_t)b * N * kMtNf;
  const double* Jo = st.jac + ((size_t)b * N + j) * kMtNf * kMtLoc;
  const double* Jp = st.jac + ((size_t)b * N + jp) * kMtNf * kMtLoc;
  const double* Ho = st.hes + ((size_t)b * N + j) * kMtLoc * kMtLoc;
  const double* Hp = st.hes + ((size_t)b * N + jp) * kMtLoc * kMtLoc;
  const double mu = scal[0];
  if (lane < 2 * kMtNi) {
    const int which = lane / kMtNi, i = lane - which * kMtNi, jj = which ? jp : j;
    const double s_ = sv[jj * kMtNi + i], z_ = zv[jj * kMtNi + i];
    const double wgt = z_ / s_;
    L.w[which][i] = wgt;
    L.zeta[which][i] = mu / s_ + wgt * (fun[jj * kMtNf + kMtNe + i] + s_);
  }
  // the inequality rows of this pair (all 18 columns) and of the previous pair (its "next node" columns) once into
  // LDS, coalesced: the triple products below read every entry 9 to 18 times
  for (int e = lane; e < kMtNi * kMtLoc; e += 64) L.Go[e / kMtLoc][e % kMtLoc] = Jo[kMtNe * kMtLoc + e];
  for (int e = lane; e < kMtNi * kMtNv; e += 64) L.Gp[e / kMtNv][e % kMtNv] = Jp[(kMtNe + e / kMtNv) * kMtLoc + 9 + e % kMtNv];
  __syncthreads();
  for (int e = lane; e < 81; e += 64) {
    const int a = e / 9, c = e - 9 * a;
    double doo = Ho[a * kMtLoc + c], cno = Ho[(9 + a) * kMtLoc + c], dnn = Hp[(9 + a) * kMtLoc + 9 + c];
#pragma unroll
    for (int i = 0; i < kMtNi; ++i) {
      const double goa = L.Go[i][a], goc = L.Go[i][c], gna = L.Go[i][9 + a];
      const double pna = L.Gp[i][a], pnc = L.Gp[i][c];
      doo += goa * L.w[0][i] * goc; cno += gna * L.w[0][i] * goc; dnn += pna * L.w[1][i] * pnc;
    }
    L.Do[e] = doo; L.Ct[e] = cno; L.Dn[e] = dnn;
  }
  __syncthreads();
  double* Dg = st.dblk + ((size_t)b * N + j) * 256;
  double* Eg = st.eblk + ((size_t)b * N + j) * 256;
  for (int e = lane; e < 256; e += 64) {
    const int i = e >> 4, c = e & 15;
    double d = 0.0, ev = 0.0;
    if (i < 9 && c < 9) {
      d = L.Do[i * 9 + c] + L.Dn[i * 9 + c];
      ev = L.Ct[i * 9 + c];
      if (i == c && i >= 5 && i < 8) { d += kMtCostDiag; ev += kMtCostOff; }
    } else if (i < 9) {
      d = Jo[(c - 9) * kMtLoc + i];          // Ao(j)'
      ev = Jo[(c - 9) * kMtLoc + 9 + i];     // An(j)'
    } else if (c < 9) {
      d = Jo[(i - 9) * kMtLoc + c];          // Ao(j)
    } else if (i == c) {
      d = -kMtEpsReg;
    }
    Dg[e] = d; Eg[e] = ev;
  }
  if (lane < 16) {
    double r;
    if (lane < 9) {
      const int a = lane;
      r = mt_cost_grad(w, N, j, a);
      for (int c = 0; c < kMtNe; ++c) r += Jo[c * kMtLoc + a] * yv[j * kMtNe + c] + Jp[c * kMtLoc + 9 + a] * yv[jp * kMtNe + c];
      for (int c = 0; c < kMtNi; ++c) r += L.Go[c][a] * L.zeta[0][c] + L.Gp[c][a] * L.zeta[1][c];
    } else {
      r = fun[j * kMtNf + (lane - 9)];
    }
    st.rhs[((size_t)b * N + j) * 16 + lane] = -r;
  }
}

// k_mt_node: Jacobian, Hessian and KKT blocks of the node pairs in ONE pass (the default; k_mt_jac_assemble,
// k_mt_hes_assemble, k_mt_prepare and k_mt_assemble stay as the cross-check path, RL_MT_UNFUSED=1).
//   The kernels it replaces were bound by memory and by their own index arithmetic: every pair's 24 x 18 Jacobian and
// 18 x 18 Hessian went to HBM and came back twice (for the pair itself and for its successor, which needs the "next
// node" columns): 25 KB per node and iteration; and with one wave per pair and "entry e = lane + 64 k" loops nine of ten
// instructions computed indices or branched.  Here
//   * a group of 16 lanes owns a RUN of kMtRun consecutive pairs and walks it; lane = COLUMN (the pair function does not
//     depend on the next node's gamma and t: columns 16, 17 of every matrix are zero), rows are compile-time loop indices,
//     so every structural decision (which of J1 / J2 / Jm, which rows are closed forms, which entries of the inequality
//     Jacobian can be non-zero) is taken by the compiler or is a lane constant;
//   * the lane keeps its column of M, Hm M, the Jacobian and the Hessian in registers; LDS holds the work array of the
//     pair (fetched one pair ahead into registers) and what other lanes read as broadcasts (M, the Jacobian);
//   * what the NEXT node's diagonal block and right-hand side need of this pair (Dn = Hess_nn + Gn' W Gn, An' y + Gn' zeta)
//     is carried to the next pair in LDS -- the group starts one pair early to have the carry of its first node
//     (1 / kMtRun more work).  Of the Jacobian only the 60 entries of the inequality rows that can be non-zero are written
//     (k_mt_dir reads them; the final residuals re-assemble the full Jacobian), the Hessian never leaves the chip.
//   The barrier parameter is updated from the residuals of ALL nodes (k_mt_prepare2, afterwards), so the right-hand
// side leaves in two pieces: rhs0 (everything but the mu / s term of zeta) and r1 = G' (1 / s); k_mt_prepare2 forms
// rhs = rhs0 - mu r1 with the mu it has just chosen.  r1[9] carries the node's max |r_d| to the convergence test.
constexpr int kMtRun = 9, kMtNodeGroups = 4;
struct MtNodeGrp {
  double hw[kMtHw];
  double wo[16], wn[16], y[8], z[kMtNi + 1], s[kMtNi + 1], fun[kMtNf];
  double M[kMtPv][16];
  double J[kMtNf][16];
  double W[kMtNi + 1], zeta[kMtNi + 1], is[kMtNi + 1];
  double gphi[16], gF[16], gpsi[16], tx[16], rt[16];
  double cDn[kMtNv][16], cvn[3][16], up[4];   // the carry: Dn, (A' y + G' W (g + s), G' (1 / s), A' y + G' z) of the next node's columns, controls
};

// can row i of the inequality Jacobian have an entry in column r?  (rows as in mt_pair_emit: 0..3 tyre ellipses and 4 power
// depend on (Y, U) of the node; 5 speed; 6, 7 force; 8, 9 steering; 10..13 the rates; 14, 15 the road edges; 16 the duration)
__host__ __device__ constexpr bool mt_g_has(int i, int r) {
  return i < 5 ? r < 8 : i == 5 ? r == 4 : i < 8 ? r == 5 : i < 10 ? r == 6 : i < 12 ? (r == 5 || r == 8 || r == 14)
       : i < 14 ? (r == 6 || r == 8 || r == 15) : i < 16 ? r == 0 : r == 8;
}
__host__ __device__ constexpr int mt_map2(int zi) { return zi >= 9 ? zi - 9 : (zi >= 5 && zi < 8 ? zi : -1); }

// Compact inequality Jacobian of a pair: slots 0..39 = rows 0..4 x columns 0..7 (tyre ellipses, power: the node's (Y, U)),
// slots 40..59 = the one to three entries of the other rows (mt_g_has), in row order.
constexpr int kMtGc = 64, kMtGcSparse = 20;
__device__ constexpr unsigned char kMtGcRow[kMtGcSparse] = {5, 6, 7, 8, 9, 10, 10, 10, 11, 11, 11, 12, 12, 12, 13, 13, 13, 14, 15, 16};
__device__ constexpr unsigned char kMtGcCol[kMtGcSparse] = {4, 5, 5, 6, 6, 5, 8, 14, 5, 8, 14, 6, 8, 15, 6, 8, 15, 0, 0, 8};

// cross-check paths (full Jacobian from k_mt_derivs<1> / k_mt_jac_assemble): gather the compact form; grid (N, B)
__global__ void __launch_bounds__(64) k_mt_gc_pack(MtProblem P, MtState st) {
  const int j = blockIdx.x, b = blockIdx.y, l = threadIdx.x, N = P.N;
  if (st.scal[(size_t)b * 16 + 5] != 0.0 || l >= 40 + kMtGcSparse) return;
  const size_t o = (size_t)b * N + j;
  const double* Jg = st.jac + (o * kMtNf + kMtNe) * kMtLoc;
  const int row = l < 40 ? l >> 3 : kMtGcRow[l - 40], col = l < 40 ? l & 7 : kMtGcCol[l - 40];
  st.gc[o * kMtGc + l] = Jg[row * kMtLoc + col];
}

static_assert(kMtHw == 21 * 16, "k_mt_node fetches the work array as 21 rows of 16 lanes");
static_assert(kMtNv == 9 && kMtNe == 7 && kMtNi == 17 && kMtLoc == 18, "k_mt_node's lane = column layout is written for 9 + 7 unknowns per node");
static_assert(kMtHesNodes * kMtDynPairs <= 64 && kMtJacNodes * kMtJacDirs <= 64, "(node, direction) lanes of the derivative kernels fit one wave");
__global__ void __launch_bounds__(64) k_mt_node(MtProblem P, MtState st) {
  __shared__ MtNodeGrp LG[kMtNodeGroups];
  const int b = blockIdx.y, lane = threadIdx.x, N = P.N;
  if (st.scal[(size_t)b * 16 + 5] != 0.0) return;
  const int c = lane & 15;
  MtNodeGrp& L = LG[lane >> 4];
  const int run0 = ((int)blockIdx.x * kMtNodeGroups + (lane >> 4)) * kMtRun;
  const int cnt = run0 < N ? (N - run0 < kMtRun ? N - run0 : kMtRun) : 0;   // a group without pairs walks along without writing
  const int j0 = cnt > 0 ? run0 : 0;
  auto pair_of = [&](int it) { return it < 0 ? (j0 == 0 ? N - 1 : j0 - 1) : (it < cnt ? j0 + it : j0); };
  // lane constants
  const int c1 = c < 8 ? c : 7;                               // the lane's column of a first-end quantity
  const int c2 = c < 8 ? c : (c >= 9 && c < 14 ? c - 9 : 0);  // ... of a second-end quantity: map2(c) where there is one
  const bool has2 = (c >= 5 && c < 8) || (c >= 9 && c < 14);
  const int q1 = 7 * c1 - c1 * (c1 - 1) / 2, q2 = 7 * c2 - c2 * (c2 - 1) / 2;
  auto tri_lane = [](int la, int cx, int qx) { return cx <= la ? qx + la : 7 * la - la * (la - 1) / 2 + cx; };   // mt_tri8(la, cx)
  double swc = P.sw[0];
#pragma unroll
  for (int k = 1; k < kMtNv; ++k) swc = c % kMtNv == k ? P.sw[k] : swc;
  const double su0 = P.sw[5], su2 = P.sw[6];
  // the inputs of a pair: 336 doubles of work array, 9 + 9 + 7 + 17 + 17 + 24 of unknowns / multipliers / slacks / functions
  double ph[21], ps[9];
  auto fetch = [&](int j) {
    const size_t o = (size_t)b * N + j, on = (size_t)b * N + (j + 1 == N ? 0 : j + 1);
    const double* hw = st.hw + o * kMtHw;
#pragma unroll
    for (int k = 0; k < 21; ++k) ph[k] = hw[c + 16 * k];
    ps[0] = c < kMtNv ? st.w[o * kMtNv + c] : 0.0;
    ps[1] = c < kMtNv ? st.w[on * kMtNv + c] : 0.0;
    ps[2] = c < kMtNe ? st.y[o * kMtNe + c] : 0.0;
    ps[3] = st.z[o * kMtNi + c]; ps[4] = st.z[o * kMtNi + 16];
    ps[5] = st.s[o * kMtNi + c]; ps[6] = st.s[o * kMtNi + 16];
    ps[7] = st.fun[o * kMtNf + c];
    ps[8] = c < 8 ? st.fun[o * kMtNf + 16 + c] : 0.0;
  };
  auto deposit = [&]() {
#pragma unroll
    for (int k = 0; k < 21; ++k) L.hw[c + 16 * k] = ph[k];
    L.wo[c] = ps[0]; L.wn[c] = ps[1];
    if (c < 8) { L.y[c] = ps[2]; L.fun[16 + c] = ps[8]; }
    L.z[c] = ps[3]; L.s[c] = ps[5]; L.fun[c] = ps[7];
    if (c == 0) { L.z[16] = ps[4]; L.s[16] = ps[6]; }
  };
#pragma unroll
  for (int k = 0; k < kMtNv; ++k) L.cDn[k][c] = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) L.cvn[k][c] = 0.0;
  fetch(pair_of(-1));
  for (int it = -1; it < kMtRun; ++it) {
    const size_t o = (size_t)b * N + pair_of(it);
    const bool own = it >= 0 && it < cnt;
    deposit();
    __syncthreads();
    if (it + 1 < kMtRun) fetch(pair_of(it + 1));
    const double* hw = L.hw;
    const double* J1 = hw + kMtHwJ1;
    const double* J2 = hw + kMtHwJ2;
    const double* Jm = hw + kMtHwJm;
    const double t = L.wo[8] * P.sw[8], t8 = t / 8.0;
    // ---- A: the lane's column of M = d(Ym, U) / d(Y, U, t, Y'); gradients at the ends; weights of the inequality rows
    double Mc[kMtPv];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const double j1 = J1[(1 + r) * kMtPv + c1], j2 = J2[(1 + r) * kMtPv + c2];
      double v = 0.0;
      if (c < 5) v = (c == r ? 0.5 : 0.0) + t8 * j1;
      else if (c < 8) v = t8 * (j1 - j2);
      else if (c == 8) v = hw[kMtHwDf + r] / 8.0;
      else if (c < 14) v = (c - 9 == r ? 0.5 : 0.0) - t8 * j2;
      Mc[r] = v;
    }
#pragma unroll
    for (int k = 5; k < kMtPv; ++k) Mc[k] = c == k ? 1.0 : 0.0;
#pragma unroll
    for (int p = 0; p < kMtPv; ++p) L.M[p][c] = Mc[p];
    {
      const double* J = c < 8 ? J1 : J2;
      const int a = c & 7;
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) s1 += L.y[cc] * P.se[cc] * J[cc * kMtPv + a];
#pragma unroll
      for (int r = 0; r < 5; ++r) s2 += hw[kMtHwGm + r] * J[(1 + r) * kMtPv + a];
      L.gphi[c] = s1; L.gF[c] = s2;
    }
    {
      const double s_ = L.s[c], z_ = L.z[c], wgt = z_ / s_;
      L.W[c] = wgt; L.zeta[c] = wgt * (L.fun[kMtNe + c] + s_); L.is[c] = 1.0 / s_;
      if (c == 0) {
        const double s6 = L.s[16], z6 = L.z[16], w6 = z6 / s6;
        L.W[16] = w6; L.zeta[16] = w6 * (L.fun[kMtNe + 16] + s6); L.is[16] = 1.0 / s6;
      }
    }
    __syncthreads();
    // ---- B: the lane's columns of Jm M and Hm M, M' gm, the products with t
    double JMc[6], Tc[kMtPv];
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) {
      double v = 0.0;
#pragma unroll
      for (int p = 0; p < kMtPv; ++p) v += Jm[cc * kMtPv + p] * Mc[p];
      JMc[cc] = v;
    }
#pragma unroll
    for (int r = 0; r < kMtPv; ++r) {
      double v = 0.0;
#pragma unroll
      for (int p = 0; p < kMtPv; ++p) v += hw[kMtHwHm + mt_tri8(r, p)] * Mc[p];
      Tc[r] = v;
    }
    double gpsi_c = 0.0;
#pragma unroll
    for (int p = 0; p < kMtPv; ++p) gpsi_c += Mc[p] * hw[kMtHwGm + p];
    double tx_c = 0.0, rt_c = 0.0;
    {
      if (c != 8 && c < 14) {
        tx_c = (2.0 / 3.0) * gpsi_c;
        if (c < 8) tx_c += L.gphi[c] / 6.0 + (t / 12.0) * L.gF[c];
        if (has2) tx_c += L.gphi[8 + c2] / 6.0 - (t / 12.0) * L.gF[8 + c2];
      }
      const double k1 = (L.z[11] - L.z[10]) / su0, k2 = (L.z[13] - L.z[12]) / su2;   // rate rows: k (u' - u) / t
      const double dF = (L.wn[5] - L.wo[5]) * P.sw[5], dD = (L.wn[6] - L.wo[6]) * P.sw[6];
      rt_c = c == 5 ? k1 / (t * t) : c == 6 ? k2 / (t * t) : c == 8 ? 2.0 * (k1 * dF + k2 * dD) / (t * t * t)
           : c == 14 ? -k1 / (t * t) : c == 15 ? -k2 / (t * t) : 0.0;
      L.gpsi[c] = gpsi_c; L.tx[c] = tx_c; L.rt[c] = rt_c;
    }
    // ---- C1: the lane's column of the 24 x 18 Jacobian (formulas of k_mt_jac_assemble)
    double Jc[kMtNf];
    {
      const double t6 = t / 6.0;
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) {
        const double j1 = J1[cc * kMtPv + c1], j2 = J2[cc * kMtPv + c2];
        double dv = 4.0 * JMc[cc];
        if (c < 5) dv += j1;
        else if (c < 8) dv += j1 + j2;
        else if (c >= 9) dv += j2;
        double v = t6 * dv;
        if (c == 8) v += (hw[kMtHwF + cc] + 4.0 * hw[kMtHwF + 12 + cc] + hw[kMtHwF + 6 + cc]) / 6.0;
        if (cc >= 1 && c == cc - 1) v += 1.0;
        if (cc >= 1 && c == cc + 8) v -= 1.0;
        v *= P.se[cc];
        Jc[cc] = c < 14 ? v * swc : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) Jc[6 + k] = c < kMtPv ? hw[kMtHwOw + k * kMtPv + c1] * swc : 0.0;
      const double ru = (L.wn[5] - L.wo[5]) * P.sw[5] / t, rd = (L.wn[6] - L.wo[6]) * P.sw[6] / t;
      const double dr1 = c == 5 ? -1.0 / t : (c == 14 ? 1.0 / t : (c == 8 ? -ru / t : 0.0));
      const double dr2 = c == 6 ? -1.0 / t : (c == 15 ? 1.0 / t : (c == 8 ? -rd / t : 0.0));
      Jc[12] = (c == 4 ? -1.0 / P.sw[4] : 0.0) * swc;
      Jc[13] = (c == 5 ? -1.0 / su0 : 0.0) * swc;
      Jc[14] = (c == 5 ? 1.0 / su0 : 0.0) * swc;
      Jc[15] = (c == 6 ? -1.0 / su2 : 0.0) * swc;
      Jc[16] = (c == 6 ? 1.0 / su2 : 0.0) * swc;
      Jc[17] = (-dr1 / su0) * swc;
      Jc[18] = (dr1 / su0) * swc;
      Jc[19] = (-dr2 / su2) * swc;
      Jc[20] = (dr2 / su2) * swc;
      Jc[21] = (c == 0 ? -1.0 / P.sw[0] : 0.0) * swc;
      Jc[22] = (c == 0 ? 1.0 / P.sw[0] : 0.0) * swc;
      Jc[23] = (c == 8 ? -1.0 : 0.0) * swc;
      double* Gc = st.gc + o * kMtGc;
#pragma unroll
      for (int row = 0; row < kMtNf; ++row) L.J[row][c] = Jc[row];
#pragma unroll
      for (int i = 0; i < 5; ++i)
        if (own && c < 8) Gc[i * 8 + c] = Jc[kMtNe + i];
    }
    __syncthreads();
    if (own) {   // the sparse rows of the inequality Jacobian, 16 + 4 entries
      double* Gc = st.gc + o * kMtGc;
      Gc[40 + c] = L.J[kMtNe + kMtGcRow[c]][kMtGcCol[c]];
      if (c < kMtGcSparse - 16) Gc[56 + c] = L.J[kMtNe + kMtGcRow[16 + c]][kMtGcCol[16 + c]];
    }
    // ---- C2: the lane's column of the 18 x 18 Hessian (formulas of k_mt_hes_assemble); rows 16, 17 are zero
    double Hc[16];
    {
      const double gpsi8 = L.gpsi[8];
#pragma unroll
      for (int la = 0; la < 16; ++la) {
        double v = 0.0;
        if (la < 14) {
          double sm = 0.0;
#pragma unroll
          for (int p = 0; p < 5; ++p) sm += L.M[p][la] * Tc[p];
          if (la >= 5 && la < 8) sm += Tc[la];            // rows 5..7 of M are unit vectors
          v = (2.0 * t / 3.0) * sm;
          if (la < 8) { const double h = hw[kMtHwH1 + tri_lane(la, c1, q1)]; v += c < 8 ? h : 0.0; }
          if (mt_map2(la) >= 0) { const double h = hw[kMtHwH2 + tri_lane(mt_map2(la), c2, q2)]; v += has2 ? h : 0.0; }
          if (la == 8) v += c == 8 ? (4.0 / 3.0) * gpsi8 : tx_c;
          else { const double x = L.tx[la]; v += c == 8 ? x : 0.0; }
          v = c < 14 ? v : 0.0;
        }
        if (la == 8) v += rt_c;
        else if (la == 5 || la == 6 || la == 14 || la == 15) { const double x = L.rt[la]; v += c == 8 ? x : 0.0; }
        Hc[la] = v * P.sw[la % kMtNv] * swc;
      }
    }
    // ---- D: Hess + G' W G (Do: rows, columns < 9; Ct: rows >= 9, columns < 9; Dn: both >= 9) and the vectors of the right-hand side
    double v0, v1, v2;
    {
      double WG[kMtNi];
#pragma unroll
      for (int i = 0; i < kMtNi; ++i) WG[i] = L.W[i] * Jc[kMtNe + i];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        double v = Hc[r];
#pragma unroll
        for (int i = 0; i < kMtNi; ++i)
          if (mt_g_has(i, r)) v += L.J[kMtNe + i][r] * WG[i];
        Hc[r] = v;
      }
      double ay = 0.0, gz0 = 0.0, g1 = 0.0, gz = 0.0;
#pragma unroll
      for (int row = 0; row < kMtNe; ++row) ay += Jc[row] * L.y[row];
#pragma unroll
      for (int i = 0; i < kMtNi; ++i) {
        const double g = Jc[kMtNe + i];
        gz0 += g * L.zeta[i]; g1 += g * L.is[i]; gz += g * L.z[i];
      }
      v0 = ay + gz0; v1 = g1; v2 = ay + gz;
    }
    // ---- E: the node's blocks and right-hand side (its own pair + the carry of the previous one)
    {
      const int ct = c >= 9 ? c - 9 : 0, cl = c < 9 ? c : 0;
      double* Dg = st.dblk + o * 256;
      double* Eg = st.eblk + o * 256;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        double d, ev = 0.0;
        if (i < kMtNv) {
          double dd = Hc[i] + L.cDn[i][cl];
          double ee = 9 + i < 16 ? Hc[9 + i] : 0.0;
          if (i >= 5 && i < 8) { dd += c == i ? kMtCostDiag : 0.0; ee += c == i ? kMtCostOff : 0.0; }
          const double jt = L.J[ct][i];                                 // Ao(j)'
          const double jt2 = 9 + i < 16 ? L.J[ct][9 + i] : 0.0;         // An(j)'
          d = c < 9 ? dd : jt; ev = c < 9 ? ee : jt2;
        } else {
          d = c < 9 ? Jc[i - 9] : (c == i ? -kMtEpsReg : 0.0);          // Ao(j)
        }
        if (own) { Dg[i * 16 + c] = d; Eg[i * 16 + c] = ev; }
      }
      double r0, r1 = 0.0, rdv = 0.0;
      if (c < kMtNv) {
        double gc = c == 8 ? 1.0 : 0.0;
        if (c >= 5 && c < 8) gc = 2e-4 * L.wo[c] + 2e-1 * (2.0 * L.wo[c] - L.wn[c] - L.up[c - 5]);
        r0 = gc + v0 + L.cvn[0][c];
        r1 = v1 + L.cvn[1][c];
        rdv = fabs(gc + v2 + L.cvn[2][c]);
      } else {
        r0 = L.fun[c - kMtNv];
      }
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) rdv = fmax(rdv, __shfl_xor(rdv, m, 16));
      if (own) {
        st.rhs[o * 16 + c] = -r0;
        st.r1[o * 16 + c] = c == kMtNv ? rdv : r1;
      }
    }
    __syncthreads();
    // the carry for the next pair: lanes 9 .. 15 hold the next node's columns (its gamma and t do not occur)
    if (c >= 9) {
#pragma unroll
      for (int a = 0; a < 7; ++a) L.cDn[a][c - 9] = Hc[9 + a];
      L.cvn[0][c - 9] = v0; L.cvn[1][c - 9] = v1; L.cvn[2][c - 9] = v2;
    } else if (c < 3) {
      L.up[c] = L.wo[5 + c];
    }
    __syncthreads();
  }
}

template <int BLOCK>
__device__ __forceinline__ double mt_block_reduce(double v, double* red, int op /*0 sum, 1 min*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = op == 0 ? wave_sum(v) : wave_min(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double r = red[0];
  for (int q = 1; q < BLOCK / 64; ++q) r = op == 0 ? r + red[q] : fmin(r, red[q]);
  __syncthreads();
  return r;
}

// NV values at once (op[v]: 0 sum, 1 min): one wave reduction each, ONE round trip through LDS and two barriers for all of
// them (the per-instance kernels are latency bound: k_mt_step spent its time in 4 + 2 kMtTrials single-value reductions of
// three barriers each).  The wave partials are combined in wave order, as mt_block_reduce does: same sums bit for bit.
template <int BLOCK, int NV>
__device__ __forceinline__ void mt_block_reduce_n(double (&v)[NV], const int (&op)[NV], double* red /* [NV * BLOCK / 64] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = op[k] == 0 ? wave_sum(v[k]) : wave_min(v[k]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) red[k * (BLOCK / 64) + wave] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double r = red[k * (BLOCK / 64)];
    for (int q = 1; q < BLOCK / 64; ++q) r = op[k] == 0 ? r + red[k * (BLOCK / 64) + q] : fmin(r, red[k * (BLOCK / 64) + q]);
    v[k] = r;
  }
}

// k_mt_prepare2 (one workgroup per instance, after k_mt_node): convergence test and barrier update as k_mt_prepare (the stationarity
// residual comes per node from k_mt_node), then the right-hand side rhs = rhs0 - mu r1.  (Measured: as a prologue of the elimination
// kernel instead of a kernel of its own it is 1.5 % faster for 1024 instances and 13 % slower for 256: a kernel of its own.)
template <int BLOCK>
__device__ __forceinline__ bool mt_prepare2(const MtProblem& P, const MtState& st, int b, int tid, double* red) {
  const int N = P.N;
  double* scal = st.scal + (size_t)b * 16;
  const double* w = st.w + (size_t)b * N * kMtNv;
  const double* sv = st.s + (size_t)b * N * kMtNi;
  const double* zv = st.z + (size_t)b * N * kMtNi;
  const double* fun = st.fun + (size_t)b * N * kMtNf;
  const double* r1 = st.r1 + (size_t)b * N * 16;
  double* rhs = st.rhs + (size_t)b * N * 16;
  double mu = scal[0];
  double kkt = 0.0, viol = 0.0, compl_ = 0.0, errmu = 0.0, lap = 0.0;
  for (int j = tid; j < N; j += BLOCK) {
    kkt = fmax(kkt, r1[(size_t)j * 16 + kMtNv]);
    lap += w[(size_t)j * kMtNv + 8] * P.sw[8];
  }
  for (int idx = tid; idx < N * kMtNe; idx += BLOCK) {
    const int j = idx / kMtNe, c = idx - j * kMtNe;
    viol = fmax(viol, fabs(fun[j * kMtNf + c]));
  }
  for (int idx = tid; idx < N * kMtNi; idx += BLOCK) {
    const int j = idx / kMtNi, c = idx - j * kMtNi;
    const double s_ = sv[idx], z_ = zv[idx];
    viol = fmax(viol, fabs(fun[j * kMtNf + kMtNe + c] + s_));
    compl_ = fmax(compl_, s_ * z_);
    errmu = fmax(errmu, fabs(s_ * z_ - mu));
  }
  {
    double v[5] = {-kkt, -viol, -compl_, -errmu, lap};
    const int op[5] = {1, 1, 1, 1, 0};
    mt_block_reduce_n<BLOCK, 5>(v, op, red);
    kkt = -v[0]; viol = -v[1]; compl_ = -v[2]; errmu = -v[3]; lap = v[4];
  }
  if (tid == 0) { scal[2] = kkt; scal[3] = viol; scal[4] = compl_; scal[11] = lap; }
  if (fmax(kkt, fmax(viol, compl_)) <= st.tol) {
    if (tid == 0) scal[5] = 1.0;
    return true;
  }
  if (fmax(fmax(kkt, viol), errmu) <= st.mu_kappa * mu) {
    const double mu_new = fmax(fmin(st.mu_fac * mu, pow(mu, st.mu_pow)), st.tol / 10.0);
    if (mu_new != mu && tid == 0) scal[14] = 0.0;
    mu = mu_new;
  }
  if (tid == 0) scal[0] = mu;
  for (int idx = tid; idx < N * kMtNv; idx += BLOCK) {
    const int j = idx / kMtNv, a = idx - j * kMtNv;
    rhs[(size_t)j * 16 + a] = fma(-mu, r1[(size_t)j * 16 + a], rhs[(size_t)j * 16 + a]);
  }
  __syncthreads();
  return false;
}


__global__ void __launch_bounds__(256) k_mt_prepare2(MtProblem P, MtState st) {
  __shared__ double red[5 * 4];
  if (st.scal[(size_t)blockIdx.x * 16 + 5] != 0.0) return;
  (void)mt_prepare2<256>(P, st, blockIdx.x, threadIdx.x, red);
}

// k_mt_kkt: two waves per instance: block elimination of the cyclic block-tridiagonal KKT system from the
// assembled blocks (retry with a larger delta until the inertia is right), solution dw, dy.
//
// The last node N-1 is the BORDER (its couplings to node 0 and to node N-2 close the lap); the chain 0 .. N-2 is
// eliminated from BOTH ENDS at once: front A (wave 0) takes 0, 1, ..., m-1, front B (wave 1) takes N-2, N-3, ...,
// m+1, each carrying its fill F towards the border, and they meet at node m = (N-1)/2, which receives the Schur
// complements of both.  The dependent chain is (N-1)/2 block steps instead of N-1; the back substitution runs from
// m outwards on both sides.  Both fronts are the same code: front B sees the chain reversed, so its "E" (coupling of
// the node to the next one of the front) is E_{j-1}' and its first border block is M[N-1][N-2] = E_{N-2}.
// All blocks stay in registers (MtBlk); the blocks of the next node are fetched while the current node is
// eliminated.  Kept for the back substitution, per node: P_j = E S_j^-1, Q_j = F_j S_j^-1 (row major), a_j = S_j^-1 r_j.
struct MtKktShare {                   // what front B hands to the meeting node, flags, and the start of the way back
  double T[256], Fn[256], Sl[256];    // blocks in the lanes' own order (lane * 4 + r)
  double rn[16], rl[16], xl[16], xm[16];
  double xs[2][2][16];
  int bad[2], nneg[2], ok;
};

__global__ void __launch_bounds__(128) k_mt_kkt(MtProblem P, MtState st) {
  __shared__ MtKktShare X;
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, N = P.N;
  const int i = lane & 15, q = lane >> 4;
  double* scal = st.scal + (size_t)b * 16;
  if (scal[5] != 0.0) return;
  const double* Dg = st.dblk + (size_t)b * N * 256;
  const double* Eg = st.eblk + (size_t)b * N * 256;
  const double* Rg = st.rhs + (size_t)b * N * 16;
  double* blk = st.blk + (size_t)b * N * 3 * 256;   // per node: P [256], Q [256], a' [16]
  double* vec = st.vec + (size_t)b * N * 16;
  double delta = scal[1];
  auto ld = [&](const double* G) { MtBlk Z;      // row-major block -> A-layout
#pragma unroll
    for (int r = 0; r < 4; ++r) Z.v[r] = G[i * 16 + 4 * r + q];
    return Z; };
  auto ld_t = [&](const double* G) { MtBlk Z;    // its transpose (a symmetric block: the coalesced way to read it)
#pragma unroll
    for (int r = 0; r < 4; ++r) Z.v[r] = G[(4 * r + q) * 16 + i];
    return Z; };
  auto st_rm = [&](double* G, const MtBlk& Z) {
#pragma unroll
    for (int r = 0; r < 4; ++r) G[i * 16 + 4 * r + q] = Z.v[r]; };
  // delta I on the unknowns' part of a diagonal block
  auto shift = [&](MtBlk& Z) {
#pragma unroll
    for (int r = 0; r < 4; ++r) Z.v[r] += (i == 4 * r + q && i < kMtNv) ? delta : 0.0; };
  const MtBlk zero = {{0.0, 0.0, 0.0, 0.0}};
  const int m = (N - 1) / 2;                         // meeting node (N >= 8: neither next to the border nor an end)
  const int cnt = wave == 0 ? m : N - 2 - m;         // nodes of this front
  auto node = [&](int s_) { return wave == 0 ? s_ : N - 2 - s_; };
  // coupling of node j to the NEXT node of its front, in the role of E_j of the ascending recursion
  auto ld_e = [&](int j) { return wave == 0 ? ld(Eg + (size_t)j * 256) : ld_t(Eg + (size_t)(j - 1) * 256); };

  int attempt = 0, fails = 0;
  bool ok = false;
  for (; attempt < 12 && !ok; ++attempt) {
    bool bad = false;
    int n_neg = 0;
    MtBlk S = ld_t(Dg + (size_t)node(0) * 256); shift(S);
    // border block of the front's first node: M[N-1][0] = E_{N-1}'  /  M[N-1][N-2] = E_{N-2}
    MtBlk F = wave == 0 ? ld_t(Eg + (size_t)(N - 1) * 256) : ld(Eg + (size_t)(N - 2) * 256);
    MtBlk SlC = zero, Tx = zero, Fx = zero;   // contributions to S_last; what the meeting node receives
    double rlC = 0.0, rx = 0.0;
    double rk[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rk[r] = Rg[(size_t)node(0) * 16 + 4 * r + q];
    MtBlk eN = ld_e(node(0)), dN = cnt > 1 ? ld_t(Dg + (size_t)node(1) * 256) : zero;   // blocks of the next step, in flight
    double rN = cnt > 1 ? Rg[(size_t)node(1) * 16 + i] : 0.0;
    for (int s_ = 0; s_ < cnt && !bad; ++s_) {
      const int j = node(s_);
      const MtBlk E = eN, dC = dN;
      const double rC = rN;
      if (s_ + 1 < cnt) {                // prefetch for the next step
        eN = ld_e(node(s_ + 1));
        dN = s_ + 2 < cnt ? ld_t(Dg + (size_t)node(s_ + 2) * 256) : zero;
        rN = s_ + 2 < cnt ? Rg[(size_t)node(s_ + 2) * 16 + i] : 0.0;
      }
      const int neg = mt_invert(S, lane);                   // S_j^-1
      if (neg < 0) { bad = true; break; }
      n_neg += neg;
      const double aj = mt_gemv(S, rk);                     // a_j = S_j^-1 r_j
      if (q == 0) vec[(size_t)j * 16 + i] = aj;
      const MtBlk Pm = mt_mul_t(S, E);                      // P_j = E S_j^-1
      const MtBlk Qm = mt_mul_t(S, F);                      // Q_j = F_j S_j^-1
      double* Bj = blk + (size_t)j * 3 * 256;
      st_rm(Bj, Pm); st_rm(Bj + 256, Qm);
      rlC -= mt_gemv(Qm, rk);                               // r_last -= Q_j r_j
      {
        const MtBlk U = mt_mul_t(F, Qm);                    // S_last -= Q_j F_j'
#pragma unroll
        for (int r = 0; r < 4; ++r) SlC.v[r] -= U.v[r];
      }
      // next node of the front: S += -P E' ; F = -Q E' ; r += -P r_j   (the last step hands them to the meeting node)
      const MtBlk M = mt_mul_t(E, Pm), Fn = mt_mul_t(E, Qm);
      const double rn = -mt_gemv(Pm, rk);
      if (s_ + 1 < cnt) {
        S = dC; shift(S);
#pragma unroll
        for (int r = 0; r < 4; ++r) { S.v[r] -= M.v[r]; F.v[r] = -Fn.v[r]; }
        mt_to_k(rC + rn, q, rk);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) { Tx.v[r] = -M.v[r]; Fx.v[r] = -Fn.v[r]; }
        rx = rn;
      }
    }
    if (wave == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { X.T[lane * 4 + r] = Tx.v[r]; X.Fn[lane * 4 + r] = Fx.v[r]; X.Sl[lane * 4 + r] = SlC.v[r]; }
      if (q == 0) { X.rn[i] = rx; X.rl[i] = rlC; }
    }
    if (lane == 0) { X.bad[wave] = bad ? 1 : 0; X.nneg[wave] = n_neg; }
    __syncthreads();
    const bool fronts_ok = X.bad[0] == 0 && X.bad[1] == 0;
    if (fronts_ok && wave == 0) {
      // ---- meeting node m, then the border
      bool okm = true;
      MtBlk Sm = ld_t(Dg + (size_t)m * 256); shift(Sm);
      MtBlk Fm, Sl = ld_t(Dg + (size_t)(N - 1) * 256);
      shift(Sl);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Sm.v[r] += Tx.v[r] + X.T[lane * 4 + r];
        Fm.v[r] = Fx.v[r] + X.Fn[lane * 4 + r];
        Sl.v[r] += SlC.v[r] + X.Sl[lane * 4 + r];
      }
      double rmk[4];
      mt_to_k(Rg[(size_t)m * 16 + i] + rx + X.rn[i], q, rmk);
      double rl = Rg[(size_t)(N - 1) * 16 + i] + rlC + X.rl[i];
      int total = X.nneg[0] + X.nneg[1];
      const int negm = mt_invert(Sm, lane);
      if (negm < 0) okm = false;
      total += negm;
      const double am = mt_gemv(Sm, rmk);
      if (q == 0) vec[(size_t)m * 16 + i] = am;
      const MtBlk Qm = mt_mul_t(Sm, Fm);
      double* Bm = blk + (size_t)m * 3 * 256;
      st_rm(Bm, zero); st_rm(Bm + 256, Qm);
      rl -= mt_gemv(Qm, rmk);
      {
        const MtBlk U = mt_mul_t(Fm, Qm);
#pragma unroll
        for (int r = 0; r < 4; ++r) Sl.v[r] -= U.v[r];
      }
      const int negl = mt_invert(Sl, lane);
      // inertia (9N, 7N, 0): the reduced Hessian is positive definite (Sylvester's law on the LDL' pivots)
      if (negl < 0 || total + negl != N * kMtNe) okm = false;
      if (okm) {
        double rlk[4];
        mt_to_k(rl, q, rlk);
        const double xl = mt_gemv(Sl, rlk);     // x_last = S_last^-1 r_last
        if (q == 0) X.xl[i] = xl;
      }
      if (lane == 0) X.ok = okm ? 1 : 0;
    }
    __syncthreads();
    if (!fronts_ok || X.ok == 0) {
      delta = fmax(10.0 * delta, 1e-4);
      ++fails;
      __syncthreads();                   // everybody has read the flags before the next attempt rewrites them
      if (delta > 1e8) break;
      continue;
    }
    ok = true;
  }
  if (threadIdx.x == 0) { scal[1] = delta; scal[10] += (double)fails; }   // refactorisations = failed attempts (also on the give-up path)
  if (!ok) {
    if (threadIdx.x == 0) scal[5] = 2.0;
    return;
  }
  // ---- back substitution: x_j = a_j - Q_j' x_last - P_j' x_(neighbour towards m), from m outwards on both sides
  double* dw = st.dw + (size_t)b * N * kMtNv;
  double* dy = st.dy + (size_t)b * N * kMtNe;
  if (wave == 0 && lane < 16) {
    const double xl = X.xl[lane];
    if (lane < kMtNv) dw[(size_t)(N - 1) * kMtNv + lane] = xl;
    else dy[(size_t)(N - 1) * kMtNe + lane - kMtNv] = xl;
  }
  // a'_j = a_j - Q_j' x_last for the nodes of this wave's side (wave 0: 0 .. m, wave 1: m+1 .. N-2), four nodes at
  // a time (lane = node q of the four, component i); written to the node's third slot, which nothing has read before
  {
    double xl[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) xl[k] = X.xl[k];
    const int j_lo = wave == 0 ? 0 : m + 1, j_hi = wave == 0 ? m + 1 : N - 1;
    for (int j0 = j_lo; j0 < j_hi; j0 += 4) {
      const int j = j0 + q;
      if (j < j_hi) {
        const double* Qg = blk + (size_t)j * 3 * 256 + 256;
        double acc = vec[(size_t)j * 16 + i];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fma(-Qg[k * 16 + i], xl[k], acc);
        blk[(size_t)j * 3 * 256 + 512 + i] = acc;
      }
    }
  }
  mt_wave_sync();
  // x_m = a'_m (P_m = 0) starts both chains: wave 0 walks m-1, ..., 0, wave 1 walks m+1, ..., N-2.  Lanes 0..15 of
  // each wave own one component each; P and a' of the node THREE steps ahead are fetched while x_j is formed (four
  // rotating register sets: a step is shorter than the latency of a load).
  if (wave == 0 && lane < 16) {
    const double x = blk[(size_t)m * 3 * 256 + 512 + lane];
    if (lane < kMtNv) dw[(size_t)m * kMtNv + lane] = x;
    else dy[(size_t)m * kMtNe + lane - kMtNv] = x;
    X.xm[lane] = x;
  }
  double (*xs)[16] = X.xs[wave];
  const int steps = wave == 0 ? m : N - 2 - m;
  auto chain_node = [&](int s_) { return wave == 0 ? m - 1 - s_ : m + 1 + s_; };
  double pb[4][17];
  auto fetch = [&](double (&dst)[17], int s_) {
    if (s_ < steps && lane < 16) {
      const double* Bn = blk + (size_t)chain_node(s_) * 3 * 256;
#pragma unroll
      for (int k = 0; k < 16; ++k) dst[k] = Bn[k * 16 + lane];
      dst[16] = Bn[512 + lane];
    }
  };
  fetch(pb[0], 0); fetch(pb[1], 1); fetch(pb[2], 2);
  __syncthreads();
  if (lane < 16) xs[0][lane] = X.xm[lane];
  mt_wave_sync();
  int cur = 0;
  for (int s0 = 0; s0 < steps; s0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s_ = s0 + u;
      if (s_ >= steps) break;
      fetch(pb[(u + 3) & 3], s_ + 3);
      if (lane < 16) {
        double x0 = pb[u][16], x1 = 0.0;
#pragma unroll
        for (int k = 0; k < 16; k += 2) { x0 = fma(-pb[u][k], xs[cur][k], x0); x1 = fma(-pb[u][k + 1], xs[cur][k + 1], x1); }
        const double x = x0 + x1;
        const int j = chain_node(s_);
        if (lane < kMtNv) dw[(size_t)j * kMtNv + lane] = x;
        else dy[(size_t)j * kMtNe + lane - kMtNv] = x;
        xs[cur ^ 1][lane] = x;
      }
      mt_wave_sync();
      cur ^= 1;
    }
  }
}

// k_mt_kkt4: the same elimination with FOUR fronts (four waves per instance).  The chain 0 .. N-2 is cut at the middle node m
// (as in k_mt_kkt) and at q1 = m / 2, q3 = m + (N - 1 - m) / 2:
//     front 0: 0 .. q1-1 ascending,   fill towards the border node L = N-1      front 2: m-1 .. q1+1 descending, fill towards m
//     front 3: m+1 .. q3-1 ascending, fill towards m                            front 1: N-2 .. q3+1 descending, fill towards L
// Interior nodes are coupled to L only through the two ends of the chain, so a front that starts next to m carries its fill
// towards m INSTEAD of towards L: every front does exactly the work of a front of k_mt_kkt, on half as many nodes.  The fronts
// meet pairwise at q1 and q3; what is left is a 4-node system (q1, q3 | m | L): wave 0 eliminates q1 and wave 1 q3 at the same
// time, wave 0 then m and L.  Back substitution: x_L, x_m, x_q1, x_q3, then the four chains away from q1 / q3.  The dependent
// chain is (N-1)/4 + 3 block steps instead of (N-1)/2 + 2.  Same factorisation up to the elimination order (inertia test as before).
struct MtKkt4Share {
  double T[4][256], Fn[4][256], Sl[4][256];       // per front: Schur complement onto its meeting node, fill meeting node -> target, sum for its target's diagonal
  double rn[4][16], rl[4][16];
  double dSm[2][256], dSL[2][256], dC[2][256];    // from the elimination of q1 (0) and q3 (1): to S_m, S_L and the block M[L][m]
  double drm[2][16], drL[2][16];
  double xL[16], xm[16], xq[2][16];
  double xs[4][2][16];
  int bad[4], nneg[4], bad2[2], nneg2[2], ok;
};

__global__ void __launch_bounds__(256, 2) k_mt_kkt4(MtProblem P, MtState st) {
  __shared__ MtKkt4Share X;
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, N = P.N;
  const int i = lane & 15, q = lane >> 4;
  double* scal = st.scal + (size_t)b * 16;
  if (scal[5] != 0.0) return;
  const double* Dg = st.dblk + (size_t)b * N * 256;
  const double* Eg = st.eblk + (size_t)b * N * 256;
  const double* Rg = st.rhs + (size_t)b * N * 16;
  double* blk = st.blk + (size_t)b * N * 3 * 256;   // per node: P [256], Q [256], a' [16]
  double* vec = st.vec + (size_t)b * N * 16;
  double delta = scal[1];
  auto ld = [&](const double* G) { MtBlk Z;
#pragma unroll
    for (int r = 0; r < 4; ++r) Z.v[r] = G[i * 16 + 4 * r + q];
    return Z; };
  auto ld_t = [&](const double* G) { MtBlk Z;
#pragma unroll
    for (int r = 0; r < 4; ++r) Z.v[r] = G[(4 * r + q) * 16 + i];
    return Z; };
  auto st_rm = [&](double* G, const MtBlk& Z) {
#pragma unroll
    for (int r = 0; r < 4; ++r) G[i * 16 + 4 * r + q] = Z.v[r]; };
  auto shift = [&](MtBlk& Z) {
#pragma unroll
    for (int r = 0; r < 4; ++r) Z.v[r] += (i == 4 * r + q && i < kMtNv) ? delta : 0.0; };
  auto lds_get = [&](const double* A) { MtBlk Z;
#pragma unroll
    for (int r = 0; r < 4; ++r) Z.v[r] = A[lane * 4 + r];
    return Z; };
  auto lds_put = [&](double* A, const MtBlk& Z) {
#pragma unroll
    for (int r = 0; r < 4; ++r) A[lane * 4 + r] = Z.v[r]; };
  const MtBlk zero = {{0.0, 0.0, 0.0, 0.0}};
  const int L_ = N - 1, m = (N - 1) / 2, q1 = m / 2, q3 = m + (N - 1 - m) / 2;
  const bool asc = wave == 0 || wave == 3;
  const int first = wave == 0 ? 0 : (wave == 1 ? N - 2 : (wave == 2 ? m - 1 : m + 1));
  const int cnt = wave == 0 ? q1 : (wave == 1 ? N - 2 - q3 : (wave == 2 ? m - 1 - q1 : q3 - m - 1));   // all >= 1 for N >= 16
  const int meet = (wave == 0 || wave == 2) ? q1 : q3;
  auto node = [&](int s_) { return asc ? first + s_ : first - s_; };
  auto ld_e = [&](int j) { return asc ? ld(Eg + (size_t)j * 256) : ld_t(Eg + (size_t)(j - 1) * 256); };   // coupling to the next node of the front

  int attempt = 0, fails = 0;
  bool ok = false;
  for (; attempt < 12 && !ok; ++attempt) {
    bool bad = false;
    int n_neg = 0;
    MtBlk S = ld_t(Dg + (size_t)first * 256); shift(S);
    // the block M[target][first node]: E_{N-1}' , E_{N-2} (border), E_{m-1}, E_m' (middle)
    MtBlk F = wave == 0 ? ld_t(Eg + (size_t)(N - 1) * 256) : (wave == 1 ? ld(Eg + (size_t)(N - 2) * 256)
            : (wave == 2 ? ld(Eg + (size_t)(m - 1) * 256) : ld_t(Eg + (size_t)m * 256)));
    MtBlk SlC = zero, Tx = zero, Fx = zero;
    double rlC = 0.0, rx = 0.0;
    double rk[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rk[r] = Rg[(size_t)first * 16 + 4 * r + q];
    MtBlk eN = ld_e(first), dN = cnt > 1 ? ld_t(Dg + (size_t)node(1) * 256) : zero;
    double rN = cnt > 1 ? Rg[(size_t)node(1) * 16 + i] : 0.0;
    for (int s_ = 0; s_ < cnt && !bad; ++s_) {
      const int j = node(s_);
      const MtBlk E = eN, dC = dN;
      const double rC = rN;
      if (s_ + 1 < cnt) {
        eN = ld_e(node(s_ + 1));
        dN = s_ + 2 < cnt ? ld_t(Dg + (size_t)node(s_ + 2) * 256) : zero;
        rN = s_ + 2 < cnt ? Rg[(size_t)node(s_ + 2) * 16 + i] : 0.0;
      }
      const int neg = mt_invert(S, lane);
      if (neg < 0) { bad = true; break; }
      n_neg += neg;
      const double aj = mt_gemv(S, rk);
      if (q == 0) vec[(size_t)j * 16 + i] = aj;
      const MtBlk Pm = mt_mul_t(S, E);
      const MtBlk Qm = mt_mul_t(S, F);
      double* Bj = blk + (size_t)j * 3 * 256;
      st_rm(Bj, Pm); st_rm(Bj + 256, Qm);
      rlC -= mt_gemv(Qm, rk);
      {
        const MtBlk U = mt_mul_t(F, Qm);
#pragma unroll
        for (int r = 0; r < 4; ++r) SlC.v[r] -= U.v[r];
      }
      const MtBlk M = mt_mul_t(E, Pm), Fn = mt_mul_t(E, Qm);
      const double rn = -mt_gemv(Pm, rk);
      if (s_ + 1 < cnt) {
        S = dC; shift(S);
#pragma unroll
        for (int r = 0; r < 4; ++r) { S.v[r] -= M.v[r]; F.v[r] = -Fn.v[r]; }
        mt_to_k(rC + rn, q, rk);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) { Tx.v[r] = -M.v[r]; Fx.v[r] = -Fn.v[r]; }
        rx = rn;
      }
    }
    lds_put(X.T[wave], Tx); lds_put(X.Fn[wave], Fx); lds_put(X.Sl[wave], SlC);
    if (q == 0) { X.rn[wave][i] = rx; X.rl[wave][i] = rlC; }
    if (lane == 0) { X.bad[wave] = bad ? 1 : 0; X.nneg[wave] = n_neg; }
    __syncthreads();
    const bool fronts_ok = X.bad[0] == 0 && X.bad[1] == 0 && X.bad[2] == 0 && X.bad[3] == 0;
    // ---- the meeting nodes: wave 0 eliminates q1 (fronts 0 and 2), wave 1 q3 (fronts 1 and 3); two targets each: L and m
    if (fronts_ok && wave < 2) {
      const int other = wave + 2;
      const MtBlk To = lds_get(X.T[other]), CL = Fx, Cm = lds_get(X.Fn[other]);   // M[L][meet] from the own front, M[m][meet] from the other
      MtBlk Sq = ld_t(Dg + (size_t)meet * 256); shift(Sq);
#pragma unroll
      for (int r = 0; r < 4; ++r) Sq.v[r] += Tx.v[r] + To.v[r];
      double rqk[4];
      mt_to_k(Rg[(size_t)meet * 16 + i] + rx + X.rn[other][i], q, rqk);
      const int negq = mt_invert(Sq, lane);
      const double aq = mt_gemv(Sq, rqk);
      if (q == 0) vec[(size_t)meet * 16 + i] = aq;
      const MtBlk Qm = mt_mul_t(Sq, Cm), QL = mt_mul_t(Sq, CL);
      double* Bq = blk + (size_t)meet * 3 * 256;
      st_rm(Bq, Qm); st_rm(Bq + 256, QL);
      MtBlk dSm = mt_mul_t(Cm, Qm), dSL = mt_mul_t(CL, QL), dCc = mt_mul_t(Cm, QL);   // Qm Cm', QL CL', QL Cm'
#pragma unroll
      for (int r = 0; r < 4; ++r) { dSm.v[r] = -dSm.v[r]; dSL.v[r] = -dSL.v[r]; dCc.v[r] = -dCc.v[r]; }
      lds_put(X.dSm[wave], dSm); lds_put(X.dSL[wave], dSL); lds_put(X.dC[wave], dCc);
      const double drm = -mt_gemv(Qm, rqk), drL = -mt_gemv(QL, rqk);
      if (q == 0) { X.drm[wave][i] = drm; X.drL[wave][i] = drL; }
      if (lane == 0) { X.bad2[wave] = negq < 0 ? 1 : 0; X.nneg2[wave] = negq < 0 ? 0 : negq; }
    }
    __syncthreads();
    if (fronts_ok && wave == 0) {
      // ---- m, then the border
      bool okm = X.bad2[0] == 0 && X.bad2[1] == 0;
      MtBlk Sm = ld_t(Dg + (size_t)m * 256); shift(Sm);
      MtBlk Sl = ld_t(Dg + (size_t)L_ * 256); shift(Sl);
      MtBlk C;
      {
        const MtBlk a2 = lds_get(X.Sl[2]), a3 = lds_get(X.Sl[3]), d0 = lds_get(X.dSm[0]), d1 = lds_get(X.dSm[1]);
        const MtBlk l1 = lds_get(X.Sl[1]), e0 = lds_get(X.dSL[0]), e1 = lds_get(X.dSL[1]);
        const MtBlk c0 = lds_get(X.dC[0]), c1 = lds_get(X.dC[1]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Sm.v[r] += a2.v[r] + a3.v[r] + d0.v[r] + d1.v[r];
          Sl.v[r] += SlC.v[r] + l1.v[r] + e0.v[r] + e1.v[r];
          C.v[r] = c0.v[r] + c1.v[r];
        }
      }
      double rmk[4];
      mt_to_k(Rg[(size_t)m * 16 + i] + X.rl[2][i] + X.rl[3][i] + X.drm[0][i] + X.drm[1][i], q, rmk);
      double rl = Rg[(size_t)L_ * 16 + i] + rlC + X.rl[1][i] + X.drL[0][i] + X.drL[1][i];
      int total = X.nneg[0] + X.nneg[1] + X.nneg[2] + X.nneg[3] + X.nneg2[0] + X.nneg2[1];
      const int negm = mt_invert(Sm, lane);
      if (negm < 0) okm = false;
      total += negm;
      const double am = mt_gemv(Sm, rmk);
      if (q == 0) vec[(size_t)m * 16 + i] = am;
      const MtBlk QLm = mt_mul_t(Sm, C);                // M[L][m] S_m^-1
      double* Bm = blk + (size_t)m * 3 * 256;
      st_rm(Bm, zero); st_rm(Bm + 256, QLm);
      rl -= mt_gemv(QLm, rmk);
      {
        const MtBlk U = mt_mul_t(C, QLm);
#pragma unroll
        for (int r = 0; r < 4; ++r) Sl.v[r] -= U.v[r];
      }
      const int negl = mt_invert(Sl, lane);
      if (negl < 0 || total + negl != N * kMtNe) okm = false;
      if (okm) {
        double rlk[4];
        mt_to_k(rl, q, rlk);
        const double xl = mt_gemv(Sl, rlk);
        if (q == 0) X.xL[i] = xl;
      }
      if (lane == 0) X.ok = okm ? 1 : 0;
    }
    __syncthreads();
    if (!fronts_ok || X.ok == 0) {
      delta = fmax(10.0 * delta, 1e-4);
      ++fails;
      __syncthreads();
      if (delta > 1e8) break;
      continue;
    }
    ok = true;
  }
  if (threadIdx.x == 0) { scal[1] = delta; scal[10] += (double)fails; }   // refactorisations = failed attempts (also on the give-up path)
  if (!ok) {
    if (threadIdx.x == 0) scal[5] = 2.0;
    return;
  }
  // ---- back substitution.  Wave 0: x_L, x_m = a_m - Q_m' x_L, x_q = a_q - Qm_q' x_m - QL_q' x_L for q1 and q3
  double* dw = st.dw + (size_t)b * N * kMtNv;
  double* dy = st.dy + (size_t)b * N * kMtNe;
  auto put_x = [&](int j, double x) {   // lanes 0..15
    if (lane < kMtNv) dw[(size_t)j * kMtNv + lane] = x;
    else dy[(size_t)j * kMtNe + lane - kMtNv] = x;
  };
  if (wave == 0 && lane < 16) {
    put_x(L_, X.xL[lane]);
    const double* Qg = blk + (size_t)m * 3 * 256 + 256;
    double xm = vec[(size_t)m * 16 + lane];
#pragma unroll
    for (int k = 0; k < 16; ++k) xm = fma(-Qg[k * 16 + lane], X.xL[k], xm);
    X.xm[lane] = xm;
    put_x(m, xm);
  }
  mt_wave_sync();
  if (wave == 0 && lane < 32) {   // lanes 0..15: q1, lanes 16..31: q3
    const int which = lane >> 4, c = lane & 15, jq = which ? q3 : q1;
    const double* Pg = blk + (size_t)jq * 3 * 256;   // Qm (towards m), then QL
    double x = vec[(size_t)jq * 16 + c];
#pragma unroll
    for (int k = 0; k < 16; ++k) x = fma(-Pg[k * 16 + c], X.xm[k], fma(-Pg[256 + k * 16 + c], X.xL[k], x));
    X.xq[which][c] = x;
    if (c < kMtNv) dw[(size_t)jq * kMtNv + c] = x;
    else dy[(size_t)jq * kMtNe + c - kMtNv] = x;
  }
  __syncthreads();
  // a'_j = a_j - Q_j' x_target for the nodes of this wave's front, four nodes at a time
  {
    double xt[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) xt[k] = wave < 2 ? X.xL[k] : X.xm[k];
    const int j_lo = asc ? first : first - (cnt - 1), j_hi = j_lo + cnt;
    for (int j0 = j_lo; j0 < j_hi; j0 += 4) {
      const int j = j0 + q;
      if (j < j_hi) {
        const double* Qg = blk + (size_t)j * 3 * 256 + 256;
        double acc = vec[(size_t)j * 16 + i];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fma(-Qg[k * 16 + i], xt[k], acc);
        blk[(size_t)j * 3 * 256 + 512 + i] = acc;
      }
    }
  }
  mt_wave_sync();
  // the chains run from the meeting node back to the front's first node: x_j = a'_j - P_j' x_(next node of the front)
  double (*xs)[16] = X.xs[wave];
  const int steps = cnt;
  auto chain_node = [&](int s_) { return node(cnt - 1 - s_); };
  double pb[4][17];
  auto fetch = [&](double (&dst)[17], int s_) {
    if (s_ < steps && lane < 16) {
      const double* Bn = blk + (size_t)chain_node(s_) * 3 * 256;
#pragma unroll
      for (int k = 0; k < 16; ++k) dst[k] = Bn[k * 16 + lane];
      dst[16] = Bn[512 + lane];
    }
  };
  fetch(pb[0], 0); fetch(pb[1], 1); fetch(pb[2], 2);
  if (lane < 16) xs[0][lane] = X.xq[(wave == 0 || wave == 2) ? 0 : 1][lane];
  mt_wave_sync();
  int cur = 0;
  for (int s0 = 0; s0 < steps; s0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s_ = s0 + u;
      if (s_ >= steps) break;
      fetch(pb[(u + 3) & 3], s_ + 3);
      if (lane < 16) {
        double x0 = pb[u][16], x1 = 0.0;
#pragma unroll
        for (int k = 0; k < 16; k += 2) { x0 = fma(-pb[u][k], xs[cur][k], x0); x1 = fma(-pb[u][k + 1], xs[cur][k + 1], x1); }
        const double x = x0 + x1;
        put_x(chain_node(s_), x);
        xs[cur ^ 1][lane] = x;
      }
      mt_wave_sync();
      cur ^= 1;
    }
  }
}

// residuals only (final report of the instances that ran into the iteration limit)
__global__ void __launch_bounds__(64) k_mt_residuals(MtProblem P, MtState st) {
  const int b = blockIdx.x, lane = threadIdx.x, N = P.N;
  double* scal = st.scal + (size_t)b * 16;
  if (scal[5] == 1.0) return;
  mt_instance(P, b);
  double kkt, viol, compl_, errmu, lap;
  mt_residuals(P, st.w + (size_t)b * N * kMtNv, st.s + (size_t)b * N * kMtNi, st.y + (size_t)b * N * kMtNe,
               st.z + (size_t)b * N * kMtNi, st.fun + (size_t)b * N * kMtNf, st.jac + (size_t)b * N * kMtNf * kMtLoc,
               scal[0], lane, kkt, viol, compl_, errmu, lap);
  if (lane == 0) { scal[2] = kkt; scal[3] = viol; scal[4] = compl_; scal[11] = lap; }
}

// ------------------------------------------------------------------------------------------------
// the step: helpers

// The step in three kernels (it was one workgroup per instance that evaluated the trial points itself: 395 registers, one
// workgroup per CU and nothing else resident beside it):
//   k_mt_dir    per inequality ROW, all instances: ds, dz, the wave's share of the step lengths and of (theta0, phi0)
//   k_mt_trial  per NODE, all instances: the trial points w + a dw, a = ap, ap / 2, ap / 4 -> the node's shares of (theta, phi)
//   k_mt_step   per instance: ap, ad, (theta0, phi0); the first acceptable of the three trial points (72 / 18 / 7 % of the steps of
//               the benchmark batch), further halvings evaluated in the kernel where all three were rejected (3 %); the update
// (Per-instance kernels wait for a slot behind the other sub-batches' node-parallel grids: the fewer of them in the chain of an
//  iteration the better -- k_mt_step_red, k_mt_step_fin and k_mt_step_back were separate until they showed as 14 % of the streams' time.)
struct MtStepPtrs {
  double *w, *sv, *yv, *zv, *ds, *dz, *scal, *filt;
  const double *dw, *dy;
};
__device__ __forceinline__ MtStepPtrs mt_step_ptrs(const MtState& st, int b, int N) {
  MtStepPtrs q;
  q.w = st.w + (size_t)b * N * kMtNv; q.sv = st.s + (size_t)b * N * kMtNi; q.yv = st.y + (size_t)b * N * kMtNe;
  q.zv = st.z + (size_t)b * N * kMtNi; q.dw = st.dw + (size_t)b * N * kMtNv; q.dy = st.dy + (size_t)b * N * kMtNe;
  q.ds = st.blk + (size_t)b * N * 3 * 256;   // [N,17] ds, then [N,17] dz (the factor blocks are dead now)
  q.dz = q.ds + (size_t)N * kMtNi;
  q.scal = st.scal + (size_t)b * 16; q.filt = st.filt + (size_t)b * 2 * kMtFilter;
  return q;
}

#ifdef RL_MT_HIST
__device__ unsigned long long g_mt_hist[16];   // diagnostic build: accepted steps by number of halvings (12 = none accepted)
#define RL_MT_COUNT(k) do { if (threadIdx.x == 0) atomicAdd(&g_mt_hist[k], 1ull); } while (0)
#else
#define RL_MT_COUNT(k) do { } while (0)
#endif
// acceptance of a trial point (theta, phi) against the current point and the filter
__device__ __forceinline__ bool mt_accept(const MtState& st, const double* filt, int nfilt, int N, double theta, double phi,
                                          double theta0, double phi0) {
  const bool fin = isfinite(theta) && isfinite(phi);
  // a step may not more than double the l1 infeasibility, whatever it does to the objective -- above a floor of
  // 1e-5 per row: next to a feasible point (theta0 ~ 0) the second-order infeasibility of any useful step is larger
  // than twice nothing, and without the floor such instances crept along with two halvings per iteration
  const double floor_ = kMtThetaFloor * (double)(N * kMtNf);
  // eth, eph: what rounding alone moves the two sums by (next to the solution the improvement asked for is smaller)
  const double eth = 1e-13 * (double)(N * kMtNf), eph = 1e-13 * fabs(phi0);
  bool acc = fin && theta <= fmax(kMtThetaGrowth * theta0, floor_) + 1e-9 &&
             (theta <= (1.0 - 1e-5) * theta0 + eth || phi <= phi0 - 1e-5 * theta0 + eph);
  // ... and once the iterate is nearly feasible (l1 infeasibility below th_filter per row) the step must be acceptable
  // to the earlier iterates of this barrier problem as well: otherwise two points, one of less infeasibility and
  // one of less objective, can be visited in turn for ever (seen on a few of 1024 instances).  Far from feasibility
  // the iteration legitimately trades one for the other, and a filter there would need a restoration phase.
  for (int e = 0; e < nfilt && acc && theta0 < st.th_filter * (double)(N * kMtNf); ++e)
    acc = theta <= (1.0 - 1e-5) * filt[2 * e] + eth || phi <= filt[2 * e + 1] - 1e-5 * filt[2 * e] + eph;
  return acc;
}

// the accepted step of length a (after `halvings` halvings of ap): iterate, damping, filter, bookkeeping
__device__ __forceinline__ void mt_take_step(const MtState& st, const MtStepPtrs& q, int N, int tid, double a, int halvings,
                                             double ap, double ad, double theta0, double phi0) {
  const double mu = q.scal[0];
  const int nfilt_total = (int)q.scal[14];
  const double az = st.dual_cap > 0.0 ? fmin(fmin(ad, 1.0), fmax(st.dual_cap * a, 1e-3)) : fmin(ad, 1.0);
  for (int idx = tid; idx < N * kMtNv; idx += 256) q.w[idx] += a * q.dw[idx];
  for (int idx = tid; idx < N * kMtNe; idx += 256) q.yv[idx] += a * q.dy[idx];
  for (int idx = tid; idx < N * kMtNi; idx += 256) {
    const double s_ = q.sv[idx] + a * q.ds[idx];
    double z_ = q.zv[idx] + az * q.dz[idx];
    z_ = fmin(fmax(z_, mu / (1e10 * s_)), 1e10 * mu / s_);
    q.sv[idx] = s_; q.zv[idx] = z_;
  }
  double delta = q.scal[1];
  delta = fmin(fmax(delta * (a > st.a_hi ? st.d_down : (a > st.a_lo ? 1.0 : st.d_up)), 1e-6), 1e3);
  __syncthreads();   // everybody has read scal
  if (tid == 0) {
    q.scal[1] = delta; q.scal[7] = a; q.scal[6] += 1.0; q.scal[8] = theta0; q.scal[9] = phi0; q.scal[12] = ap; q.scal[13] = (double)halvings;
    const int slot = nfilt_total % kMtFilter;   // the point just left joins the filter
    q.filt[2 * slot] = theta0; q.filt[2 * slot + 1] = phi0;
    q.scal[14] = (double)(nfilt_total + 1);
  }
}

// grid (blocks of 64 rows, B): one inequality row per thread: ds = -r_g - G dw, dz = -(s z - mu + z ds) / s, and the wave's
// share of (ap, ad, theta0, phi0) -> part[b][block][4]  (part = the start of the instance's derivative work array hw, dead after k_mt_node)
constexpr int kMtDirBlocks(int N) { return (N * kMtNi + 63) / 64; }
__global__ void __launch_bounds__(64) k_mt_dir(MtProblem P, MtState st) {
  const int b = blockIdx.y, N = P.N, idx = blockIdx.x * 64 + threadIdx.x;
  const MtStepPtrs q = mt_step_ptrs(st, b, N);
  if (q.scal[5] != 0.0) return;
  double ap = 1.0, ad = 1.0, th = 0.0, ph = 0.0;
  if (idx < N * kMtNi) {
    const double* fun = st.fun + (size_t)b * N * kMtNf;
    const double mu = q.scal[0];
    const int j = idx / kMtNi, c = idx - j * kMtNi, jn = j + 1 == N ? 0 : j + 1;
    // G dw over the entries the row can have (the compact form of k_mt_node / k_mt_gc_pack)
    const double* G = st.gc + ((size_t)b * N + j) * kMtGc;
    const double* dwo = q.dw + (size_t)j * kMtNv;
    const double* dwn = q.dw + (size_t)jn * kMtNv;
    double gd = 0.0;
    if (c < 5) {
#pragma unroll
      for (int a = 0; a < 8; ++a) gd += G[c * 8 + a] * dwo[a];
    } else if (c == 5) gd = G[40] * dwo[4];
    else if (c < 8) gd = G[35 + c] * dwo[5];
    else if (c < 10) gd = G[35 + c] * dwo[6];
    else if (c < 12) { const double* g3 = G + 45 + 3 * (c - 10); gd = g3[0] * dwo[5] + g3[1] * dwo[8] + g3[2] * dwn[5]; }
    else if (c < 14) { const double* g3 = G + 51 + 3 * (c - 12); gd = g3[0] * dwo[6] + g3[1] * dwo[8] + g3[2] * dwn[6]; }
    else if (c < 16) gd = G[43 + c] * dwo[0];
    else gd = G[59] * dwo[8];
    const double s_ = q.sv[idx], z_ = q.zv[idx], rg = fun[j * kMtNf + kMtNe + c] + s_;
    const double d_s = -rg - gd;
    const double d_z = -(s_ * z_ - mu + z_ * d_s) / s_;
    q.ds[idx] = d_s; q.dz[idx] = d_z;
    if (d_s < 0.0) ap = -0.995 * s_ / d_s;
    if (d_z < 0.0) ad = -0.995 * z_ / d_z;
    ap = fmin(ap, 1.0); ad = fmin(ad, 1.0);
    th = fabs(rg);
    ph = -mu * log(s_);
  }
  ap = wave_min(ap); ad = wave_min(ad); th = wave_sum(th); ph = wave_sum(ph);
  if (threadIdx.x == 0) {
    double* o = st.hw + (size_t)b * N * kMtHw + (size_t)blockIdx.x * 4;
    o[0] = ap; o[1] = ad; o[2] = th; o[3] = ph;
  }
}

// (theta, phi) share of the pair j at the trial point w + a dw, s + a ds
__device__ __forceinline__ void mt_trial_node(const MtProblem& P, const MtStepPtrs& q, int N, int j, double a, double mu,
                                              double& theta, double& phi) {
  const int jn = j + 1 == N ? 0 : j + 1;
  double wo[kMtNv], wn[kMtNv], eq[kMtNe], g[kMtNi];
#pragma unroll
  for (int k = 0; k < kMtNv; ++k) {
    wo[k] = q.w[(size_t)j * kMtNv + k] + a * q.dw[(size_t)j * kMtNv + k];
    wn[k] = q.w[(size_t)jn * kMtNv + k] + a * q.dw[(size_t)jn * kMtNv + k];
  }
  mt_pair<double>(P, j, wo, wn, eq, g);
  double c = wo[8];
#pragma unroll
  for (int k = 5; k < 8; ++k) c += 1e-4 * wo[k] * wo[k] + 1e-1 * (wn[k] - wo[k]) * (wn[k] - wo[k]);
  phi += c;
#pragma unroll
  for (int k = 0; k < kMtNe; ++k) theta += fabs(eq[k]);
#pragma unroll
  for (int k = 0; k < kMtNi; ++k) {
    const double s_ = q.sv[(size_t)j * kMtNi + k] + a * q.ds[(size_t)j * kMtNi + k];
    theta += fabs(g[k] + s_);
    phi -= mu * log(s_);
  }
}

// grid (node blocks, B): the first kMtTrials trial points, one node pair per thread -> vec[b][j][2 k .. 2 k + 1], k = trial.
// Round 4: ONE trial point (a = ap: 72 % of the accepted steps of the benchmark batch take it: histogram of a -DRL_MT_HIST build);
// the 28 % that need ap / 2, ap / 4, ... evaluate them in k_mt_step, whose halving loop forms the same per-node shares and
// adds them in the same order, so the decisions are bit for bit those of the three-point build (RL_MT_TRIALS=3 rebuilds it).
#ifndef RL_MT_TRIALS
#define RL_MT_TRIALS 1
#endif
constexpr int kMtTrials = RL_MT_TRIALS;
__global__ void __launch_bounds__(64) k_mt_trial(MtProblem P, MtState st) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y, N = P.N;
  const MtStepPtrs q = mt_step_ptrs(st, b, N);
  if (q.scal[5] != 0.0) return;
  // the primal step length: every workgroup takes the minimum over the shares of k_mt_dir itself (220 values; a minimum does not depend on the order)
  double ap = 1.0;
  {
    const double* part = st.hw + (size_t)b * N * kMtHw;
    for (int k = threadIdx.x; k < kMtDirBlocks(N); k += 64) ap = fmin(ap, part[4 * k]);
    ap = wave_min(ap);
  }
  if (j >= N) return;
  mt_instance(P, b);
  double* o = st.vec + ((size_t)b * N + j) * 16;
  double a = ap;
  const double mu = q.scal[0];
  for (int k = 0; k < kMtTrials; ++k) {
    double theta = 0.0, phi = 0.0;
    mt_trial_node(P, q, N, j, a, mu, theta, phi);
    o[2 * k] = theta; o[2 * k + 1] = phi;
    a *= 0.5;
  }
}

// The step decision in two stages (round 6).  72 % of the steps of the benchmark batch take the first trial point (a = ap); the others
// used to evaluate ap / 2, ap / 4, ... INSIDE k_mt_step -- one workgroup per instance, four nodes per thread one after the other,
// while the workgroups of the instances that were done sat idle: the kernel lasted as long as its slowest instance (865 us per
// iteration against 115 us for k_mt_trial's whole grid).  Now k_mt_step decides on the first point and, where it is rejected, only
// marks the instance (scal[15]) and leaves (ap, ad, theta0, phi0) in the free slots of node 0's trial record; k_mt_trial_b --
// k_mt_trial's grid, returning at once for unmarked instances -- evaluates ap / 2 and ap / 4 per node, and k_mt_step_b decides on
// those (and, for the 3 % that need more, goes on halving in the kernel as before).  The per-node shares are formed and added
// exactly as before (mt_trial_node, thread-strided sums, mt_block_reduce_n): the decisions and the iterates are the same bits.
static_assert(kMtTrials == 1, "the deferred halvings are written for one precomputed trial point");
constexpr int kMtStash = 8;   // vec[b][0][8 .. 11]: ap, ad, theta0, phi0 of an instance whose first trial point was rejected

__global__ void __launch_bounds__(256, 2) k_mt_step(MtProblem P, MtState st) {
  __shared__ double red[(4 + 2 * kMtTrials) * 4];
  const int b = blockIdx.x, tid = threadIdx.x, N = P.N;
  const MtStepPtrs q = mt_step_ptrs(st, b, N);
  if (q.scal[5] != 0.0) return;
  mt_instance(P, b);
  // ---- the step lengths and (theta0, phi0) of the current point from the shares of k_mt_dir
  double ap = 1.0, ad = 1.0, theta0 = 0.0, phi0 = 0.0;
  {
    const double* fun = st.fun + (size_t)b * N * kMtNf;
    const double* part = st.hw + (size_t)b * N * kMtHw;
    for (int k = tid; k < kMtDirBlocks(N); k += 256) {
      ap = fmin(ap, part[4 * k]); ad = fmin(ad, part[4 * k + 1]); theta0 += part[4 * k + 2]; phi0 += part[4 * k + 3];
    }
    for (int idx = tid; idx < N * kMtNe; idx += 256) {
      const int j = idx / kMtNe, c = idx - j * kMtNe;
      theta0 += fabs(fun[j * kMtNf + c]);
    }
    for (int idx = tid; idx < N; idx += 256) {
      const double* wj = q.w + (size_t)idx * kMtNv;
      const double* wn = q.w + (size_t)(idx + 1 == N ? 0 : idx + 1) * kMtNv;
      double c = wj[8];
      for (int a = 5; a < 8; ++a) c += 1e-4 * wj[a] * wj[a] + 1e-1 * (wn[a] - wj[a]) * (wn[a] - wj[a]);
      phi0 += c;
    }
  }
  // ---- the trial point of k_mt_trial
  double* tp = st.vec + (size_t)b * N * 16;
  double th0 = 0.0, ph0 = 0.0;
  for (int j = tid; j < N; j += 256) { th0 += tp[(size_t)j * 16]; ph0 += tp[(size_t)j * 16 + 1]; }
  {
    double v[6] = {ap, ad, theta0, phi0, th0, ph0};
    const int op[6] = {1, 1, 0, 0, 0, 0};
    mt_block_reduce_n<256, 6>(v, op, red);
    ap = v[0]; ad = v[1]; theta0 = v[2]; phi0 = v[3]; th0 = v[4]; ph0 = v[5];
  }
  const int nfilt_total = (int)q.scal[14], nfilt = nfilt_total < kMtFilter ? nfilt_total : kMtFilter;
  if (mt_accept(st, q.filt, nfilt, N, th0, ph0, theta0, phi0)) {
    RL_MT_COUNT(0);
    mt_take_step(st, q, N, tid, ap, 0, ap, ad, theta0, phi0);
    return;
  }
  // rejected: the halvings are evaluated by k_mt_trial_b / k_mt_step_b
  __syncthreads();   // everybody has read scal and the trial record
  if (tid == 0) { tp[kMtStash] = ap; tp[kMtStash + 1] = ad; tp[kMtStash + 2] = theta0; tp[kMtStash + 3] = phi0; q.scal[15] = 1.0; }
}

// grid as k_mt_trial: for the instances k_mt_step has marked, the trial points ap / 2 and ap / 4 per node -> vec[b][j][2 .. 5]
__global__ void __launch_bounds__(64) k_mt_trial_b(MtProblem P, MtState st) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y, N = P.N;
  const MtStepPtrs q = mt_step_ptrs(st, b, N);
  if (q.scal[5] != 0.0 || q.scal[15] == 0.0) return;
  if (j >= N) return;
  mt_instance(P, b);
  const double ap = st.vec[(size_t)b * N * 16 + kMtStash];
  const double mu = q.scal[0];
  double* o = st.vec + ((size_t)b * N + j) * 16;
  double a = 0.5 * ap;
  for (int k = 1; k <= 2; ++k) {
    double theta = 0.0, phi = 0.0;
    mt_trial_node(P, q, N, j, a, mu, theta, phi);
    o[2 * k] = theta; o[2 * k + 1] = phi;   // (node 0's record also holds the stash, in slots 8 .. 11)
    a *= 0.5;
  }
}

__global__ void __launch_bounds__(256, 2) k_mt_step_b(MtProblem P, MtState st) {
  __shared__ double red[4 * 4];
  const int b = blockIdx.x, tid = threadIdx.x, N = P.N;
  const MtStepPtrs q = mt_step_ptrs(st, b, N);
  if (q.scal[5] != 0.0 || q.scal[15] == 0.0) return;
  mt_instance(P, b);
  const double mu = q.scal[0];
  const double* tp = st.vec + (size_t)b * N * 16;
  const double ap = tp[kMtStash], ad = tp[kMtStash + 1], theta0 = tp[kMtStash + 2], phi0 = tp[kMtStash + 3];
  double th[2] = {0.0, 0.0}, ph[2] = {0.0, 0.0};
  for (int j = tid; j < N; j += 256) {
#pragma unroll
    for (int k = 0; k < 2; ++k) { th[k] += tp[(size_t)j * 16 + 2 + 2 * k]; ph[k] += tp[(size_t)j * 16 + 3 + 2 * k]; }
  }
  {
    double v[4] = {th[0], ph[0], th[1], ph[1]};
    const int op[4] = {0, 0, 0, 0};
    mt_block_reduce_n<256, 4>(v, op, red);
    th[0] = v[0]; ph[0] = v[1]; th[1] = v[2]; ph[1] = v[3];
  }
  const int nfilt_total = (int)q.scal[14], nfilt = nfilt_total < kMtFilter ? nfilt_total : kMtFilter;
  double a = 0.5 * ap;
  int taken = -1;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (taken < 0) {
      if (mt_accept(st, q.filt, nfilt, N, th[k], ph[k], theta0, phi0)) taken = 1 + k;
      else a *= 0.5;
    }
  }
  // ---- further halvings, evaluated here (3 % of the steps)
  for (int trial = 3; trial < 12 && taken < 0; ++trial) {
    double theta = 0.0, phi = 0.0;
    for (int j = tid; j < N; j += 256) {   // per-node shares from zero, then added: the association of k_mt_trial's path
      double tj = 0.0, pj = 0.0;
      mt_trial_node(P, q, N, j, a, mu, tj, pj);
      theta += tj; phi += pj;
    }
    {
      double v[2] = {theta, phi};
      const int op[2] = {0, 0};
      mt_block_reduce_n<256, 2>(v, op, red);
      theta = v[0]; phi = v[1];
    }
    if (mt_accept(st, q.filt, nfilt, N, theta, phi, theta0, phi0)) taken = trial;
    else a *= 0.5;
  }
  __syncthreads();
  if (tid == 0) q.scal[15] = 0.0;
  if (taken < 0) {
    // no acceptable step: more damping, same point (the next iteration re-solves with the larger delta)
    const double delta = fmax(10.0 * q.scal[1], 1e-4);
    __syncthreads();
    if (tid == 0) { q.scal[1] = delta; q.scal[7] = 0.0; q.scal[6] += 1.0; if (delta > 1e6) q.scal[5] = 2.0; }
    RL_MT_COUNT(12);
    return;
  }
  RL_MT_COUNT(taken);
  mt_take_step(st, q, N, tid, a, taken, ap, ad, theta0, phi0);
}

// physical X [B,N,6], U [B,N,4], T [B,N]  <->  scaled unknowns w [B,N,9]
__global__ void k_mt_pack(MtProblem P, MtState st, const double* X, const double* U, const double* T) {
  const int b = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x, N = P.N;
  if (j >= N) return;
  const size_t o = (size_t)b * N + j;
  double* w = st.w + o * kMtNv;
  for (int c = 0; c < 5; ++c) w[c] = X[o * 6 + 1 + c] / P.sw[c];
  w[5] = U[o * 4 + 0] / P.sw[5]; w[6] = U[o * 4 + 2] / P.sw[6]; w[7] = U[o * 4 + 3] / P.sw[7];
  w[8] = T[o] / P.sw[8];
}
__global__ void k_mt_unpack(MtProblem P, MtState st, double* X, double* U, double* T) {
  const int b = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x, N = P.N;
  if (j >= N) return;
  const size_t o = (size_t)b * N + j;
  const double* w = st.w + o * kMtNv;
  X[o * 6 + 0] = P.s[j];
  for (int c = 0; c < 5; ++c) X[o * 6 + 1 + c] = w[c] * P.sw[c];
  U[o * 4 + 0] = w[5] * P.sw[5]; U[o * 4 + 1] = 0.0; U[o * 4 + 2] = w[6] * P.sw[6]; U[o * 4 + 3] = w[7] * P.sw[7];
  T[o] = w[8] * P.sw[8];
}

// per-instance report [B,12] (include/rl_mincurv.h: rl_mintime_solve_batch)
__global__ void k_mt_stats(MtState st, double* stats) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= st.B) return;
  const double* q = st.scal + (size_t)b * 16;
  double* o = stats + (size_t)b * 12;
  o[0] = q[6]; o[1] = q[2]; o[2] = q[3]; o[3] = q[4]; o[4] = q[11]; o[5] = q[5]; o[6] = q[0]; o[7] = q[1];
  o[8] = q[7]; o[9] = q[10]; o[10] = q[12]; o[11] = q[13];
}

// initial slacks / multipliers from the functions at w0:  s = max(-g, 1e-2), z = mu0 / s, y = 0
__global__ void k_mt_init(MtProblem P, MtState st, double mu0, double delta0) {
  const int b = blockIdx.y, idx = blockIdx.x * blockDim.x + threadIdx.x, N = P.N;
  if (idx < N * kMtNi) {
    const int j = idx / kMtNi, c = idx - j * kMtNi;
    const double g = st.fun[((size_t)b * N + j) * kMtNf + kMtNe + c];
    const double s = fmax(-g, 1e-2);
    st.s[(size_t)b * N * kMtNi + idx] = s;
    st.z[(size_t)b * N * kMtNi + idx] = mu0 / s;
  }
  if (idx < N * kMtNe) st.y[(size_t)b * N * kMtNe + idx] = 0.0;
  if (idx < 16) st.scal[(size_t)b * 16 + idx] = idx == 0 ? mu0 : (idx == 1 ? delta0 : 0.0);
}

}  // namespace rl
