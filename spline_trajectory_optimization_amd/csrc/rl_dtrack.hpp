// rl_dtrack.hpp -- first slice of the min-time double-track NLP (BASELINE config 5):
// the model's FUNCTION EVALUATION for B candidate solutions x N nodes -- vehicle dynamics in the
// curvilinear (Frenet) frame, Hermite-Simpson defects, tyre friction ellipses, load-transfer
// residual, actuator constraints and the objective -- i.e. what an NLP/SQP solver calls at every
// iterate.  The solve itself (batched SQP with a block-banded KKT) is not built yet.
//
// Restates models/double_track.py:10-140 (dynamics: the double-track model of
// doi:10.1080/00423114.2019.1704804 with the reference's simplified Pacejka curve, tanh-blended
// drive/brake force, and its quirks, e.g. `lr` in BOTH axle loads, :68-79) and :143-204 (node
// constraints), min_time_optimizer.py:93-163 (cost, pairing of node i-1 with node i, curvature at
// the left node), utils/utils.py:10-18 (yaw / abscissa alignment).  CPU checker:
// oracle/double_track.py (numpy).  One thread per (instance, node pair); everything is FP64
// transcendental-heavy arithmetic on 11 + 11 inputs -> 17 outputs: VALU bound.
#pragma once
#include <hip/hip_runtime.h>

namespace rl {

// parameter slots (include/rl_mincurv.h: RL_DT_*)
enum DtParam {
  DT_KD_F, DT_KB_F, DT_MASS, DT_JZZ, DT_LF, DT_LR, DT_TWF, DT_TWR, DT_DELTA_MAX, DT_FR, DT_HCOG,
  DT_KROLL_F, DT_CL_F, DT_CL_R, DT_RHO, DT_A, DT_CD, DT_MU, DT_BF, DT_CF, DT_BR, DT_CR, DT_PMAX,
  DT_FD_MAX, DT_FB_MAX, DT_TD, DT_TB, DT_TDELTA, DT_NPARAM
};
constexpr int kDtNx = 6, kDtNu = 4, kDtNeq = 8, kDtNineq = 14;
constexpr double kDtGravity = 9.8;  // double_track.py:7

struct DtArgs {
  double p[DT_NPARAM];
  int B, N;
  const double* s;       // [N] abscissa of the nodes
  const double* kappa;   // [N] centre-line curvature at the nodes
  const double* left;    // [N] distance to the left boundary (> 0)
  const double* right;   // [N] distance to the right boundary (< 0)
  double margin;         // vehicle_width / 2 + safety_margin (min_time_optimizer.py:134)
  double track_length;
  const double* X;       // [B,N,6]  s, n, xi, omega, beta, v   (physical units)
  const double* U;       // [B,N,4]  force, (unused), delta, gamma_y
  const double* T;       // [B,N]    duration of the interval that STARTS at the node
  double* eq;            // [B,N,8]  Hermite-Simpson defect (6), load-transfer residual, abscissa pin
  double* ineq;          // [B,N,14] g <= 0: 4 tyre ellipses, power, 1 - v, force lo/hi, steer lo/hi, force rate,
                         //          steer rate, lateral position lo/hi
  double* cost_part;     // [B,N]    per-node objective terms (summed on the host side of the ABI)
};

struct DtTyres { double fx[4], fy[4], fz[4]; };  // fl, fr, rl, rr

__device__ __forceinline__ void dt_dynamics(const double* p, const double x[6], const double u[4], double k,
                                            double f[6], DtTyres& ty) {
  const double n = x[1], phi = x[2], omega = x[3], beta = x[4], v = x[5];
  const double fd = u[0] * (tanh(u[0]) * 0.5 + 0.5);   // :17
  const double fb = u[0] * (tanh(-u[0]) * 0.5 + 0.5);  // :18
  const double delta = u[2], gam = u[3];
  const double m = p[DT_MASS], lf = p[DT_LF], lr = p[DT_LR], l = lf + lr;
  const double roll = 0.5 * p[DT_FR] * m * kDtGravity;
  const double fxf = 0.5 * p[DT_KD_F] * fd + 0.5 * p[DT_KB_F] * fb - roll * lr / l;                  // :58
  const double fxr = 0.5 * (1 - p[DT_KD_F]) * fd + 0.5 * (1 - p[DT_KB_F]) * fb - roll * lf / l;      // :61
  const double v2 = v * v;
  const double ax = (fd + fb - 0.5 * p[DT_CD] * p[DT_A] * v2 - p[DT_FR] * m * kDtGravity) / m;       // :67
  const double stat = 0.5 * m * kDtGravity * lr / l, pitch = 0.5 * p[DT_HCOG] / l * m * ax;
  const double fzf = stat - pitch + 0.25 * p[DT_CL_F] * p[DT_RHO] * p[DT_A] * v2;                    // :70-72
  const double fzr = stat + pitch + 0.25 * p[DT_CL_R] * p[DT_RHO] * p[DT_A] * v2;                    // :75-77 (lr, as written)
  ty.fz[0] = fzf - p[DT_KROLL_F] * gam; ty.fz[1] = fzf + p[DT_KROLL_F] * gam;
  ty.fz[2] = fzr - (1 - p[DT_KROLL_F]) * gam; ty.fz[3] = fzr + (1 - p[DT_KROLL_F]) * gam;
  double sb, cb;
  sincos(beta, &sb, &cb);
  const double vs = v * sb, vc = v * cb;
  const double afl = delta - atan((lf * omega + vs) / (vc - 0.5 * p[DT_TWF] * omega));               // :83-90
  const double afr = delta - atan((lf * omega + vs) / (vc + 0.5 * p[DT_TWF] * omega));
  const double arl = atan((lr * omega - vs) / (vc - 0.5 * p[DT_TWR] * omega));
  const double arr = atan((lr * omega - vs) / (vc + 0.5 * p[DT_TWR] * omega));
  const double mu = p[DT_MU];
  ty.fy[0] = mu * ty.fz[0] * sin(p[DT_CF] * atan(p[DT_BF] * afl));                                   // :104-107
  ty.fy[1] = mu * ty.fz[1] * sin(p[DT_CF] * atan(p[DT_BF] * afr));
  ty.fy[2] = mu * ty.fz[2] * sin(p[DT_CR] * atan(p[DT_BR] * arl));
  ty.fy[3] = mu * ty.fz[3] * sin(p[DT_CR] * atan(p[DT_BR] * arr));
  ty.fx[0] = fxf; ty.fx[1] = fxf; ty.fx[2] = fxr; ty.fx[3] = fxr;
  double sd, cd_, sdb, cdb;
  sincos(delta, &sd, &cd_);
  sincos(delta - beta, &sdb, &cdb);
  const double fxF = ty.fx[0] + ty.fx[1], fxR = ty.fx[2] + ty.fx[3];
  const double fyF = ty.fy[0] + ty.fy[1], fyR = ty.fy[2] + ty.fy[3];
  const double drag = 0.5 * p[DT_CD] * p[DT_RHO] * p[DT_A] * v2;
  const double v_dot = (fxR * cb + fxF * cdb + fyR * sb - fyF * sdb - drag * cb) / m;                // :110-113
  const double beta_dot = -omega + (-fxR * sb + fxF * sdb + fyR * cb + fyF * cdb + drag * sb) / (m * v);  // :114-117
  const double omega_dot = ((ty.fx[3] - ty.fx[2]) * p[DT_TWR] / 2 - fyR * lr +
                            ((ty.fx[1] - ty.fx[0]) * cd_ + (ty.fy[0] - ty.fy[1]) * sd) * p[DT_TWF] / 2 +
                            (fyF * cd_ + fxF * sd) * lf) / p[DT_JZZ];                                // :118-121
  double s_dot = v * cos(phi + beta);
  const double n_dot = v * sin(phi + beta);
  s_dot /= (1 - n * k);                                                                               // :128
  const double phi_dot = omega - k * s_dot;                                                           // :129
  f[0] = s_dot; f[1] = n_dot; f[2] = phi_dot; f[3] = omega_dot; f[4] = beta_dot; f[5] = v_dot;
}

__device__ __forceinline__ double dt_sign(double x) { return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : 0.0); }

__global__ void __launch_bounds__(128) k_dt_eval_nodes(DtArgs a) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;  // left node of the pair (j, j+1)
  const int b = blockIdx.y;
  if (j >= a.N) return;
  const int jn = j + 1 == a.N ? 0 : j + 1;
  const double* p = a.p;
  const size_t o = (size_t)b * a.N + j, on = (size_t)b * a.N + jn;
  double x[6], xn[6], u[4], un[4];
#pragma unroll
  for (int c = 0; c < 6; ++c) { x[c] = a.X[o * 6 + c]; xn[c] = a.X[on * 6 + c]; }
#pragma unroll
  for (int c = 0; c < 4; ++c) { u[c] = a.U[o * 4 + c]; un[c] = a.U[on * 4 + c]; }
  const double t = a.T[o], k = a.kappa[j];
  // next state brought next to this one (utils.py:10-18)
  {
    const double d = xn[2] - x[2];
    xn[2] = atan2(sin(d), cos(d)) + x[2];
    const double ds = x[0] - xn[0], kk = fabs(ds) + a.track_length / 2.0;
    xn[0] = xn[0] + (kk - fmod(kk, a.track_length)) * dt_sign(ds);
  }
  double f1[6], f2[6], fm[6], xm[6];
  DtTyres ty, ty2;
  dt_dynamics(p, x, u, k, f1, ty);
  dt_dynamics(p, xn, u, k, f2, ty2);
#pragma unroll
  for (int c = 0; c < 6; ++c) xm[c] = 0.5 * (x[c] + xn[c]) + (t / 8.0) * (f1[c] - f2[c]);  // :170
  dt_dynamics(p, xm, u, k, fm, ty2);
  double* eq = a.eq + o * kDtNeq;
#pragma unroll
  for (int c = 0; c < 6; ++c) eq[c] = x[c] + (t / 6.0) * (f1[c] + 4 * fm[c] + f2[c]) - xn[c];  // :172
  const double delta = u[2], gam = u[3], v = x[5];
  double sd, cd_;
  sincos(delta, &sd, &cd_);
  eq[6] = gam - p[DT_HCOG] / (0.5 * (p[DT_TWF] + p[DT_TWR])) *
                    (ty.fy[2] + ty.fy[3] + (ty.fx[0] + ty.fx[1]) * sd + (ty.fy[0] + ty.fy[1]) * cd_);  // :183-184
  eq[7] = x[0] - a.s[j];                                                                              // :130
  double* g = a.ineq + o * kDtNineq;
#pragma unroll
  for (int w = 0; w < 4; ++w) {                                                                       // :178-180
    const double qx = ty.fx[w] / (p[DT_MU] * ty.fz[w]), qy = ty.fy[w] / (p[DT_MU] * ty.fz[w]);
    g[w] = qx * qx + qy * qy - 1.0;
  }
  const double fd = u[0] * (tanh(u[0]) * 0.5 + 0.5);
  g[4] = v * fd - p[DT_PMAX];                                                                         // :187
  g[5] = 1.0 - v;                                                                                     // :188
  g[6] = p[DT_FB_MAX] - u[0]; g[7] = u[0] - p[DT_FD_MAX];                                             // :193
  g[8] = -p[DT_DELTA_MAX] - delta; g[9] = delta - p[DT_DELTA_MAX];                                    // :194
  const double ru = (un[0] - u[0]) / t, rd = (un[2] - delta) / t;                                     // :199-201
  // the two rate constraints are two-sided; the more violated side of each is reported
  g[10] = fmax(p[DT_FB_MAX] / p[DT_TB] - ru, ru - p[DT_FD_MAX] / p[DT_TD]);
  g[11] = fmax(-p[DT_DELTA_MAX] / p[DT_TDELTA] - rd, rd - p[DT_DELTA_MAX] / p[DT_TDELTA]);
  g[12] = (a.right[j] + a.margin) - x[1]; g[13] = x[1] - (a.left[j] - a.margin);                      // :131-136
  // objective (min_time_optimizer.py:119-123) in the reference's scaled controls
  const double su[4] = {p[DT_FD_MAX], fabs(p[DT_FB_MAX]), p[DT_DELTA_MAX], p[DT_MASS] * 50.0};
  double c = t;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const double a0 = u[q] / su[q], a1 = un[q] / su[q];
    c += 1e-4 * a1 * a1 + 1e-1 * (a1 - a0) * (a1 - a0);
  }
  a.cost_part[o] = c;
}

}  // namespace rl
