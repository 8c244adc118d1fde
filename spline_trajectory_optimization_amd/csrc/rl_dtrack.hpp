// rl_dtrack.hpp -- first slice of the min-time double-track NLP (BASELINE config 5):
// the model's FUNCTION EVALUATION for B candidate solutions x N nodes -- vehicle dynamics in the
// curvilinear (Frenet) frame, Hermite-Simpson defects, tyre friction ellipses, load-transfer
// residual, actuator constraints and the objective -- i.e. what an NLP/SQP solver calls at every
// iterate.  The solve built on it: rl_mintime.hpp.
//
// Restates models/double_track.py:10-140 (dynamics: the double-track model of
// doi:10.1080/00423114.2019.1704804 with the reference's simplified Pacejka curve, tanh-blended
// drive/brake force, and its quirks, e.g. `lr` in BOTH axle loads, :68-79) and :143-204 (node
// constraints), min_time_optimizer.py:93-163 (cost, pairing of node i-1 with node i, curvature at
// the left node), utils/utils.py:10-18 (yaw / abscissa alignment).  CPU checker:
// oracle/dt_checker.py (numpy).  One thread per (instance, node pair); everything is FP64
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

// ---- scalar types: double for the values, Dual<ND> (value + ND directional derivatives, forward
// mode) for the Jacobian, and Dual<ND, Dual<ND>> (forward over forward) for second derivatives -- the
// Hessian of the Lagrangian the min-time solver needs (rl_mintime.hpp).  The model code below is written
// once against these overloads; every operation is generic in the component type T.
template <int ND, typename T = double>
struct Dual {
  T v;
  T d[ND];
};
#define RL_DUAL_FOR for (int i_ = 0; i_ < ND; ++i_)
#define RL_DT template <int ND, typename T> __device__ __forceinline__
__device__ __forceinline__ double m_val(double x) { return x; }
RL_DT double m_val(const Dual<ND, T>& x) { return m_val(x.v); }
RL_DT Dual<ND, T> operator+(Dual<ND, T> a, const Dual<ND, T>& b) { a.v = a.v + b.v; RL_DUAL_FOR a.d[i_] = a.d[i_] + b.d[i_]; return a; }
RL_DT Dual<ND, T> operator-(Dual<ND, T> a, const Dual<ND, T>& b) { a.v = a.v - b.v; RL_DUAL_FOR a.d[i_] = a.d[i_] - b.d[i_]; return a; }
RL_DT Dual<ND, T> operator-(Dual<ND, T> a) { a.v = -a.v; RL_DUAL_FOR a.d[i_] = -a.d[i_]; return a; }
RL_DT Dual<ND, T> operator*(const Dual<ND, T>& a, const Dual<ND, T>& b) { Dual<ND, T> r; r.v = a.v * b.v; RL_DUAL_FOR r.d[i_] = a.d[i_] * b.v + a.v * b.d[i_]; return r; }
RL_DT Dual<ND, T> operator/(const Dual<ND, T>& a, const Dual<ND, T>& b) { Dual<ND, T> r; const T ib = 1.0 / b.v; r.v = a.v * ib; RL_DUAL_FOR r.d[i_] = (a.d[i_] - r.v * b.d[i_]) * ib; return r; }
RL_DT Dual<ND, T> operator+(Dual<ND, T> a, double b) { a.v = a.v + b; return a; }
RL_DT Dual<ND, T> operator+(double b, Dual<ND, T> a) { a.v = a.v + b; return a; }
RL_DT Dual<ND, T> operator-(Dual<ND, T> a, double b) { a.v = a.v - b; return a; }
RL_DT Dual<ND, T> operator-(double b, Dual<ND, T> a) { a.v = b - a.v; RL_DUAL_FOR a.d[i_] = -a.d[i_]; return a; }
RL_DT Dual<ND, T> operator*(Dual<ND, T> a, double b) { a.v = a.v * b; RL_DUAL_FOR a.d[i_] = a.d[i_] * b; return a; }
RL_DT Dual<ND, T> operator*(double b, Dual<ND, T> a) { a.v = a.v * b; RL_DUAL_FOR a.d[i_] = a.d[i_] * b; return a; }
RL_DT Dual<ND, T> operator/(Dual<ND, T> a, double b) { const double ib = 1.0 / b; a.v = a.v * ib; RL_DUAL_FOR a.d[i_] = a.d[i_] * ib; return a; }
RL_DT Dual<ND, T> operator/(double b, const Dual<ND, T>& a) { Dual<ND, T> r; r.v = b / a.v; const T f = -(r.v / a.v); RL_DUAL_FOR r.d[i_] = f * a.d[i_]; return r; }
// f(a) with f(a.v) = v and f'(a.v) = dv given in the component type
RL_DT Dual<ND, T> chain(Dual<ND, T> a, const T& v, const T& dv) { a.v = v; RL_DUAL_FOR a.d[i_] = a.d[i_] * dv; return a; }
// ---- the model's elementary functions on plain doubles.  The derivative kernels of the min-time solve evaluate
// them once per direction pair and node (113 times per node and iteration): the library versions (185 / 85
// instructions for sincos / atan, with a large-argument path that the model's angles never take) were half of those
// kernels' instructions.  These are the fdlibm kernels (Sun Microsystems' k_sin.c / k_cos.c / s_atan.c coefficients)
// written branch-free for the wavefront: Cody-Waite reduction with a two-part pi/2 (the model's arguments are angles
// of a few radians; the absolute error grows like 1e-16 |x|, there is no large-argument path), polynomial on
// [-pi/4, pi/4], quadrant by select; atan with its four break points as selects and one division.  Accuracy ~1 ulp
// (the reference's numpy / CasADi values differ from any libm by as much).
__device__ __forceinline__ void m_sincos(double x, double& s, double& c) {
  const double k = rint(x * 6.36619772367581382433e-01);
  double r = fma(-k, 1.57079632673412561417e+00, x);
  r = fma(-k, 6.07710050650619224932e-11, r);
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                                2.75573137070700676789e-06), -1.98412698298579493134e-04),
                               8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double sr = fma(z * r, ps, r);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                                -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                               -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double cr = 1.0 - fma(-z * z, pc, 0.5 * z);   // 1 - (z/2 - z^2 pc)
  const int n = (int)fmin(fmax(k, -1e9), 1e9) & 3;
  const double ss = (n & 1) ? cr : sr, cc = (n & 1) ? sr : cr;
  s = (n & 2) ? -ss : ss;
  c = ((n + 1) & 2) ? -cc : cc;
}
__device__ __forceinline__ double m_sin(double x) { double s, c; m_sincos(x, s, c); return s; }
__device__ __forceinline__ double m_cos(double x) { double s, c; m_sincos(x, s, c); return c; }
__device__ __forceinline__ double m_tanh(double x) { return tanh(x); }
__device__ __forceinline__ double m_atan(double x) {
  const double ax = fabs(x);
  const bool r0 = ax < 0.4375, r1 = ax < 0.6875, r2 = ax < 1.1875, r3 = ax < 2.4375;
  // t = num / den:  x | (2x-1)/(2+x) | (x-1)/(x+1) | (x-1.5)/(1+1.5x) | -1/x
  const double num = r0 ? ax : (r1 ? 2.0 * ax - 1.0 : (r2 ? ax - 1.0 : (r3 ? ax - 1.5 : -1.0)));
  const double den = r0 ? 1.0 : (r1 ? 2.0 + ax : (r2 ? ax + 1.0 : (r3 ? fma(1.5, ax, 1.0) : ax)));
  const double hi = r0 ? 0.0 : (r1 ? 4.63647609000806093515e-01 : (r2 ? 7.85398163397448278999e-01
                                   : (r3 ? 9.82793723247329054082e-01 : 1.57079632679489655800e+00)));
  const double lo = r0 ? 0.0 : (r1 ? 2.26987774529616870924e-17 : (r2 ? 3.06161699786838301793e-17
                                   : (r3 ? 1.39033110312309984516e-17 : 6.12323399573676603587e-17)));
  const double t = num / den;
  const double z = t * t, w = z * z;
  const double s1 = z * fma(w, fma(w, fma(w, fma(w, fma(w, 1.62858201153657823623e-02, 4.97687799461593236017e-02),
                                                 6.66107313738753120669e-02), 9.09088713343650656196e-02),
                                   1.42857142725034663711e-01), 3.33333333333329318027e-01);
  const double s2 = w * fma(w, fma(w, fma(w, fma(w, -3.65315727442169155270e-02, -5.83357013379057348645e-02),
                                          -7.69187620504482999495e-02), -1.11111104054623557880e-01),
                            -1.99999999998764832476e-01);
  const double res = hi - ((t * (s1 + s2) - lo) - t);
  return x < 0.0 ? -res : res;   // a NaN argument falls through every comparison and comes back as NaN
}
__device__ __forceinline__ double m_abs(double x) { return fabs(x); }
__device__ __forceinline__ double m_atan2(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ double m_fmod(double x, double m) { return fmod(x, m); }
__device__ __forceinline__ double m_max(double a, double b) { return fmax(a, b); }
// sine and cosine of a dual number share ONE sincos of the value at every nesting level (the derivative of
// either needs the other): a nested dual costs one range reduction instead of four
RL_DT void m_sincos(const Dual<ND, T>& x, Dual<ND, T>& s, Dual<ND, T>& c) {
  T sv, cv; m_sincos(x.v, sv, cv); s = chain(x, sv, cv); c = chain(x, cv, -sv);
}
RL_DT Dual<ND, T> m_sin(const Dual<ND, T>& x) { T sv, cv; m_sincos(x.v, sv, cv); return chain(x, sv, cv); }
RL_DT Dual<ND, T> m_cos(const Dual<ND, T>& x) { T sv, cv; m_sincos(x.v, sv, cv); return chain(x, cv, -sv); }
RL_DT Dual<ND, T> m_tanh(const Dual<ND, T>& x) { const T t = m_tanh(x.v); return chain(x, t, 1.0 - t * t); }
RL_DT Dual<ND, T> m_atan(const Dual<ND, T>& x) { return chain(x, m_atan(x.v), 1.0 / (1.0 + x.v * x.v)); }
RL_DT Dual<ND, T> m_abs(Dual<ND, T> x) { const double sg = m_val(x.v) < 0.0 ? -1.0 : 1.0; x.v = m_abs(x.v); RL_DUAL_FOR x.d[i_] = x.d[i_] * sg; return x; }
RL_DT Dual<ND, T> m_atan2(const Dual<ND, T>& y, const Dual<ND, T>& x) {
  Dual<ND, T> r; r.v = m_atan2(y.v, x.v); const T q = 1.0 / (x.v * x.v + y.v * y.v);
  RL_DUAL_FOR r.d[i_] = (x.v * y.d[i_] - y.v * x.d[i_]) * q; return r;
}
RL_DT Dual<ND, T> m_fmod(Dual<ND, T> x, double m) { x.v = m_fmod(x.v, m); return x; }
RL_DT Dual<ND, T> m_max(const Dual<ND, T>& a, const Dual<ND, T>& b) { return m_val(a) >= m_val(b) ? a : b; }
#undef RL_DUAL_FOR
#undef RL_DT

template <typename S>
struct DtTyres { S fx[4], fy[4], fz[4]; };  // fl, fr, rl, rr

template <typename S>
__device__ __forceinline__ void dt_dynamics(const double* p, const S (&x)[6], const S (&u)[4], double k,
                                            S (&f)[6], DtTyres<S>& ty) {
  const S n = x[1], phi = x[2], omega = x[3], beta = x[4], v = x[5];
  const S fd = u[0] * (m_tanh(u[0]) * 0.5 + 0.5);    // :17
  const S fb = u[0] * (m_tanh(-u[0]) * 0.5 + 0.5);   // :18
  const S delta = u[2], gam = u[3];
  const double m = p[DT_MASS], lf = p[DT_LF], lr = p[DT_LR], l = lf + lr;
  const double roll = 0.5 * p[DT_FR] * m * kDtGravity;
  const S fxf = 0.5 * p[DT_KD_F] * fd + 0.5 * p[DT_KB_F] * fb - roll * lr / l;                   // :58
  const S fxr = 0.5 * (1 - p[DT_KD_F]) * fd + 0.5 * (1 - p[DT_KB_F]) * fb - roll * lf / l;       // :61
  const S v2 = v * v;
  const S ax = (fd + fb - 0.5 * p[DT_CD] * p[DT_A] * v2 - p[DT_FR] * m * kDtGravity) / m;        // :67
  const double stat = 0.5 * m * kDtGravity * lr / l;
  const S pitch = 0.5 * p[DT_HCOG] / l * m * ax;
  const S fzf = stat - pitch + 0.25 * p[DT_CL_F] * p[DT_RHO] * p[DT_A] * v2;                     // :70-72
  const S fzr = stat + pitch + 0.25 * p[DT_CL_R] * p[DT_RHO] * p[DT_A] * v2;                     // :75-77 (lr, as written)
  ty.fz[0] = fzf - p[DT_KROLL_F] * gam; ty.fz[1] = fzf + p[DT_KROLL_F] * gam;
  ty.fz[2] = fzr - (1 - p[DT_KROLL_F]) * gam; ty.fz[3] = fzr + (1 - p[DT_KROLL_F]) * gam;
  S sb, cb;
  m_sincos(beta, sb, cb);
  const S vs = v * sb, vc = v * cb;
  const S afl = delta - m_atan((lf * omega + vs) / (vc - 0.5 * p[DT_TWF] * omega));              // :83-90
  const S afr = delta - m_atan((lf * omega + vs) / (vc + 0.5 * p[DT_TWF] * omega));
  const S arl = m_atan((lr * omega - vs) / (vc - 0.5 * p[DT_TWR] * omega));
  const S arr = m_atan((lr * omega - vs) / (vc + 0.5 * p[DT_TWR] * omega));
  const double mu = p[DT_MU];
  ty.fy[0] = mu * ty.fz[0] * m_sin(p[DT_CF] * m_atan(p[DT_BF] * afl));                           // :104-107
  ty.fy[1] = mu * ty.fz[1] * m_sin(p[DT_CF] * m_atan(p[DT_BF] * afr));
  ty.fy[2] = mu * ty.fz[2] * m_sin(p[DT_CR] * m_atan(p[DT_BR] * arl));
  ty.fy[3] = mu * ty.fz[3] * m_sin(p[DT_CR] * m_atan(p[DT_BR] * arr));
  ty.fx[0] = fxf; ty.fx[1] = fxf; ty.fx[2] = fxr; ty.fx[3] = fxr;
  S sd, cd_, sdb, cdb;
  m_sincos(delta, sd, cd_);
  m_sincos(delta - beta, sdb, cdb);
  const S fxF = ty.fx[0] + ty.fx[1], fxR = ty.fx[2] + ty.fx[3];
  const S fyF = ty.fy[0] + ty.fy[1], fyR = ty.fy[2] + ty.fy[3];
  const S drag = 0.5 * p[DT_CD] * p[DT_RHO] * p[DT_A] * v2;
  const S v_dot = (fxR * cb + fxF * cdb + fyR * sb - fyF * sdb - drag * cb) / m;                 // :110-113
  const S beta_dot = -omega + (-fxR * sb + fxF * sdb + fyR * cb + fyF * cdb + drag * sb) / (m * v);  // :114-117
  const S omega_dot = ((ty.fx[3] - ty.fx[2]) * (p[DT_TWR] / 2) - fyR * lr +
                       ((ty.fx[1] - ty.fx[0]) * cd_ + (ty.fy[0] - ty.fy[1]) * sd) * (p[DT_TWF] / 2) +
                       (fyF * cd_ + fxF * sd) * lf) / p[DT_JZZ];                                  // :118-121
  S spb, cpb;
  m_sincos(phi + beta, spb, cpb);
  const S s_dot = v * cpb / (1 - n * k);                                                         // :122, :128
  const S n_dot = v * spb;
  const S phi_dot = omega - k * s_dot;                                                           // :129
  f[0] = s_dot; f[1] = n_dot; f[2] = phi_dot; f[3] = omega_dot; f[4] = beta_dot; f[5] = v_dot;
}

__device__ __forceinline__ double dt_sign(double x) { return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : 0.0); }

// everything a node pair contributes: 8 equality residuals, 14 inequality values, the objective terms
template <typename S>
__device__ __forceinline__ void dt_pair(const DtArgs& a, int j, const S (&x)[6], const S (&u)[4], S t,
                                        S (&xn)[6], const S (&un)[4], S (&eq)[kDtNeq], S (&g)[kDtNineq], S& cost) {
  const double* p = a.p;
  const double k = a.kappa[j];
  // next state brought next to this one (utils.py:10-18)
  {
    const S d = xn[2] - x[2];
    S sdd, cdd;
    m_sincos(d, sdd, cdd);
    xn[2] = m_atan2(sdd, cdd) + x[2];
    const S ds = x[0] - xn[0], kk = m_abs(ds) + a.track_length / 2.0;
    xn[0] = xn[0] + (kk - m_fmod(kk, a.track_length)) * dt_sign(m_val(ds));
  }
  S f1[6], f2[6], fm[6], xm[6];
  DtTyres<S> ty, ty2;
  dt_dynamics(p, x, u, k, f1, ty);
  dt_dynamics(p, xn, u, k, f2, ty2);
#pragma unroll
  for (int c = 0; c < 6; ++c) xm[c] = 0.5 * (x[c] + xn[c]) + (t / 8.0) * (f1[c] - f2[c]);  // :170
  dt_dynamics(p, xm, u, k, fm, ty2);
#pragma unroll
  for (int c = 0; c < 6; ++c) eq[c] = x[c] + (t / 6.0) * (f1[c] + 4.0 * fm[c] + f2[c]) - xn[c];  // :172
  const S delta = u[2], gam = u[3], v = x[5];
  S sd, cd_;
  m_sincos(delta, sd, cd_);
  eq[6] = gam - p[DT_HCOG] / (0.5 * (p[DT_TWF] + p[DT_TWR])) *
                    (ty.fy[2] + ty.fy[3] + (ty.fx[0] + ty.fx[1]) * sd + (ty.fy[0] + ty.fy[1]) * cd_);  // :183-184
  eq[7] = x[0] - a.s[j];                                                                              // :130
#pragma unroll
  for (int w = 0; w < 4; ++w) {                                                                       // :178-180
    const S qx = ty.fx[w] / (p[DT_MU] * ty.fz[w]), qy = ty.fy[w] / (p[DT_MU] * ty.fz[w]);
    g[w] = qx * qx + qy * qy - 1.0;
  }
  const S fd = u[0] * (m_tanh(u[0]) * 0.5 + 0.5);
  g[4] = v * fd - p[DT_PMAX];                                                                         // :187
  g[5] = 1.0 - v;                                                                                     // :188
  g[6] = p[DT_FB_MAX] - u[0]; g[7] = u[0] - p[DT_FD_MAX];                                             // :193
  g[8] = -p[DT_DELTA_MAX] - delta; g[9] = delta - p[DT_DELTA_MAX];                                    // :194
  const S ru = (un[0] - u[0]) / t, rd = (un[2] - delta) / t;                                          // :199-201
  // the two rate constraints are two-sided; the more violated side of each is reported
  g[10] = m_max(p[DT_FB_MAX] / p[DT_TB] - ru, ru - p[DT_FD_MAX] / p[DT_TD]);
  g[11] = m_max(-p[DT_DELTA_MAX] / p[DT_TDELTA] - rd, rd - p[DT_DELTA_MAX] / p[DT_TDELTA]);
  g[12] = (a.right[j] + a.margin) - x[1]; g[13] = x[1] - (a.left[j] - a.margin);                      // :131-136
  // objective (min_time_optimizer.py:119-123) in the reference's scaled controls
  const double su[4] = {p[DT_FD_MAX], fabs(p[DT_FB_MAX]), p[DT_DELTA_MAX], p[DT_MASS] * 50.0};
  cost = t;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const S a0 = u[q] / su[q], a1 = un[q] / su[q];
    cost = cost + 1e-4 * a1 * a1 + 1e-1 * (a1 - a0) * (a1 - a0);
  }
}

__global__ void __launch_bounds__(128) k_dt_eval_nodes(DtArgs a) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;  // left node of the pair (j, j+1)
  const int b = blockIdx.y;
  if (j >= a.N) return;
  const int jn = j + 1 == a.N ? 0 : j + 1;
  const size_t o = (size_t)b * a.N + j, on = (size_t)b * a.N + jn;
  double x[6], xn[6], u[4], un[4];
#pragma unroll
  for (int c = 0; c < 6; ++c) { x[c] = a.X[o * 6 + c]; xn[c] = a.X[on * 6 + c]; }
#pragma unroll
  for (int c = 0; c < 4; ++c) { u[c] = a.U[o * 4 + c]; un[c] = a.U[on * 4 + c]; }
  double eq[kDtNeq], g[kDtNineq], cost;
  dt_pair<double>(a, j, x, u, a.T[o], xn, un, eq, g, cost);
#pragma unroll
  for (int c = 0; c < kDtNeq; ++c) a.eq[o * kDtNeq + c] = eq[c];
#pragma unroll
  for (int c = 0; c < kDtNineq; ++c) a.ineq[o * kDtNineq + c] = g[c];
  a.cost_part[o] = cost;
}

// Jacobian of the node-pair functions with respect to the 21 local variables
//   [ x(6) | u(4) | t | x_next(6) | u_next(4) ]
// by forward-mode differentiation, kDtJacND directions per thread (blockIdx.z selects the slice):
// jac_eq [B,N,8,21], jac_ineq [B,N,14,21], grad_cost [B,N,21] (the node pair's objective terms).
constexpr int kDtNvar = 21, kDtJacND = 3;
struct DtJacArgs {
  DtArgs f;
  double* jac_eq;
  double* jac_ineq;
  double* grad_cost;
};

__global__ void __launch_bounds__(128) k_dt_eval_jac(DtJacArgs ja) {
  const DtArgs& a = ja.f;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y, v0 = blockIdx.z * kDtJacND;
  if (j >= a.N) return;
  const int jn = j + 1 == a.N ? 0 : j + 1;
  const size_t o = (size_t)b * a.N + j, on = (size_t)b * a.N + jn;
  using D = Dual<kDtJacND>;
  D x[6], xn[6], u[4], un[4], t;
  auto seed = [&](D& q, double val, int var) {
    q.v = val;
#pragma unroll
    for (int i = 0; i < kDtJacND; ++i) q.d[i] = (var == v0 + i) ? 1.0 : 0.0;
  };
#pragma unroll
  for (int c = 0; c < 6; ++c) { seed(x[c], a.X[o * 6 + c], c); seed(xn[c], a.X[on * 6 + c], 11 + c); }
#pragma unroll
  for (int c = 0; c < 4; ++c) { seed(u[c], a.U[o * 4 + c], 6 + c); seed(un[c], a.U[on * 4 + c], 17 + c); }
  seed(t, a.T[o], 10);
  D eq[kDtNeq], g[kDtNineq], cost;
  dt_pair<D>(a, j, x, u, t, xn, un, eq, g, cost);
#pragma unroll
  for (int i = 0; i < kDtJacND; ++i) {
    const int var = v0 + i;
    if (var < kDtNvar) {
#pragma unroll
      for (int c = 0; c < kDtNeq; ++c) ja.jac_eq[(o * kDtNeq + c) * kDtNvar + var] = eq[c].d[i];
#pragma unroll
      for (int c = 0; c < kDtNineq; ++c) ja.jac_ineq[(o * kDtNineq + c) * kDtNvar + var] = g[c].d[i];
      ja.grad_cost[o * kDtNvar + var] = cost.d[i];
    }
  }
}

}  // namespace rl
