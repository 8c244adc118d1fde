// rl_global_xy.hpp -- a15, second formulation: the GLOBAL min-curvature QP with BOTH coordinates of every control point free
// (julia/spline_traj_opt.ipynb L247-281, L410-432: Z = [c_x; c_y], one lateral and one longitudinal row per sample, bounds
// from the track widths and +-1 m; with the corrections of SURVEY.md App. A.6 -- second-derivative basis in the cost, the
// linear term kept, one normal convention).  CPU twin with the maths: oracle/mincurv_oracle.c: orc_global_mincurv_xy.
//
//   unknowns   z = (zx_j, zy_j), j < np = n - k, interleaved (index 2 j + c): control point j is c0_j + z_j
//   rows       per sample i:  lo_i <= n0_i . (r_i - p0_i) <= hi_i   (widths - margin)
//                             -lon <= t0_i . (r_i - p0_i) <= lon     (t0 = (n0y, -n0x))
//              both rows touch the k+1 control points of the sample's knot span, in both coordinates: the normal matrix is
//              cyclic-banded with half-bandwidth 2k+1 and made of 2x2 blocks  sum_i b_a(i) b_b(i) M_i,
//              M_i = dm_lat n0 n0' + dm_lon t0 t0'  -- three numbers per sample
//   cost       Gauss-Newton on sum kappa_i^2 in both coordinates, re-linearised n_outer times; the step of a linearisation is
//              taken whole or halved, whichever has the smaller sum kappa^2 (Gauss-Newton 2-cycles on near-straights)
//   QP         the Mehrotra predictor-corrector iteration of rl_global.hpp, 4N bound rows
//
// Mapping to the machine (one workgroup = one instance)
//   * thread = kXYRows samples (tid, tid + NT, ...); the interior-point state of their 2 x kXYRows rows stays in registers.
//   * what a sample contributes to the normal matrix and to the right-hand sides is a handful of WEIGHTS (3 for the matrix,
//     2 per vector), written to LDS.  The sums over the samples of a knot span,  S[a][b] = sum_i b_a(i) b_b(i) w_i  and
//     s[a] = sum_i b_a(i) v_i,  are formed by (span, table column) tasks that read the weights from LDS and the
//     instance-independent basis products from an L2-resident table of the track, in sample order -- bit-reproducible, no
//     chunk tables, no partial sums through LDS.  (Tried and dropped: the same sums on the matrix cores,
//     v_mfma_f64_16x16x4_f64 with four samples per instruction -- a 6 x 6 result uses 14 % of the 16 x 16 tile: 88 k cycles
//     per interior-point iteration against the vector units' few thousand.)  A short pass adds the k+1 span sums that make up
//     a 2x2 block and writes it into the factorisation's storage.
//   * the cyclic band is FOLDED (unknown order 0, nz-1, 1, nz-2, ...) into a plain band of half-bandwidth HB = 2 (2k+1), stored
//     by COLUMNS in LDS (column c: L[c+1..c+HB][c], a zero, D_c), so that every address of a column step is a per-lane constant
//     plus the column's base and rows outside the matrix read zeros -- no predication anywhere.  L D L' by wave 0, four
//     entries of the trailing update per lane, while the other waves form the right-hand side.  The triangular solves keep
//     the vector in registers and broadcast the pivot with v_readlane.
#pragma once
#include "rl_global2.hpp"

namespace rl {

#ifndef RL_XY_ROWS
#define RL_XY_ROWS 4
#endif
constexpr int kXYRows = RL_XY_ROWS;     // samples per thread
constexpr int kXYMaxNz = 192;  // unknowns: three rows per lane in the solves
// (interior-point exit rule: rl_device.hpp, ipm_done with kIpmTolTwoCoords -- tighter than the one-offset kernels': with both
// coordinates free the cost is nearly flat along the line and an under-converged iterate moves the result)

struct GlobalXYArgs {
  TrackDev trk;
  const int* span_first;   // [np+1] first sample of knot span s (control points s .. s+k); span_first[np] = N
  const int* chunk_first;  // [nch+1] the samples of a span cut into CHUNKS of bounded length (Monza, N = 2000: one span has 281
                           //         samples, the mean is 33 -- a task per span waits for that one): first sample of chunk j
  const int* span_chunk0;  // [np+1] first chunk of span s (chunks of a span are consecutive; an empty span has none)
  int nch;
  const double* bbx;       // [N][NE + K1]: basis products D0[a][i] D0[b][i] (a >= b, column a (a+1)/2 + b), then D0[a][i]
  int np;
  const double* widths;    // [B][N][2] (w_left, w_right)
  double margin, lon;
  int n_outer, max_ipm;
  double* out_ctrl;        // [B][n][2]
  double* out_xy;          // [B][N][2] or null
  double* out_z;           // [B][np][2] or null
  double* out_stats;       // [B][8]
};

struct GlobalXYLayout {  // offsets in doubles
  int xs, zs, dxs, cs, c0s, qv, rdP, rd, rhs, dinv, Pc, Lf, sums, wbuf, red, sfirst, cfirst, sch0, total;
};

__host__ __device__ inline GlobalXYLayout global_xy_layout(int k, int n, int np, int N, int nch) {
  const int K1 = k + 1, NE = K1 * (K1 + 1) / 2, H1 = 2 * k + 3, HB = 2 * (2 * k + 1), CW = HB + 2, nz = 2 * np;   // (H1 odd: LDS banks)
  GlobalXYLayout L;
  int o = 0;
  auto take = [&](int c) { const int r = o; o += (c + 1) & ~1; return r; };   // 16-byte aligned pieces
  L.xs = take(2 * n); L.zs = take(2 * n); L.dxs = take(2 * n); L.cs = take(2 * n); L.c0s = take(2 * n);
  L.qv = take(nz); L.rdP = take(nz); L.rd = take(nz); L.rhs = take(nz); L.dinv = take(nz);
  L.Pc = take(nz * H1); L.Lf = take((nz + HB + 1) * CW);
  const int ssum = nch * NE * 3, part = 4 * ((nch * K1) | 1);
  L.sums = take(ssum > part ? ssum : part);      // chunk sums of the matrix / of the vectors (never live together)
  L.wbuf = take(4 * N);
  L.red = take(64);
  L.sfirst = take((np + 2) / 2 + 1);
  L.cfirst = take((nch + 2) / 2 + 1);
  L.sch0 = take((np + 2) / 2 + 1);
  L.total = o;
  return L;
}

template <int K>
__global__ void k_global_xy_pairs(TrackDev tr, double* __restrict__ bbx) {
  constexpr int K1 = K + 1, NE = K1 * (K1 + 1) / 2, ROW = NE + K1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= tr.N) return;
  double b[K1];
#pragma unroll
  for (int a = 0; a < K1; ++a) b[a] = tr.D[(size_t)a * tr.N + i];
  int e = 0;
#pragma unroll
  for (int a = 0; a < K1; ++a)
#pragma unroll
    for (int c = 0; c <= a; ++c) bbx[(size_t)i * ROW + e++] = b[a] * b[c];
#pragma unroll
  for (int a = 0; a < K1; ++a) bbx[(size_t)i * ROW + NE + a] = b[a];
}

// A zero the compiler cannot see through, re-made wherever it is called.  Added to the index of a table or LDS access it keeps
// the address computation INSIDE the loop that uses it: hoisted out of the interior-point loop, the ~100 loop-invariant
// addresses of this kernel (six basis rows x four samples x three tables, ...) are spilled to scratch and every load first
// waits for its own address to come back from memory.
__device__ __forceinline__ int xy_opaque_zero() {
  int z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  return z;
}
__device__ __forceinline__ double xy_uniform(double v) {   // a wave-uniform value into scalar registers
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// position of cyclic unknown u in the folded order 0, nz-1, 1, nz-2, ... and back (nz even)
__device__ __forceinline__ int xy_fold(int u, int nz) { return u < (nz >> 1) ? 2 * u : 2 * (nz - 1 - u) + 1; }
__device__ __forceinline__ int xy_unfold(int r, int nz) { return (r & 1) ? nz - 1 - (r >> 1) : (r >> 1); }

// Column-major band: column c occupies CW = HB + 2 doubles,  LT[c CW + d - 1] = K[c+d][c] (d = 1..HB),  LT[c CW + HB] = 0,
// LT[c CW + HB + 1] = K[c][c].  HB + 1 zero columns follow the nz real ones, so rows past the matrix read as zeros.
// (CW even on purpose: the lanes of a trailing update and of the backward solve step by CW - 1 doubles, which must be odd
// for the LDS banks.)
template <int HB>
__device__ __forceinline__ double* xy_entry(double* LT, int r, int c) {   // r >= c
  return LT + c * (HB + 2) + (r == c ? HB + 1 : r - c - 1);
}

// In-place L D L'.  Afterwards the sub-diagonal slots hold L, the diagonal slots D, dinv[c] = 1/D_c.  ONE wave; lane = EPL
// entries (t1 >= t2) of the trailing update of the current column (rows c+1+t1, c+1+t2); lanes < HB also scale the column.
// No predication: entries whose rows lie past the matrix have a zero multiplier and rewrite zeros.
template <int HB>
__device__ __forceinline__ void fband_factor(double* LT, double* dinv, int nz, int lane) {
  constexpr int CW = HB + 2, NENT = HB * (HB + 1) / 2, EPL = (NENT + kWave - 1) / kWave;
  int o1[EPL], o2[EPL], oo[EPL];
#pragma unroll
  for (int q = 0; q < EPL; ++q) {
    const int e = lane + kWave * q;
    int t1 = 0;
    while ((t1 + 1) * (t1 + 2) / 2 <= e) ++t1;
    const int t2 = e - t1 * (t1 + 1) / 2;
    if (e < NENT) {
      o1[q] = t1; o2[q] = t2;                                                     // L[c+1+t1][c], L[c+1+t2][c]
      oo[q] = t1 == t2 ? (1 + t1) * CW + HB + 1 : (1 + t2) * CW + (t1 - t2 - 1);  // K[c+1+t1][c+1+t2]
    } else {
      o1[q] = HB; o2[q] = HB; oo[q] = CW + HB;                                    // zeros: 0 - 0 * 0
    }
  }
  // per-lane pointers that advance by one column per step (an index plus the column's base costs an extra add per access, and
  // this wave is alone on its SIMD: every instruction is ~8 cycles)
  double* pc = LT + (lane < HB ? lane : HB);
  double *p1[EPL], *p2[EPL], *po[EPL];
#pragma unroll
  for (int q = 0; q < EPL; ++q) { p1[q] = LT + o1[q]; p2[q] = LT + o2[q]; po[q] = LT + oo[q]; }
  const double* pd = LT + HB + 1;
  for (int c = 0; c < nz; ++c) {
    const double inv = frcp(*pd);
    const double uc = *pc;
    double u1[EPL], u2[EPL], own[EPL];
#pragma unroll
    for (int q = 0; q < EPL; ++q) { u1[q] = *p1[q]; u2[q] = *p2[q]; own[q] = *po[q]; }
    wave_lds_fence();
#pragma unroll
    for (int q = 0; q < EPL; ++q) *po[q] = fma(-u1[q], u2[q] * inv, own[q]);
    *pc = uc * inv;
    if (lane == 0) dinv[c] = inv;
    wave_lds_fence();
    pd += CW; pc += CW;
#pragma unroll
    for (int q = 0; q < EPL; ++q) { p1[q] += CW; p2[q] += CW; po[q] += CW; }
  }
}

// (L D L') x = rhs with the factor above.  ONE wave; folded row r = lane + 64 q in registers (NG groups); rhs and out in
// CYCLIC order.  Per column: one readlane pair, and per row group one subtract, one unsigned minimum (rows outside the
// band land on the column's zero), one LDS read, one fma.  The factor's entries of the NEXT column are read while the current
// one is applied: the addresses do not depend on the solution, and a step that waits for its own LDS read costs 185 cycles
// instead of ~60 (measured: 45 k -> see DESIGN_HISTORY.md 3b cycles per solve).
template <int HB, int NG>
__device__ __forceinline__ void fband_solve(const double* LT, const double* dinv, int nz, int lane, const double* rhs, double* out) {
  constexpr int CW = HB + 2;
  double y[NG], ln[NG];
  int rcl[NG];
#pragma unroll
  for (int q = 0; q < NG; ++q) {
    const int r = lane + kWave * q;
    y[q] = r < nz ? rhs[xy_unfold(r, nz)] : 0.0;
    rcl[q] = min(r, nz) * CW;          // (rows past the matrix: a zero column)
  }
  // L y = rhs: y_r -= L[r][c] y_c, L[r][c] = LT[c CW + (r - c - 1)]
  auto fetch_f = [&](int c) {
    const double* col = LT + c * CW;
#pragma unroll
    for (int q = 0; q < NG; ++q) ln[q] = col[min((unsigned)(lane + kWave * q - c - 1), (unsigned)HB)];
  };
  fetch_f(0);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int c1 = min(nz - 1, kWave * (g + 1));
    for (int c = kWave * g; c < c1; ++c) {
      double lc[NG];
#pragma unroll
      for (int q = 0; q < NG; ++q) lc[q] = ln[q];
      fetch_f(c + 1);                    // (column nz - 1 has no rows below: its entries are zeros)
      const double yc = lane_bcast(y[g], c - kWave * g);
#pragma unroll
      for (int q = 0; q < NG; ++q) y[q] = fma(-lc[q], yc, y[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < NG; ++q) { const int r = lane + kWave * q; if (r < nz) y[q] *= dinv[r]; }
  // L' x = y: x_r -= L[c][r] x_c, L[c][r] = LT[r CW + (c - r - 1)]
  auto fetch_b = [&](int c) {
#pragma unroll
    for (int q = 0; q < NG; ++q) {     // (rows outside the band read ONE zero: their own pads lie 24 doubles apart -- bank conflicts)
      const unsigned idx = (unsigned)(c - lane - kWave * q - 1);
      ln[q] = LT[idx < (unsigned)HB ? rcl[q] + idx : HB];
    }
  };
  fetch_b(nz - 1);
#pragma unroll
  for (int g = NG - 1; g >= 0; --g) {
    const int c0 = max(1, kWave * g);
    for (int c = min(nz, kWave * (g + 1)) - 1; c >= c0; --c) {
      double lc[NG];
#pragma unroll
      for (int q = 0; q < NG; ++q) lc[q] = ln[q];
      fetch_b(c - 1);                    // (c - 1 = 0: nothing above row 0, zeros)
      const double xc = lane_bcast(y[g], c - kWave * g);
#pragma unroll
      for (int q = 0; q < NG; ++q) y[q] = fma(-lc[q], xc, y[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < NG; ++q) { const int r = lane + kWave * q; if (r < nz) out[xy_unfold(r, nz)] = y[q]; }
}

// Span sums.  A task is (span s, column p of the track's table bbx): p < NE a basis pair (a >= b), p >= NE a single basis
// function; it runs over the samples of its span in order.  The lanes of a wave are consecutive columns of a few spans, so a
// load instruction reads a few contiguous runs of the table and the weights come as LDS broadcasts.  The table values of
// kXYBatch samples are requested together -- the loop is bound by the L2 latency otherwise (measured: 380 k cycles per
// interior-point iteration when each term waited for its own load).
//   MODE 0:  sums[(s NE + p) 3 + c] = sum_i bb_p(i) w_c(i),  c < 3, weights at wb[c N + i]
//   MODE 1/2: sums[n VP + s K1 + a] = sum_i b_a(i) v_n(i),  n < 4 / n < 2, the 2-vectors at wb2[i], wb2[N + i];
//            VP = (np K1) | 1: component planes -- consecutive tasks write consecutive words
// "Span" here is a CHUNK of a span (GlobalXYArgs::chunk_first): the callers add the chunks of a span in order.
// Tasks t0, t0 + nth, ... of this thread.
constexpr int kXYBatch = 12;
template <int K, int MODE>
__device__ __forceinline__ void xy_span_sums(const double* __restrict__ bbx, int N, int np, const int* sfirst, const double* wb,
                                             double* sums, int t0, int nth) {   // (np, sfirst: the CHUNKS and their first samples)
  constexpr int K1 = K + 1, NE = K1 * (K1 + 1) / 2, ROW = NE + K1, NT_ = MODE == 0 ? NE : K1, U = kXYBatch;
  const double2* wb2 = reinterpret_cast<const double2*>(wb);
  if (MODE == 0) {
    // a task = three basis pairs of one span (adjacent table columns): the three weights of a sample are read from LDS once
    // for nine products, and one round of tasks covers the track (np (NE / 3) <= threads).  With one pair per task the serial
    // LDS round trips of the weights were most of the 118 k cycles this pass took per interior-point iteration.
    constexpr int PM = 3, NGR = (NE + PM - 1) / PM, US = 4;
    for (int task = t0; task < np * NGR; task += nth) {
      const int sp = task / NGR, g = task - sp * NGR;
      const int p0 = g * PM, cnt = min(PM, NE - p0);
      const int e0 = sfirst[sp], e1 = sfirst[sp + 1];
      const int oz = xy_opaque_zero();
      const double* __restrict__ col = bbx + (p0 + oz);
      const double* wbo = wb + oz;
      double acc[PM][3];
#pragma unroll
      for (int q = 0; q < PM; ++q) { acc[q][0] = 0.0; acc[q][1] = 0.0; acc[q][2] = 0.0; }
      for (int base = e0; base < e1; base += US) {
        double tv[US][PM], wv[US][3];
#pragma unroll
        for (int u = 0; u < US; ++u) {
          const double* __restrict__ row = col + (size_t)min(base + u, e1 - 1) * ROW;
#pragma unroll
          for (int q = 0; q < PM; ++q) tv[u][q] = row[q < cnt ? q : 0];
        }
#pragma unroll
        for (int u = 0; u < US; ++u) {
          const int i = min(base + u, e1 - 1);
          wv[u][0] = wbo[i]; wv[u][1] = wbo[N + i]; wv[u][2] = wbo[2 * N + i];
        }
#pragma unroll
        for (int u = 0; u < US; ++u) {
          const bool in = base + u < e1;     // past the span: zero weights
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const double w = in ? wv[u][c] : 0.0;
#pragma unroll
            for (int q = 0; q < PM; ++q) acc[q][c] = fma(tv[u][q], w, acc[q][c]);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < PM; ++q)
        if (q < cnt) { double* o = sums + (sp * NE + p0 + q) * 3; o[0] = acc[q][0]; o[1] = acc[q][1]; o[2] = acc[q][2]; }
    }
    return;
  }
  for (int task = t0; task < np * NT_; task += nth) {
    const int sp = task / NT_, p = task - sp * NT_;
    const int e0 = sfirst[sp], e1 = sfirst[sp + 1];
    const int oz = xy_opaque_zero();
    const double* __restrict__ col = bbx + ((MODE == 0 ? p : NE + p) + oz);
    const double* wbo = wb + oz;
    const double2* wb2o = wb2 + oz;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int base = e0; base < e1; base += U) {
      // past the span: the last sample again, with a zero table value -- no branch in the batch, so that the U table loads
      // are issued together
      double tv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) tv[u] = col[(size_t)min(base + u, e1 - 1) * ROW];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = min(base + u, e1 - 1);
        const double t = base + u < e1 ? tv[u] : 0.0;
        if (MODE == 0) {
          a0 = fma(t, wbo[i], a0); a1 = fma(t, wbo[N + i], a1); a2 = fma(t, wbo[2 * N + i], a2);
        } else if (MODE == 1) {
          const double2 v0 = wb2o[i], v1 = wb2o[N + i];
          a0 = fma(t, v0.x, a0); a1 = fma(t, v0.y, a1); a2 = fma(t, v1.x, a2); a3 = fma(t, v1.y, a3);
        } else {
          const double2 v0 = wb2o[i];
          a0 = fma(t, v0.x, a0); a1 = fma(t, v0.y, a1);
        }
      }
    }
    if (MODE == 0) { double* o = sums + task * 3; o[0] = a0; o[1] = a1; o[2] = a2; }
    else { const int VP = (np * K1) | 1; double* o = sums + task; o[0] = a0; o[VP] = a1; o[2 * VP] = a2; o[3 * VP] = a3; }
  }
}

// ------------------------------------------------------------------------------------------------
template <int K, int NT>
__global__ void __launch_bounds__(NT) k_global_xy(GlobalXYArgs a) {
  constexpr int K1 = K + 1, NE = K1 * (K1 + 1) / 2, HC = 2 * K + 1, H1 = HC + 2, HB = 2 * HC, CW = HB + 2;   // H1: row stride of Pc (odd; HC + 1 slots used)
  constexpr int R = kXYRows;
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: wave-level control flow stays on the scalar unit)
  constexpr int nw = NT >> 6;
  const int b = blockIdx.x;
  const int N = a.trk.N, n = a.trk.n, np = a.np, nz = 2 * np;
  const GlobalXYLayout L = global_xy_layout(K, n, np, N, a.nch);
  double2* xs = reinterpret_cast<double2*>(lds + L.xs);
  double2* zs = reinterpret_cast<double2*>(lds + L.zs);
  double2* dxs = reinterpret_cast<double2*>(lds + L.dxs);
  double2* cs = reinterpret_cast<double2*>(lds + L.cs);
  double2* c0s = reinterpret_cast<double2*>(lds + L.c0s);
  double *qv = lds + L.qv, *rdP = lds + L.rdP, *rd = lds + L.rd, *rhs = lds + L.rhs, *dinv = lds + L.dinv;
  double *Pc = lds + L.Pc, *Lf = lds + L.Lf, *sums = lds + L.sums, *wb = lds + L.wbuf, *red = lds + L.red;
  double2* wb2 = reinterpret_cast<double2*>(wb);
  int* sfirst = reinterpret_cast<int*>(lds + L.sfirst);
  int* cfirst = reinterpret_cast<int*>(lds + L.cfirst);
  int* sch0 = reinterpret_cast<int*>(lds + L.sch0);
  const int nch = a.nch;
  double* xsf = reinterpret_cast<double*>(xs);     // flat views: unknown 2 j + c
  double* dxf = reinterpret_cast<double*>(dxs);
  double* zsf = reinterpret_cast<double*>(zs);

  // block reduction of two sums, a maximum and a minimum: two barriers
  auto reduce4 = [&](double& s1, double& s2, double& vmax, double& vmin) {
    s1 = wave_sum(s1); s2 = wave_sum(s2); vmax = wave_max(vmax); vmin = wave_min(vmin);
    if (lane == 0) { red[wave] = s1; red[16 + wave] = s2; red[32 + wave] = vmax; red[48 + wave] = vmin; }
    __syncthreads();
    double a1 = 0.0, a2 = 0.0, mx = -INFINITY, mn = INFINITY;
    for (int w = 0; w < nw; ++w) { a1 += red[w]; a2 += red[16 + w]; mx = fmax(mx, red[32 + w]); mn = fmin(mn, red[48 + w]); }
    __syncthreads();
    s1 = xy_uniform(a1); s2 = xy_uniform(a2); vmax = xy_uniform(mx); vmin = xy_uniform(mn);
  };
  auto wrap = [&](double2* v) {   // periodic copies of the first k control points; call between barriers
    if (tid < K) v[np + tid] = v[tid];
  };

  const double* __restrict__ D0 = a.trk.D;
  const double* __restrict__ D1 = a.trk.D + (size_t)K1 * N;
  const double* __restrict__ D2 = a.trk.D + (size_t)2 * K1 * N;
  const double* __restrict__ bbx = a.bbx;
  const double* __restrict__ wid = a.widths + (size_t)b * N * 2;

  // ---- this thread's samples
  // sample r of this thread: the lanes of a wave hold CONSECUTIVE samples, so that everything indexed by the sample -- table
  // reads, weights, A dx in LDS -- is coalesced / conflict-free (R consecutive samples per thread: 8-way LDS bank conflicts)
#define XY_SMP(r) (tid + (r) * NT)
  bool ok[R];
  int j0[R];
  double nx[R], ny[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = XY_SMP(r);
    ok[r] = i < N;
    j0[r] = 0; nx[r] = 0.0; ny[r] = 1.0;
    if (ok[r]) {
      j0[r] = a.trk.ell[i] - K;
      nx[r] = a.trk.base[(size_t)2 * N + i]; ny[r] = a.trk.base[(size_t)3 * N + i];
    }
  }
  auto lat_bounds = [&](int r, double& lo_, double& hi_) {   // (read where needed: two registers per sample less to keep)
    const double2 w = reinterpret_cast<const double2*>(wid)[XY_SMP(r) + xy_opaque_zero()];
    lo_ = -(w.y - a.margin); hi_ = w.x - a.margin;
  };
  for (int j = tid; j < n; j += NT) {
    c0s[j] = make_double2(a.trk.c0[j], a.trk.c0[n + j]);
    zs[j] = make_double2(0.0, 0.0);
  }
  for (int j = tid; j <= np; j += NT) { sfirst[j] = a.span_first[j]; sch0[j] = a.span_chunk0[j]; }
  for (int j = tid; j <= nch; j += NT) cfirst[j] = a.chunk_first[j];
  __syncthreads();

#ifdef RL_XY_PROFILE   // diagnostic build: out_z[b][0..11] = cycles (s_memtime) per phase, summed over the solve
  long long pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt0 = clock64();
#define XY_STAMP(slot) { const long long now_ = clock64(); pt[slot] += now_ - pt0; pt0 = now_; }
#else
#define XY_STAMP(slot)
#endif
  // interior-point state: [kind][sample], kind 0 lateral, 1 longitudinal
  double sl[2][R], su[2][R], ll[2][R], lu[2][R], rpl[2][R], rpu[2][R];
  double k2_first = 0.0, k2_last = 0.0, last_step = 0.0;
  int total_it = 0, n_halved = 0;

  // the basis values of the thread's samples (coalesced: sample index fastest in the table) and their products with a vector
  // of LDS control-point pairs: (px, py) = sum_al b_al v[j0 + al]
  auto load_b = [&](double (&bv)[K1][R]) {
    const int oz = xy_opaque_zero();
#pragma unroll
    for (int al = 0; al < K1; ++al) {
#pragma unroll
      for (int r = 0; r < R; ++r) bv[al][r] = D0[(size_t)al * N + min(XY_SMP(r), N - 1) + oz];
    }
  };
  auto dot_b = [&](const double (&bv)[K1][R], int r, const double2* v, double& px, double& py) {
    px = 0.0; py = 0.0;
    const double2* vo = v + xy_opaque_zero();
#pragma unroll
    for (int al = 0; al < K1; ++al) {
      const double2 x = vo[j0[r] + al];
      px = fma(bv[al][r], x.x, px); py = fma(bv[al][r], x.y, py);
    }
  };
  // gathered(j, c, q) = sum_al sums[((j - al) K1 + al) 4 + 2 q + c]: the k+1 spans that contain control point j
  const int vplane = (nch * K1) | 1;      // xy_span_sums: plane stride of the vector sums
  auto gathered = [&](int j, int c, int q) {
    double s = 0.0;
#pragma unroll
    for (int al = 0; al < K1; ++al) {
      int sp = j - al; if (sp < 0) sp += np;
      for (int ch = sch0[sp]; ch < sch0[sp + 1]; ++ch) s += sums[(2 * q + c) * vplane + ch * K1 + al];
    }
    return s;
  };

  for (int outer = 0;; ++outer) {
    // ---- current control points
    for (int j = tid; j < n; j += NT) cs[j] = make_double2(c0s[j].x + zs[j].x, c0s[j].y + zs[j].y);
    for (int q = tid; q < nz * H1; q += NT) Pc[q] = 0.0;
    __syncthreads();
    // ---- curvature, its derivatives with respect to (x', y', x'', y''), the Gauss-Newton residual
    double k2p = 0.0;
    double gres[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      gres[r] = 0.0;
      if (ok[r]) {
        const int oz = xy_opaque_zero();
        const int i = XY_SMP(r) + oz;
        double dx = 0, dy = 0, ddx = 0, ddy = 0, zdx = 0, zdy = 0, zddx = 0, zddy = 0;
#pragma unroll
        for (int al = 0; al < K1; ++al) {
          const double b1 = D1[(size_t)al * N + i], b2 = D2[(size_t)al * N + i];
          const double2 c = cs[j0[r] + al + oz], z = zs[j0[r] + al + oz];
          dx = fma(c.x, b1, dx); dy = fma(c.y, b1, dy); ddx = fma(c.x, b2, ddx); ddy = fma(c.y, b2, ddy);
          zdx = fma(z.x, b1, zdx); zdy = fma(z.y, b1, zdy); zddx = fma(z.x, b2, zddx); zddy = fma(z.y, b2, zddy);
        }
        const double s2 = dx * dx + dy * dy, inv3 = 1.0 / (s2 * sqrt(s2));
        const double kp = (dx * ddy - dy * ddx) * inv3;
        k2p = fma(kp, kp, k2p);
        const double g1x = ddy * inv3 - 3.0 * kp * dx / s2, g1y = -ddx * inv3 - 3.0 * kp * dy / s2;
        const double g2x = -dy * inv3, g2y = dx * inv3;
        gres[r] = kp - (g1x * zdx + g2x * zddx + g1y * zdy + g2y * zddy);
        wb2[i] = make_double2(g1x, g2x);
        wb2[N + i] = make_double2(g1y, g2y);
      }
    }
    {
      double z1 = 0.0, vmx = 0.0, vmn = 0.0;
      reduce4(k2p, z1, vmx, vmn);          // (its barriers also publish the weights)
    }
    if (outer == 0) k2_first = k2p;
    k2_last = k2p;
    if (outer == a.n_outer) break;
    // ---- P = 2 G'G as a cyclic band: Pc[(2 j1 + c1) H1 + dd] = P[2 j1 + c1][2 j1 + c1 - dd]
    // (a task per block runs over whole spans; split into thirds of each span it was 8 k cycles shorter here and 17 k longer
    // in the row passes -- the register allocation of this kernel is at its limit)
    for (int task = tid; task < np * K1; task += NT) {
      const int j1 = task / K1, d = task - j1 * K1;
      double sxx = 0.0, sxy = 0.0, syx = 0.0, syy = 0.0;
      for (int al = d; al <= K; ++al) {
        int s = j1 - al; if (s < 0) s += np;
        const int be = al - d;
        const int oz = xy_opaque_zero();
        const double* __restrict__ p1a = D1 + ((size_t)al * N + oz); const double* __restrict__ p2a = D2 + ((size_t)al * N + oz);
        const double* __restrict__ p1b = D1 + ((size_t)be * N + oz); const double* __restrict__ p2b = D2 + ((size_t)be * N + oz);
        const int e0 = sfirst[s], e1 = sfirst[s + 1];
#pragma unroll 2
        for (int i = e0; i < e1; ++i) {
          const double2 gx = wb2[i], gy = wb2[N + i];
          const double b1a = p1a[i], b2a = p2a[i], b1b = p1b[i], b2b = p2b[i];
          const double Gax = fma(gx.x, b1a, gx.y * b2a), Gay = fma(gy.x, b1a, gy.y * b2a);
          const double Gbx = fma(gx.x, b1b, gx.y * b2b), Gby = fma(gy.x, b1b, gy.y * b2b);
          sxx = fma(Gax, Gbx, sxx); sxy = fma(Gax, Gby, sxy); syx = fma(Gay, Gbx, syx); syy = fma(Gay, Gby, syy);
        }
      }
      Pc[(2 * j1) * H1 + 2 * d] = 2.0 * sxx;
      if (d > 0) Pc[(2 * j1) * H1 + 2 * d - 1] = 2.0 * sxy;
      Pc[(2 * j1 + 1) * H1 + 2 * d + 1] = 2.0 * syx;
      Pc[(2 * j1 + 1) * H1 + 2 * d] = 2.0 * syy;
    }
    __syncthreads();
    // ---- q = 2 G'(kappa - G z): the weights become g * residual
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (ok[r]) {
        const int i = XY_SMP(r);
        double2 gx = wb2[i], gy = wb2[N + i];
        gx.x *= gres[r]; gx.y *= gres[r]; gy.x *= gres[r]; gy.y *= gres[r];
        wb2[i] = gx; wb2[N + i] = gy;
      }
    }
    __syncthreads();
    for (int task = tid; task < nch * K1; task += NT) {
      const int s = task / K1, al = task - s * K1;
      const double* __restrict__ p1 = D1 + (size_t)al * N; const double* __restrict__ p2 = D2 + (size_t)al * N;
      double a0 = 0.0, a1 = 0.0;
      const int e0 = cfirst[s], e1 = cfirst[s + 1];
#pragma unroll 4
      for (int i = e0; i < e1; ++i) {
        const double2 gx = wb2[i], gy = wb2[N + i];
        const double b1 = p1[i], b2 = p2[i];
        a0 += fma(gx.x, b1, gx.y * b2); a1 += fma(gy.x, b1, gy.y * b2);
      }
      sums[task] = a0; sums[vplane + task] = a1;
    }
    __syncthreads();
    for (int u = tid; u < nz; u += NT) qv[u] = 2.0 * gathered(u >> 1, u & 1, 0);
    for (int j = tid; j < n; j += NT) xs[j] = zs[j];
    double tr = 0.0, qinf = 0.0, dummy = 0.0, dummy2 = 0.0;
    for (int u = tid; u < nz; u += NT) tr += Pc[u * H1];
    reduce4(tr, dummy2, qinf, dummy);
    const double sc = (double)nz / tr;
    for (int q = tid; q < nz * H1; q += NT) Pc[q] = Pc[q] * sc + ((q % H1) == 0 ? 1e-9 : 0.0);
    qinf = 0.0;
    for (int u = tid; u < nz; u += NT) { qv[u] *= sc; qinf = fmax(qinf, fabs(qv[u])); }
    tr = 0.0;
    reduce4(tr, dummy2, qinf, dummy);
    // ---- interior point from x = z
    {
    double bv[K1][R];
    load_b(bv);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (ok[r]) {
        double px, py;
        dot_b(bv, r, xs, px, py);
        double lo_r, hi_r;
        lat_bounds(r, lo_r, hi_r);
#pragma unroll
        for (int kd = 0; kd < 2; ++kd) {
          const double ax = kd == 0 ? nx[r] * px + ny[r] * py : ny[r] * px - nx[r] * py;
          const double lo_ = kd == 0 ? lo_r : -a.lon, hi_ = kd == 0 ? hi_r : a.lon;
          if (outer == 0) {
            sl[kd][r] = fmax(ax - lo_, 1e-2); su[kd][r] = fmax(hi_ - ax, 1e-2); ll[kd][r] = 1.0; lu[kd][r] = 1.0;
          } else {
            sl[kd][r] = fmax(sl[kd][r], 1e-2); su[kd][r] = fmax(su[kd][r], 1e-2);
            ll[kd][r] = fmax(ll[kd][r], 1e-2); lu[kd][r] = fmax(lu[kd][r], 1e-2);
          }
          rpl[kd][r] = ax - lo_ - sl[kd][r]; rpu[kd][r] = hi_ - ax - su[kd][r];
        }
      }
    }
    }
    XY_STAMP(7)
    for (int it = 0; it < a.max_ipm; ++it) {
      // ---- P x + q; the factorisation's storage cleared
      for (int u = tid; u < nz; u += NT) {
        double s = qv[u];
#pragma unroll
        for (int dd = 0; dd <= HC; ++dd) {
          int um = u - dd; if (um < 0) um += nz;
          s = fma(Pc[u * H1 + dd], xsf[um], s);
          if (dd > 0) { int up = u + dd; if (up >= nz) up -= nz; s = fma(Pc[up * H1 + dd], xsf[up], s); }
        }
        rdP[u] = s;
      }
      for (int q = tid; q < (nz + HB + 1) * CW; q += NT) Lf[q] = 0.0;
      // ---- rows: complementarity and the three matrix weights per sample
      double mu = 0.0, rpmax = 0.0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (ok[r]) {
          double dm[2];
#pragma unroll
          for (int kd = 0; kd < 2; ++kd) {
            dm[kd] = ll[kd][r] * frcp(sl[kd][r]) + lu[kd][r] * frcp(su[kd][r]);
            mu += sl[kd][r] * ll[kd][r] + su[kd][r] * lu[kd][r];
            rpmax = fmax(rpmax, fmax(fabs(rpl[kd][r]), fabs(rpu[kd][r])));
          }
          const int i = XY_SMP(r) + xy_opaque_zero();
          const double x2 = nx[r] * nx[r], y2 = ny[r] * ny[r], xy = nx[r] * ny[r];
          wb[i] = dm[0] * x2 + dm[1] * y2;
          wb[N + i] = (dm[0] - dm[1]) * xy;
          wb[2 * N + i] = dm[0] * y2 + dm[1] * x2;
        }
      }
      {
        double z1 = 0.0, mn = 0.0;
        reduce4(mu, z1, rpmax, mn);          // (barriers: weights, rdP and the cleared storage are visible)
      }
      const double mu_sum = mu;
      mu /= (double)(4 * N);
      XY_STAMP(1)
      // ---- span sums of  b_a b_b M
      xy_span_sums<K, 0>(bbx, N, nch, cfirst, wb, sums, tid, NT);
      __syncthreads();
      XY_STAMP(2)
      // ---- 2x2 blocks of  P + A'DA  into the folded band
      for (int task = tid; task < np * K1; task += NT) {
        const int j1 = task / K1, d = task - j1 * K1;
        double axx = 0.0, axy = 0.0, ayy = 0.0;
#pragma unroll
        for (int al = 0; al <= K; ++al) {
          if (al >= d) {
            int s = j1 - al; if (s < 0) s += np;
            for (int ch = sch0[s]; ch < sch0[s + 1]; ++ch) {
              const double* sp = sums + (ch * NE + al * (al + 1) / 2 + (al - d)) * 3;
              axx += sp[0]; axy += sp[1]; ayy += sp[2];
            }
          }
        }
        int j2 = j1 - d; if (j2 < 0) j2 += np;
        auto store = [&](int ui, int uj, double v) {
          const int fi = xy_fold(ui, nz), fj = xy_fold(uj, nz);
          *xy_entry<HB>(Lf, max(fi, fj), min(fi, fj)) = v;
        };
        store(2 * j1, 2 * j2, axx + Pc[(2 * j1) * H1 + 2 * d]);
        if (d > 0) store(2 * j1, 2 * j2 + 1, axy + Pc[(2 * j1) * H1 + 2 * d - 1]);
        store(2 * j1 + 1, 2 * j2, axy + Pc[(2 * j1 + 1) * H1 + 2 * d + 1]);
        store(2 * j1 + 1, 2 * j2 + 1, ayy + Pc[(2 * j1 + 1) * H1 + 2 * d]);
      }
      // ---- weights of A'e and A'(lu - ll) (the matrix weights are no longer needed: barrier above)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (ok[r]) {
          double e[2], dl[2];
#pragma unroll
          for (int kd = 0; kd < 2; ++kd) {
            const double ql = ll[kd][r] * frcp(sl[kd][r]), qu = lu[kd][r] * frcp(su[kd][r]);
            e[kd] = qu * rpu[kd][r] - ql * rpl[kd][r]; dl[kd] = lu[kd][r] - ll[kd][r];
          }
          const int i = XY_SMP(r) + xy_opaque_zero();
          wb2[i] = make_double2(e[0] * nx[r] + e[1] * ny[r], e[0] * ny[r] - e[1] * nx[r]);
          wb2[N + i] = make_double2(dl[0] * nx[r] + dl[1] * ny[r], dl[0] * ny[r] - dl[1] * nx[r]);
        }
      }
      __syncthreads();
      XY_STAMP(3)
      // ---- wave 0 factorises while the others form the span sums of the two vectors
      if (wave == 0) fband_factor<HB>(Lf, dinv, nz, lane);
      else xy_span_sums<K, 1>(bbx, N, nch, cfirst, wb, sums, tid - kWave, NT - kWave);
      __syncthreads();
      XY_STAMP(4)
      double rdmax = 0.0;
      for (int u = tid; u < nz; u += NT) {
        const double s1 = gathered(u >> 1, u & 1, 0), s2 = gathered(u >> 1, u & 1, 1);
        const double rr = rdP[u] + s2;
        rd[u] = rr;
        rhs[u] = s1 - rdP[u];
        rdmax = fmax(rdmax, fabs(rr));
      }
      {
        double d0 = 0.0, d1 = 0.0, mn = 0.0;
        reduce4(d0, d1, rdmax, mn);
      }
      {
        // the exit rule (rl_device.hpp: ipm_done): the complementarity alone, 1e-9 before the last linearisation, 1e-10 on it (going
        // on past it only drives mu down -- 1e-156 was seen -- and the residual UP with the normal matrix's conditioning)
        (void)rdmax; (void)rpmax; (void)qinf;
        const bool done = ipm_done(kIpmTolTwoCoords, outer + 1 >= a.n_outer, mu);
        if (done) break;
      }
      ++total_it;
      // ---- affine direction
      if (wave == 0) { if (nz <= 2 * kWave) fband_solve<HB, 2>(Lf, dinv, nz, lane, rhs, dxf); else fband_solve<HB, 3>(Lf, dinv, nz, lane, rhs, dxf); }
      __syncthreads();
      wrap(dxs);
      __syncthreads();
      XY_STAMP(5)
      // one row of the step equations: the affine step from A dx_aff (smu = 0, no second-order term), or the final step
      // from A dx with the second-order term of the affine one.  Everything is recomputed from the few numbers kept per row
      // (A dx_aff, A dx): keeping the deltas themselves would spill.
      // A dx_aff lives in the upper half of the weight buffer (free after the vector sums), A dx in the lower half (free after
      // the corrector's sums): 32 registers less across the row passes
      double* adxa_l = wb + 2 * N;
      double* adxc_l = wb;
      auto affine_row = [&](int kd, int r, double i_sl, double i_su, double& d_sl, double& d_su, double& d_ll, double& d_lu) {
        const double adx = adxa_l[kd * N + XY_SMP(r) + xy_opaque_zero()];
        d_sl = adx + rpl[kd][r]; d_su = rpu[kd][r] - adx;
        d_ll = (-(sl[kd][r] * ll[kd][r]) - ll[kd][r] * d_sl) * i_sl;
        d_lu = (-(su[kd][r] * lu[kd][r]) - lu[kd][r] * d_su) * i_su;
      };
      double c1 = 0.0, c2 = 0.0, rmax = 0.0;
      {
        double bv[K1][R];
        load_b(bv);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (ok[r]) {
            double px, py;
            dot_b(bv, r, dxs, px, py);
#pragma unroll
            for (int kd = 0; kd < 2; ++kd) {
              const double adx = kd == 0 ? nx[r] * px + ny[r] * py : ny[r] * px - nx[r] * py;
              adxa_l[kd * N + XY_SMP(r)] = adx;
              const double i_sl = frcp(sl[kd][r]), i_su = frcp(su[kd][r]);
              const double d_sl = adx + rpl[kd][r], d_su = rpu[kd][r] - adx;
              const double d_ll = (-(sl[kd][r] * ll[kd][r]) - ll[kd][r] * d_sl) * i_sl;
              const double d_lu = (-(su[kd][r] * lu[kd][r]) - lu[kd][r] * d_su) * i_su;
              rmax = fmax(rmax, fmax(fmax(-d_sl * i_sl, -d_su * i_su),
                                     fmax(-d_ll * __builtin_amdgcn_rcp(ll[kd][r]), -d_lu * __builtin_amdgcn_rcp(lu[kd][r]))));
              c1 += sl[kd][r] * d_ll + ll[kd][r] * d_sl + su[kd][r] * d_lu + lu[kd][r] * d_su;
              c2 += d_sl * d_ll + d_su * d_lu;
            }
          }
        }
      }
      {
        double mn = 0.0;
        reduce4(c1, c2, rmax, mn);
      }
      XY_STAMP(6)
      // step to the boundary: alpha = min(1, 0.995 / max(-delta / value))
      const double a_aff = rmax > 0.995 ? 0.995 / rmax : 1.0;
      const double mu_aff = (mu_sum + a_aff * c1 + a_aff * a_aff * c2) / (double)(4 * N);
      const double ratio = mu_aff / mu, sigma = ratio * ratio * ratio;
      const double smu = sigma * mu;
      // ---- corrector right-hand side
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (ok[r]) {
          double wv[2];
#pragma unroll
          for (int kd = 0; kd < 2; ++kd) {
            const double i_sl = frcp(sl[kd][r]), i_su = frcp(su[kd][r]);
            double d_sl, d_su, d_ll, d_lu;
            affine_row(kd, r, i_sl, i_su, d_sl, d_su, d_ll, d_lu);
            const double rcl = sl[kd][r] * ll[kd][r] - smu + d_sl * d_ll, rcu = su[kd][r] * lu[kd][r] - smu + d_su * d_lu;
            wv[kd] = (-rcl - ll[kd][r] * rpl[kd][r]) * i_sl - (-rcu - lu[kd][r] * rpu[kd][r]) * i_su;
          }
          wb2[XY_SMP(r) + xy_opaque_zero()] = make_double2(wv[0] * nx[r] + wv[1] * ny[r], wv[0] * ny[r] - wv[1] * nx[r]);
        }
      }
      __syncthreads();
      XY_STAMP(11)
      xy_span_sums<K, 2>(bbx, N, nch, cfirst, wb, sums, tid, NT);
      __syncthreads();
      for (int u = tid; u < nz; u += NT) rhs[u] = gathered(u >> 1, u & 1, 0) - rd[u];
      __syncthreads();
      XY_STAMP(8)
      if (wave == 0) { if (nz <= 2 * kWave) fband_solve<HB, 2>(Lf, dinv, nz, lane, rhs, dxf); else fband_solve<HB, 3>(Lf, dinv, nz, lane, rhs, dxf); }
      __syncthreads();
      wrap(dxs);
      __syncthreads();
      XY_STAMP(10)
      // ---- step length and update
      auto final_row = [&](int kd, int r, double& d_sl, double& d_su, double& d_ll, double& d_lu, double& i_sl, double& i_su) {
        i_sl = frcp(sl[kd][r]); i_su = frcp(su[kd][r]);
        double a_sl, a_su, a_ll, a_lu;
        affine_row(kd, r, i_sl, i_su, a_sl, a_su, a_ll, a_lu);
        const double rcl = sl[kd][r] * ll[kd][r] - smu + a_sl * a_ll, rcu = su[kd][r] * lu[kd][r] - smu + a_su * a_lu;
        const double adx = adxc_l[kd * N + XY_SMP(r) + xy_opaque_zero()];
        d_sl = adx + rpl[kd][r]; d_su = rpu[kd][r] - adx;
        d_ll = (-rcl - ll[kd][r] * d_sl) * i_sl; d_lu = (-rcu - lu[kd][r] * d_su) * i_su;
      };
      rmax = 0.0;
      {
        double bv[K1][R];
        load_b(bv);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (ok[r]) {
            double px, py;
            dot_b(bv, r, dxs, px, py);
#pragma unroll
            for (int kd = 0; kd < 2; ++kd) {
              adxc_l[kd * N + XY_SMP(r)] = kd == 0 ? nx[r] * px + ny[r] * py : ny[r] * px - nx[r] * py;
              double d_sl, d_su, d_ll, d_lu, i_sl, i_su;
              final_row(kd, r, d_sl, d_su, d_ll, d_lu, i_sl, i_su);
              rmax = fmax(rmax, fmax(fmax(-d_sl * i_sl, -d_su * i_su),
                                     fmax(-d_ll * __builtin_amdgcn_rcp(ll[kd][r]), -d_lu * __builtin_amdgcn_rcp(lu[kd][r]))));
            }
          }
        }
      }
      {
        double z0 = 0.0, z1 = 0.0, mn = 0.0;
        reduce4(z0, z1, rmax, mn);
      }
      const double alpha = rmax > 0.995 ? 0.995 / rmax : 1.0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (ok[r]) {
#pragma unroll
          for (int kd = 0; kd < 2; ++kd) {
            double d_sl, d_su, d_ll, d_lu, i_sl, i_su;
            final_row(kd, r, d_sl, d_su, d_ll, d_lu, i_sl, i_su);
            sl[kd][r] = fma(alpha, d_sl, sl[kd][r]);
            su[kd][r] = fma(alpha, d_su, su[kd][r]);
            ll[kd][r] = fma(alpha, d_ll, ll[kd][r]);
            lu[kd][r] = fma(alpha, d_lu, lu[kd][r]);
            rpl[kd][r] *= 1.0 - alpha; rpu[kd][r] *= 1.0 - alpha;
          }
        }
      }
      for (int j = tid; j < n; j += NT) xs[j] = make_double2(fma(alpha, dxs[j].x, xs[j].x), fma(alpha, dxs[j].y, xs[j].y));
      __syncthreads();
      XY_STAMP(9)
    }
    // ---- whole step or half step: the smaller sum kappa^2 (ties: the whole step)
    double cw = 0.0, ch = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (ok[r]) {
        const int oz = xy_opaque_zero();
        const int i = XY_SMP(r) + oz;
        double dx = 0, dy = 0, ddx = 0, ddy = 0, sdx = 0, sdy = 0, sddx = 0, sddy = 0;
#pragma unroll
        for (int al = 0; al < K1; ++al) {
          const double b1 = D1[(size_t)al * N + i], b2 = D2[(size_t)al * N + i];
          const double2 c = cs[j0[r] + al + oz], x = xs[j0[r] + al + oz], z = zs[j0[r] + al + oz];
          const double ex = x.x - z.x, ey = x.y - z.y;
          dx = fma(c.x, b1, dx); dy = fma(c.y, b1, dy); ddx = fma(c.x, b2, ddx); ddy = fma(c.y, b2, ddy);
          sdx = fma(ex, b1, sdx); sdy = fma(ey, b1, sdy); sddx = fma(ex, b2, sddx); sddy = fma(ey, b2, sddy);
        }
        {
          const double ax = dx + sdx, ay = dy + sdy, bx = ddx + sddx, by = ddy + sddy;
          const double s2 = ax * ax + ay * ay, kp = (ax * by - ay * bx) / (s2 * sqrt(s2));
          cw = fma(kp, kp, cw);
        }
        {
          const double ax = fma(0.5, sdx, dx), ay = fma(0.5, sdy, dy), bx = fma(0.5, sddx, ddx), by = fma(0.5, sddy, ddy);
          const double s2 = ax * ax + ay * ay, kp = (ax * by - ay * bx) / (s2 * sqrt(s2));
          ch = fma(kp, kp, ch);
        }
      }
    }
    {
      double mx = 0.0, mn = 0.0;
      reduce4(cw, ch, mx, mn);
    }
    const double tau = ch < cw ? 0.5 : 1.0;
    if (tau != 1.0) ++n_halved;
    double stepmax = 0.0;
    for (int u = tid; u < nz; u += NT) {
      const double zn = fma(tau, xsf[u] - zsf[u], zsf[u]);
      stepmax = fmax(stepmax, fabs(zn - zsf[u]));
      zsf[u] = zn;
    }
    {
      double z0 = 0.0, z1 = 0.0, mn = 0.0;
      reduce4(z0, z1, stepmax, mn);
    }
    last_step = stepmax;
    wrap(zs);
    __syncthreads();
  }

  // ---- outputs: control points, line samples, offsets, statistics
  for (int j = tid; j < n; j += NT) reinterpret_cast<double2*>(a.out_ctrl)[(size_t)b * n + j] = cs[j];
  if (a.out_z)
    for (int j = tid; j < np; j += NT) reinterpret_cast<double2*>(a.out_z)[(size_t)b * np + j] = zs[j];
  double viol = -INFINITY, nlat = 0.0, nlon = 0.0;
  double bvo[K1][R];
  load_b(bvo);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (ok[r]) {
      const int i = XY_SMP(r);
      double x, y;
      dot_b(bvo, r, cs, x, y);
      if (a.out_xy) reinterpret_cast<double2*>(a.out_xy)[(size_t)b * N + i] = make_double2(x, y);
      const double ex = x - a.trk.base[i], ey = y - a.trk.base[(size_t)N + i];
      const double lat = ex * nx[r] + ey * ny[r], lg = ex * ny[r] - ey * nx[r];
      double lo_r, hi_r;
      lat_bounds(r, lo_r, hi_r);
      viol = fmax(viol, fmax(fmax(lo_r - lat, lat - hi_r), fabs(lg) - a.lon));
      if (lat - lo_r < 1e-6 || hi_r - lat < 1e-6) nlat += 1.0;
      if (a.lon - fabs(lg) < 1e-6) nlon += 1.0;
    }
  }
  {
    double mn = 0.0;
    reduce4(nlat, nlon, viol, mn);
  }
  if (tid == 0) {
    double* st = a.out_stats + (size_t)b * 8;
    st[0] = (double)total_it; st[1] = k2_first; st[2] = k2_last; st[3] = fmax(viol, 0.0);
    st[4] = last_step; st[5] = nlat; st[6] = nlon; st[7] = (double)n_halved;
#ifdef RL_XY_PROFILE
    if (a.out_z) for (int q = 0; q < 12; ++q) a.out_z[(size_t)b * np * 2 + q] = (double)pt[q];
#endif
  }
#undef XY_SMP
#undef XY_STAMP
}

}  // namespace rl
