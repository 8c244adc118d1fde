// rl_global2.hpp -- a15, fast path: the global min-curvature QP of rl_global.hpp (same formulation,
// same CPU twin oracle/mincurv_oracle.c: orc_global_mincurv) as a WAVE-SPECIALISED workgroup.
//
//   * waves 0 .. nw-2 ("row waves"): thread = one chunk of <= R consecutive samples of ONE knot span.
//     The chunk's constraint rows A (instance independent) and its interior-point state (slacks, duals)
//     stay in registers for the whole solve; bounds lo/hi are in LDS.
//   * wave nw-1 ("linear-algebra wave"): owns the np x np normal matrix.  The cyclic band
//     (half-bandwidth K) is FOLDED (unknown order 0, np-1, 1, np-2, ...) into a plain band of
//     half-bandwidth 2K, stored one matrix row per lane, both triangles, column c in register slot
//     c mod (4K+1).  L D L' and both triangular solves then run entirely in registers: the pivot
//     column is broadcast with v_readlane, every update is lane-local, no LDS, no barrier inside.
//   The two roles are two separate code paths (register allocation = max, not sum) that meet at the
//   same sequence of workgroup barriers; the barrier protocol is spelled out at k_global_qp2.
//
//   * chunk partials -> span sums go through an LDS staging area in rounds (fixed summation order:
//     run-to-run reproducible, independent of the position in the batch).
//
// Shapes the fast path does not cover (more chunks than 7 waves, LDS overflow) use k_global_qp.
#pragma once
#include "rl_global.hpp"

namespace rl {

// keeps the scheduler from interleaving the unrolled per-row bodies (register pressure)
#define RL_ROW_FENCE() __builtin_amdgcn_sched_barrier(0)

// span outputs staged per flush round: 17 (two rounds for the 33 outputs) where the LDS allows it,
// 12 with three groups of matrix rows (the folded matrix is three times as large there)
__host__ __device__ constexpr int g2_round(int groups) { return groups == 1 ? 17 : 12; }
constexpr int kG2Block = 512;  // 7 row waves + 1 linear-algebra wave; 256 VGPRs per lane

struct Global2Layout {  // offsets in doubles
  int xs, xs2, dxa, dxs, avs, qv, rdP, rd, rhs, cxs, cys, nus, Pc, band, itab, Ssum, Spart, lohi, red, ctl, sch, total;
};

__host__ __device__ inline Global2Layout global2_layout(int k, int n, int np, int N, int groups, int nrow) {
  const int K1 = k + 1, NO = K1 * (K1 + 1) / 2 + 2 * K1, NS = 4 * k + 1;
  Global2Layout L;
  int o = 0;
  auto take = [&](int c) { const int r = o; o += (c + 1) & ~1; return r; };
  L.xs = take(np); L.xs2 = take(np); L.dxa = take(np); L.dxs = take(np); L.avs = take(np); L.qv = take(np); L.rdP = take(np);
  L.rd = take(np); L.rhs = take(np);
  L.cxs = take(n); L.cys = take(n); L.nus = take(2 * np);
  L.Pc = take(np * K1);
  L.band = take(np * K1);   // the normal matrix as a cyclic band, band[j][d] = M[j][j-d]: the np (K+1) DISTINCT values of it
  L.itab = take(groups == 1 ? (NS * 64 + NS * 32 + 3) / 4 : 0);   // int16 index of every image entry in the band (TwoFront)
  L.Ssum = take(np * NO);
  // the folded matrix aliases the staging area; one group: the two-front image [NS][64] + the middle block [NS][32]
  const int spart = nrow * g2_round(groups), mf = groups == 1 ? NS * 64 + NS * 32 + 128 : NS * 64 * groups;  // + 128: broadcast scratch
  L.Spart = take(spart > mf ? spart : mf);
  L.lohi = take(2 * N);
  L.red = take(4 * 16);
  L.ctl = take(8);
  L.sch = take((np + 2) / 2 + 1);
  L.total = o;
  return L;
}

// 1/x to double precision without the IEEE division sequence (x is a positive, normal number here)
__device__ __forceinline__ double frcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// ------------------------------------------------------------------------------------------------
// Folded band in registers.  Matrix row p lives in lane p & 63 of group p >> 6; slot s of row p holds
// column  col(p, s) = the one column == s (mod NS) in [p - BW, p + BW].  Rows >= np are zero.
//
// Every step is templated on the group Q of the pivot column so that all register indices are
// static; the rows a column touches lie in group Q and, where the window crosses a multiple of 64,
// in the neighbouring group (same lanes, wrapped).  No branch inside a column step.
template <int BW, int G>
struct FoldBand {
  static constexpr int NS = 2 * BW + 1;
  double S[G][NS];
  double dinv[G];

  __device__ __forceinline__ void load(const double* Mf, int lane) {
#pragma unroll
    for (int q = 0; q < G; ++q) {
      dinv[q] = 0.0;
#pragma unroll
      for (int s = 0; s < NS; ++s) S[q][s] = Mf[s * (64 * G) + 64 * q + lane];
    }
  }

  // rows (c, c + BW] <-> lanes of group Q (and Q + 1); rows [c - BW, c) <-> group Q (and Q - 1)
  template <int Q>
  static __device__ __forceinline__ bool below(int lane, int c, int dq) { return (unsigned)(lane + 64 * (Q + dq) - c - 1) < (unsigned)BW; }
  template <int Q>
  static __device__ __forceinline__ bool above(int lane, int c, int dq) { return (unsigned)(c - 1 - (lane + 64 * (Q + dq))) < (unsigned)BW; }

  // One column of L D L'.  Afterwards slot s holds, in the rows below the pivot, L[r][c]; in the rows
  // above it, the mirrored entry times 1/d_c (what the backward sweep multiplies with); dinv = 1/d.
  template <int Q>
  __device__ __forceinline__ void factor_col(int c, int s, int lane) {
    constexpr bool NEXT = Q + 1 < G, PREV = Q > 0;
    const int lc = c & 63;
    const double inv = frcp(lane_bcast(S[Q][s], lc));
    const bool m0 = below<Q>(lane, c, 0), m1 = NEXT && below<Q>(lane, c, 1);
    const double uv0 = m0 ? S[Q][s] : 0.0;
    const double uv1 = m1 ? S[NEXT ? Q + 1 : Q][s] : 0.0;
    const double uvc = m1 ? uv1 : uv0;  // the pivot column in lane space (the two windows are disjoint)
    dinv[Q] = lane == lc ? inv : dinv[Q];
    S[Q][s] = m0 ? uv0 * inv : (above<Q>(lane, c, 0) ? S[Q][s] * inv : S[Q][s]);
    if (NEXT) S[Q + 1][s] = m1 ? uv1 * inv : S[Q + 1][s];
    if (PREV) S[Q - 1][s] = above<Q>(lane, c, -1) ? S[Q - 1][s] * inv : S[Q - 1][s];
    // all broadcasts first (independent v_readlane pairs), then the independent updates: a single
    // wave issues in order, so interleaving them would expose the readlane -> VALU latency BW times.
    // (uv * u) * inv rounds identically in both triangles: the matrix stays bitwise symmetric.
    double u[BW];
#pragma unroll
    for (int j = 1; j <= BW; ++j) u[j - 1] = lane_bcast(uvc, (c + j) & 63);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 1; j <= BW; ++j) {
      S[Q][(s + j) % NS] = fma(-(uv0 * u[j - 1]), inv, S[Q][(s + j) % NS]);
      if (NEXT) S[Q + 1][(s + j) % NS] = fma(-(uv1 * u[j - 1]), inv, S[Q + 1][(s + j) % NS]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  template <int Q>
  __device__ __forceinline__ void forward_col(int c, int s, int lane, double (&b)[G]) const {
    constexpr bool NEXT = Q + 1 < G;
    const double bc = lane_bcast(b[Q], c & 63);
    b[Q] = fma(-(below<Q>(lane, c, 0) ? S[Q][s] : 0.0), bc, b[Q]);
    if (NEXT) b[Q + 1] = fma(-(below<Q>(lane, c, 1) ? S[Q + 1][s] : 0.0), bc, b[Q + 1]);
  }
  template <int Q>
  __device__ __forceinline__ void backward_col(int c, int s, int lane, double (&b)[G]) const {
    constexpr bool PREV = Q > 0;
    const double tc = lane_bcast(b[Q], c & 63);
    b[Q] = fma(-(above<Q>(lane, c, 0) ? S[Q][s] : 0.0), tc, b[Q]);
    if (PREV) b[Q - 1] = fma(-(above<Q>(lane, c, -1) ? S[Q - 1][s] : 0.0), tc, b[Q - 1]);
  }

#define RL_FB_DISPATCH(CALL)                                   \
  if (G == 1) { CALL(0); }                                     \
  else { const int qc = c >> 6;                                \
         if (qc == 0) { CALL(0); } else if (qc == 1) { CALL(G > 1 ? 1 : 0); } else { CALL(G > 2 ? 2 : 0); } }

  __device__ __forceinline__ void factor(int np, int lane) {
    static_assert(G <= 3, "three groups of 64 rows at most");
    for (int c0 = 0; c0 < np; c0 += NS) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int c = c0 + s;
        if (c >= np) break;
#define RL_FB_CALL(Q) factor_col<(Q)>(c, s, lane)
        RL_FB_DISPATCH(RL_FB_CALL)
#undef RL_FB_CALL
      }
    }
  }

  // (L D L') x = b, b in registers (row = lane + 64 q), overwritten by x
  __device__ __forceinline__ void solve(int np, int lane, double (&b)[G]) const {
    for (int c0 = 0; c0 < np; c0 += NS) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int c = c0 + s;
        if (c >= np) break;
#define RL_FB_CALL(Q) forward_col<(Q)>(c, s, lane, b)
        RL_FB_DISPATCH(RL_FB_CALL)
#undef RL_FB_CALL
      }
    }
    for (int c0 = ((np - 1) / NS) * NS; c0 >= 0; c0 -= NS) {
#pragma unroll
      for (int s = NS - 1; s >= 0; --s) {
        const int c = c0 + s;
        if (c < np) {
#define RL_FB_CALL(Q) backward_col<(Q)>(c, s, lane, b)
          RL_FB_DISPATCH(RL_FB_CALL)
#undef RL_FB_CALL
        }
      }
    }
#pragma unroll
    for (int q = 0; q < G; ++q) b[q] *= dinv[q];
  }
#undef RL_FB_DISPATCH
};

__device__ __forceinline__ int fold_pos(int j, int np) { return j < (np + 1) / 2 ? 2 * j : 2 * (np - 1 - j) + 1; }
__device__ __forceinline__ int fold_inv(int p, int np) { return (p & 1) ? np - 1 - (p >> 1) : (p >> 1); }

// ------------------------------------------------------------------------------------------------
// np <= 64: TWO ELIMINATION FRONTS IN ONE WAVE.  The folded band is a plain band (half-bandwidth BW) of np rows; a
// column step only touches the BW rows below its pivot, so two fronts that start at the two ends of the band -- columns
// 0, 1, 2, ... and columns np-1, np-2, ... -- are independent until they come within BW rows of each other.  One wave runs
// both with ONE instruction stream: lanes 0..31 hold rows 0..31 of the matrix, lanes 32..63 rows 0..31 of the
// index-REVERSED matrix (row p~ = np-1-p, column q~ = np-1-q), both in the slot layout of FoldBand (slot = column mod NS), so
// step t of either front reads and writes the same register slots; what differs between the halves is only WHICH lane a
// value is fetched from (lane t + j or 32 + t + j: broadcast reads from a 1 KB LDS scratch instead of v_readlane).  After
// T = (np - 2 BW) / 2 steps each front has reached the MIDDLE block of np - 2T <= 2 BW + 1 rows: its Schur complement is
// gathered through LDS into a dense image (lane = row, slot = column) and eliminated by the same wave.  The dependent
// chain of a factorisation is T + (np - 2T) = 41 column steps instead of np = 61 (Monza, s = 100), of a solve 92 instead
// of 122, and no second wave, no extra workgroup barrier and no row wave is given up for it.
//   measured   (Monza N = 2000, 1024 instances, per interior-point iteration, -DRL_G2_PROFILE): factorisation 11.2 us ->
//              8.9 us INCLUDING the forward substitution of the first right-hand side, the remaining solve work 6.6 -> 5.7 us;
//              kernel 14.5 -> 14.3 ms.  The steps were halved-ish, the time was not: a wave that is alone on its SIMD issues
//              one FP64 VALU / LDS instruction per ~8 cycles whatever it depends on (timing builds without the Newton steps
//              of 1/sqrt, or without any cross-lane traffic at all, are 6 % / 17 % faster only), so the linear algebra of an
//              instance costs (instructions) x 8 cycles on 1/8 of the CU's issue slots while the row waves wait.  What would
//              change that is a second INSTANCE in the workgroup whose row passes fill those slots (DESIGN_HISTORY.md section 9).
//   ordering   P M P' = L D L' with P = (top columns, bottom columns, middle): an ordinary symmetric permutation, so
//              the factorisation is as stable as the one-front one; rounding differs (another elimination order).
//   storage    as FoldBand: below the pivot L[r][c]; above it the mirrored entry times 1/d_c for eliminated columns,
//              unscaled (d_r L[c][r]) for the middle columns, which the fronts never reach.
template <int BW>
struct TwoFront {
  static constexpr int NS = 2 * BW + 1, MID = 2 * BW + 1;
  double S[NS];    // fronts image
  double Sb[NS];   // after the factorisation: the bottom front's rows once more, in lanes 0..31 (for the solves)
  double Sm[MID];  // middle image: lane i < mid holds row T + i of the Schur complement, slot j = column T + j
  double dinv, dinvb, dinvm;
  double yp, ym;   // L^-1 (first right-hand side): fronts (lane packed) and middle block, formed DURING the factorisation

  static __host__ __device__ __forceinline__ int fronts(int np) { return np > 2 * BW ? (np - 2 * BW) / 2 : 0; }

  static __device__ __forceinline__ double perm(double v, int src_lane) {  // per-lane source: v of lane src_lane
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
  }
  // value of lane t for the top half, of lane 32 + t for the bottom half (t wave-uniform): two scalar broadcasts and a
  // select -- a handful of cycles, where ds_bpermute would put its ~100 cycles on the dependent chain of the pivots
  static __device__ __forceinline__ double half_bcast(double v, int t, bool bottom) {
    const double a = lane_bcast(v, t), b = lane_bcast(v, 32 + t);
    return bottom ? b : a;
  }
  // lanes 32..63 of v in lanes 0..31 (v_permlane32_swap: a half exchange, no LDS)
  static __device__ __forceinline__ double upper_half(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)r1[1], (int)r0[1]);
  }
  // column (front orientation) that slot s of row i holds: the one == s (mod NS) in [i - BW, i + BW]
  static __device__ __forceinline__ int slot_col(int i, int s) {
    int m = (s - i + BW) % NS;
    if (m < 0) m += NS;
    return i - BW + m;
  }
  // 1 / sqrt(x) to double precision (x positive, normal): v_rsq_f64 + two Newton steps, no division
  static __device__ __forceinline__ double frsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = fma(fma(-hx * y, y, 0.5), y, y);
    y = fma(fma(-hx * y, y, 0.5), y, y);
    return y;
  }

  static __device__ __forceinline__ int band_index(int p, int col, int np, int K) {   // matrix (folded) row / column -> band[]
    const int j1 = fold_inv(p, np), j2 = fold_inv(col, np);
    int d = j1 - j2;
    if (d < 0) d += np;
    if (d <= K) return j1 * (K + 1) + d;
    if (np - d <= K) return j2 * (K + 1) + (np - d);
    return -1;
  }
  // Where every entry of the two register images comes from in the cyclic band (instance independent; built once per
  // kernel into LDS by all threads): >= 0 index into band[], -1 zero, -2 one (identity padding of the middle block).
  // Entry order = the images' order: fronts [NS][64 lanes], then middle block [MID][32].
  static __device__ __forceinline__ void build_index_table(int tid, int nt, int np, short* itab) {
    constexpr int K = BW / 2;
    const int T = fronts(np), mid = np - 2 * T;
    for (int task = tid; task < NS * 64 + MID * 32; task += nt) {
      int ix = -1;
      if (task < NS * 64) {
        const int s = task >> 6, l = task & 63, i = l & 31;
        const bool bottom = l >= 32;
        if (T > 0 && i < T + BW) {
          const int q = slot_col(i, s);
          if (q >= 0 && q < T + BW) ix = band_index(bottom ? np - 1 - i : i, bottom ? np - 1 - q : q, np, K);
        }
      } else {
        const int m = task - NS * 64, j = m >> 5, i = m & 31;
        if (i < mid && j < mid) ix = band_index(T + i, T + j, np, K);
        else if (i == j && i < MID) ix = -2;
      }
      itab[task] = (short)ix;
    }
  }
  // the images from the band (all threads; after the band is complete)
  static __device__ __forceinline__ void fill_images(int tid, int nt, const short* itab, const double* band, double* Mf) {
    for (int task = tid; task < NS * 64 + MID * 32; task += nt) {
      const int ix = itab[task];
      const double v = band[ix < 0 ? 0 : ix];
      Mf[task] = ix >= 0 ? v : (ix == -2 ? 1.0 : 0.0);
    }
  }
  __device__ __forceinline__ void load(const double* Mf, int lane) {
#pragma unroll
    for (int s = 0; s < NS; ++s) S[s] = Mf[s * 64 + lane];
    dinv = 0.0; dinvm = 0.0;
  }

  // One column of both fronts.  The update is z z' with z = (pivot column) / sqrt(d): ONE fma per entry, and bitwise
  // symmetric because z_r z_q is the same product in both triangles.  What is STORED is the L D L' form of FoldBand (the
  // solves are unchanged): below the pivot z / sqrt(d) = L[r][c], above it the mirrored entry times 1/d, dinv = 1/d.
  // The pivots form the one dependent chain of the factorisation (d_t -> 1/sqrt -> column t -> d_t+1); everything else hangs
  // off it.  So the NEXT pivot is taken out of the update loop: lane t+1 holds both its diagonal and its entry of column t,
  // d_t+1 = S[t+1][t+1] - z_t+1^2 is one lane-local fma, and its 1/sqrt is under way while the BW row updates of column t
  // wait for their LDS operands.  rs = 1/sqrt(d_t) comes in, 1/sqrt(d_t+1) goes out.
  // s = t mod NS: a constant in every caller's unrolled loop, so that all register indices are static.
  __device__ __forceinline__ double front_col(int s, int t, int i, bool bottom, int half, double rs, double* zb, int lane) {
    const double inv = rs * rs;
    const bool below = (unsigned)(i - t - 1) < (unsigned)BW, above = (unsigned)(t - 1 - i) < (unsigned)BW;
    const double z = below ? S[s] * rs : 0.0;
    const double rs_next = frsqrt(half_bcast(fma(-z, z, S[(s + 1) % NS]), t + 1, bottom));
    // The BW rows below the pivot need z of the BW lanes after the pivot's, per half: through LDS -- every lane stores its z
    // (and its y, for the forward substitution that rides along), then reads the values it needs at addresses that are the
    // same for a whole half (broadcast reads, two per instruction): 7 LDS instructions where per-lane ds_bpermute takes 22,
    // and a lone wave pays ~8 cycles for each
    double u[BW];
    zb[lane] = z; zb[64 + lane] = yp;
    wave_lds_fence();
    const double* zr = zb + half + t;
#pragma unroll
    for (int j = 1; j <= BW; ++j) u[j - 1] = zr[j];
    const double yt = zr[64];
    const double w = z * rs;                     // L[r][t]
    dinv = i == t ? inv : dinv;
    S[s] = below ? w : (above ? S[s] * inv : S[s]);
#pragma unroll
    for (int j = 1; j <= BW; ++j) S[(s + j) % NS] = fma(-z, u[j - 1], S[(s + j) % NS]);
    yp = fma(-w, yt, yp);
    return rs_next;
  }

  template <int c>
  __device__ __forceinline__ double mid_col(int lane, double rs) {
    const double inv = rs * rs;
    const bool below = lane > c && lane < MID;
    const double z = below ? Sm[c] * rs : 0.0;
    double rs_next = 0.0;
    if constexpr (c + 1 < MID) rs_next = frsqrt(lane_bcast(fma(-z, z, Sm[c + 1]), c + 1));
    const double w = z * rs;
    const double yc = lane_bcast(ym, c);
    dinvm = lane == c ? inv : dinvm;
    Sm[c] = below ? w : (lane < c ? Sm[c] * inv : Sm[c]);
    if constexpr (c + 1 < MID) {
#pragma unroll
      for (int j = c + 1; j < MID; ++j) Sm[j] = fma(-z, lane_bcast(z, j), Sm[j]);
    }
    ym = fma(-w, yc, ym);
    return rs_next;
  }
  template <int c>
  __device__ __forceinline__ void mid_cols(int lane, double rs) {
    const double rs_next = mid_col<c>(lane, rs);
    if constexpr (c + 1 < MID) mid_cols<c + 1>(lane, rs_next);
  }

  // right-hand side of the middle block: what the fronts left in its first / last BW rows, the original elsewhere
  static __device__ __forceinline__ double mid_rhs(int np, int T, int mid, int lane, double from_top_v, double from_bot_v,
                                                   const double* rhs) {
    const int pm = T + lane;                              // matrix row of middle lane `lane`
    const bool from_top = T > 0 && lane < BW, from_bot = T > 0 && lane < mid && pm >= np - T - BW;
    return lane < mid ? (from_top ? from_top_v : (from_bot ? from_bot_v : rhs[fold_inv(pm, np)])) : 0.0;
  }

  // Factorisation + L^-1 rhs.  Mm: the middle block as assembled (g2_assemble_tf), [slot j][32 lanes]; the blocks the
  // fronts have updated are written over it here, then the whole block is read back one row per lane.
  __device__ __forceinline__ void factor(int np, int lane, double* Mm, const double* rhs) {
    const int T = fronts(np), mid = np - 2 * T, i = lane & 31, half = lane & 32;
    const bool bottom = half != 0;
    yp = (T > 0 && i < T + BW) ? rhs[fold_inv(bottom ? np - 1 - i : i, np)] : 0.0;
    double rs = T > 0 ? frsqrt(half_bcast(S[0], 0, bottom)) : 0.0;
    for (int t0 = 0; t0 < T; t0 += NS) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int t = t0 + s;
        if (t < T) rs = front_col(s, t, i, bottom, half, rs, Mm + NS * 32, lane);   // (a guard, not a break: the loop must unroll completely)
      }
    }
    if (T > 0) {
      if (i >= T && i < T + BW) {
        // slot s of row i holds column i - BW + ((s - s0) mod NS), s0 = (i - BW) mod NS: one modulo per lane, then the
        // columns of consecutive slots are consecutive (with one wrap)
        int s0 = (i - BW) % NS;
        if (s0 < 0) s0 += NS;
        const int qb = i - BW - s0, base = bottom ? (np - 1 - T) * 33 - i : i - T * 33;   // element (mr, mc) at mc * 32 + mr
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const int q = qb + s + (s < s0 ? NS : 0);
          if (q >= T && q < T + BW) Mm[bottom ? base - 32 * q : base + 32 * q] = S[s];
        }
      }
      wave_lds_fence();
    }
#pragma unroll
    for (int j = 0; j < MID; ++j) Sm[j] = Mm[j * 32 + i];
    {
      const int pm = T + lane;
      const double g1 = perm(yp, (T > 0 && lane < BW) ? pm : 0);
      const double g2 = perm(yp, (T > 0 && lane < mid && pm >= np - T - BW) ? 32 + (np - 1 - pm) : 0);
      ym = mid_rhs(np, T, mid, lane, g1, g2, rhs);
    }
    mid_cols<0>(lane, frsqrt(lane_bcast(Sm[0], 0)));
    // The factorisation packs the two fronts into the two halves of the wave because it is ISSUE bound: one instruction
    // updates both.  The triangular solves are LATENCY bound (one dependent fma per column), so for them the bottom front's
    // rows are laid beside the top front's in lanes 0..31: two independent chains that hide each other's latency, and the
    // value a column needs is a plain v_readlane of lane t for both.
#pragma unroll
    for (int s = 0; s < NS; ++s) Sb[s] = upper_half(S[s]);
    dinvb = upper_half(dinv);
  }

  // L y = rhs for a further right-hand side: (top front, bottom front, middle) parts of y
  __device__ __forceinline__ void forward(int np, int lane, const double* rhs, double& bt, double& bb, double& bm) const {
    const int T = fronts(np), mid = np - 2 * T, i = lane & 31;
    const bool rowok = T > 0 && i < T + BW;
    bt = rowok ? rhs[fold_inv(i, np)] : 0.0;            // top front: matrix row i
    bb = rowok ? rhs[fold_inv(np - 1 - i, np)] : 0.0;   // bottom front: matrix row np-1-i
    for (int t0 = 0; t0 < T; t0 += NS) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int t = t0 + s;
        if (t < T) {
          const double ct = lane_bcast(bt, t), cb = lane_bcast(bb, t);
          const bool below = (unsigned)(i - t - 1) < (unsigned)BW;
          bt = fma(-(below ? S[s] : 0.0), ct, bt);
          bb = fma(-(below ? Sb[s] : 0.0), cb, bb);
        }
      }
    }
    {
      const int pm = T + lane;
      const double g1 = perm(bt, (T > 0 && lane < BW) ? pm : 0);
      const double g2 = perm(bb, (T > 0 && lane < mid && pm >= np - T - BW) ? np - 1 - pm : 0);
      bm = mid_rhs(np, T, mid, lane, g1, g2, rhs);
    }
#pragma unroll
    for (int c = 0; c < MID; ++c) {
      const double bc = lane_bcast(bm, c);
      bm = fma(-((lane > c && lane < MID) ? Sm[c] : 0.0), bc, bm);
    }
  }
  // the y of the first right-hand side, as the factorisation left it, in the layout of forward()
  __device__ __forceinline__ void first_y(double& bt, double& bb, double& bm) const { bt = yp; bb = upper_half(yp); bm = ym; }

  // D L' x = y; x to LDS in the natural order of the unknowns (fold_inv of the matrix row)
  __device__ __forceinline__ void backward(int np, int lane, double bt, double bb, double bm, double* x) const {
    const int T = fronts(np), mid = np - 2 * T, i = lane & 31;
#pragma unroll
    for (int c = MID - 1; c >= 0; --c) {
      const double tc = lane_bcast(bm, c);
      bm = fma(-(lane < c ? Sm[c] : 0.0), tc, bm);
    }
    bm *= dinvm;
    if (lane < mid) x[fold_inv(T + lane, np)] = bm;
    if (T > 0) {
      // the solution of the middle rows next to each front goes where the fronts' columns expect it
      const bool blk = i >= T && i < T + BW;
      const double g1 = perm(bm, blk ? i - T : 0), g2 = perm(bm, blk ? np - 1 - i - T : 0);
      bt = blk ? g1 : bt;
      bb = blk ? g2 : bb;
      // middle columns first (unscaled entries times x), then the fronts' own columns (scaled entries times d x)
      for (int t0 = ((T + BW - 1) / NS) * NS; t0 >= 0; t0 -= NS) {
#pragma unroll
        for (int s = NS - 1; s >= 0; --s) {
          const int t = t0 + s;
          if (t < T + BW) {
            const double ct = lane_bcast(bt, t), cb = lane_bcast(bb, t);
            const bool above = (unsigned)(t - 1 - i) < (unsigned)BW && i < T;
            bt = fma(-(above ? S[s] : 0.0), ct, bt);
            bb = fma(-(above ? Sb[s] : 0.0), cb, bb);
          }
        }
      }
      if (lane < T) {
        x[fold_inv(lane, np)] = bt * dinv;
        x[fold_inv(np - 1 - lane, np)] = bb * dinvb;
      }
    }
  }
};

// ------------------------------------------------------------------------------------------------
// pieces shared by both roles (every thread of the workgroup calls them with its own tid)

// Ssum[sp * NO + out0 + o] = sum over the chunks of span sp, in chunk order, of Spart[ch * GR + o]
template <int GR>
__device__ __forceinline__ void g2_reduce_round(int tid, int nt, int np, const int* sch, const double* Spart,
                                                double* Ssum, int NO, int out0, int w) {
  for (int task = tid; task < np * w; task += nt) {
    const int sp = task / w, o = task - sp * w;
    double sum = 0.0;
    int ch = sch[sp];
    const int ce = sch[sp + 1];
    for (; ch + 4 <= ce; ch += 4) {  // four loads in flight; the additions keep the chunk order
      const double v0 = Spart[ch * GR + o], v1 = Spart[(ch + 1) * GR + o];
      const double v2 = Spart[(ch + 2) * GR + o], v3 = Spart[(ch + 3) * GR + o];
      sum += v0; sum += v1; sum += v2; sum += v3;
    }
    for (; ch < ce; ++ch) sum += Spart[ch * GR + o];
    Ssum[sp * NO + out0 + o] = sum;
  }
}

template <int K>
__device__ __forceinline__ double g2_band_entry(const double* Pc, const double* Ssum, int np, int j, int d) {
  constexpr int K1 = K + 1, NO = K1 * (K1 + 1) / 2 + 2 * K1;
  double sum = Pc[j * K1 + d];
  for (int al = d; al <= K; ++al) {
    int sp = j - al;
    if (sp < 0) sp += np;
    sum += Ssum[sp * NO + al * (al + 1) / 2 + (al - d)];
  }
  return sum;
}

// normal matrix P + A'DA into the folded slot layout Mf[s][row] (all slots written, zeros included),
// dual residual rd and the first right-hand side
template <int K, int G>
__device__ __forceinline__ void g2_assemble(int tid, int nt, int np, const double* Pc, const double* Ssum,
                                            const double* rdP, double* Mf, double* rd, double* rhs) {
  constexpr int K1 = K + 1, NE = K1 * (K1 + 1) / 2, NO = NE + 2 * K1, BW = 2 * K, NS = 2 * BW + 1;
  for (int task = tid; task < NS * 64 * G; task += nt) {
    const int s = task / (64 * G), p = task - s * (64 * G);
    int m = (s - p + BW) % NS;
    if (m < 0) m += NS;
    const int col = p - BW + m;
    double v = 0.0;
    if (p < np && col >= 0 && col < np) {
      const int j1 = fold_inv(p, np), j2 = fold_inv(col, np);
      int d = j1 - j2;
      if (d < 0) d += np;
      if (d <= K) v = g2_band_entry<K>(Pc, Ssum, np, j1, d);
      else if (np - d <= K) v = g2_band_entry<K>(Pc, Ssum, np, j2, np - d);
    }
    Mf[task] = v;
  }
  for (int j = tid; j < np; j += nt) {
    double s1 = 0.0, s2 = 0.0;
    for (int al = 0; al <= K; ++al) {
      int sp = j - al;
      if (sp < 0) sp += np;
      s1 += Ssum[sp * NO + NE + al]; s2 += Ssum[sp * NO + NE + K1 + al];
    }
    rd[j] = rdP[j] + s2;
    rhs[j] = s1 - rdP[j];
  }
}

// np <= 64 (two-front factorisation): the normal matrix P + A'DA has only np (K+1) distinct entries -- the cyclic band
// band[j][d] = M[j][j-d] -- so that is all that is assembled; the linear-algebra wave picks its register image out of it
// (TwoFront::load).  Plus the dual residual rd and the first right-hand side.
template <int K>
__device__ __forceinline__ void g2_assemble_band(int tid, int nt, int np, const double* Pc, const double* Ssum,
                                                 const double* rdP, double* band, double* rd, double* rhs) {
  constexpr int K1 = K + 1, NE = K1 * (K1 + 1) / 2, NO = NE + 2 * K1;
  for (int task = tid; task < np * K1; task += nt) {
    const int j = task / K1, d = task - j * K1;
    band[task] = g2_band_entry<K>(Pc, Ssum, np, j, d);
  }
  for (int j = tid; j < np; j += nt) {
    double s1 = 0.0, s2 = 0.0;
    for (int al = 0; al <= K; ++al) {
      int sp = j - al;
      if (sp < 0) sp += np;
      s1 += Ssum[sp * NO + NE + al]; s2 += Ssum[sp * NO + NE + K1 + al];
    }
    rd[j] = rdP[j] + s2;
    rhs[j] = s1 - rdP[j];
  }
}

// The chunk's constraint rows: register resident (AREG, R <= 6: 12 R VGPRs), or, for long chunks
// (N up to ~4400 with R = 10), re-read from the L2-resident table in every pass.
template <int K1, int R, bool AREG>
struct ChunkRows;
template <int K1, int R>
struct ChunkRows<K1, R, true> {
  double a[R][K1];
  __device__ __forceinline__ void load(const double* A, int row0, int cnt) {
    const double* Ap = A;
    asm volatile("" : "+s"(Ap));  // keeps the loads where they are (after the register-hungry curvature pass)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int al = 0; al < K1; ++al) a[r][al] = r < cnt ? Ap[(size_t)(row0 + r) * K1 + al] : 0.0;
  }
  __device__ __forceinline__ void get(int r, double (&o)[K1]) const {
#pragma unroll
    for (int al = 0; al < K1; ++al) o[al] = a[r][al];
  }
};
template <int K1, int R>
struct ChunkRows<K1, R, false> {
  const double* p;
  int cnt;
  __device__ __forceinline__ void load(const double* A, int row0, int cnt_) { p = A + (size_t)row0 * K1; cnt = cnt_; }
  __device__ __forceinline__ void get(int r, double (&o)[K1]) const {
    static_assert(K1 % 2 == 0, "rows are read as double2");
    const double2* q = reinterpret_cast<const double2*>(p + (size_t)r * K1);
#pragma unroll
    for (int al = 0; al < K1; al += 2) {
      const double2 v = r < cnt ? q[al / 2] : make_double2(0.0, 0.0);
      o[al] = v.x; o[al + 1] = v.y;
    }
  }
};

// ------------------------------------------------------------------------------------------------
// Barrier protocol.  Both roles execute exactly this sequence (R = row waves, L = linear-algebra wave):
//
//   per outer iteration
//     [Oa] all: control points from the offsets                                  -> barrier
//     [Ob] R: curvature rows, Gauss-Newton partials (27 outputs)                 -> flush (3 rounds:
//          R writes Spart | barrier | all reduce | barrier)    -- the first barrier also publishes
//          sum kappa^2; every thread leaves the outer loop here when outer == n_outer
//     [Oc] L: P, q (scaled), x = a.   R: lo/hi, initial slacks and duals         -> barrier
//     per interior-point iteration
//       [I1] R: residuals, A'DA / A'e / A'(lu-ll) partials (33 outputs).  L: P x + q   -> flush (3 rounds)
//       [I2] all: folded normal matrix, rd, rhs                                   -> barrier
//       [I3] L: convergence test -> ctl[0]; factor; affine solve -> dxs          -> barrier
//            all: leave the loop when ctl[0] != 0
//       [I4] R: affine step: ratios, complementarity sums -> red; corrector partials A'w0 and
//            A'(1/sl - 1/su) (2(K+1) outputs) -> Spart                            -> barrier
//       [I5] all: reduce                                                          -> barrier
//       [I6] L: alpha_aff, sigma -> ctl[1]; second right-hand side; solve -> dxs -> barrier
//       [I7] R: step ratios -> red                                                -> barrier
//       [I8] all: alpha.  R: update slacks/duals.  L: x += alpha dx              -> barrier
//     [Od] L: a = x, step size                                                   -> barrier
template <int K, int R, int G, bool AREG>
__global__ void __launch_bounds__(kG2Block) k_global_qp2(GlobalArgs a) {
  constexpr int K1 = K + 1, NE = K1 * (K1 + 1) / 2, NO = NE + 2 * K1, BW = 2 * K, GR = g2_round(G);
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const int nt = blockDim.x, nrw = (nt >> 6) - 1, nrow = nrw * kWave;
  const bool la = wave == nrw;
  const int b = blockIdx.x;
  const int N = a.trk.N, n = a.trk.n, np = a.np;
  const Global2Layout L = global2_layout(K, n, np, N, G, nrow);
  double *xsA = lds + L.xs, *xsB = lds + L.xs2, *dxa = lds + L.dxa, *dxs = lds + L.dxs, *avs = lds + L.avs, *qv = lds + L.qv, *rdP = lds + L.rdP;
  double *rd = lds + L.rd, *rhs = lds + L.rhs, *cxs = lds + L.cxs, *cys = lds + L.cys, *nus = lds + L.nus;
  double *Pc = lds + L.Pc, *band = lds + L.band, *Ssum = lds + L.Ssum, *Spart = lds + L.Spart, *Mf = lds + L.Spart;
  short* itab = reinterpret_cast<short*>(lds + L.itab);
  double *lohi = lds + L.lohi, *red = lds + L.red, *ctl = lds + L.ctl;
  int* sch = reinterpret_cast<int*>(lds + L.sch);
  const double* __restrict__ wid = a.widths + (size_t)b * N * 2;

  for (int j = tid; j < np; j += nt) { nus[2 * j] = a.nu[2 * j]; nus[2 * j + 1] = a.nu[2 * j + 1]; avs[j] = 0.0; }
  for (int j = tid; j <= np; j += nt) sch[j] = a.span_ch0[j];
  if constexpr (G == 1) TwoFront<BW>::build_index_table(tid, nt, np, itab);
  for (int i = tid; i < N; i += nt) {
    const double2 w = reinterpret_cast<const double2*>(wid)[i];
    reinterpret_cast<double2*>(lohi)[i] = make_double2(-(w.y - a.margin), w.x - a.margin);
  }
  __syncthreads();

  // block-level combine of the row waves' partials (fixed order)
  auto red_sum = [&](int kind) { double s = 0.0; for (int w = 0; w < nrw; ++w) s += red[kind * 16 + w]; return s; };
  auto red_max = [&](int kind) { double s = -INFINITY; for (int w = 0; w < nrw; ++w) s = fmax(s, red[kind * 16 + w]); return s; };
  double k2_first = 0.0, k2_last = 0.0;
  int total_it = 0;

  if (!la) {
    // =================================================================================== row waves
    const bool has = tid < a.nc;
    int row0 = 0, cnt = 0, sp0 = 0;
    if (has) {
      const int2 c = reinterpret_cast<const int2*>(a.chunk)[tid];
      row0 = c.x & 0xffffff; cnt = (unsigned)c.x >> 24; sp0 = c.y;
    }
    int J[K1];
#pragma unroll
    for (int al = 0; al < K1; ++al) { J[al] = sp0 + al; if (J[al] >= np) J[al] -= np; }
    ChunkRows<K1, R, AREG> rows;
    const double* __restrict__ D1 = a.trk.D + (size_t)K1 * N;
    const double* __restrict__ D2 = a.trk.D + (size_t)2 * K1 * N;
    double sl[R] = {}, su[R] = {}, ll[R] = {}, lu[R] = {};
#ifdef RL_G2_PROFILE
    long long rT[6] = {0, 0, 0, 0, 0, 0}, rk = wall_clock64();
#define G2_LAP(i) { const long long n_ = wall_clock64(); rT[i] += n_ - rk; rk = n_; }
#else
#define G2_LAP(i)
#endif

    for (int outer = 0;; ++outer) {
      // [Oa]
      for (int j = tid; j < n; j += nt) {
        const int jj = j >= np ? j - np : j;
        cxs[j] = fma(avs[jj], nus[2 * jj], a.trk.c0[jj]);
        cys[j] = fma(avs[jj], nus[2 * jj + 1], a.trk.c0[n + jj]);
      }
      __syncthreads();
      // [Ob]
      {
        double acc[NE + K1];
#pragma unroll
        for (int o = 0; o < NE + K1; ++o) acc[o] = 0.0;
        double k2p = 0.0;
        if (has) {
          double cxJ[K1], cyJ[K1];
#pragma unroll
          for (int al = 0; al < K1; ++al) { cxJ[al] = cxs[sp0 + al]; cyJ[al] = cys[sp0 + al]; }
#pragma unroll 1
          for (int r = 0; r < cnt; ++r) {
            const int i = row0 + r;
            double b1[K1], b2[K1], dx = 0, dy = 0, ddx = 0, ddy = 0;
#pragma unroll
            for (int al = 0; al < K1; ++al) {
              b1[al] = D1[(size_t)al * N + i]; b2[al] = D2[(size_t)al * N + i];
              dx = fma(cxJ[al], b1[al], dx); dy = fma(cyJ[al], b1[al], dy);
              ddx = fma(cxJ[al], b2[al], ddx); ddy = fma(cyJ[al], b2[al], ddy);
            }
            const double s2 = dx * dx + dy * dy, inv3 = 1.0 / (s2 * sqrt(s2));
            const double kp = (dx * ddy - dy * ddx) * inv3;
            k2p = fma(kp, kp, k2p);
            double Gr[K1], ga = 0.0;
#pragma unroll
            for (int al = 0; al < K1; ++al) {
              const double nx = nus[2 * J[al]], ny = nus[2 * J[al] + 1];
              Gr[al] = ((b1[al] * nx) * ddy + dx * (b2[al] * ny) - (b1[al] * ny) * ddx - dy * (b2[al] * nx)) * inv3 -
                       3.0 * kp * (dx * b1[al] * nx + dy * b1[al] * ny) / s2;
              ga = fma(Gr[al], avs[J[al]], ga);
            }
            const double res = kp - ga;
            int e = 0;
#pragma unroll
            for (int al = 0; al < K1; ++al) {
#pragma unroll
              for (int be = 0; be <= al; ++be) { acc[e] = fma(Gr[al], Gr[be], acc[e]); ++e; }
              acc[NE + al] = fma(Gr[al], res, acc[NE + al]);
            }
          }
        }
        k2p = wave_sum(k2p);
        if (lane == 0) red[wave] = k2p;
#pragma unroll
        for (int r0 = 0; r0 < NE + K1; r0 += GR) {
          const int w = NE + K1 - r0 < GR ? NE + K1 - r0 : GR;
          if (has) {
#pragma unroll
            for (int o = 0; o < GR; ++o)
              if (r0 + o < NE + K1) Spart[tid * GR + o] = acc[r0 + o];
          }
          __syncthreads();
          if (r0 == 0) {
            k2_last = red_sum(0);
            if (outer == 0) k2_first = k2_last;
            if (outer == a.n_outer) break;
          }
          g2_reduce_round<GR>(tid, nt, np, sch, Spart, Ssum, NO, r0, w);
          __syncthreads();
        }
        if (outer == a.n_outer) break;
      }
      // [Oc]  the chunk's constraint rows are (re)loaded here, after the register-hungry curvature
      // pass, so that they are not live across it; the laundered pointer keeps the loads in the loop
      rows.load(a.A, row0, cnt);
      if (has) {
        double xJ[K1];
#pragma unroll
        for (int al = 0; al < K1; ++al) xJ[al] = avs[J[al]];
#pragma unroll
        for (int r = 0; r < R; ++r) { RL_ROW_FENCE();
          double Av[K1], ax = 0.0;
          rows.get(r, Av);
#pragma unroll
          for (int al = 0; al < K1; ++al) ax = fma(Av[al], xJ[al], ax);
          const double2 lh = r < cnt ? reinterpret_cast<const double2*>(lohi)[row0 + r] : make_double2(-1.0, 1.0);
          if (outer == 0) {
            sl[r] = fmax(ax - lh.x, 1e-2); su[r] = fmax(lh.y - ax, 1e-2);
            ll[r] = 1.0; lu[r] = 1.0;
          } else {  // warm start from the previous linearisation's slacks and duals
            sl[r] = fmax(sl[r], 1e-2); su[r] = fmax(su[r], 1e-2);
            ll[r] = fmax(ll[r], 1e-2); lu[r] = fmax(lu[r], 1e-2);
          }
        }
      }
      __syncthreads();
      const double *xc = xsA, *xn = xsB;  // x is double buffered: [I8] reads it while the LA wave writes the next one
      for (int it = 0; it < a.max_ipm; ++it) {
        G2_LAP(5);
        // [I1]  three sweeps over the rows, one per flush round: only GR accumulators are live at a time
        {
          double dmv[R], ev[R], dlv[R];
#pragma unroll
          for (int r0 = 0; r0 < NO; r0 += GR) {
            const int w = NO - r0 < GR ? NO - r0 : GR;
            double acc[GR];
#pragma unroll
            for (int o = 0; o < GR; ++o) acc[o] = 0.0;
            if (r0 == 0) {
              double mu = 0.0, rpmax = 0.0;
              if (has) {
                double xJ[K1];
#pragma unroll
                for (int al = 0; al < K1; ++al) xJ[al] = xc[J[al]];
#pragma unroll
                for (int r = 0; r < R; ++r) { RL_ROW_FENCE();
                  dmv[r] = 0.0; ev[r] = 0.0; dlv[r] = 0.0;
                  if (r < cnt) {
                    double Av[K1], ax = 0.0;
                    rows.get(r, Av);
#pragma unroll
                    for (int al = 0; al < K1; ++al) ax = fma(Av[al], xJ[al], ax);
                    const double2 lh = reinterpret_cast<const double2*>(lohi)[row0 + r];
                    const double rpl = ax - lh.x - sl[r], rpu = lh.y - ax - su[r];
                    const double ql = ll[r] * frcp(sl[r]), qu = lu[r] * frcp(su[r]);
                    dmv[r] = ql + qu; ev[r] = qu * rpu - ql * rpl; dlv[r] = lu[r] - ll[r];
                    mu += sl[r] * ll[r] + su[r] * lu[r];
                    rpmax = fmax(rpmax, fmax(fabs(rpl), fabs(rpu)));
                  }
                }
              }
              mu = wave_sum(mu); rpmax = wave_max(rpmax);
              if (lane == 0) { red[wave] = mu; red[32 + wave] = rpmax; }
            }
            if (has) {
#pragma unroll
              for (int r = 0; r < R; ++r) { RL_ROW_FENCE();
                // rows beyond cnt have A = 0 and dm = e = dl = 0: they add nothing
                double Av[K1];
                rows.get(r, Av);
                int o = 0;
#pragma unroll
                for (int al = 0; al < K1; ++al) {
                  const double wa = dmv[r] * Av[al];
#pragma unroll
                  for (int be = 0; be <= al; ++be) {
                    if (o >= r0 && o < r0 + GR) acc[o - r0] = fma(wa, Av[be], acc[o - r0]);
                    ++o;
                  }
                  if (NE + al >= r0 && NE + al < r0 + GR) acc[NE + al - r0] = fma(Av[al], ev[r], acc[NE + al - r0]);
                  if (NE + K1 + al >= r0 && NE + K1 + al < r0 + GR)
                    acc[NE + K1 + al - r0] = fma(Av[al], dlv[r], acc[NE + K1 + al - r0]);
                }
              }
#pragma unroll
              for (int o = 0; o < GR; ++o)
                if (r0 + o < NO) Spart[tid * GR + o] = acc[o];
            }
            __syncthreads();
            g2_reduce_round<GR>(tid, nt, np, sch, Spart, Ssum, NO, r0, w);
            __syncthreads();
          }
        }
        G2_LAP(0);
        // [I2]
        if constexpr (G == 1) {
          g2_assemble_band<K>(tid, nt, np, Pc, Ssum, rdP, band, rd, rhs);
          __syncthreads();
          TwoFront<BW>::fill_images(tid, nt, itab, band, Mf);
        } else g2_assemble<K, G>(tid, nt, np, Pc, Ssum, rdP, Mf, rd, rhs);
        __syncthreads();
        G2_LAP(2);   // assembly of the normal matrix (all threads) up to its barrier
        // [I3]
        __syncthreads();
        G2_LAP(1);
        if (ctl[0] != 0.0) break;
        ++total_it;
        // [I4]..[I8]  Nothing but the slacks and duals is carried across a barrier: every pass rebuilds
        // the row quantities it needs from x, the affine direction dxa and the final direction dxs.
        struct RowAff { double rpl, rpu, isl, isu, d_sl, d_su, d_ll, d_lu; };
        auto row_affine = [&](int r, const double (&Av)[K1], const double (&xJ)[K1], const double (&aJ)[K1]) {
          RowAff f;
          double ax = 0.0, adx = 0.0;
#pragma unroll
          for (int al = 0; al < K1; ++al) { ax = fma(Av[al], xJ[al], ax); adx = fma(Av[al], aJ[al], adx); }
          const double2 lh = reinterpret_cast<const double2*>(lohi)[row0 + r];
          f.rpl = ax - lh.x - sl[r]; f.rpu = lh.y - ax - su[r];
          f.isl = frcp(sl[r]); f.isu = frcp(su[r]);
          f.d_sl = adx + f.rpl; f.d_su = f.rpu - adx;
          f.d_ll = (-(sl[r] * ll[r]) - ll[r] * f.d_sl) * f.isl;
          f.d_lu = (-(su[r] * lu[r]) - lu[r] * f.d_su) * f.isu;
          return f;
        };
        // [I4]+[I5]  one pass: step-length / complementarity statistics of the affine direction AND the
        // corrector partials.  The corrector weight is affine in sigma*mu,
        //   w = w0 + smu (1/sl - 1/su),   w0 = (-(sl ll + dsl dll) - ll rpl)/sl - (-(su lu + dsu dlu) - lu rpu)/su,
        // so A'w0 and A'(1/sl - 1/su) are accumulated side by side (2 (K+1) outputs) and the LA wave, which
        // computes sigma from the statistics, combines them.
        {
          double c1 = 0.0, c2 = 0.0, rmax = 0.0;
          double acc[2 * K1];
#pragma unroll
          for (int o = 0; o < 2 * K1; ++o) acc[o] = 0.0;
          if (has) {
            double xJ[K1], aJ[K1];
#pragma unroll
            for (int al = 0; al < K1; ++al) { xJ[al] = xc[J[al]]; aJ[al] = dxa[J[al]]; }
#pragma unroll
            for (int r = 0; r < R; ++r) {
              if (r < cnt) {
                double Av[K1];
                rows.get(r, Av);
                const RowAff f = row_affine(r, Av, xJ, aJ);
                rmax = fmax(rmax, fmax(fmax(-f.d_sl * f.isl, -f.d_su * f.isu),
                                       fmax(-f.d_ll * __builtin_amdgcn_rcp(ll[r]), -f.d_lu * __builtin_amdgcn_rcp(lu[r]))));
                c1 += sl[r] * f.d_ll + ll[r] * f.d_sl + su[r] * f.d_lu + lu[r] * f.d_su;
                c2 += f.d_sl * f.d_ll + f.d_su * f.d_lu;
                const double w0 = (-(sl[r] * ll[r] + f.d_sl * f.d_ll) - ll[r] * f.rpl) * f.isl -
                                  (-(su[r] * lu[r] + f.d_su * f.d_lu) - lu[r] * f.rpu) * f.isu;
                const double w1 = f.isl - f.isu;
#pragma unroll
                for (int al = 0; al < K1; ++al) {
                  acc[al] = fma(Av[al], w0, acc[al]);
                  acc[K1 + al] = fma(Av[al], w1, acc[K1 + al]);
                }
              }
            }
#pragma unroll
            for (int o = 0; o < 2 * K1; ++o) Spart[tid * GR + o] = acc[o];
          }
          c1 = wave_sum(c1); c2 = wave_sum(c2); rmax = wave_max(rmax);
          if (lane == 0) { red[wave] = c1; red[16 + wave] = c2; red[32 + wave] = rmax; }
        }
        __syncthreads();
        g2_reduce_round<GR>(tid, nt, np, sch, Spart, Ssum, NO, NE, 2 * K1);
        __syncthreads();
        G2_LAP(3);
        // [I6]
        __syncthreads();
        G2_LAP(4);
        const double smu = ctl[1];
        // [I7], [I8]
        auto row_final = [&](int r, const double (&xJ)[K1], const double (&aJ)[K1], const double (&dJ)[K1],
                             double& d_sl, double& d_su, double& d_ll, double& d_lu, double& isl, double& isu) {
          double Av[K1];
          rows.get(r, Av);
          const RowAff f = row_affine(r, Av, xJ, aJ);
          double adx = 0.0;
#pragma unroll
          for (int al = 0; al < K1; ++al) adx = fma(Av[al], dJ[al], adx);
          const double rcl = sl[r] * ll[r] - smu + f.d_sl * f.d_ll, rcu = su[r] * lu[r] - smu + f.d_su * f.d_lu;
          d_sl = adx + f.rpl; d_su = f.rpu - adx;
          d_ll = (-rcl - ll[r] * d_sl) * f.isl; d_lu = (-rcu - lu[r] * d_su) * f.isu;
          isl = f.isl; isu = f.isu;
        };
        {
          double rmax = 0.0;
          if (has) {
            double xJ[K1], aJ[K1], dJ[K1];
#pragma unroll
            for (int al = 0; al < K1; ++al) { xJ[al] = xc[J[al]]; aJ[al] = dxa[J[al]]; dJ[al] = dxs[J[al]]; }
#pragma unroll
            for (int r = 0; r < R; ++r) {
              if (r < cnt) {
                double d_sl, d_su, d_ll, d_lu, isl, isu;
                row_final(r, xJ, aJ, dJ, d_sl, d_su, d_ll, d_lu, isl, isu);
                rmax = fmax(rmax, fmax(fmax(-d_sl * isl, -d_su * isu),
                                       fmax(-d_ll * __builtin_amdgcn_rcp(ll[r]), -d_lu * __builtin_amdgcn_rcp(lu[r]))));
              }
            }
          }
          rmax = wave_max(rmax);
          if (lane == 0) red[32 + wave] = rmax;
        }
        __syncthreads();
        {
          const double rmax = red_max(2);
          const double alpha = rmax > 0.995 ? 0.995 / rmax : 1.0;
          if (has) {
            double xJ[K1], aJ[K1], dJ[K1];
#pragma unroll
            for (int al = 0; al < K1; ++al) { xJ[al] = xc[J[al]]; aJ[al] = dxa[J[al]]; dJ[al] = dxs[J[al]]; }
#pragma unroll
            for (int r = 0; r < R; ++r) {
              if (r < cnt) {
                double d_sl, d_su, d_ll, d_lu, isl, isu;
                row_final(r, xJ, aJ, dJ, d_sl, d_su, d_ll, d_lu, isl, isu);
                sl[r] = fma(alpha, d_sl, sl[r]); su[r] = fma(alpha, d_su, su[r]);
                ll[r] = fma(alpha, d_ll, ll[r]); lu[r] = fma(alpha, d_lu, lu[r]);
              }
            }
          }
        }
        __syncthreads();
        { const double* t_ = xc; xc = xn; xn = t_; }
      }
      // [Od]
      __syncthreads();
    }
#ifdef RL_G2_PROFILE
    if (tid == 0) for (int i = 0; i < 6; ++i) ctl[1 + i] = (double)rT[i];
#endif
    // ---- outputs of the row waves: line samples and the largest bound violation
    {
      const double* __restrict__ D0 = a.trk.D;
      double viol = -INFINITY, nact = 0.0;
      if (has) {
        for (int r = 0; r < cnt; ++r) {
          const int i = row0 + r;
          double x = 0.0, y = 0.0;
#pragma unroll
          for (int al = 0; al < K1; ++al) {
            const double h = D0[(size_t)al * N + i];
            x = fma(cxs[sp0 + al], h, x); y = fma(cys[sp0 + al], h, y);
          }
          if (a.out_xy) reinterpret_cast<double2*>(a.out_xy)[(size_t)b * N + i] = make_double2(x, y);
          const double lat = (x - a.trk.base[i]) * a.trk.base[(size_t)2 * N + i] +
                             (y - a.trk.base[(size_t)N + i]) * a.trk.base[(size_t)3 * N + i];
          const double2 lh = reinterpret_cast<const double2*>(lohi)[i];
          viol = fmax(viol, fmax(lh.x - lat, lat - lh.y));
          if (lat - lh.x < 1e-6 || lh.y - lat < 1e-6) nact += 1.0;
        }
      }
      viol = wave_max(viol); nact = wave_sum(nact);
      if (lane == 0) { red[32 + wave] = viol; red[wave] = nact; }
    }
    __syncthreads();
  } else {
    // ========================================================================= linear-algebra wave
    FoldBand<BW, G> fb;   // np > 64 (G == 3): the folded band over G groups of 64 rows (unused, hence eliminated, for G == 1)
    TwoFront<BW> tf;      // np <= 64 (G == 1): two fronts in this one wave
    double last_step = 0.0;
#ifdef RL_G2_PROFILE
    long long tF = 0, tS = 0, tA = 0, tk = 0;
#define G2_TIC() tk = wall_clock64()
#define G2_TOC(acc) acc += wall_clock64() - tk
#else
#define G2_TIC()
#define G2_TOC(acc)
#endif
    for (int outer = 0;; ++outer) {
      // [Oa]
      for (int j = tid; j < n; j += nt) {
        const int jj = j >= np ? j - np : j;
        cxs[j] = fma(avs[jj], nus[2 * jj], a.trk.c0[jj]);
        cys[j] = fma(avs[jj], nus[2 * jj + 1], a.trk.c0[n + jj]);
      }
      __syncthreads();
      // [Ob]
      bool done = false;
#pragma unroll
      for (int r0 = 0; r0 < NE + K1; r0 += GR) {
        const int w = NE + K1 - r0 < GR ? NE + K1 - r0 : GR;
        __syncthreads();
        if (r0 == 0) {
          k2_last = red_sum(0);
          if (outer == 0) k2_first = k2_last;
          if (outer == a.n_outer) { done = true; break; }
        }
        g2_reduce_round<GR>(tid, nt, np, sch, Spart, Ssum, NO, r0, w);
        __syncthreads();
      }
      if (done) break;
      // [Oc]  P = sc 2 G'G + eps I (cyclic band: Pc[j][d] = P[j][j-d]),  q = sc 2 G'(kappa - G a)
      double qinf = 0.0;
      {
        double tr = 0.0;
        for (int task = lane; task < np * K1; task += kWave) {
          const int j = task / K1, d = task - j * K1;
          double sum = 0.0;
          for (int al = d; al <= K; ++al) {
            int sp = j - al;
            if (sp < 0) sp += np;
            sum += Ssum[sp * NO + al * (al + 1) / 2 + (al - d)];
          }
          Pc[task] = 2.0 * sum;
          if (d == 0) tr += 2.0 * sum;
        }
        tr = wave_sum(tr);
        const double sc = (double)np / tr;
        for (int task = lane; task < np * K1; task += kWave) {
          const int d = task % K1;
          Pc[task] = Pc[task] * sc + (d == 0 ? 1e-9 : 0.0);
        }
        for (int j = lane; j < np; j += kWave) {
          double sum = 0.0;
          for (int al = 0; al <= K; ++al) {
            int sp = j - al;
            if (sp < 0) sp += np;
            sum += Ssum[sp * NO + NE + al];
          }
          const double q = 2.0 * sum * sc;
          qv[j] = q;
          xsA[j] = avs[j];
          qinf = fmax(qinf, fabs(q));
        }
        qinf = wave_max(qinf);
      }
      __syncthreads();
      double *xc = xsA, *xn = xsB;
      for (int it = 0; it < a.max_ipm; ++it) {
        // [I1]  rdP = P x + q
        for (int j = lane; j < np; j += kWave) {
          double s = qv[j];
#pragma unroll
          for (int d = 0; d <= K; ++d) {
            int jm = j - d; if (jm < 0) jm += np;
            s = fma(Pc[j * K1 + d], xc[jm], s);
            if (d > 0) { int jp = j + d; if (jp >= np) jp -= np; s = fma(Pc[jp * K1 + d], xc[jp], s); }
          }
          rdP[j] = s;
        }
#pragma unroll
        for (int r0 = 0; r0 < NO; r0 += GR) {
          const int w = NO - r0 < GR ? NO - r0 : GR;
          __syncthreads();
          g2_reduce_round<GR>(tid, nt, np, sch, Spart, Ssum, NO, r0, w);
          __syncthreads();
        }
        // [I2]
        const double mu_sum = red_sum(0), rpmax = red_max(2);
        if constexpr (G == 1) {
          g2_assemble_band<K>(tid, nt, np, Pc, Ssum, rdP, band, rd, rhs);
          __syncthreads();
          TwoFront<BW>::fill_images(tid, nt, itab, band, Mf);
        } else g2_assemble<K, G>(tid, nt, np, Pc, Ssum, rdP, Mf, rd, rhs);
        __syncthreads();
        // [I3]
        bool conv;
        {
          double rdmax = 0.0;
          for (int j = lane; j < np; j += kWave) rdmax = fmax(rdmax, fabs(rd[j]));
          rdmax = wave_max(rdmax);
          const double mu = mu_sum / (double)(2 * N);
          // the exit rule (rl_device.hpp: ipm_done): the complementarity alone, 1e-7 before the last linearisation, 1e-10 on it
          (void)rdmax; (void)rpmax; (void)qinf;
          conv = ipm_done(kIpmTolOneOffset, outer + 1 >= a.n_outer, mu);
          if (lane == 0) ctl[0] = conv ? 1.0 : 0.0;
        }
        double bx[G];
        if (!conv) {
          if constexpr (G == 1) {
            G2_TIC();
            tf.load(Mf, lane);
            tf.factor(np, lane, Mf + (2 * BW + 1) * 64, rhs);   // + forward substitution of the affine right-hand side
            G2_TOC(tF);
            G2_TIC();
            double yt_, yb_, ym_;
            tf.first_y(yt_, yb_, ym_);
            tf.backward(np, lane, yt_, yb_, ym_, dxa);
            G2_TOC(tS);
          } else {
            G2_TIC();
            fb.load(Mf, lane);
            fb.factor(np, lane);
            G2_TOC(tF);
            G2_TIC();
#pragma unroll
            for (int q = 0; q < G; ++q) { const int p = lane + 64 * q; bx[q] = p < np ? rhs[fold_inv(p, np)] : 0.0; }
            fb.solve(np, lane, bx);
            G2_TOC(tS);
#pragma unroll
            for (int q = 0; q < G; ++q) { const int p = lane + 64 * q; if (p < np) dxa[fold_inv(p, np)] = bx[q]; }
          }
        }
        __syncthreads();
        if (conv) break;
        ++total_it;
        // [I4]+[I5]
        __syncthreads();
        g2_reduce_round<GR>(tid, nt, np, sch, Spart, Ssum, NO, NE, 2 * K1);
        __syncthreads();
        // [I6]  sigma from the affine statistics, second right-hand side, solve
        double smu;
        {
          const double c1 = red_sum(0), c2 = red_sum(1), rmax = red_max(2);
          const double aaff = rmax > 0.995 ? 0.995 / rmax : 1.0;
          const double mu = mu_sum / (double)(2 * N);
          const double mu_aff = (mu_sum + aaff * c1 + aaff * aaff * c2) / (double)(2 * N);
          const double ratio = mu_aff / mu;
          smu = ratio * ratio * ratio * mu;
          if (lane == 0) ctl[1] = smu;
        }
        if constexpr (G == 1) {
          // second right-hand side in the natural order of the unknowns, through LDS (rhs is free after the first solve)
          if (lane < np) {
            const int j = lane;
            double v = 0.0, v1 = 0.0;
            for (int al = 0; al <= K; ++al) {
              int sp = j - al;
              if (sp < 0) sp += np;
              v += Ssum[sp * NO + NE + al];
              v1 += Ssum[sp * NO + NE + K1 + al];
            }
            rhs[j] = fma(smu, v1, v) - rd[j];
          }
          wave_lds_fence();
          G2_TIC();
          double yt_, yb_, ym_;
          tf.forward(np, lane, rhs, yt_, yb_, ym_);
          tf.backward(np, lane, yt_, yb_, ym_, dxs);
          G2_TOC(tS);
        } else {
#pragma unroll
          for (int q = 0; q < G; ++q) {
            const int p = lane + 64 * q;
            double v = 0.0;
            if (p < np) {
              const int j = fold_inv(p, np);
              double v1 = 0.0;
              for (int al = 0; al <= K; ++al) {
                int sp = j - al;
                if (sp < 0) sp += np;
                v += Ssum[sp * NO + NE + al];
                v1 += Ssum[sp * NO + NE + K1 + al];
              }
              v = fma(smu, v1, v) - rd[j];
            }
            bx[q] = v;
          }
          G2_TIC();
          fb.solve(np, lane, bx);
          G2_TOC(tS);
#pragma unroll
          for (int q = 0; q < G; ++q) { const int p = lane + 64 * q; if (p < np) dxs[fold_inv(p, np)] = bx[q]; }
        }
        __syncthreads();
        // [I7]
        __syncthreads();
        // [I8]
        {
          const double rmax = red_max(2);
          const double alpha = rmax > 0.995 ? 0.995 / rmax : 1.0;
          for (int j = lane; j < np; j += kWave) xn[j] = fma(alpha, dxs[j], xc[j]);
        }
        __syncthreads();
        { double* t_ = xc; xc = xn; xn = t_; }
      }
      // [Od]
      {
        double stepmax = 0.0;
        for (int j = lane; j < np; j += kWave) { stepmax = fmax(stepmax, fabs(xc[j] - avs[j])); avs[j] = xc[j]; }
        last_step = wave_max(stepmax);
      }
      __syncthreads();
    }
    // ---- outputs of the linear-algebra wave: control points, offsets, statistics
    for (int j = lane; j < n; j += kWave)
      reinterpret_cast<double2*>(a.out_ctrl)[(size_t)b * n + j] = make_double2(cxs[j], cys[j]);
    if (a.out_a)
      for (int j = lane; j < np; j += kWave) a.out_a[(size_t)b * np + j] = avs[j];
    __syncthreads();
    if (lane == 0) {
      double* st = a.out_stats + (size_t)b * 8;
      st[0] = (double)total_it; st[1] = k2_first; st[2] = k2_last; st[3] = red_max(2); st[4] = last_step;
      st[5] = red_sum(0); st[6] = 0.0; st[7] = 0.0;
#ifdef RL_G2_PROFILE
      st[5] = (double)tF; st[6] = (double)tS; st[7] = ctl[6]; st[1] = ctl[1]; st[2] = ctl[2]; st[3] = ctl[3]; st[4] = ctl[4] + 1e-3 * ctl[5];
#endif
    }
  }
}

}  // namespace rl
