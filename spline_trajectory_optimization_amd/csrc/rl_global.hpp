// rl_global.hpp -- a15: the GLOBAL min-curvature QP (BASELINE.json north_star:
// "QP with the lateral offset from the centre line bounded by the left/right track widths, banded
// KKT structure from the spline basis factorised in LDS, one track instance per workgroup").
//
// The reference has no working global solver (its Julia notebook prototype does not converge,
// SURVEY.md App. A.6), so the formulation is this build's own; its CPU twin is
// oracle/mincurv_oracle.c: orc_global_mincurv, and the comment block there states the maths.
//
//   unknowns   a_j (j < np = n - k): periodic control point j slides along the fixed unit normal
//              nu_j of the centre line at its Greville site
//   rows       lo_i <= A_i . a <= hi_i, one per sample; row i touches the k+1 control points of its
//              knot span, so A'DA is cyclic-banded with half-bandwidth k
//   cost       Gauss-Newton on sum kappa_i^2, re-linearised n_outer times
//   QP         Mehrotra predictor-corrector interior point, common primal/dual step
//
// Mapping to the machine
//   * one workgroup = one instance; thread = one CHUNK of <= kGRows consecutive samples of ONE knot
//     span.  The interior-point state of those rows (slacks, duals, primal residuals) never leaves
//     the thread's registers, and because a chunk has a single span its k+1 unknowns are read from
//     LDS once per phase.
//   * A'DA, A'w and G'G are sums over rows of one span: every thread reduces its chunk in
//     registers, the chunk partials of a span are added in chunk order through a small LDS
//     staging area (fixed order -> run-to-run reproducible), and span sums are gathered into the
//     cyclic band.
//   * the np x np normal matrix is stored as a band (rows < np-k) plus k dense border rows (the
//     periodic wrap) in LDS and factorised L D L' by wave 0, one lane per entry of the trailing
//     update; the triangular solves keep the vector in registers and broadcast the pivot with
//     v_readlane, so only L is read from LDS.
//   * the constraint rows A (instance independent) are a track-level table that stays in L2.
#pragma once
#include "rl_device.hpp"

namespace rl {

// (the interior-point exit rule of all global-QP kernels: rl_device.hpp, ipm_done)

constexpr int kGRows = 8;     // samples per chunk == per thread
constexpr int kGRound = 11;   // span outputs staged per flush round
constexpr int kGMaxNp = 192;  // 3 rows per lane in the triangular solves

struct GlobalArgs {
  TrackDev trk;
  const double* A;       // [N][k+1] constraint rows: D0[a][i] * (n0_i . nu_j)
  const double* nu;      // [np][2]
  const int* chunk;      // [nc][2]: (first row | count << 24, span)
  const int* span_ch0;   // [np+1] first chunk of every span (chunks of a span are consecutive)
  int nc, np;
  const double* widths;  // [B][N][2] (w_left, w_right)
  double margin;
  int n_outer, max_ipm;
  double* out_ctrl;      // [B][n][2]
  double* out_xy;        // [B][N][2] or null
  double* out_a;         // [B][np] or null
  double* out_stats;     // [B][8]: ipm iterations, sum kappa^2 first / last, bound violation, last step
};

struct GlobalLayout {  // offsets in doubles
  int xs, dxs, avs, qv, rdP, rd, rhs, dinv, cxs, cys, nus, Pc, Lb, W, Ssum, Spart, red, sch, total;
};

__host__ __device__ inline GlobalLayout global_layout(int k, int n, int np, int block) {
  const int K1 = k + 1, NO = K1 * (K1 + 1) / 2 + 2 * K1;
  GlobalLayout L;
  int o = 0;
  auto take = [&](int c) { const int r = o; o += c; return r; };
  L.xs = take(np); L.dxs = take(np); L.avs = take(np); L.qv = take(np); L.rdP = take(np);
  L.rd = take(np); L.rhs = take(np); L.dinv = take(np);
  L.cxs = take(n); L.cys = take(n); L.nus = take(2 * np);
  L.Pc = take(np * K1); L.Lb = take(np * K1); L.W = take(k * np);
  L.Ssum = take(np * NO); L.Spart = take(block * kGRound);
  L.red = take(3 * 16);
  L.sch = take((np + 2) / 2 + 1);  // np+1 ints
  L.total = o;
  return L;
}

// ------------------------------------------------------------------------------------------------
// track-level tables
template <int K>
__global__ void k_global_normals(TrackDev tr, int np, double* __restrict__ nu) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= np) return;
  double g = 0.0;
  for (int q = 1; q <= K; ++q) g += tr.t[j + q];
  g /= (double)K;
  g -= floor(g);
  const int l = find_interval(tr.t, K, tr.n, g);
  double h[K + 1];
  deboor<K>(tr.t, g, l, 1, h);
  double dx = 0, dy = 0;
#pragma unroll
  for (int a = 0; a <= K; ++a) { dx += tr.c0[l - K + a] * h[a]; dy += tr.c0[tr.n + l - K + a] * h[a]; }
  const double s = sqrt(dx * dx + dy * dy);
  nu[2 * j] = -dy / s;
  nu[2 * j + 1] = dx / s;
}

template <int K>
__global__ void k_global_rows(TrackDev tr, int np, const double* __restrict__ nu, double* __restrict__ A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int N = tr.N;
  if (i >= N) return;
  const int s = tr.ell[i] - K;
  const double n0x = tr.base[(size_t)2 * N + i], n0y = tr.base[(size_t)3 * N + i];
#pragma unroll
  for (int a = 0; a <= K; ++a) {
    int j = s + a;
    if (j >= np) j -= np;
    A[(size_t)i * (K + 1) + a] = tr.D[(size_t)a * N + i] * (n0x * nu[2 * j] + n0y * nu[2 * j + 1]);
  }
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double lane_bcast(double v, int l) {  // l wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// In-place L D L' of the bordered band: rows r < m = np-K hold K[r][r-d] at Lb[r*(K+1)+d], the K
// border rows hold K[m+q][c] at W[q*np+c].  Afterwards the strictly lower entries are L, the
// diagonal slots still hold D and dinv[c] = 1/D_c.  ONE wave; lane = one entry (t1,t2) of the
// trailing update of the current column, whose at most 2K rows are the band rows below the pivot
// and the border rows.
template <int K>
__device__ __forceinline__ void band_factor(double* Lb, double* W, double* dinv, int np, int lane) {
  constexpr int K1 = K + 1;
  const int m = np - K;
  int t1 = 0;
  while ((t1 + 1) * (t1 + 2) / 2 <= lane) ++t1;
  const int t2 = lane - t1 * (t1 + 1) / 2;
  auto E = [&](int r, int q) -> double* { return r < m ? Lb + r * K1 + (r - q) : W + (r - m) * np + q; };
  for (int c = 0; c < np; ++c) {
    const int nb = c < m ? min(K, m - 1 - c) : 0;
    const int bs = max(m, c + 1);
    const int T = nb + np - bs;
    const bool pv = t1 < T;
    const int r1 = t1 < nb ? c + 1 + t1 : bs + (t1 - nb);
    const int r2 = t2 < nb ? c + 1 + t2 : bs + (t2 - nb);
    const double inv = 1.0 / *E(c, c);
    double u1 = 0.0, u2 = 0.0, own = 0.0;
    if (pv) { u1 = *E(r1, c); u2 = *E(r2, c); own = *E(r1, r2); }
    wave_lds_fence();
    if (pv) {
      *E(r1, r2) = fma(-u1, u2 * inv, own);
      if (t1 == t2) *E(r1, c) = u1 * inv;
    }
    if (lane == 0) dinv[c] = inv;
    wave_lds_fence();
  }
}

// Solves (L D L') x = rhs with the factor above; ONE wave, x in registers (row = lane + 64 q).
template <int K>
__device__ __forceinline__ void band_solve(const double* Lb, const double* W, const double* dinv, int np,
                                           int lane, const double* rhs, double* out) {
  constexpr int K1 = K + 1;
  constexpr int RLM = kGMaxNp / kWave;
  const int m = np - K;
  double y[RLM];
#pragma unroll
  for (int q = 0; q < RLM; ++q) { const int r = lane + kWave * q; y[q] = r < np ? rhs[r] : 0.0; }
  auto pivot = [&](int c) {
    const int seg = c >> 6;
    const double v = seg == 0 ? y[0] : (seg == 1 ? y[1] : y[2]);
    return lane_bcast(v, c & (kWave - 1));
  };
  static_assert(RLM == 3, "pivot() selects among three row groups");
  for (int c = 0; c + 1 < np; ++c) {  // L y = rhs, column sweep
    const double yc = pivot(c);
#pragma unroll
    for (int q = 0; q < RLM; ++q) {
      if (kWave * q < np) {
        const int r = lane + kWave * q;
        if (r < np && r > c && (r >= m || r - c <= K)) {
          const double l = r < m ? Lb[r * K1 + (r - c)] : W[(r - m) * np + c];
          y[q] = fma(-l, yc, y[q]);
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < RLM; ++q) { const int r = lane + kWave * q; if (r < np) y[q] *= dinv[r]; }
  for (int c = np - 1; c > 0; --c) {  // L' x = y
    const double xc = pivot(c);
#pragma unroll
    for (int q = 0; q < RLM; ++q) {
      if (kWave * q < c) {
        const int r = lane + kWave * q;
        if (r < c && (c >= m || c - r <= K)) {
          const double l = c < m ? Lb[c * K1 + (c - r)] : W[(c - m) * np + r];
          y[q] = fma(-l, xc, y[q]);
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < RLM; ++q) { const int r = lane + kWave * q; if (r < np) out[r] = y[q]; }
}

// Chunk partials -> span sums.  acc[0..NOUT) of every chunk thread go through Spart in rounds of
// kGRound outputs; task (span, output) adds the partials of the span's chunks in chunk order and
// stores the sum at Ssum[span * NO + out0 + output].
template <int NOUT, int NACC>
__device__ __forceinline__ void flush_outputs(const double (&acc)[NACC], bool has, int tid, int nthreads,
                                              int np, const int* sch, double* Spart, double* Ssum,
                                              int NO, int out0) {
#pragma unroll
  for (int r0 = 0; r0 < NOUT; r0 += kGRound) {
    const int w = NOUT - r0 < kGRound ? NOUT - r0 : kGRound;
    if (has) {
#pragma unroll
      for (int o = 0; o < kGRound; ++o)
        if (r0 + o < NOUT) Spart[tid * kGRound + o] = acc[r0 + o];
    }
    __syncthreads();
    for (int task = tid; task < np * w; task += nthreads) {
      const int sp = task / w, o = task - sp * w;
      double sum = 0.0;
      for (int ch = sch[sp]; ch < sch[sp + 1]; ++ch) sum += Spart[ch * kGRound + o];
      Ssum[sp * NO + out0 + r0 + o] = sum;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
template <int K, int MAXB>
__global__ void __launch_bounds__(MAXB) k_global_qp(GlobalArgs a) {
  constexpr int K1 = K + 1, NE = K1 * (K1 + 1) / 2, NO = NE + 2 * K1, R = kGRows;
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const int nt = blockDim.x, nw = nt >> 6;
  const int b = blockIdx.x;
  const int N = a.trk.N, n = a.trk.n, np = a.np, m = np - K;
  const GlobalLayout L = global_layout(K, n, np, nt);
  double *xs = lds + L.xs, *dxs = lds + L.dxs, *avs = lds + L.avs, *qv = lds + L.qv, *rdP = lds + L.rdP;
  double *rd = lds + L.rd, *rhs = lds + L.rhs, *dinv = lds + L.dinv, *cxs = lds + L.cxs, *cys = lds + L.cys;
  double *nus = lds + L.nus, *Pc = lds + L.Pc, *Lb = lds + L.Lb, *W = lds + L.W, *Ssum = lds + L.Ssum;
  double *Spart = lds + L.Spart, *red = lds + L.red;
  int* sch = reinterpret_cast<int*>(lds + L.sch);

  // block reductions: up to three values at once, two barriers
  auto reduce3 = [&](double& vsum, double& vmax, double& vmin) {
    vsum = wave_sum(vsum); vmax = wave_max(vmax); vmin = wave_min(vmin);
    if (lane == 0) { red[wave] = vsum; red[16 + wave] = vmax; red[32 + wave] = vmin; }
    __syncthreads();
    double s = 0.0, mx = -INFINITY, mn = INFINITY;
    for (int w = 0; w < nw; ++w) { s += red[w]; mx = fmax(mx, red[16 + w]); mn = fmin(mn, red[32 + w]); }
    __syncthreads();
    vsum = s; vmax = mx; vmin = mn;
  };
  auto span_of = [&](int j, int al) { const int sp = j - al; return sp < 0 ? sp + np : sp; };

  // this thread's chunk
  const bool has = tid < a.nc;
  int row0 = 0, cnt = 0, sp0 = 0;
  if (has) {
    const int2 c = reinterpret_cast<const int2*>(a.chunk)[tid];
    row0 = c.x & 0xffffff; cnt = (unsigned)c.x >> 24; sp0 = c.y;
  }
  int J[K1];
#pragma unroll
  for (int al = 0; al < K1; ++al) { J[al] = sp0 + al; if (J[al] >= np) J[al] -= np; }

  for (int j = tid; j < np; j += nt) { nus[2 * j] = a.nu[2 * j]; nus[2 * j + 1] = a.nu[2 * j + 1]; avs[j] = 0.0; }
  for (int j = tid; j <= np; j += nt) sch[j] = a.span_ch0[j];
  __syncthreads();

  const double* __restrict__ wid = a.widths + (size_t)b * N * 2;
  const double* __restrict__ D0 = a.trk.D;
  const double* __restrict__ D1 = a.trk.D + (size_t)K1 * N;
  const double* __restrict__ D2 = a.trk.D + (size_t)2 * K1 * N;
  double k2_first = 0.0, k2_last = 0.0, last_step = 0.0;
  int total_it = 0;

  // interior-point state of this thread's rows
  double sl[R], su[R], ll[R], lu[R], rpl[R], rpu[R];

  for (int outer = 0;; ++outer) {
    // ---- current control points
    for (int j = tid; j < n; j += nt) {
      const int jj = j >= np ? j - np : j;
      cxs[j] = fma(avs[jj], nus[2 * jj], a.trk.c0[jj]);
      cys[j] = fma(avs[jj], nus[2 * jj + 1], a.trk.c0[n + jj]);
    }
    __syncthreads();
    // ---- curvature and its Gauss-Newton rows; G'G and G'(kappa - G a) per chunk
    double k2p = 0.0;
    {
      double acc[NE + K1];
#pragma unroll
      for (int o = 0; o < NE + K1; ++o) acc[o] = 0.0;
      if (has) {
        double cxJ[K1], cyJ[K1], nxJ[K1], nyJ[K1], avJ[K1];
#pragma unroll
        for (int al = 0; al < K1; ++al) {
          cxJ[al] = cxs[sp0 + al]; cyJ[al] = cys[sp0 + al];
          nxJ[al] = nus[2 * J[al]]; nyJ[al] = nus[2 * J[al] + 1]; avJ[al] = avs[J[al]];
        }
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
          double G[K1], ga = 0.0;
#pragma unroll
          for (int al = 0; al < K1; ++al) {
            const double nx = nxJ[al], ny = nyJ[al];
            G[al] = ((b1[al] * nx) * ddy + dx * (b2[al] * ny) - (b1[al] * ny) * ddx - dy * (b2[al] * nx)) * inv3 -
                    3.0 * kp * (dx * b1[al] * nx + dy * b1[al] * ny) / s2;
            ga = fma(G[al], avJ[al], ga);
          }
          const double res = kp - ga;
          int e = 0;
#pragma unroll
          for (int al = 0; al < K1; ++al) {
#pragma unroll
            for (int be = 0; be <= al; ++be) { acc[e] = fma(G[al], G[be], acc[e]); ++e; }
            acc[NE + al] = fma(G[al], res, acc[NE + al]);
          }
        }
      }
      double vmx = 0.0, vmn = 0.0;
      reduce3(k2p, vmx, vmn);
      if (outer == 0) k2_first = k2p;
      k2_last = k2p;
      if (outer == a.n_outer) break;
      flush_outputs<NE + K1>(acc, has, tid, nt, np, sch, Spart, Ssum, NO, 0);
    }
    // ---- P = sc 2 G'G + eps I (cyclic band: Pc[j][d] = P[j][j-d]),  q = sc 2 G'(kappa - G a)
    for (int task = tid; task < np * K1; task += nt) {
      const int j = task / K1, d = task - j * K1;
      double sum = 0.0;
      for (int al = d; al <= K; ++al) sum += Ssum[span_of(j, al) * NO + al * (al + 1) / 2 + (al - d)];
      Pc[task] = 2.0 * sum;
    }
    for (int j = tid; j < np; j += nt) {
      double sum = 0.0;
      for (int al = 0; al <= K; ++al) sum += Ssum[span_of(j, al) * NO + NE + al];
      qv[j] = 2.0 * sum;
      xs[j] = avs[j];
    }
    __syncthreads();
    double tr = 0.0, qinf = 0.0, dummy = 0.0;
    for (int j = tid; j < np; j += nt) tr += Pc[j * K1];
    reduce3(tr, qinf, dummy);
    const double sc = (double)np / tr;
    qinf = 0.0;
    for (int task = tid; task < np * K1; task += nt) {
      const int d = task % K1;
      Pc[task] = Pc[task] * sc + (d == 0 ? 1e-9 : 0.0);
    }
    for (int j = tid; j < np; j += nt) { qv[j] *= sc; qinf = fmax(qinf, fabs(qv[j])); }
    tr = 0.0;
    reduce3(tr, qinf, dummy);
    // ---- interior point from x = a
    if (has) {
      double xJ[K1];
#pragma unroll
      for (int al = 0; al < K1; ++al) xJ[al] = xs[J[al]];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (r < cnt) {
          const int i = row0 + r;
          const double* __restrict__ Ar = a.A + (size_t)i * K1;
          double ax = 0.0;
#pragma unroll
          for (int al = 0; al < K1; ++al) ax = fma(Ar[al], xJ[al], ax);
          const double2 w = reinterpret_cast<const double2*>(wid)[i];
          const double lo = -(w.y - a.margin), hi = w.x - a.margin;
          if (outer == 0) {
            sl[r] = fmax(ax - lo, 1e-2); su[r] = fmax(hi - ax, 1e-2);
            ll[r] = 1.0; lu[r] = 1.0;
          } else {  // warm start from the previous linearisation's slacks and duals
            sl[r] = fmax(sl[r], 1e-2); su[r] = fmax(su[r], 1e-2);
            ll[r] = fmax(ll[r], 1e-2); lu[r] = fmax(lu[r], 1e-2);
          }
          rpl[r] = ax - lo - sl[r]; rpu[r] = hi - ax - su[r];
        }
      }
    }
    for (int it = 0; it < a.max_ipm; ++it) {
      // ---- residuals, complementarity, and the span sums of A'DA, A'e, A'(lu - ll)
      for (int j = tid; j < np; j += nt) {
        double s = qv[j];
#pragma unroll
        for (int d = 0; d <= K; ++d) {
          int jm = j - d; if (jm < 0) jm += np;
          s = fma(Pc[j * K1 + d], xs[jm], s);
          if (d > 0) { int jp = j + d; if (jp >= np) jp -= np; s = fma(Pc[jp * K1 + d], xs[jp], s); }
        }
        rdP[j] = s;
      }
      for (int q = tid; q < K * np; q += nt) W[q] = 0.0;
      double mu = 0.0, rpmax = 0.0, amin = 1.0;
      {
        double acc[NO];
#pragma unroll
        for (int o = 0; o < NO; ++o) acc[o] = 0.0;
        if (has) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            if (r < cnt) {
              const double* __restrict__ Ar = a.A + (size_t)(row0 + r) * K1;
              double Av[K1];
#pragma unroll
              for (int al = 0; al < K1; ++al) Av[al] = Ar[al];
              const double ql = ll[r] / sl[r], qu = lu[r] / su[r];
              const double dm = ql + qu, e = qu * rpu[r] - ql * rpl[r], dl = lu[r] - ll[r];
              mu += sl[r] * ll[r] + su[r] * lu[r];
              rpmax = fmax(rpmax, fmax(fabs(rpl[r]), fabs(rpu[r])));
              int o = 0;
#pragma unroll
              for (int al = 0; al < K1; ++al) {
                const double wa = dm * Av[al];
#pragma unroll
                for (int be = 0; be <= al; ++be) { acc[o] = fma(wa, Av[be], acc[o]); ++o; }
                acc[NE + al] = fma(Av[al], e, acc[NE + al]);
                acc[NE + K1 + al] = fma(Av[al], dl, acc[NE + K1 + al]);
              }
            }
          }
        }
        reduce3(mu, rpmax, amin);
        flush_outputs<NO>(acc, has, tid, nt, np, sch, Spart, Ssum, NO, 0);
      }
      const double mu_sum = mu;
      mu /= (double)(2 * N);
      // ---- dual residual, first right-hand side, normal matrix into the band + border storage
      double rdmax = 0.0;
      for (int j = tid; j < np; j += nt) {
        double s1 = 0.0, s2 = 0.0;
        for (int al = 0; al <= K; ++al) {
          const int base = span_of(j, al) * NO + NE + al;
          s1 += Ssum[base]; s2 += Ssum[base + K1];
        }
        const double r = rdP[j] + s2;
        rd[j] = r;
        rhs[j] = s1 - rdP[j];
        rdmax = fmax(rdmax, fabs(r));
      }
      for (int task = tid; task < np * K1; task += nt) {
        const int j = task / K1, d = task - j * K1;
        double sum = Pc[task];
        for (int al = d; al <= K; ++al) sum += Ssum[span_of(j, al) * NO + al * (al + 1) / 2 + (al - d)];
        int r = j, c = j - d;
        if (c < 0) { r = c + np; c = j; }
        if (r < m) Lb[r * K1 + (r - c)] = sum; else W[(r - m) * np + c] = sum;
      }
      double d0 = 0.0, d1 = 0.0;
      reduce3(d0, rdmax, d1);
      (void)rdmax; (void)rpmax; (void)qinf;   // the residuals no longer decide anything (rl_device.hpp: ipm_done)
      if (ipm_done(kIpmTolOneOffset, outer + 1 >= a.n_outer, mu)) break;
      ++total_it;
      // ---- factor, affine direction
      if (wave == 0) {
        band_factor<K>(Lb, W, dinv, np, lane);
        band_solve<K>(Lb, W, dinv, np, lane, rhs, dxs);
      }
      __syncthreads();
      double pl[R], pu[R];
      double c1 = 0.0, c2 = 0.0;
      amin = 1.0;
      if (has) {
        double dJ[K1];
#pragma unroll
        for (int al = 0; al < K1; ++al) dJ[al] = dxs[J[al]];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (r < cnt) {
            const double* __restrict__ Ar = a.A + (size_t)(row0 + r) * K1;
            double adx = 0.0;
#pragma unroll
            for (int al = 0; al < K1; ++al) adx = fma(Ar[al], dJ[al], adx);
            const double d_sl = adx + rpl[r], d_su = rpu[r] - adx;
            const double d_ll = (-(sl[r] * ll[r]) - ll[r] * d_sl) / sl[r];
            const double d_lu = (-(su[r] * lu[r]) - lu[r] * d_su) / su[r];
            if (d_sl < 0.0) amin = fmin(amin, 0.995 * (-sl[r] / d_sl));
            if (d_su < 0.0) amin = fmin(amin, 0.995 * (-su[r] / d_su));
            if (d_ll < 0.0) amin = fmin(amin, 0.995 * (-ll[r] / d_ll));
            if (d_lu < 0.0) amin = fmin(amin, 0.995 * (-lu[r] / d_lu));
            pl[r] = d_sl * d_ll; pu[r] = d_su * d_lu;
            c1 += sl[r] * d_ll + ll[r] * d_sl + su[r] * d_lu + lu[r] * d_su;
            c2 += pl[r] + pu[r];
          }
        }
      }
      // mu_aff = sum (s + alpha ds)(l + alpha dl) / 2N, expanded in alpha so that one reduction does
      double c2s = c2, dmx = 0.0;
      reduce3(c1, dmx, amin);
      {
        double z0 = 0.0, z1 = 0.0;
        reduce3(c2s, z0, z1);
      }
      const double mu_aff = (mu_sum + amin * c1 + amin * amin * c2s) / (double)(2 * N);
      const double ratio = mu_aff / mu, sigma = ratio * ratio * ratio;
      const double smu = sigma * mu;
      // ---- corrector right-hand side
      {
        double acc[K1];
#pragma unroll
        for (int al = 0; al < K1; ++al) acc[al] = 0.0;
        if (has) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            if (r < cnt) {
              const double* __restrict__ Ar = a.A + (size_t)(row0 + r) * K1;
              const double rcl = sl[r] * ll[r] - smu + pl[r], rcu = su[r] * lu[r] - smu + pu[r];
              const double wv = (-rcl - ll[r] * rpl[r]) / sl[r] - (-rcu - lu[r] * rpu[r]) / su[r];
#pragma unroll
              for (int al = 0; al < K1; ++al) acc[al] = fma(Ar[al], wv, acc[al]);
            }
          }
        }
        flush_outputs<K1>(acc, has, tid, nt, np, sch, Spart, Ssum, NO, NE);
      }
      for (int j = tid; j < np; j += nt) {
        double s1 = 0.0;
        for (int al = 0; al <= K; ++al) s1 += Ssum[span_of(j, al) * NO + NE + al];
        rhs[j] = s1 - rd[j];
      }
      __syncthreads();
      if (wave == 0) band_solve<K>(Lb, W, dinv, np, lane, rhs, dxs);
      __syncthreads();
      // ---- step length and update
      double adxv[R], dllv[R], dluv[R];
      amin = 1.0;
      if (has) {
        double dJ[K1];
#pragma unroll
        for (int al = 0; al < K1; ++al) dJ[al] = dxs[J[al]];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (r < cnt) {
            const double* __restrict__ Ar = a.A + (size_t)(row0 + r) * K1;
            double adx = 0.0;
#pragma unroll
            for (int al = 0; al < K1; ++al) adx = fma(Ar[al], dJ[al], adx);
            const double d_sl = adx + rpl[r], d_su = rpu[r] - adx;
            const double rcl = sl[r] * ll[r] - smu + pl[r], rcu = su[r] * lu[r] - smu + pu[r];
            const double d_ll = (-rcl - ll[r] * d_sl) / sl[r], d_lu = (-rcu - lu[r] * d_su) / su[r];
            if (d_sl < 0.0) amin = fmin(amin, 0.995 * (-sl[r] / d_sl));
            if (d_su < 0.0) amin = fmin(amin, 0.995 * (-su[r] / d_su));
            if (d_ll < 0.0) amin = fmin(amin, 0.995 * (-ll[r] / d_ll));
            if (d_lu < 0.0) amin = fmin(amin, 0.995 * (-lu[r] / d_lu));
            adxv[r] = adx; dllv[r] = d_ll; dluv[r] = d_lu;
          }
        }
      }
      {
        double z0 = 0.0, z1 = 0.0;
        reduce3(z0, z1, amin);
      }
      const double alpha = amin;
      if (has) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (r < cnt) {
            sl[r] = fma(alpha, adxv[r] + rpl[r], sl[r]);
            su[r] = fma(alpha, rpu[r] - adxv[r], su[r]);
            ll[r] = fma(alpha, dllv[r], ll[r]);
            lu[r] = fma(alpha, dluv[r], lu[r]);
            rpl[r] *= 1.0 - alpha; rpu[r] *= 1.0 - alpha;
          }
        }
      }
      for (int j = tid; j < np; j += nt) xs[j] = fma(alpha, dxs[j], xs[j]);
      __syncthreads();
    }
    // ---- accept the QP solution as the next linearisation point
    double stepmax = 0.0, z0 = 0.0, z1 = 0.0;
    for (int j = tid; j < np; j += nt) stepmax = fmax(stepmax, fabs(xs[j] - avs[j]));
    reduce3(z0, stepmax, z1);
    last_step = stepmax;
    for (int j = tid; j < np; j += nt) avs[j] = xs[j];
    __syncthreads();
  }

  // ---- outputs: control points, line samples, offsets, statistics
  for (int j = tid; j < n; j += nt)
    reinterpret_cast<double2*>(a.out_ctrl)[(size_t)b * n + j] = make_double2(cxs[j], cys[j]);
  if (a.out_a)
    for (int j = tid; j < np; j += nt) a.out_a[(size_t)b * np + j] = avs[j];
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
      const double2 w = reinterpret_cast<const double2*>(wid)[i];
      viol = fmax(viol, fmax(-(w.y - a.margin) - lat, lat - (w.x - a.margin)));
      if (lat + (w.y - a.margin) < 1e-6 || (w.x - a.margin) - lat < 1e-6) nact += 1.0;
    }
  }
  double z1 = 0.0;
  reduce3(nact, viol, z1);
  if (tid == 0) {
    double* st = a.out_stats + (size_t)b * 8;
    st[0] = (double)total_it; st[1] = k2_first; st[2] = k2_last; st[3] = viol; st[4] = last_step;
    st[5] = nact; st[6] = 0.0; st[7] = 0.0;
  }
}

}  // namespace rl
