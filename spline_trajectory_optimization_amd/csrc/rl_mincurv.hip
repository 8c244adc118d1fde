// rl_mincurv.hip -- C ABI (include/rl_mincurv.h) over the HIP kernels in rl_kernels.hpp.
// Built for gfx950 only:  hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC
//
// There is deliberately no CPU path in this library: every entry point launches HIP kernels and
// fails with RL_ERR_HIP when no device is usable.
#include "../../include/rl_mincurv.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "rl_kernels.hpp"
#include "rl_qss_df.hpp"
#include "rl_global.hpp"
#include "rl_global2.hpp"
#include "rl_global_xy.hpp"
#include "rl_dtrack.hpp"
#include "rl_mintime.hpp"

namespace {

thread_local std::string g_err;
double* g_dbg_buf = nullptr;   // rl_debug_dump_enable: step / window dump of the sweep kernels (tests)
size_t g_dbg_len = 0;
int g_dbg_instances = 0;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define RL_HIP(expr)                                                                      \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      return fail(RL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));         \
    }                                                                                     \
  } while (0)

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    release();
    n = count;
    if (count == 0) return hipSuccess;
    return hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  ~DevBuf() { release(); }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

}  // namespace

struct rl_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int max_lds = 65536;
  int num_cu = 0;
  bool force_global_v1 = false;  // test hook: RL_GLOBAL_V1=1 keeps the generic kernel
  int arith = RL_ARITH_REFERENCE;   // rl_ctx_set_arith: arithmetic of the sweep; the reference-order arithmetic since round 6
  bool arith_explicit = false;      // set by rl_ctx_set_arith / RL_ARITH: an explicit choice fails where it does not exist, the default follows
  int np_raise_at_start = 0;     // rl_ctx_set_numpy_raise: the reference-order sweep starts with np.seterr(all='raise') in effect
  // test hooks of the QSS simulator (rl_ctx_set_option; defaults from RL_QSS_DF / RL_QSS_V1 / RL_QSS_DF_WAVES / RL_QSS_DF_BAIL_AT, read
  // once in rl_ctx_create): which kernel (-1 = by the rounds the batch takes, 0 = list order, 1 = dataflow), waves per instance of
  // the dataflow kernel, the iteration at which it hands every instance back (0 = never)
  int qss_kernel = -1, qss_df_waves = 4, qss_df_bail_at = 0;
  // largest dynamic-LDS size already granted per kernel (hipFuncSetAttribute is issued only when a call needs more)
  // Device scratch owned by the context (grow-only): the *_dev entry points of the QSS simulator and the
  // min-time solve carve their work arrays out of it, so that steady-state calls allocate nothing.
  void* arena = nullptr;
  size_t arena_cap = 0;
  // The arena is handed out from offset 0 by every *_dev call, so two calls may only overlap in time if they are
  // ordered on the device: each call records `arena_ev` behind its last use, and a call made after
  // rl_ctx_set_stream switched to another stream first makes that stream wait for the event (Arena::begin).
  hipEvent_t arena_ev = nullptr;
  hipStream_t arena_stream = nullptr;
  bool arena_busy = false;       // arena_ev has been recorded at least once
  // second in-order queue of the min-time solve (half batches side by side), forked from / joined into `stream`
  static constexpr int kMaxGroups = 8;
  hipStream_t aux_stream[kMaxGroups - 1] = {};
  hipEvent_t ev_fork = nullptr, ev_join[kMaxGroups - 1] = {};
  bool mt_poll = false;          // set by the HOST entry point of the min-time solve around its call of the _dev one: poll for early exit
  // the poll never drains the queues: the status words of a chunk of 8 iterations are copied by a side stream into pinned
  // memory while the next chunk is already enqueued, and looked at one chunk later
  hipStream_t poll_stream = nullptr;
  hipEvent_t ev_chunk[kMaxGroups] = {}, ev_poll[2] = {nullptr, nullptr};
  double* poll_host = nullptr;   // pinned, 2 x poll_cap doubles
  size_t poll_cap = 0;
  bool mt_hes_sweep = false;     // RL_MT_HES_SWEEP=1: the Hessian by k_mt_derivs<2> instead of the chain-rule kernels
  bool mt_kkt4 = true;           // RL_MT_KKT4=0: the elimination with two fronts per instance (k_mt_kkt) instead of four (k_mt_kkt4, N >= 64)
  bool mt_unfused = false;       // RL_MT_UNFUSED=1: Jacobian / Hessian / block assembly by the four separate kernels instead of k_mt_node
  int mt_groups = 3;             // RL_MT_GROUPS=1..8 (round 3, 1024 instances: 1 / 2 / 3 / 4 / 8 streams 1.23 / 1.29 / 1.21 / 1.19 / 1.19 s)
  // Device staging blocks of the HOST-pointer entry points (PoolBuf): handed out best-fit, returned at the end of
  // the call, freed with the context -- a second call of the same shape allocates nothing.
  struct PoolBlock { void* p; size_t cap; bool used; };
  std::vector<PoolBlock> pool;
};

namespace {
// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the FUNCTION on a device, shared by every context of the
// process: one process-wide table per (device, kernel) that only ever raises the value, instead of one call per launch
// (a per-context cache would go stale as soon as a second context asked for less).
hipError_t grant_dyn_lds(rl_ctx* ctx, const void* fn, size_t bytes) {
  static std::mutex mu;
  static std::vector<std::pair<std::pair<int, const void*>, size_t>> granted;
  std::lock_guard<std::mutex> lock(mu);
  const std::pair<int, const void*> key(ctx->device, fn);
  for (auto& e : granted)
    if (e.first == key) {
      if (e.second >= bytes) return hipSuccess;
      const hipError_t r = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (r == hipSuccess) e.second = bytes;
      return r;
    }
  const hipError_t r = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (r == hipSuccess) granted.emplace_back(key, bytes);
  return r;
}
// per-call device staging of the host-pointer entry points, from the context's pool (every such entry point ends
// with a stream synchronisation, so a block is idle when it is returned)
template <typename T>
struct PoolBuf {
  rl_ctx* ctx;
  T* p = nullptr;
  size_t n = 0;
  int slot = -1;
  explicit PoolBuf(rl_ctx* c) : ctx(c) {}
  hipError_t alloc(size_t count) {
    release();
    n = count;
    if (count == 0) return hipSuccess;
    const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
    int best = -1;
    for (int i = 0; i < (int)ctx->pool.size(); ++i)
      if (!ctx->pool[i].used && ctx->pool[i].cap >= bytes && ctx->pool[i].cap <= 4 * bytes + 4096 &&   // never park a small request on a huge block
          (best < 0 || ctx->pool[i].cap < ctx->pool[best].cap)) best = i;
    if (best < 0) {
      if (ctx->pool.size() >= 96) {   // shapes keep changing: drop what is idle before growing further
        for (auto& blk : ctx->pool)
          if (!blk.used && blk.p) { (void)hipFree(blk.p); blk.p = nullptr; blk.cap = 0; }
      }
      void* q = nullptr;
      hipError_t e = hipMalloc(&q, bytes);
      if (e != hipSuccess) {          // out of memory: give back what is idle, then try once more
        (void)hipGetLastError();
        for (auto& blk : ctx->pool)
          if (!blk.used && blk.p) { (void)hipFree(blk.p); blk.p = nullptr; blk.cap = 0; }
        e = hipMalloc(&q, bytes);
      }
      if (e != hipSuccess) return e;
      for (int i = 0; i < (int)ctx->pool.size() && best < 0; ++i)
        if (!ctx->pool[i].p) { ctx->pool[i] = {q, bytes, false}; best = i; }
      if (best < 0) { ctx->pool.push_back({q, bytes, false}); best = (int)ctx->pool.size() - 1; }
    }
    ctx->pool[best].used = true;
    slot = best;
    p = static_cast<T*>(ctx->pool[best].p);
    return hipSuccess;
  }
  void release() {
    if (slot >= 0) ctx->pool[slot].used = false;
    slot = -1; p = nullptr; n = 0;
  }
  // Success paths end with a stream synchronisation, so the block is idle here.  An early error return does not: wait for
  // whatever was already enqueued (copies into / out of this block) before the block can be handed to the next call.
  ~PoolBuf() {
    if (slot >= 0 && hipStreamQuery(ctx->stream) == hipErrorNotReady) (void)hipStreamSynchronize(ctx->stream);
    release();
  }
  PoolBuf(const PoolBuf&) = delete;
  PoolBuf& operator=(const PoolBuf&) = delete;
};

// sequential carve-out of the context's scratch arena; reserve() first with the total
struct Arena {
  rl_ctx* ctx; size_t off = 0;
  explicit Arena(rl_ctx* c) : ctx(c) {}
  static size_t pad(size_t b) { return (b + 255) & ~(size_t)255; }
  // order this call behind the previous user of the arena if that one ran on another stream
  hipError_t begin() {
    if (!ctx->arena_ev) {
      const hipError_t e = hipEventCreateWithFlags(&ctx->arena_ev, hipEventDisableTiming);
      if (e != hipSuccess) return e;
    }
    if (ctx->arena_busy && ctx->arena_stream != ctx->stream) return hipStreamWaitEvent(ctx->stream, ctx->arena_ev, 0);
    return hipSuccess;
  }
  // record "the arena is free again once everything enqueued so far on the context's stream has run"
  hipError_t end() {
    const hipError_t e = hipEventRecord(ctx->arena_ev, ctx->stream);
    if (e == hipSuccess) { ctx->arena_busy = true; ctx->arena_stream = ctx->stream; }
    return e;
  }
  hipError_t reserve(size_t bytes) {
    hipError_t e = begin();
    if (e != hipSuccess) return e;
    if (bytes <= ctx->arena_cap) return hipSuccess;
    if (ctx->arena_busy) { e = hipEventSynchronize(ctx->arena_ev); if (e != hipSuccess) return e; }   // the last user, whatever its stream
    e = hipStreamSynchronize(ctx->stream);                     // nothing in flight may still use the old block
    if (e != hipSuccess) return e;
    if (ctx->arena) (void)hipFree(ctx->arena);
    ctx->arena = nullptr; ctx->arena_cap = 0;
    e = hipMalloc(&ctx->arena, bytes);
    if (e == hipSuccess) ctx->arena_cap = bytes;
    return e;
  }
  template <typename T> T* take(size_t count) {
    T* p = reinterpret_cast<T*>(static_cast<char*>(ctx->arena) + off);
    off += pad(count * sizeof(T));
    return p;
  }
};
}  // namespace

struct rl_track {
  rl_ctx* ctx = nullptr;
  int k = 0, n = 0, nt = 0, N = 0;
  std::vector<double> t_host;
  DevBuf<double> t, c0, D, base;
  DevBuf<int> ell, sup;
  // reference-order mode (RL_ARITH_REFERENCE): unfused de Boor tables and the initial line with its two normal
  // directions, built by the first solve that asks for them, rebuilt when the control points change
  mutable DevBuf<double> Ds, base_s;
  mutable bool strict_valid = false;
  mutable DevBuf<double> aux;   // reference-order sweep, per instance: table control points + per-sample flags (numpy's error state)
  DevBuf<double> ringL, ringR;  // shared rings as (x,y) pairs
  int nL = 0, nR = 0;
  double length = 0.0;          // BSplineTrajectory._length of the INITIAL spline (rl_track_set_length); 0 = not given
  // Device scratch owned by the track (sweep residency variant, global-QP tables): a track is therefore
  // SINGLE-STREAM -- two contexts / streams must not run solves on the same rl_track concurrently
  // (include/rl_mincurv.h, "Threading").
  mutable DevBuf<double> gscratch;  // for the non-LDS-resident sweep variant
  // tables of the global QP (a15), built on first use and dropped when the centre line changes
  mutable DevBuf<double> gq_A, gq_nu;
  mutable DevBuf<int> gq_chunk, gq_span;    // chunks of <= rl::kGRows rows (k_global_qp)
  mutable DevBuf<int> gq2_chunk, gq2_span;  // chunks of <= gq2_rows rows (k_global_qp2), if they fit
  mutable int gq_nc = 0, gq2_nc = 0, gq2_rows = 0;  // gq2_rows = 0: the fast path does not fit
  mutable bool gq_valid = false;
  // tables of the two-coordinate global QP (rl_global_xy.hpp)
  mutable DevBuf<int> gxy_span, gxy_chunk, gxy_sch0;
  mutable int gxy_nch = 0;
  mutable DevBuf<double> gxy_bbx;
  mutable bool gxy_valid = false;
  rl::TrackDev dev() const {
    rl::TrackDev d;
    d.k = k; d.n = n; d.nt = nt; d.N = N;
    d.t = t.p; d.c0 = c0.p; d.ell = ell.p; d.D = D.p; d.sup = sup.p; d.base = base.p;
    d.Ds = Ds.p; d.base_s = base_s.p;
    return d;
  }
};

namespace {

bool degree_supported(int k) { return k == 3 || k == 5; }

// dispatch a callable templated on the spline degree
template <typename F3, typename F5>
int by_degree(int k, F3&& f3, F5&& f5) {
  if (k == 3) return f3();
  if (k == 5) return f5();
  return fail(RL_ERR_UNSUPPORTED, "spline degree must be 3 or 5");
}

int check_spline(const double* t, int nt, const double* cx, const double* cy, int k) {
  if (!t || !cx || !cy) return fail(RL_ERR_ARG, "null spline pointer");
  if (!degree_supported(k)) return fail(RL_ERR_UNSUPPORTED, "spline degree must be 3 or 5");
  if (nt < 2 * (k + 1)) return fail(RL_ERR_ARG, "knot vector too short");
  return RL_OK;
}

struct SweepPlan {
  bool rings_in_lds;
  bool sigma_in_lds;
  size_t lds_bytes;
  size_t gscratch_doubles;  // per instance
  int block;
};

SweepPlan plan_sweep(const rl_ctx* ctx, int n, int N, int nL, int nR, int B, bool joint = false, bool strict = false) {
  SweepPlan p;
  p.block = 256;
  // Residency of the per-instance state (ring vertices 16 B x (nL + nR), crossings 8 B x 2N):
  //   all in LDS      : lowest latency per step, but ~120 KB at N = 2000 -> one workgroup per CU;
  //   all in global   : ~31 KB of LDS -> four workgroups per CU (VGPR-limited), state served by L1/L2;
  //   crossings in LDS, rings in global (RL_FORCE_RESIDENCY=2 only): ~63 KB -> two workgroups per CU, no
  //                     global writes inside the sweep (measured slower than all-global; DESIGN_HISTORY.md section 6).
  // The kernel is FP64-issue / latency bound, so once a batch offers more than ~2 workgroups per
  // CU the occupancy wins (measured: 12.5 ms vs 20.2 ms for 1024 instances, N = 2000).
  // RL_FORCE_GLOBAL_RINGS=0/1 and RL_FORCE_RESIDENCY=0/1/2 override (tests cover the variants).
  const char* force = getenv("RL_FORCE_GLOBAL_RINGS");
  int want = B >= 2 * ctx->num_cu ? 0 : 1;    // 0 all global, 1 all LDS, 2 crossings in LDS
  if (force && (force[0] == '0' || force[0] == '1')) want = force[0] == '1' ? 0 : 1;
  if (const char* r = getenv("RL_FORCE_RESIDENCY")) if (r[0] >= '0' && r[0] <= '2') want = r[0] - '0';
  if (strict && want == 2) want = 0;   // the reference-order mode exists all-LDS and all-global
  rl::SweepLds in = rl::sweep_lds_layout(n, N, nL, nR, true, true, joint, strict);
  if (want == 1 && in.total * sizeof(double) > (size_t)ctx->max_lds) want = 0;
  if (want == 2) {
    rl::SweepLds mid = rl::sweep_lds_layout(n, N, nL, nR, false, true, joint);
    if (mid.total * sizeof(double) > (size_t)ctx->max_lds) want = 0;
  }
  p.rings_in_lds = want == 1;
  p.sigma_in_lds = want != 0;
  rl::SweepLds L = rl::sweep_lds_layout(n, N, nL, nR, p.rings_in_lds, p.sigma_in_lds, joint, strict);
  p.lds_bytes = L.total * sizeof(double);
  // crossings (fast mode: one double per sample and side) or bound points (reference-order mode: two)
  p.gscratch_doubles = (p.sigma_in_lds ? 0 : (size_t)2 * ((N + 1) & ~1) * (strict ? 2 : 1)) +
                       (p.rings_in_lds ? 0 : (size_t)2 * (nL + rl::kRingPad) + (size_t)2 * (nR + rl::kRingPad));
  return p;
}

template <int K, int BLOCK, bool RL, bool JOINT = false, bool DUMP = false, bool SL = RL, bool STRICT = false, bool RAISE = false,
          bool LITE = false>
int launch_sweep_t(const rl_ctx* ctx, const rl::SweepArgs& a, size_t lds) {
  auto kern = rl::k_sweep<K, BLOCK, RL, JOINT, DUMP, SL, STRICT, RAISE, LITE>;
  RL_HIP(grant_dyn_lds(const_cast<rl_ctx*>(ctx), reinterpret_cast<const void*>(kern), lds));
  hipLaunchKernelGGL(kern, dim3(a.B), dim3(BLOCK), lds, ctx->stream, a);
  RL_HIP(hipGetLastError());
  return RL_OK;
}

int launch_sweep(const rl_ctx* ctx, int k, const SweepPlan& p, const rl::SweepArgs& a, bool joint = false, bool strict = false,
                 bool lite = false) {
  if (strict) {   // RL_ARITH_REFERENCE / _BRANCH: degree-5 splines (the reference's wrap is written for k = 5, optimizer.py:281-285)
#ifdef RL_STAMPS
    if (k == 5 && !joint && !lite && !p.rings_in_lds)   // diagnostic build: the plain reference-order kernel with its phase stamps in a.dbg
      return launch_sweep_t<5, 256, false, false, false, false, true>(ctx, a, p.lds_bytes);
#endif
    if (k != 5 || a.dbg || (joint && lite)) return fail(RL_ERR_UNSUPPORTED, "reference-order / branch arithmetic: degree-5 splines, no step dump; the sliding-window driver in the reference-order arithmetic only");
    if (joint)    // run_joint_min_curvature_qp in the reference-order arithmetic
      return p.rings_in_lds ? launch_sweep_t<5, 256, true, true, false, true, true>(ctx, a, p.lds_bytes)
                            : launch_sweep_t<5, 256, false, true, false, false, true>(ctx, a, p.lds_bytes);
    if (lite)     // RL_ARITH_BRANCH
      return p.rings_in_lds ? launch_sweep_t<5, 256, true, false, false, true, true, false, true>(ctx, a, p.lds_bytes)
                            : launch_sweep_t<5, 256, false, false, false, false, true, false, true>(ctx, a, p.lds_bytes);
    // the plain kernel flags the instances that need numpy's error state modelled (a sample of exactly zero curvature while
    // numpy is in raise mode); the RAISE instantiation behind it redoes those and returns at once on all others
    if (int rc = p.rings_in_lds ? launch_sweep_t<5, 256, true, false, false, true, true>(ctx, a, p.lds_bytes)
                                : launch_sweep_t<5, 256, false, false, false, false, true>(ctx, a, p.lds_bytes)) return rc;
    return p.rings_in_lds ? launch_sweep_t<5, 256, true, false, false, true, true, true>(ctx, a, p.lds_bytes)
                          : launch_sweep_t<5, 256, false, false, false, false, true, true>(ctx, a, p.lds_bytes);
  }
  if (joint) {
    if (k != 5) return fail(RL_ERR_UNSUPPORTED, "the sliding-window variant is built for degree 5 (span 5)");
    return p.rings_in_lds ? launch_sweep_t<5, 256, true, true>(ctx, a, p.lds_bytes)
                          : launch_sweep_t<5, 256, false, true>(ctx, a, p.lds_bytes);
  }
#ifdef RL_ABLATION
  if (k == 5 && !joint && !p.rings_in_lds && !a.dbg && getenv("RL_SWEEP_BLOCK") && atoi(getenv("RL_SWEEP_BLOCK")) == 512)
    return p.sigma_in_lds ? launch_sweep_t<5, 512, false, false, false, true>(ctx, a, p.lds_bytes)
                          : launch_sweep_t<5, 512, false>(ctx, a, p.lds_bytes);
#endif
  if (k == 5 && !joint && !p.rings_in_lds && p.sigma_in_lds)
    return launch_sweep_t<5, 256, false, false, false, true>(ctx, a, p.lds_bytes);
#ifdef RL_STAMPS
  if (k == 5 && !p.rings_in_lds && !p.sigma_in_lds) return launch_sweep_t<5, 256, false>(ctx, a, p.lds_bytes);  // stamps go to a.dbg
#endif
  if (k == 5) {
    if (a.dbg)  // recording instantiation (test aid): same source, one extra store block per step
      return p.rings_in_lds ? launch_sweep_t<5, 256, true, false, true>(ctx, a, p.lds_bytes)
                            : launch_sweep_t<5, 256, false, false, true>(ctx, a, p.lds_bytes);
    return p.rings_in_lds ? launch_sweep_t<5, 256, true>(ctx, a, p.lds_bytes)
                          : launch_sweep_t<5, 256, false>(ctx, a, p.lds_bytes);
  }
  if (k == 3) {
    return p.rings_in_lds ? launch_sweep_t<3, 256, true>(ctx, a, p.lds_bytes)
                          : launch_sweep_t<3, 256, false>(ctx, a, p.lds_bytes);
  }
  return fail(RL_ERR_UNSUPPORTED, "spline degree must be 3 or 5");
}

// tables of the reference-order mode (rl_kernels.hpp: k_build_tables_strict), built on first use
int ensure_strict_tables(const rl_ctx* ctx, const rl_track* trk) {
  if (trk->strict_valid) return RL_OK;
  const int k = trk->k, N = trk->N;
  if (k != 5) return fail(RL_ERR_UNSUPPORTED, "reference-order arithmetic: degree-5 splines");
  if (!trk->Ds.p) RL_HIP(trk->Ds.alloc((size_t)rl::StrictRows<5>::total * N));
  if (!trk->base_s.p) RL_HIP(trk->base_s.alloc((size_t)6 * N));
  const dim3 grid((N + 127) / 128), block(128);
  hipLaunchKernelGGL(rl::k_build_tables_strict<5>, grid, block, 0, ctx->stream, trk->t.p, trk->nt, trk->c0.p, N,
                     trk->Ds.p, trk->base_s.p);
  RL_HIP(hipGetLastError());
  // like the other table builders: complete (and known to have succeeded) before the tables count as valid -- a later call
  // may arrive on another stream (rl_ctx_set_stream), with no event ordering it behind this launch
  RL_HIP(hipStreamSynchronize(ctx->stream));
  trk->strict_valid = true;
  return RL_OK;
}

}  // namespace

static int sweep_single(rl_ctx* ctx, rl_track* trk, const int* i_start, int max_iter, double* cx, double* cy,
                        double* points, int* n_success, rl_stats* stats, bool joint);

extern "C" {

int rl_version(void) { return RL_VERSION; }

int rl_debug_dump_enable(int instances) {
  if (instances < 0) return fail(RL_ERR_ARG, "instances < 0");
  g_dbg_instances = instances;
  return RL_OK;
}

int rl_debug_read(double* out, long long n) {
  if (!g_dbg_buf || !out || n < 0 || (size_t)n > g_dbg_len) return fail(RL_ERR_ARG, "no debug dump of that size");
  RL_HIP(hipMemcpy(out, g_dbg_buf, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  return RL_OK;
}

const char* rl_last_error(void) { return g_err.c_str(); }

int rl_ctx_create(int device_id, rl_ctx** out) {
  if (!out) return fail(RL_ERR_ARG, "out is null");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(RL_ERR_HIP, "no HIP device available (this library has no CPU fallback)");
  if (device_id < 0 || device_id >= count) return fail(RL_ERR_ARG, "bad device id");
  RL_HIP(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  RL_HIP(hipGetDeviceProperties(&prop, device_id));
  rl_ctx* c = new rl_ctx();
  c->device = device_id;
  c->max_lds = (int)prop.sharedMemPerBlock;  // 160 KiB on gfx950
  {
    int optin = 0;
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, device_id) == hipSuccess &&
        optin > c->max_lds)
      c->max_lds = optin;
  }
  c->num_cu = prop.multiProcessorCount;
  if (const char* v = getenv("RL_GLOBAL_V1")) c->force_global_v1 = v[0] == '1';
  if (const char* v = getenv("RL_ARITH")) {
    c->arith = (v[0] == 'r' || v[0] == '1') ? RL_ARITH_REFERENCE : ((v[0] == 'b' || v[0] == '2') ? RL_ARITH_BRANCH : RL_ARITH_FAST);
    c->arith_explicit = true;
  }
  if (const char* v = getenv("RL_QSS_DF")) if (v[0] == '0' || v[0] == '1') c->qss_kernel = v[0] - '0';
  if (const char* v = getenv("RL_QSS_V1")) if (v[0] == '1') c->qss_kernel = 0;
  if (const char* v = getenv("RL_QSS_DF_WAVES")) { const int w = atoi(v); if (w == 1 || w == 2 || w == 4) c->qss_df_waves = w; }
  if (const char* v = getenv("RL_QSS_DF_BAIL_AT")) { const int g = atoi(v); if (g >= 0) c->qss_df_bail_at = g; }
  if (const char* v = getenv("RL_MT_HES_SWEEP")) c->mt_hes_sweep = v[0] == '1';
  if (const char* v = getenv("RL_MT_UNFUSED")) c->mt_unfused = v[0] == '1';
  if (const char* v = getenv("RL_MT_KKT4")) c->mt_kkt4 = v[0] == '1';
  if (const char* v = getenv("RL_MT_GROUPS")) { const int g = atoi(v); if (g >= 1 && g <= rl_ctx::kMaxGroups) c->mt_groups = g; }
  if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    delete c;
    return fail(RL_ERR_HIP, "hipEventCreate failed");
  }
  *out = c;
  return RL_OK;
}

void rl_ctx_destroy(rl_ctx* ctx) {
  if (!ctx) return;
  if (ctx->arena) (void)hipFree(ctx->arena);
  if (ctx->arena_ev) (void)hipEventDestroy(ctx->arena_ev);
  for (auto& blk : ctx->pool)
    if (blk.p) (void)hipFree(blk.p);
  for (int g = 0; g < rl_ctx::kMaxGroups - 1; ++g) {
    if (ctx->aux_stream[g]) (void)hipStreamDestroy(ctx->aux_stream[g]);
    if (ctx->ev_join[g]) (void)hipEventDestroy(ctx->ev_join[g]);
  }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->poll_stream) (void)hipStreamDestroy(ctx->poll_stream);
  for (auto& e : ctx->ev_chunk) if (e) (void)hipEventDestroy(e);
  for (auto& e : ctx->ev_poll) if (e) (void)hipEventDestroy(e);
  if (ctx->poll_host) (void)hipHostFree(ctx->poll_host);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  delete ctx;
}

int rl_ctx_set_stream(rl_ctx* ctx, void* hip_stream) {
  if (!ctx) return fail(RL_ERR_ARG, "ctx is null");
  ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
  return RL_OK;
}

int rl_ctx_set_arith(rl_ctx* ctx, int arith) {
  if (!ctx) return fail(RL_ERR_ARG, "ctx is null");
  if (arith == RL_ARITH_DEFAULT) { ctx->arith = RL_ARITH_REFERENCE; ctx->arith_explicit = false; return RL_OK; }
  if (arith != RL_ARITH_FAST && arith != RL_ARITH_REFERENCE && arith != RL_ARITH_BRANCH)
    return fail(RL_ERR_ARG, "arith must be RL_ARITH_DEFAULT, RL_ARITH_FAST, RL_ARITH_REFERENCE or RL_ARITH_BRANCH");
  ctx->arith = arith;
  ctx->arith_explicit = true;
  return RL_OK;
}

int rl_ctx_get_arith(const rl_ctx* ctx) { return ctx ? (ctx->arith_explicit ? ctx->arith : RL_ARITH_DEFAULT) : RL_ERR_ARG; }

int rl_ctx_set_option(rl_ctx* ctx, const char* name, int value) {
  if (!ctx || !name) return fail(RL_ERR_ARG, "null argument");
  const std::string k(name);
  if (k == "qss_kernel") { if (value < -1 || value > 1) return fail(RL_ERR_ARG, "qss_kernel: -1 (auto), 0 (list order), 1 (dataflow)"); ctx->qss_kernel = value; }
  else if (k == "qss_df_waves") { if (value != 1 && value != 2 && value != 4) return fail(RL_ERR_ARG, "qss_df_waves: 1, 2 or 4"); ctx->qss_df_waves = value; }
  else if (k == "qss_df_bail_at") { if (value < 0) return fail(RL_ERR_ARG, "qss_df_bail_at >= 0"); ctx->qss_df_bail_at = value; }
  else return fail(RL_ERR_ARG, "unknown option '" + k + "'");
  return RL_OK;
}

int rl_ctx_set_numpy_raise(rl_ctx* ctx, int on) {
  if (!ctx) return fail(RL_ERR_ARG, "ctx is null");
  ctx->np_raise_at_start = on ? 1 : 0;
  return RL_OK;
}

int rl_debug_cr_heading(rl_ctx* ctx, const double* dx, const double* dy, int n, double* out) {
  if (!ctx || !dx || !dy || !out || n <= 0) return fail(RL_ERR_ARG, "bad argument");
  RL_HIP(hipSetDevice(ctx->device));
  PoolBuf<double> ddx(ctx), ddy(ctx), dout(ctx);
  RL_HIP(ddx.alloc(n)); RL_HIP(ddy.alloc(n)); RL_HIP(dout.alloc((size_t)5 * n));
  RL_HIP(hipMemcpyAsync(ddx.p, dx, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  RL_HIP(hipMemcpyAsync(ddy.p, dy, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(rl::k_debug_cr_heading, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ddx.p, ddy.p, n, dout.p);
  RL_HIP(hipGetLastError());
  RL_HIP(hipMemcpyAsync(out, dout.p, (size_t)5 * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipStreamSynchronize(ctx->stream));
  return RL_OK;
}

int rl_ctx_synchronize(rl_ctx* ctx) {
  if (!ctx) return fail(RL_ERR_ARG, "ctx is null");
  RL_HIP(hipStreamSynchronize(ctx->stream));
  return RL_OK;
}

// ------------------------------------------------------------------------------------------------
int rl_spline_eval(rl_ctx* ctx, const double* t, int nt, const double* cx, const double* cy, int k,
                   const double* u, int N, int der_max, double* out) {
  if (!ctx || !u || !out) return fail(RL_ERR_ARG, "null argument");
  if (int rc = check_spline(t, nt, cx, cy, k)) return rc;
  if (N <= 0 || der_max < 0 || der_max > 2) return fail(RL_ERR_ARG, "bad N / der_max");
  RL_HIP(hipSetDevice(ctx->device));
  const int n = nt - k - 1;
  PoolBuf<double> dt(ctx), dcx(ctx), dcy(ctx), du(ctx), dout(ctx);
  RL_HIP(dt.alloc(nt)); RL_HIP(dcx.alloc(n)); RL_HIP(dcy.alloc(n)); RL_HIP(du.alloc(N));
  RL_HIP(dout.alloc((size_t)2 * (der_max + 1) * N));
  RL_HIP(hipMemcpyAsync(dt.p, t, nt * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  RL_HIP(hipMemcpyAsync(dcx.p, cx, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  RL_HIP(hipMemcpyAsync(dcy.p, cy, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  RL_HIP(hipMemcpyAsync(du.p, u, (size_t)N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  const dim3 grid((N + 255) / 256), block(256);
  int rc = by_degree(
      k,
      [&] { hipLaunchKernelGGL(rl::k_spline_eval<3>, grid, block, 0, ctx->stream, dt.p, nt, dcx.p, dcy.p, du.p, N, der_max, dout.p); return RL_OK; },
      [&] { hipLaunchKernelGGL(rl::k_spline_eval<5>, grid, block, 0, ctx->stream, dt.p, nt, dcx.p, dcy.p, du.p, N, der_max, dout.p); return RL_OK; });
  if (rc) return rc;
  RL_HIP(hipGetLastError());
  RL_HIP(hipMemcpyAsync(out, dout.p, dout.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipStreamSynchronize(ctx->stream));
  return RL_OK;
}

int rl_sample_along(rl_ctx* ctx, const double* t, int nt, const double* cx, const double* cy, int k,
                    double length, const double* u, int N, double* points) {
  if (!ctx || !u || !points) return fail(RL_ERR_ARG, "null argument");
  if (int rc = check_spline(t, nt, cx, cy, k)) return rc;
  if (N <= 0) return fail(RL_ERR_ARG, "bad N");
  RL_HIP(hipSetDevice(ctx->device));
  const int n = nt - k - 1;
  PoolBuf<double> dt(ctx), dcx(ctx), dcy(ctx), du(ctx), dpts(ctx), dseg(ctx);
  RL_HIP(dt.alloc(nt)); RL_HIP(dcx.alloc(n)); RL_HIP(dcy.alloc(n)); RL_HIP(du.alloc(N));
  RL_HIP(dpts.alloc((size_t)N * RL_NCOL)); RL_HIP(dseg.alloc(N));
  RL_HIP(hipMemcpyAsync(dt.p, t, nt * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  RL_HIP(hipMemcpyAsync(dcx.p, cx, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  RL_HIP(hipMemcpyAsync(dcy.p, cy, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  RL_HIP(hipMemcpyAsync(du.p, u, (size_t)N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  const dim3 grid((N + 127) / 128), block(128);
  int rc = by_degree(
      k,
      [&] { hipLaunchKernelGGL(rl::k_sample_geometry<3>, grid, block, 0, ctx->stream, dt.p, nt, dcx.p, dcy.p, du.p, N, dpts.p, dseg.p); return RL_OK; },
      [&] { hipLaunchKernelGGL(rl::k_sample_geometry<5>, grid, block, 0, ctx->stream, dt.p, nt, dcx.p, dcy.p, du.p, N, dpts.p, dseg.p); return RL_OK; });
  if (rc) return rc;
  RL_HIP(hipGetLastError());
  hipLaunchKernelGGL(rl::k_sample_cumsum, dim3(1), dim3(64), 0, ctx->stream, dseg.p, N, length, dpts.p);
  RL_HIP(hipGetLastError());
  RL_HIP(hipMemcpyAsync(points, dpts.p, dpts.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipStreamSynchronize(ctx->stream));
  return RL_OK;
}

int rl_fill_bounds(rl_ctx* ctx, double* points, int N, const double* ringL, int nL,
                   const double* ringR, int nR, double max_dist) {
  if (!ctx || !points || !ringL || !ringR) return fail(RL_ERR_ARG, "null argument");
  if (N <= 0 || nL < 2 || nR < 2) return fail(RL_ERR_ARG, "bad sizes");
  RL_HIP(hipSetDevice(ctx->device));
  PoolBuf<double> dpts(ctx), dL(ctx), dR(ctx);
  RL_HIP(dpts.alloc((size_t)N * RL_NCOL)); RL_HIP(dL.alloc((size_t)2 * nL)); RL_HIP(dR.alloc((size_t)2 * nR));
  RL_HIP(hipMemcpyAsync(dpts.p, points, dpts.n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  RL_HIP(hipMemcpyAsync(dL.p, ringL, dL.n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  RL_HIP(hipMemcpyAsync(dR.p, ringR, dR.n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(rl::k_fill_bounds, dim3((2 * N + 63) / 64), dim3(64), 0, ctx->stream, dpts.p, N,
                     reinterpret_cast<const double2*>(dL.p), nL,
                     reinterpret_cast<const double2*>(dR.p), nR, max_dist);
  RL_HIP(hipGetLastError());
  RL_HIP(hipMemcpyAsync(points, dpts.p, dpts.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipStreamSynchronize(ctx->stream));
  return RL_OK;
}

// ------------------------------------------------------------------------------------------------
int rl_track_create(rl_ctx* ctx, const double* t, int nt, const double* cx0, const double* cy0,
                    int k, int N, rl_track** out) {
  if (!ctx || !out) return fail(RL_ERR_ARG, "null argument");
  if (int rc = check_spline(t, nt, cx0, cy0, k)) return rc;
  if (N <= 0) return fail(RL_ERR_ARG, "bad N");
  const int n = nt - k - 1;
  if (n < 2 * k + 1) return fail(RL_ERR_ARG, "too few control points for the periodic wrap");
  RL_HIP(hipSetDevice(ctx->device));
  rl_track* trk = new rl_track();
  trk->ctx = ctx;
  trk->k = k; trk->n = n; trk->nt = nt; trk->N = N;
  trk->t_host.assign(t, t + nt);
  // support ranges with the reference's own mask: (u >= t[j]) & (u < t[j+k+1]), u_i = i * (1/N)
  std::vector<int> sup(2 * (size_t)n);
  const double step = 1.0 / (double)N;
  for (int j = 0; j < n; ++j) {
    const double ts = t[j], te = t[j + k + 1];
    int a = 0;
    while (a < N && !((double)a * step >= ts)) ++a;
    int b = a;
    while (b < N && (double)b * step < te) ++b;
    sup[2 * j] = a;
    sup[2 * j + 1] = b;
  }
  hipError_t e = hipSuccess;
  auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  ok(trk->t.alloc(nt)); ok(trk->c0.alloc((size_t)2 * n)); ok(trk->D.alloc((size_t)3 * (k + 1) * N));
  ok(trk->base.alloc((size_t)4 * N)); ok(trk->ell.alloc(N)); ok(trk->sup.alloc((size_t)2 * n));
  if (e == hipSuccess) ok(hipMemcpyAsync(trk->t.p, t, nt * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (e == hipSuccess) ok(hipMemcpyAsync(trk->c0.p, cx0, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (e == hipSuccess) ok(hipMemcpyAsync(trk->c0.p + n, cy0, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (e == hipSuccess) ok(hipMemcpyAsync(trk->sup.p, sup.data(), sup.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  if (e == hipSuccess) {
    const dim3 grid((N + 127) / 128), block(128);
    if (k == 3)
      hipLaunchKernelGGL(rl::k_build_tables<3>, grid, block, 0, ctx->stream, trk->t.p, nt, trk->c0.p, N, trk->ell.p, trk->D.p, trk->base.p);
    else
      hipLaunchKernelGGL(rl::k_build_tables<5>, grid, block, 0, ctx->stream, trk->t.p, nt, trk->c0.p, N, trk->ell.p, trk->D.p, trk->base.p);
    ok(hipGetLastError());
    ok(hipStreamSynchronize(ctx->stream));
  }
  if (e != hipSuccess) {
    delete trk;
    return fail(RL_ERR_HIP, std::string("rl_track_create: ") + hipGetErrorString(e));
  }
  *out = trk;
  return RL_OK;
}

void rl_track_destroy(rl_track* trk) { delete trk; }

int rl_track_set_rings(rl_track* trk, const double* ringL, int nL, const double* ringR, int nR) {
  if (!trk || !ringL || !ringR) return fail(RL_ERR_ARG, "null argument");
  if (nL < 3 || nR < 3) return fail(RL_ERR_ARG, "a ring needs at least 3 vertices");
  RL_HIP(hipSetDevice(trk->ctx->device));
  // shapely's LinearRing drops an explicit closing vertex; do the same
  if (ringL[0] == ringL[2 * (nL - 1)] && ringL[1] == ringL[2 * (nL - 1) + 1]) --nL;
  if (ringR[0] == ringR[2 * (nR - 1)] && ringR[1] == ringR[2 * (nR - 1) + 1]) --nR;
  RL_HIP(trk->ringL.alloc((size_t)2 * nL));
  RL_HIP(trk->ringR.alloc((size_t)2 * nR));
  RL_HIP(hipMemcpy(trk->ringL.p, ringL, (size_t)2 * nL * sizeof(double), hipMemcpyHostToDevice));
  RL_HIP(hipMemcpy(trk->ringR.p, ringR, (size_t)2 * nR * sizeof(double), hipMemcpyHostToDevice));
  trk->nL = nL;
  trk->nR = nR;
  return RL_OK;
}

int rl_track_set_length(rl_track* trk, double length) {
  if (!trk) return fail(RL_ERR_ARG, "null argument");
  if (!(length >= 0.0)) return fail(RL_ERR_ARG, "length must be >= 0");
  trk->length = length;
  return RL_OK;
}

int rl_track_set_control_points(rl_track* trk, const double* cx0, const double* cy0) {
  if (!trk || !cx0 || !cy0) return fail(RL_ERR_ARG, "null argument");
  rl_ctx* ctx = trk->ctx;
  RL_HIP(hipSetDevice(ctx->device));
  const int n = trk->n, N = trk->N;
  RL_HIP(hipMemcpyAsync(trk->c0.p, cx0, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  RL_HIP(hipMemcpyAsync(trk->c0.p + n, cy0, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  const dim3 grid((N + 127) / 128), block(128);
  if (trk->k == 3)
    hipLaunchKernelGGL(rl::k_build_tables<3>, grid, block, 0, ctx->stream, trk->t.p, trk->nt, trk->c0.p, N, trk->ell.p, trk->D.p, trk->base.p);
  else
    hipLaunchKernelGGL(rl::k_build_tables<5>, grid, block, 0, ctx->stream, trk->t.p, trk->nt, trk->c0.p, N, trk->ell.p, trk->D.p, trk->base.p);
  RL_HIP(hipGetLastError());
  RL_HIP(hipStreamSynchronize(ctx->stream));
  trk->gq_valid = false;
  trk->gxy_valid = false;
  trk->strict_valid = false;
  return RL_OK;
}

// ------------------------------------------------------------------------------------------------
int rl_mincurv_cost(rl_ctx* ctx, const rl_track* trk, const int* idx, int n_idx, const double* z,
                    double* H, double* g, int* M) {
  if (!ctx || !trk || !idx || !H || !g) return fail(RL_ERR_ARG, "null argument");
  if (n_idx <= 0) return fail(RL_ERR_ARG, "n_idx <= 0");
  for (int q = 0; q < n_idx; ++q)
    if (idx[q] < 0 || idx[q] >= trk->n) return fail(RL_ERR_ARG, "control point index out of range");
  RL_HIP(hipSetDevice(ctx->device));
  PoolBuf<int> didx(ctx), dM(ctx);
  PoolBuf<double> dz(ctx), dH(ctx), dg(ctx);
  RL_HIP(didx.alloc(n_idx)); RL_HIP(dM.alloc(n_idx));
  RL_HIP(dH.alloc((size_t)4 * n_idx)); RL_HIP(dg.alloc((size_t)2 * n_idx));
  RL_HIP(hipMemcpyAsync(didx.p, idx, n_idx * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  if (z) {
    RL_HIP(dz.alloc((size_t)2 * n_idx));
    RL_HIP(hipMemcpyAsync(dz.p, z, (size_t)2 * n_idx * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  }
  if (ctx->arith == RL_ARITH_REFERENCE && trk->k == 5) if (int rc = ensure_strict_tables(ctx, trk)) return rc;
  const rl::TrackDev td = trk->dev();
  const double* cx = trk->c0.p;
  const double* cy = trk->c0.p + trk->n;
  if (ctx->arith == RL_ARITH_REFERENCE && trk->k == 5)     // the oracle's bits (orc_min_curvature_cost)
    hipLaunchKernelGGL(rl::k_cost_strict<5>, dim3(n_idx), dim3(256), 0, ctx->stream, td, cx, cy, didx.p, dz.p, dH.p, dg.p, dM.p);
  else if (trk->k == 3)
    hipLaunchKernelGGL(rl::k_cost<3>, dim3(n_idx), dim3(256), 0, ctx->stream, td, cx, cy, didx.p, dz.p, dH.p, dg.p, dM.p);
  else
    hipLaunchKernelGGL(rl::k_cost<5>, dim3(n_idx), dim3(256), 0, ctx->stream, td, cx, cy, didx.p, dz.p, dH.p, dg.p, dM.p);
  RL_HIP(hipGetLastError());
  RL_HIP(hipMemcpyAsync(H, dH.p, dH.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipMemcpyAsync(g, dg.p, dg.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (M) RL_HIP(hipMemcpyAsync(M, dM.p, n_idx * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipStreamSynchronize(ctx->stream));
  return RL_OK;
}

int rl_track_constraint(rl_ctx* ctx, const rl_track* trk, const double* points, int idx,
                        double* b, double* lba, double* uba, int* M) {
  if (!ctx || !trk || !points || !b || !lba || !uba || !M) return fail(RL_ERR_ARG, "null argument");
  if (idx < 0 || idx >= trk->n) return fail(RL_ERR_ARG, "control point index out of range");
  RL_HIP(hipSetDevice(ctx->device));
  const int N = trk->N, k = trk->k;
  // support size on the host (same mask as rl_track_create)
  const double step = 1.0 / (double)N;
  const double ts = trk->t_host[idx], te = trk->t_host[idx + k + 1];
  int s0 = 0;
  while (s0 < N && !((double)s0 * step >= ts)) ++s0;
  int s1 = s0;
  while (s1 < N && (double)s1 * step < te) ++s1;
  const int m = s1 - s0;
  *M = m;
  if (m == 0) return RL_OK;
  PoolBuf<double> dpts(ctx), db(ctx), dl(ctx), du(ctx);
  RL_HIP(dpts.alloc((size_t)N * RL_NCOL)); RL_HIP(db.alloc(m)); RL_HIP(dl.alloc((size_t)2 * m)); RL_HIP(du.alloc((size_t)2 * m));
  RL_HIP(hipMemcpyAsync(dpts.p, points, dpts.n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (ctx->arith == RL_ARITH_REFERENCE && k == 5) if (int rc = ensure_strict_tables(ctx, trk)) return rc;
  const rl::TrackDev td = trk->dev();
  const double* cx = trk->c0.p;
  const double* cy = trk->c0.p + trk->n;
  const dim3 grid((m + 255) / 256), block(256);
  if (ctx->arith == RL_ARITH_REFERENCE && k == 5)          // the oracle's bits (orc_track_constraint)
    hipLaunchKernelGGL(rl::k_constraint_strict<5>, grid, block, 0, ctx->stream, td, cx, cy, dpts.p, idx, db.p, dl.p, du.p);
  else if (k == 3)
    hipLaunchKernelGGL(rl::k_constraint<3>, grid, block, 0, ctx->stream, td, cx, cy, dpts.p, idx, db.p, dl.p, du.p);
  else
    hipLaunchKernelGGL(rl::k_constraint<5>, grid, block, 0, ctx->stream, td, cx, cy, dpts.p, idx, db.p, dl.p, du.p);
  RL_HIP(hipGetLastError());
  RL_HIP(hipMemcpyAsync(b, db.p, m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipMemcpyAsync(lba, dl.p, (size_t)2 * m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipMemcpyAsync(uba, du.p, (size_t)2 * m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipStreamSynchronize(ctx->stream));
  return RL_OK;
}

// ------------------------------------------------------------------------------------------------
static int solve_batch_common(rl_ctx* ctx, const rl_track* trk, int form, const double* in_dev,
                              int B, const int* i_start, int max_iter, int search,
                              double* out_ctrl, double* out_xy, double* out_points, int* n_success,
                              int* status, rl_stats* stats, SweepPlan* plan_out, bool joint = false) {
  if (!ctx || !trk || !i_start || !out_ctrl) return fail(RL_ERR_ARG, "null argument");
  if (!out_xy && !out_points) return fail(RL_ERR_ARG, "need out_xy or out_points");
  if (B <= 0) return fail(RL_ERR_ARG, "B <= 0");
  if (max_iter <= 0 || max_iter > RL_MAX_ITER) return fail(RL_ERR_ARG, "max_iter out of range");
  if (search < RL_SEARCH_BRUTE || search > RL_SEARCH_WINDOWED) return fail(RL_ERR_ARG, "bad search mode");
  const int n = trk->n, N = trk->N, k = trk->k;
  const int i_min = k / 2, i_max = n - (k - k / 2) - (joint ? 5 : 0);
  if (i_max <= i_min) return fail(RL_ERR_ARG, "too few control points");
  for (int j = 0; j < max_iter; ++j)
    if (i_start[j] < i_min || i_start[j] >= i_max)
      return fail(RL_ERR_ARG, joint ? "i_start outside [k//2, n-(k-k//2)-span) (optimizer.py:176-178)"
                                    : "i_start outside [k//2, n-(k-k//2)) (optimizer.py:301-303)");
  if (joint) {  // the window rows live in registers: 3 per thread
    int widest = 0;
    const double step = 1.0 / (double)N;
    for (int kk = i_min; kk < i_max; ++kk) {
      const double ts = trk->t_host[kk], te = trk->t_host[kk + 4 + k + 1];
      int a0 = 0;
      while (a0 < N && !((double)a0 * step >= ts)) ++a0;
      int b0 = a0;
      while (b0 < N && (double)b0 * step < te) ++b0;
      if (b0 - a0 > widest) widest = b0 - a0;
    }
    if (widest > rl::kJointRowsPerThread * 256)
      return fail(RL_ERR_UNSUPPORTED, "sliding-window variant: a window spans more than 768 samples");
  }
  // the DEFAULT arithmetic is the reference-order one where it exists (degree-5 splines, no step dump) and the fast one
  // elsewhere; an arithmetic chosen with rl_ctx_set_arith / RL_ARITH is taken literally (and fails where it does not exist).
  // rl_stats.reserved[0] names the arithmetic that ran.
  int arith = ctx->arith;
  if (!ctx->arith_explicit && arith == RL_ARITH_REFERENCE && (k != 5 || g_dbg_instances > 0)) arith = RL_ARITH_FAST;
  const bool lite = arith == RL_ARITH_BRANCH;
  const bool strict = arith == RL_ARITH_REFERENCE || lite;    // the branch mode shares the reference-order kernel's tables and state
  if (strict) {
    if (joint && lite) return fail(RL_ERR_UNSUPPORTED, "the branch arithmetic (rl_ctx_set_arith) covers run_min_curvature_qp; the sliding-window driver exists in the fast and in the reference-order arithmetic");
    RL_HIP(hipSetDevice(ctx->device));
    if (int rc = ensure_strict_tables(ctx, trk)) return rc;
  }
  rl::SweepArgs a;
  std::memset(&a, 0, sizeof(a));
  a.tr = trk->dev();
  a.B = B;
  a.form = form;
  a.in = in_dev;
  if (form == RL_BOUNDS_SHARED_RINGS) {
    if (trk->nL == 0) return fail(RL_ERR_ARG, "rl_track_set_rings was not called");
    a.ringL = reinterpret_cast<const double2*>(trk->ringL.p);
    a.ringR = reinterpret_cast<const double2*>(trk->ringR.p);
    a.nL = trk->nL; a.nR = trk->nR;
  } else if (form == RL_BOUNDS_WIDTHS || form == RL_BOUNDS_POINTS) {
    if (!in_dev) return fail(RL_ERR_ARG, "bounds input is null");
    a.nL = N; a.nR = N;
  } else {
    return fail(RL_ERR_ARG, "bad bounds_form");
  }
  a.max_iter = max_iter;
  for (int j = 0; j < max_iter; ++j) a.i_start[j] = i_start[j];
  a.search = search;
  a.max_dist = 100.0;  // race_track.py:104
#ifdef RL_ABLATION
  if (const char* dbg = getenv("RL_DEBUG_FLAGS")) a.debug = atoi(dbg);
#endif
#ifdef RL_STAMPS
  const bool dbg_ok = true;
#else
  const bool dbg_ok = !strict;
#endif
  if (g_dbg_instances > 0 && (joint || k == 5) && dbg_ok) {
    const int ninst = std::min(g_dbg_instances, B);
    size_t need = joint ? (size_t)max_iter * (size_t)(i_max - i_min) * (48 + 9 * rl::kJointRowsPerThread * 256 + 2 * n)
                        : (size_t)ninst * max_iter * 2 * (size_t)(i_max - i_min) * (rl::kSweepDumpHead + 2 * n);
#ifdef RL_STAMPS
    need = std::max(need, (size_t)B * 4 * 16);   // diagnostic build: per-wave phase stamps of every instance
#endif
    a.dbg_instances = ninst;
    if (g_dbg_len < need) {
      if (g_dbg_buf) (void)hipFree(g_dbg_buf);
      RL_HIP(hipMalloc(reinterpret_cast<void**>(&g_dbg_buf), need * sizeof(double)));
      g_dbg_len = need;
    }
    RL_HIP(hipMemsetAsync(g_dbg_buf, 0, need * sizeof(double), ctx->stream));
    a.dbg = g_dbg_buf;
  }
  a.out_ctrl = out_ctrl; a.out_xy = out_xy; a.out_points = out_points;
  a.n_success = n_success; a.status = status;
  RL_HIP(hipSetDevice(ctx->device));
  SweepPlan p = plan_sweep(ctx, n, N, a.nL, a.nR, B, joint, strict);
  if (p.lds_bytes > (size_t)ctx->max_lds) return fail(RL_ERR_UNSUPPORTED, "problem does not fit LDS");
  if (p.sigma_in_lds && !p.rings_in_lds && (joint || k != 5 || a.dbg)) {   // the mixed residency exists for the k = 5 sweep only
    p.sigma_in_lds = false;
    p.lds_bytes = rl::sweep_lds_layout(n, N, a.nL, a.nR, false, false, joint).total * sizeof(double);
    p.gscratch_doubles += (size_t)2 * ((N + 1) & ~1);
  }
  if (p.gscratch_doubles) {
    const size_t need = p.gscratch_doubles * (size_t)B;
    if (trk->gscratch.n < need) RL_HIP(trk->gscratch.alloc(need));
    a.gscratch = trk->gscratch.p;
    a.gscratch_stride = p.gscratch_doubles;
  }
  if (strict && !lite) {
    // per instance: [2 cpad] reserved | [ceil(N/16)*2] doubles of per-sample flag bytes | [N] double2 snapshot of the table's X, Y
    const size_t per = (size_t)2 * ((n + 1) & ~1) + (size_t)2 * ((N + 15) / 16) + (size_t)2 * N;
    const size_t flags = ((size_t)B + 1) / 2;    // [B] ints behind the per-instance blocks
    if (trk->aux.n < per * (size_t)B + flags) RL_HIP(trk->aux.alloc(per * (size_t)B + flags));
    a.aux = trk->aux.p;
    a.aux_stride = per;
    a.np_raise_at_start = ctx->np_raise_at_start;
    a.raise_flag = reinterpret_cast<int*>(trk->aux.p + per * (size_t)B);
    RL_HIP(hipMemsetAsync(a.raise_flag, 0, (size_t)B * sizeof(int), ctx->stream));
  }
  if (stats) {
    stats->lds_bytes = (int)p.lds_bytes;
    stats->block_threads = p.block;
    stats->rings_in_lds = p.rings_in_lds ? 1 : (p.sigma_in_lds ? 2 : 0);
    stats->reserved[0] = arith;
  }
  if (plan_out) *plan_out = p;
  return launch_sweep(ctx, k, p, a, joint, strict, lite);
}

int rl_mincurv_solve_batch_dev(rl_ctx* ctx, const rl_track* trk, int bounds_form, const double* in,
                               int B, const int* i_start, int max_iter, int search,
                               double* out_ctrl, double* out_xy, int* n_success, int* status,
                               rl_stats* stats) {
  if (!out_xy) return fail(RL_ERR_ARG, "out_xy is null");
  return solve_batch_common(ctx, trk, bounds_form, in, B, i_start, max_iter, search, out_ctrl, out_xy,
                            nullptr, n_success, status, stats, nullptr);
}

int rl_mincurv_solve_batch_host(rl_ctx* ctx, const rl_track* trk, int bounds_form, const double* in,
                                int B, const int* i_start, int max_iter, int search,
                                double* out_ctrl, double* out_xy, int* n_success, int* status,
                                rl_stats* stats) {
  if (!ctx || !trk || !out_ctrl || !out_xy) return fail(RL_ERR_ARG, "null argument");
  if (B <= 0) return fail(RL_ERR_ARG, "B <= 0");
  RL_HIP(hipSetDevice(ctx->device));
  const int n = trk->n, N = trk->N;
  const int cols = bounds_form == RL_BOUNDS_WIDTHS ? 2 : (bounds_form == RL_BOUNDS_POINTS ? 4 : 0);
  PoolBuf<double> din(ctx), dctrl(ctx), dxy(ctx);
  PoolBuf<int> dns(ctx), dst(ctx);
  if (cols) {
    if (!in) return fail(RL_ERR_ARG, "bounds input is null");
    RL_HIP(din.alloc((size_t)B * N * cols));
    RL_HIP(hipMemcpyAsync(din.p, in, din.n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  }
  RL_HIP(dctrl.alloc((size_t)B * n * 2)); RL_HIP(dxy.alloc((size_t)B * N * 2));
  RL_HIP(dns.alloc((size_t)B * 2 * (max_iter > 0 ? max_iter : 1))); RL_HIP(dst.alloc(B));
  RL_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  int rc = solve_batch_common(ctx, trk, bounds_form, din.p, B, i_start, max_iter, search, dctrl.p, dxy.p,
                              nullptr, dns.p, dst.p, stats, nullptr);
  if (rc) return rc;
  RL_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  RL_HIP(hipMemcpyAsync(out_ctrl, dctrl.p, dctrl.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipMemcpyAsync(out_xy, dxy.p, dxy.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (n_success) RL_HIP(hipMemcpyAsync(n_success, dns.p, (size_t)B * 2 * max_iter * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  if (status) RL_HIP(hipMemcpyAsync(status, dst.p, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipStreamSynchronize(ctx->stream));
  if (stats) {
    float ms = 0.f;
    RL_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    stats->kernel_ms = ms;
  }
  return RL_OK;
}

// ------------------------------------------------------------------------------------------------
// a15: global min-curvature QP (rl_global.hpp)
extern "C++" {
namespace {

int host_find_interval(const std::vector<double>& t, int k, int n, double x) {
  int lo = k, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (x >= t[mid]) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// Track-level tables: constraint rows, control-point normals, and the partition of the sample
// grid into chunks of <= kGRows consecutive samples of one knot span.
int global_tables(const rl_ctx* ctx, const rl_track* trk) {
  if (trk->gq_valid) return RL_OK;
  const int k = trk->k, n = trk->n, N = trk->N, np = n - k;
  if (np < 2 * (k + 1)) return fail(RL_ERR_UNSUPPORTED, "too few control points for the global QP");
  if (np > rl::kGMaxNp) return fail(RL_ERR_UNSUPPORTED, "global QP: more than 192 free control points");
  if (N >= (1 << 24)) return fail(RL_ERR_UNSUPPORTED, "global QP: N too large");
  std::vector<int> rows(np, 0);
  const double step = 1.0 / (double)N;
  for (int i = 0; i < N; ++i) ++rows[host_find_interval(trk->t_host, k, n, (double)i * step) - k];
  auto partition = [&](int R, std::vector<int>& chunk, std::vector<int>& span) {
    chunk.clear(); span.assign(np + 1, 0);
    int row = 0;
    for (int s = 0; s < np; ++s) {
      span[s] = (int)chunk.size() / 2;
      for (int r = 0; r < rows[s]; r += R) {
        const int c = std::min(R, rows[s] - r);
        chunk.push_back((row + r) | (c << 24));
        chunk.push_back(s);
      }
      row += rows[s];
    }
    span[np] = (int)chunk.size() / 2;
    return span[np];
  };
  std::vector<int> chunk, span, chunk2, span2;
  const int nc = partition(rl::kGRows, chunk, span);
  int rows2 = 0, nc2 = 0;
  for (int R : {5, 6, 10}) {  // fewest rows per thread whose chunks fit the row waves of k_global_qp2
    nc2 = partition(R, chunk2, span2);
    if (nc2 <= rl::kG2Block - 64) { rows2 = R; break; }
  }
  if (nc > 1024) return fail(RL_ERR_UNSUPPORTED, "global QP: N / 8 + n exceeds one workgroup");
  RL_HIP(trk->gq2_chunk.alloc(chunk2.size())); RL_HIP(trk->gq2_span.alloc(span2.size()));
  RL_HIP(hipMemcpyAsync(trk->gq2_chunk.p, chunk2.data(), chunk2.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  RL_HIP(hipMemcpyAsync(trk->gq2_span.p, span2.data(), span2.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  RL_HIP(trk->gq_A.alloc((size_t)N * (k + 1))); RL_HIP(trk->gq_nu.alloc((size_t)2 * np));
  RL_HIP(trk->gq_chunk.alloc(chunk.size())); RL_HIP(trk->gq_span.alloc(span.size()));
  RL_HIP(hipMemcpyAsync(trk->gq_chunk.p, chunk.data(), chunk.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  RL_HIP(hipMemcpyAsync(trk->gq_span.p, span.data(), span.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  const rl::TrackDev td = trk->dev();
  if (k == 3) {
    hipLaunchKernelGGL(rl::k_global_normals<3>, dim3((np + 63) / 64), dim3(64), 0, ctx->stream, td, np, trk->gq_nu.p);
    hipLaunchKernelGGL(rl::k_global_rows<3>, dim3((N + 127) / 128), dim3(128), 0, ctx->stream, td, np, trk->gq_nu.p, trk->gq_A.p);
  } else {
    hipLaunchKernelGGL(rl::k_global_normals<5>, dim3((np + 63) / 64), dim3(64), 0, ctx->stream, td, np, trk->gq_nu.p);
    hipLaunchKernelGGL(rl::k_global_rows<5>, dim3((N + 127) / 128), dim3(128), 0, ctx->stream, td, np, trk->gq_nu.p, trk->gq_A.p);
  }
  RL_HIP(hipGetLastError());
  RL_HIP(hipStreamSynchronize(ctx->stream));  // chunk / span are host vectors
  trk->gq_nc = nc;
  trk->gq2_nc = nc2;
  trk->gq2_rows = rows2;
  trk->gq_valid = true;
  return RL_OK;
}

template <int K, int MAXB>
int launch_global_t(const rl_ctx* ctx, const rl::GlobalArgs& a, int B, int block, size_t lds) {
  auto kern = rl::k_global_qp<K, MAXB>;
  RL_HIP(grant_dyn_lds(const_cast<rl_ctx*>(ctx), reinterpret_cast<const void*>(kern), lds));
  hipLaunchKernelGGL(kern, dim3(B), dim3(block), lds, ctx->stream, a);
  RL_HIP(hipGetLastError());
  return RL_OK;
}

template <int K, int R, int G, bool AREG>
int launch_global2_t(const rl_ctx* ctx, const rl::GlobalArgs& a, int B, int block, size_t lds) {
  auto kern = rl::k_global_qp2<K, R, G, AREG>;
  RL_HIP(grant_dyn_lds(const_cast<rl_ctx*>(ctx), reinterpret_cast<const void*>(kern), lds));
  hipLaunchKernelGGL(kern, dim3(B), dim3(block), lds, ctx->stream, a);
  RL_HIP(hipGetLastError());
  return RL_OK;
}

int global_common(rl_ctx* ctx, const rl_track* trk, const double* widths, int B, double margin, int n_outer,
                  double* out_ctrl, double* out_xy, double* out_a, double* out_stats, rl_stats* stats) {
  if (!ctx || !trk || !widths || !out_ctrl || !out_stats) return fail(RL_ERR_ARG, "null argument");
  if (B <= 0) return fail(RL_ERR_ARG, "B <= 0");
  if (n_outer < 0 || n_outer > 64) return fail(RL_ERR_ARG, "n_outer out of range");
  if (!degree_supported(trk->k)) return fail(RL_ERR_UNSUPPORTED, "spline degree must be 3 or 5");
  RL_HIP(hipSetDevice(ctx->device));
  if (int rc = global_tables(ctx, trk)) return rc;
  rl::GlobalArgs a;
  a.trk = trk->dev();
  a.A = trk->gq_A.p; a.nu = trk->gq_nu.p; a.chunk = trk->gq_chunk.p; a.span_ch0 = trk->gq_span.p;
  a.nc = trk->gq_nc; a.np = trk->n - trk->k;
  a.widths = widths; a.margin = margin; a.n_outer = n_outer; a.max_ipm = 80;
  a.out_ctrl = out_ctrl; a.out_xy = out_xy; a.out_a = out_a; a.out_stats = out_stats;
  // fast path: wave-specialised kernel (rl_global2.hpp) when the chunks fit 7 row waves and the LDS
  if (!ctx->force_global_v1 && trk->gq2_rows != 0 && a.np <= 192) {
    const int groups = a.np <= 64 ? 1 : 3;
    const int block2 = ((trk->gq2_nc + 63) / 64 + 1) * 64;
    const size_t lds2 = (size_t)rl::global2_layout(trk->k, trk->n, a.np, trk->N, groups, block2 - 64).total * sizeof(double);
    if (lds2 <= (size_t)ctx->max_lds) {
      // R = 5, 6: constraint rows register resident; R = 10 (N up to ~4400): re-read from L2, one group only
      const int R = trk->gq2_rows;
      if (R != 10 || groups == 1) {
        a.chunk = trk->gq2_chunk.p; a.span_ch0 = trk->gq2_span.p; a.nc = trk->gq2_nc;
        if (stats) { stats->lds_bytes = (int)lds2; stats->block_threads = block2; stats->rings_in_lds = 0; }
#define RL_G2_LAUNCH(KK, RR, GG, AR) return launch_global2_t<KK, RR, GG, AR>(ctx, a, B, block2, lds2)
        if (trk->k == 3) {
          if (R == 5) { if (groups == 1) RL_G2_LAUNCH(3, 5, 1, true); RL_G2_LAUNCH(3, 5, 3, true); }
          if (R == 6) { if (groups == 1) RL_G2_LAUNCH(3, 6, 1, true); RL_G2_LAUNCH(3, 6, 3, true); }
          RL_G2_LAUNCH(3, 10, 1, false);
        }
        if (R == 5) { if (groups == 1) RL_G2_LAUNCH(5, 5, 1, true); RL_G2_LAUNCH(5, 5, 3, true); }
        if (R == 6) { if (groups == 1) RL_G2_LAUNCH(5, 6, 1, true); RL_G2_LAUNCH(5, 6, 3, true); }
        RL_G2_LAUNCH(5, 10, 1, false);
#undef RL_G2_LAUNCH
      }
    }
  }
  const int block = std::max(64, (a.nc + 63) / 64 * 64);
  const size_t lds = (size_t)rl::global_layout(trk->k, trk->n, a.np, block).total * sizeof(double);
  if (lds > (size_t)ctx->max_lds) return fail(RL_ERR_UNSUPPORTED, "global QP does not fit LDS");
  if (stats) { stats->lds_bytes = (int)lds; stats->block_threads = block; stats->rings_in_lds = 0; }
  if (trk->k == 3) {
    if (block <= 384) return launch_global_t<3, 384>(ctx, a, B, block, lds);
    if (block <= 640) return launch_global_t<3, 640>(ctx, a, B, block, lds);
    return launch_global_t<3, 1024>(ctx, a, B, block, lds);
  }
  if (block <= 384) return launch_global_t<5, 384>(ctx, a, B, block, lds);
  if (block <= 640) return launch_global_t<5, 640>(ctx, a, B, block, lds);
  return launch_global_t<5, 1024>(ctx, a, B, block, lds);
}

// ---- the two-coordinate formulation (rl_global_xy.hpp)
int global_xy_tables(const rl_ctx* ctx, const rl_track* trk) {
  if (trk->gxy_valid) return RL_OK;
  const int k = trk->k, n = trk->n, N = trk->N, np = n - k;
  if (np < 2 * (k + 1)) return fail(RL_ERR_UNSUPPORTED, "too few control points for the global QP");
  if (2 * np > rl::kXYMaxNz) return fail(RL_ERR_UNSUPPORTED, "global QP (dof 2): more than 96 free control points");
  std::vector<int> first(np + 1, 0);
  const double step = 1.0 / (double)N;
  int prev = -1;
  for (int i = 0; i < N; ++i) {   // spans are consecutive sample ranges; empty spans get the next span's first sample
    const int s = host_find_interval(trk->t_host, k, n, (double)i * step) - k;
    for (int q = prev + 1; q <= s; ++q) first[q] = i;
    prev = std::max(prev, s);
  }
  for (int q = prev + 1; q <= np; ++q) first[q] = N;
  RL_HIP(trk->gxy_span.alloc(first.size()));
  RL_HIP(hipMemcpyAsync(trk->gxy_span.p, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  // chunks: the samples of a span in pieces of <= CL samples.  The adaptive knots give spans of very different lengths (Monza
  // s = 100, N = 2000: 2 .. 281 samples), and a task per span waits for the longest.  The shortest CL whose chunk sums fit LDS.
  std::vector<int> cf, sc0(np + 1, 0);
  for (int CL = 33;; CL = (CL + CL / 2) | 1) {   // (odd: the chunks of a long span then start on different LDS banks)
    cf.clear();
    for (int s = 0; s < np; ++s) {
      sc0[s] = (int)cf.size();
      for (int m = first[s]; m < first[s + 1]; m += CL) cf.push_back(m);
    }
    sc0[np] = (int)cf.size();
    cf.push_back(N);
    const size_t lds = (size_t)rl::global_xy_layout(k, n, np, N, sc0[np]).total * sizeof(double);
    if (lds <= (size_t)ctx->max_lds || CL >= N) break;
  }
  trk->gxy_nch = sc0[np];
  RL_HIP(trk->gxy_chunk.alloc(cf.size())); RL_HIP(trk->gxy_sch0.alloc(sc0.size()));
  RL_HIP(hipMemcpyAsync(trk->gxy_chunk.p, cf.data(), cf.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  RL_HIP(hipMemcpyAsync(trk->gxy_sch0.p, sc0.data(), sc0.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  const int K1 = k + 1;
  RL_HIP(trk->gxy_bbx.alloc((size_t)N * (K1 * (K1 + 1) / 2 + K1)));
  const rl::TrackDev td = trk->dev();
  if (k == 3) hipLaunchKernelGGL(rl::k_global_xy_pairs<3>, dim3((N + 127) / 128), dim3(128), 0, ctx->stream, td, trk->gxy_bbx.p);
  else hipLaunchKernelGGL(rl::k_global_xy_pairs<5>, dim3((N + 127) / 128), dim3(128), 0, ctx->stream, td, trk->gxy_bbx.p);
  RL_HIP(hipGetLastError());
  RL_HIP(hipStreamSynchronize(ctx->stream));  // `first`, `cf`, `sc0` are host vectors
  trk->gxy_valid = true;
  return RL_OK;
}

template <int K, int NT>
int launch_global_xy_t(const rl_ctx* ctx, const rl::GlobalXYArgs& a, int B, size_t lds) {
  auto kern = rl::k_global_xy<K, NT>;
  RL_HIP(grant_dyn_lds(const_cast<rl_ctx*>(ctx), reinterpret_cast<const void*>(kern), lds));
  hipLaunchKernelGGL(kern, dim3(B), dim3(NT), lds, ctx->stream, a);
  RL_HIP(hipGetLastError());
  return RL_OK;
}

int global_xy_common(rl_ctx* ctx, const rl_track* trk, const double* widths, int B, double margin, double lon, int n_outer,
                     double* out_ctrl, double* out_xy, double* out_z, double* out_stats, rl_stats* stats) {
  if (!ctx || !trk || !widths || !out_ctrl || !out_stats) return fail(RL_ERR_ARG, "null argument");
  if (B <= 0) return fail(RL_ERR_ARG, "B <= 0");
  if (n_outer < 0 || n_outer > 64) return fail(RL_ERR_ARG, "n_outer out of range");
  if (!(lon > 0.0)) return fail(RL_ERR_ARG, "the longitudinal bound must be positive");
  if (!degree_supported(trk->k)) return fail(RL_ERR_UNSUPPORTED, "spline degree must be 3 or 5");
  RL_HIP(hipSetDevice(ctx->device));
  if (int rc = global_xy_tables(ctx, trk)) return rc;
  rl::GlobalXYArgs a;
  a.trk = trk->dev();
  a.span_first = trk->gxy_span.p; a.chunk_first = trk->gxy_chunk.p; a.span_chunk0 = trk->gxy_sch0.p; a.nch = trk->gxy_nch;
  a.bbx = trk->gxy_bbx.p; a.np = trk->n - trk->k;
  a.widths = widths; a.margin = margin; a.lon = lon; a.n_outer = n_outer; a.max_ipm = 80;
  a.out_ctrl = out_ctrl; a.out_xy = out_xy; a.out_z = out_z; a.out_stats = out_stats;
  const int block = trk->N <= 512 * rl::kXYRows ? 512 : 1024;
  if (trk->N > block * rl::kXYRows) return fail(RL_ERR_UNSUPPORTED, "global QP (dof 2): more than 4096 samples");
  const size_t lds = (size_t)rl::global_xy_layout(trk->k, trk->n, a.np, trk->N, a.nch).total * sizeof(double);
  if (lds > (size_t)ctx->max_lds) return fail(RL_ERR_UNSUPPORTED, "global QP (dof 2) does not fit LDS");
  if (stats) { stats->lds_bytes = (int)lds; stats->block_threads = block; stats->rings_in_lds = 0; }
  if (trk->k == 3) return block == 512 ? launch_global_xy_t<3, 512>(ctx, a, B, lds) : launch_global_xy_t<3, 1024>(ctx, a, B, lds);
  return block == 512 ? launch_global_xy_t<5, 512>(ctx, a, B, lds) : launch_global_xy_t<5, 1024>(ctx, a, B, lds);
}

}  // namespace
}  // extern "C++"

int rl_mincurv_global_xy_batch_dev(rl_ctx* ctx, const rl_track* trk, const double* widths, int B, double margin, double lon,
                                   int n_outer, double* out_ctrl, double* out_xy, double* out_z, double* out_stats,
                                   rl_stats* stats) {
  return global_xy_common(ctx, trk, widths, B, margin, lon, n_outer, out_ctrl, out_xy, out_z, out_stats, stats);
}

int rl_mincurv_global_xy_batch_host(rl_ctx* ctx, const rl_track* trk, const double* widths, int B, double margin, double lon,
                                    int n_outer, double* out_ctrl, double* out_xy, double* out_z, double* out_stats,
                                    rl_stats* stats) {
  if (!ctx || !trk || !widths || !out_ctrl || !out_stats) return fail(RL_ERR_ARG, "null argument");
  if (B <= 0) return fail(RL_ERR_ARG, "B <= 0");
  RL_HIP(hipSetDevice(ctx->device));
  const int n = trk->n, N = trk->N, np = n - trk->k;
  PoolBuf<double> dw(ctx), dctrl(ctx), dxy(ctx), dz(ctx), dst(ctx);
  RL_HIP(dw.alloc((size_t)B * N * 2)); RL_HIP(dctrl.alloc((size_t)B * n * 2));
  if (out_xy) RL_HIP(dxy.alloc((size_t)B * N * 2));
  if (out_z) RL_HIP(dz.alloc((size_t)B * np * 2));
  RL_HIP(dst.alloc((size_t)B * 8));
  RL_HIP(hipMemcpyAsync(dw.p, widths, dw.n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (int rc = global_xy_tables(ctx, trk)) return rc;  // keeps the table build out of the timed region
  RL_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  if (int rc = global_xy_common(ctx, trk, dw.p, B, margin, lon, n_outer, dctrl.p, dxy.p, dz.p, dst.p, stats)) return rc;
  RL_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  RL_HIP(hipMemcpyAsync(out_ctrl, dctrl.p, dctrl.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (out_xy) RL_HIP(hipMemcpyAsync(out_xy, dxy.p, dxy.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (out_z) RL_HIP(hipMemcpyAsync(out_z, dz.p, dz.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipMemcpyAsync(out_stats, dst.p, dst.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipStreamSynchronize(ctx->stream));
  if (stats) {
    float ms = 0.f;
    RL_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    stats->kernel_ms = ms;
  }
  return RL_OK;
}

int rl_mincurv_global_batch_dev(rl_ctx* ctx, const rl_track* trk, const double* widths, int B,
                                double margin, int n_outer, double* out_ctrl, double* out_xy,
                                double* out_a, double* out_stats, rl_stats* stats) {
  return global_common(ctx, trk, widths, B, margin, n_outer, out_ctrl, out_xy, out_a, out_stats, stats);
}

int rl_mincurv_global_batch_host(rl_ctx* ctx, const rl_track* trk, const double* widths, int B,
                                 double margin, int n_outer, double* out_ctrl, double* out_xy,
                                 double* out_a, double* out_stats, rl_stats* stats) {
  if (!ctx || !trk || !widths || !out_ctrl || !out_stats) return fail(RL_ERR_ARG, "null argument");
  if (B <= 0) return fail(RL_ERR_ARG, "B <= 0");
  RL_HIP(hipSetDevice(ctx->device));
  const int n = trk->n, N = trk->N, np = n - trk->k;
  PoolBuf<double> dw(ctx), dctrl(ctx), dxy(ctx), da(ctx), dst(ctx);
  RL_HIP(dw.alloc((size_t)B * N * 2)); RL_HIP(dctrl.alloc((size_t)B * n * 2));
  if (out_xy) RL_HIP(dxy.alloc((size_t)B * N * 2));
  if (out_a) RL_HIP(da.alloc((size_t)B * np));
  RL_HIP(dst.alloc((size_t)B * 8));
  RL_HIP(hipMemcpyAsync(dw.p, widths, dw.n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (int rc = global_tables(ctx, trk)) return rc;  // keeps the table build out of the timed region
  RL_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  if (int rc = global_common(ctx, trk, dw.p, B, margin, n_outer, dctrl.p, dxy.p, da.p, dst.p, stats)) return rc;
  RL_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  RL_HIP(hipMemcpyAsync(out_ctrl, dctrl.p, dctrl.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (out_xy) RL_HIP(hipMemcpyAsync(out_xy, dxy.p, dxy.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (out_a) RL_HIP(hipMemcpyAsync(out_a, da.p, da.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipMemcpyAsync(out_stats, dst.p, dst.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipStreamSynchronize(ctx->stream));
  if (stats) {
    float ms = 0.f;
    RL_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    stats->kernel_ms = ms;
  }
  return RL_OK;
}

int rl_mincurv_sweep(rl_ctx* ctx, rl_track* trk, const int* i_start, int max_iter,
                     double* cx, double* cy, double* points, int* n_success, rl_stats* stats) {
  return sweep_single(ctx, trk, i_start, max_iter, cx, cy, points, n_success, stats, false);
}

int rl_mincurv_sweep_joint(rl_ctx* ctx, rl_track* trk, const int* i_start, int max_iter,
                           double* cx, double* cy, double* points, int* n_success, rl_stats* stats) {
  return sweep_single(ctx, trk, i_start, max_iter, cx, cy, points, n_success, stats, true);
}

int rl_qss_sim_dev(rl_ctx* ctx, double* points, int B, int N, const double* acc_x, const double* acc_c,
                   int acc_m, const double* dcc_x, const double* dcc_c, int dcc_m, const double* params,
                   int* iters) {
  if (!ctx || !points || !acc_x || !acc_c || !dcc_x || !dcc_c || !params || !iters)
    return fail(RL_ERR_ARG, "null argument");
  if (B <= 0 || N < 2 || acc_m < 1 || dcc_m < 1) return fail(RL_ERR_ARG, "bad sizes");
  if (N >= 65535) return fail(RL_ERR_UNSUPPORTED, "qss: owner indices are 16 bits");
  RL_HIP(hipSetDevice(ctx->device));
  const int cap = 4 * N + 16;
  const size_t lds = ((size_t)2 * N + 5 * (acc_m + dcc_m) + 2) * sizeof(double) + (size_t)rl::kQssStamp * sizeof(int) +
                     (((size_t)N * sizeof(unsigned short) + 7) & ~(size_t)7);  // speed, lon acc, tables, stamp buckets, owner
  if (lds > (size_t)ctx->max_lds) return fail(RL_ERR_UNSUPPORTED, "qss: trajectory too long for the LDS-resident profile");
  // The lookup tables are tiny host arrays (scipy CubicSpline pieces).  When they fit they travel BY VALUE in the kernel
  // arguments, so the call is asynchronous and keeps no pointer to caller memory; longer tables are staged through the arena
  // with pageable copies (the runtime then blocks the host until it has read them).
  const int tab_n = (acc_m + 1) + 4 * acc_m + (dcc_m + 1) + 4 * dcc_m;
  const bool by_value = tab_n <= rl::kQssTabMax;
  Arena ar(ctx);
  RL_HIP(ar.reserve((by_value ? 0 : Arena::pad((acc_m + 1) * 8) + Arena::pad((size_t)4 * acc_m * 8) + Arena::pad((dcc_m + 1) * 8) +
                                    Arena::pad((size_t)4 * dcc_m * 8)) +
                    Arena::pad((size_t)B * cap * 5 * 4) + Arena::pad((size_t)B * cap * 4) + Arena::pad((size_t)B * 3 * N * 8)));
  double *dax = nullptr, *dac = nullptr, *ddx = nullptr, *ddc = nullptr;
  if (!by_value) {
    dax = ar.take<double>(acc_m + 1); dac = ar.take<double>((size_t)4 * acc_m);
    ddx = ar.take<double>(dcc_m + 1); ddc = ar.take<double>((size_t)4 * dcc_m);
    RL_HIP(hipMemcpyAsync(dax, acc_x, (acc_m + 1) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    RL_HIP(hipMemcpyAsync(dac, acc_c, (size_t)4 * acc_m * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    RL_HIP(hipMemcpyAsync(ddx, dcc_x, (dcc_m + 1) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    RL_HIP(hipMemcpyAsync(ddc, dcc_c, (size_t)4 * dcc_m * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  }
  int* dfl = ar.take<int>((size_t)B * cap * 5); int* dnw = ar.take<int>((size_t)B * cap);
  double* dcst = ar.take<double>((size_t)B * 3 * N);
  rl::QssArgs a;
  a.points = points; a.B = B; a.N = N;
  a.acc_x = dax; a.acc_c = dac; a.acc_m = acc_m;
  a.dcc_x = ddx; a.dcc_c = ddc; a.dcc_m = dcc_m;
  a.max_lon_acc = params[0]; a.max_lon_dcc = params[1]; a.max_left_acc = params[2];
  a.max_right_acc = params[3]; a.max_speed = params[4]; a.max_jerk = params[5];
  a.flags = dfl; a.fresh = dnw; a.cst = dcst; a.cap = cap; a.iters = iters;
  a.tab_n = 0; a.redo = 0; a.dbg = nullptr; a.df_bail_at = ctx->qss_df_bail_at;
  if (by_value) {
    double* q = a.tab;
    for (int i = 0; i <= acc_m; ++i) *q++ = acc_x[i];
    for (int i = 0; i < 4 * acc_m; ++i) *q++ = acc_c[i];
    for (int i = 0; i <= dcc_m; ++i) *q++ = dcc_x[i];
    for (int i = 0; i < 4 * dcc_m; ++i) *q++ = dcc_c[i];
    a.tab_n = tab_n;
  }
  // The dataflow kernel takes the sizes its LDS tables hold; what it hands back (iters = -2: front or iteration counts beyond its
  // tables) is re-run by the list-order kernel right behind it, which returns at once for every other instance.
  // Which kernel: the dataflow kernel finishes an instance 2 - 4.5 times sooner but its tables fill the LDS (one instance per CU at
  // N = 2000, three at N = 500), the list-order kernel holds four and more per CU.  Compare the number of ROUNDS the batch
  // takes on the chip (measured, DESIGN_HISTORY.md 3c); RL_QSS_DF=1 / 0 forces one or the other (tests run both).
  bool use_df = rl::df_supported(N, acc_m, dcc_m, (size_t)ctx->max_lds);
  if (use_df) {
    const long long per_cu_df = (long long)((size_t)ctx->max_lds / rl::df_layout(N, acc_m, dcc_m).bytes);
    long long per_cu_list = (long long)((size_t)ctx->max_lds / lds);
    per_cu_list = per_cu_list > 8 ? 8 : per_cu_list;
    const long long cus = ctx->num_cu > 0 ? ctx->num_cu : 1;
    const long long rounds_df = (B + per_cu_df * cus - 1) / (per_cu_df * cus), rounds_list = (B + per_cu_list * cus - 1) / (per_cu_list * cus);
    // a round of the dataflow kernel takes from 1/5 (N = 2000: 29 against 168 ms) to 1/2 (N = 500: 4.4 against 9.3 ms) of a
    // list-order round (DESIGN_HISTORY.md 3c): the ratio the comparison allows grows with N between those two measurements
    const double ratio = N <= 500 ? 2.0 : (N >= 2000 ? 5.0 : 2.0 + 3.0 * (double)(N - 500) / 1500.0);
    use_df = (double)rounds_df <= ratio * (double)rounds_list;
  }
  if (ctx->qss_kernel == 0) use_df = false;
  if (ctx->qss_kernel == 1) use_df = rl::df_supported(N, acc_m, dcc_m, (size_t)ctx->max_lds);
  if (use_df) {
    const size_t lds_df = rl::df_layout(N, acc_m, dcc_m).bytes;
    int* ddbg = nullptr;
#ifdef RL_ABLATION
    if (const char* dbg = getenv("RL_QSS_DEBUG")) if (dbg[0] == '1') { RL_HIP(hipMalloc(&ddbg, (size_t)B * 12 * sizeof(int))); a.dbg = ddbg; }
#endif
    // waves of the (one) workgroup an instance's tables leave room for: four is the fastest at every size measured
    // (DESIGN_HISTORY.md 3c); one and two exist for the tests (rl_ctx_set_option "qss_df_waves")
    if (ctx->qss_df_waves == 1) {
      RL_HIP(grant_dyn_lds(ctx, reinterpret_cast<const void*>(rl::k_qss_dfw<1>), lds_df));
      hipLaunchKernelGGL(rl::k_qss_dfw<1>, dim3(B), dim3(64), lds_df, ctx->stream, a);
    } else if (ctx->qss_df_waves == 2) {
      RL_HIP(grant_dyn_lds(ctx, reinterpret_cast<const void*>(rl::k_qss_dfw<2>), lds_df));
      hipLaunchKernelGGL(rl::k_qss_dfw<2>, dim3(B), dim3(128), lds_df, ctx->stream, a);
    } else {
      RL_HIP(grant_dyn_lds(ctx, reinterpret_cast<const void*>(rl::k_qss_dfw<4>), lds_df));
      hipLaunchKernelGGL(rl::k_qss_dfw<4>, dim3(B), dim3(256), lds_df, ctx->stream, a);
    }
    RL_HIP(hipGetLastError());
    if (ddbg) {   // diagnostic build only: synchronous
      std::vector<int> hd((size_t)B * 12);
      RL_HIP(hipStreamSynchronize(ctx->stream));
      const hipError_t ce = hipMemcpy(hd.data(), ddbg, hd.size() * sizeof(int), hipMemcpyDeviceToHost);
      (void)hipFree(ddbg);
      RL_HIP(ce);
      long long tot[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int handed = 0;
      for (int i = 0; i < B; ++i) { for (int q = 0; q < 8; ++q) tot[q] += hd[(size_t)i * 12 + q]; handed += hd[(size_t)i * 12 + 7] != 0; }
      fprintf(stderr, "k_qss_dfw: B=%d N=%d lds=%zu  per instance: passes %.0f chunks %.0f examinations %.0f steps %.0f longest queue %.0f numberings %.0f spawned %.0f; handed back %d (reason of instance 0: %d)\n",
              B, N, lds_df, (double)tot[0] / B, (double)tot[1] / B, (double)tot[2] / B, (double)tot[3] / B, (double)tot[4] / B, (double)tot[5] / B, (double)tot[6] / B, handed, hd[7]);
      fprintf(stderr, "k_qss_dfw: thread 0 of instance 0, thousands of clock64 ticks: examination + step %d, records (+ waits) %d, list lane + next agents + wake-ups %d, end of pass %d\n", hd[8], hd[9], hd[10], hd[11]);
      a.dbg = nullptr;
    }
    a.redo = 1;
  }
  RL_HIP(grant_dyn_lds(ctx, reinterpret_cast<const void*>(rl::k_qss_sim), lds));
  hipLaunchKernelGGL(rl::k_qss_sim, dim3(B), dim3(64), lds, ctx->stream, a);
  RL_HIP(hipGetLastError());
  RL_HIP(ar.end());
  return RL_OK;
}

int rl_qss_sim(rl_ctx* ctx, double* points, int B, int N, const double* acc_x, const double* acc_c,
               int acc_m, const double* dcc_x, const double* dcc_c, int dcc_m, const double* params,
               int* iters) {
  if (!ctx || !points || !iters) return fail(RL_ERR_ARG, "null argument");
  if (B <= 0 || N < 2) return fail(RL_ERR_ARG, "bad sizes");
  RL_HIP(hipSetDevice(ctx->device));
  PoolBuf<double> dpts(ctx);
  PoolBuf<int> dit(ctx);
  RL_HIP(dpts.alloc((size_t)B * N * RL_NCOL)); RL_HIP(dit.alloc(B));
  RL_HIP(hipMemcpyAsync(dpts.p, points, dpts.n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (int rc = rl_qss_sim_dev(ctx, dpts.p, B, N, acc_x, acc_c, acc_m, dcc_x, dcc_c, dcc_m, params, dit.p)) return rc;
  RL_HIP(hipMemcpyAsync(points, dpts.p, dpts.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipMemcpyAsync(iters, dit.p, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipStreamSynchronize(ctx->stream));
  return RL_OK;
}

int rl_dt_eval_nodes(rl_ctx* ctx, const double* model, int B, int N, const double* s, const double* kappa,
                     const double* left, const double* right, double margin, double track_length,
                     const double* X, const double* U, const double* T, double* eq, double* ineq,
                     double* cost) {
  static_assert((int)RL_DT_NPARAM == (int)rl::DT_NPARAM, "parameter tables out of step");
  if (!ctx || !model || !s || !kappa || !left || !right || !X || !U || !T || !eq || !ineq || !cost)
    return fail(RL_ERR_ARG, "null argument");
  if (B <= 0 || N < 2 || !(track_length > 0.0)) return fail(RL_ERR_ARG, "bad sizes");
  RL_HIP(hipSetDevice(ctx->device));
  const size_t bn = (size_t)B * N;
  PoolBuf<double> ds(ctx), dk(ctx), dl(ctx), dr(ctx), dX(ctx), dU(ctx), dT(ctx), deq(ctx), dg(ctx), dc(ctx);
  RL_HIP(ds.alloc(N)); RL_HIP(dk.alloc(N)); RL_HIP(dl.alloc(N)); RL_HIP(dr.alloc(N));
  RL_HIP(dX.alloc(bn * 6)); RL_HIP(dU.alloc(bn * 4)); RL_HIP(dT.alloc(bn));
  RL_HIP(deq.alloc(bn * rl::kDtNeq)); RL_HIP(dg.alloc(bn * rl::kDtNineq)); RL_HIP(dc.alloc(bn));
  auto up = [&](PoolBuf<double>& d, const double* h) {
    return hipMemcpyAsync(d.p, h, d.n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  };
  RL_HIP(up(ds, s)); RL_HIP(up(dk, kappa)); RL_HIP(up(dl, left)); RL_HIP(up(dr, right));
  RL_HIP(up(dX, X)); RL_HIP(up(dU, U)); RL_HIP(up(dT, T));
  rl::DtArgs a;
  for (int i = 0; i < rl::DT_NPARAM; ++i) a.p[i] = model[i];
  a.B = B; a.N = N; a.s = ds.p; a.kappa = dk.p; a.left = dl.p; a.right = dr.p; a.margin = margin;
  a.track_length = track_length; a.X = dX.p; a.U = dU.p; a.T = dT.p;
  a.eq = deq.p; a.ineq = dg.p; a.cost_part = dc.p;
  hipLaunchKernelGGL(rl::k_dt_eval_nodes, dim3((N + 127) / 128, B), dim3(128), 0, ctx->stream, a);
  RL_HIP(hipGetLastError());
  std::vector<double> part(bn);
  RL_HIP(hipMemcpyAsync(eq, deq.p, deq.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipMemcpyAsync(ineq, dg.p, dg.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipMemcpyAsync(part.data(), dc.p, bn * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipStreamSynchronize(ctx->stream));
  for (int b = 0; b < B; ++b) {  // the objective is a plain sum over the nodes, in node order
    double c = 0.0;
    for (int j = 0; j < N; ++j) c += part[(size_t)b * N + j];
    cost[b] = c;
  }
  return RL_OK;
}

int rl_dt_eval_jac(rl_ctx* ctx, const double* model, int B, int N, const double* s, const double* kappa,
                   const double* left, const double* right, double margin, double track_length,
                   const double* X, const double* U, const double* T, double* jac_eq, double* jac_ineq,
                   double* grad_cost) {
  if (!ctx || !model || !s || !kappa || !left || !right || !X || !U || !T || !jac_eq || !jac_ineq || !grad_cost)
    return fail(RL_ERR_ARG, "null argument");
  if (B <= 0 || N < 2 || !(track_length > 0.0)) return fail(RL_ERR_ARG, "bad sizes");
  RL_HIP(hipSetDevice(ctx->device));
  const size_t bn = (size_t)B * N;
  PoolBuf<double> ds(ctx), dk(ctx), dl(ctx), dr(ctx), dX(ctx), dU(ctx), dT(ctx), dje(ctx), dji(ctx), dgc(ctx);
  RL_HIP(ds.alloc(N)); RL_HIP(dk.alloc(N)); RL_HIP(dl.alloc(N)); RL_HIP(dr.alloc(N));
  RL_HIP(dX.alloc(bn * 6)); RL_HIP(dU.alloc(bn * 4)); RL_HIP(dT.alloc(bn));
  RL_HIP(dje.alloc(bn * rl::kDtNeq * rl::kDtNvar)); RL_HIP(dji.alloc(bn * rl::kDtNineq * rl::kDtNvar));
  RL_HIP(dgc.alloc(bn * rl::kDtNvar));
  auto up = [&](PoolBuf<double>& d, const double* h) {
    return hipMemcpyAsync(d.p, h, d.n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  };
  RL_HIP(up(ds, s)); RL_HIP(up(dk, kappa)); RL_HIP(up(dl, left)); RL_HIP(up(dr, right));
  RL_HIP(up(dX, X)); RL_HIP(up(dU, U)); RL_HIP(up(dT, T));
  rl::DtJacArgs ja;
  rl::DtArgs& a = ja.f;
  for (int i = 0; i < rl::DT_NPARAM; ++i) a.p[i] = model[i];
  a.B = B; a.N = N; a.s = ds.p; a.kappa = dk.p; a.left = dl.p; a.right = dr.p; a.margin = margin;
  a.track_length = track_length; a.X = dX.p; a.U = dU.p; a.T = dT.p;
  a.eq = nullptr; a.ineq = nullptr; a.cost_part = nullptr;
  ja.jac_eq = dje.p; ja.jac_ineq = dji.p; ja.grad_cost = dgc.p;
  const int slices = (rl::kDtNvar + rl::kDtJacND - 1) / rl::kDtJacND;
  hipLaunchKernelGGL(rl::k_dt_eval_jac, dim3((N + 127) / 128, B, slices), dim3(128), 0, ctx->stream, ja);
  RL_HIP(hipGetLastError());
  RL_HIP(hipMemcpyAsync(jac_eq, dje.p, dje.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipMemcpyAsync(jac_ineq, dji.p, dji.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipMemcpyAsync(grad_cost, dgc.p, dgc.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipStreamSynchronize(ctx->stream));
  return RL_OK;
}

int rl_mintime_solve_batch_dev(rl_ctx* ctx, const double* model, int B, int N, const double* s, const double* kappa,
                               const double* left, const double* right, int bounds_per_instance, double margin,
                               double track_length, double average_track_width, double speed_cap, double* X, double* U,
                               double* T, int max_iter, double tol, double* stats) {
  if (!ctx || !model || !s || !kappa || !left || !right || !X || !U || !T || !stats) return fail(RL_ERR_ARG, "null argument");
  if (B <= 0 || N < 8 || !(track_length > 0.0) || max_iter < 1 || !(tol > 0.0)) return fail(RL_ERR_ARG, "bad sizes (N >= 8 nodes)");
  if (!(average_track_width > 0.0) || !(speed_cap > 0.0)) return fail(RL_ERR_ARG, "bad scales");
  RL_HIP(hipSetDevice(ctx->device));
  rl::MtProblem P;
  for (int i = 0; i < rl::DT_NPARAM; ++i) P.p[i] = model[i];
  P.N = N; P.bounds_per_instance = bounds_per_instance ? 1 : 0;
  P.margin = margin; P.track_length = track_length;
  // min_time_optimizer.py:109-113: scale_x = (1, average_track_width, 1, 1, 0.5, speed_cap),
  // scale_u = (Fd_max, |Fb_max|, delta_max, 50 mass), scale_t = 1
  const double sx[6] = {1.0, average_track_width, 1.0, 1.0, 0.5, speed_cap};
  const double su[4] = {model[RL_DT_FD_MAX], std::fabs(model[RL_DT_FB_MAX]), model[RL_DT_DELTA_MAX], model[RL_DT_MASS] * 50.0};
  for (int c = 0; c < 5; ++c) P.sw[c] = sx[1 + c];
  P.sw[5] = su[0]; P.sw[6] = su[2]; P.sw[7] = su[3]; P.sw[8] = 1.0;
  for (int c = 0; c < 6; ++c) P.se[c] = 1.0 / sx[c];
  P.se[6] = 1.0 / su[3];
  P.s = s; P.kappa = kappa; P.left = left; P.right = right;
  const size_t bn = (size_t)B * N;
  const size_t counts[] = {bn * rl::kMtNv, bn * rl::kMtNi, bn * rl::kMtNe, bn * rl::kMtNi, bn * rl::kMtNf,
                           bn * rl::kMtNf * rl::kMtLoc, bn * rl::kMtLoc * rl::kMtLoc, bn * rl::kMtNv, bn * rl::kMtNe,
                           bn * 3 * 256, bn * 16, (size_t)B * 16, bn * 256, bn * 256, bn * 16, (size_t)B * 2 * rl::kMtFilter, bn * rl::kMtHw, bn * 16, bn * rl::kMtGc};
  size_t total = 0;
  for (size_t c : counts) total += Arena::pad(c * sizeof(double));
  Arena ar(ctx);
  RL_HIP(ar.reserve(total));
  rl::MtState st;
  st.B = B; st.N = N;
  st.w = ar.take<double>(counts[0]); st.s = ar.take<double>(counts[1]); st.y = ar.take<double>(counts[2]);
  st.z = ar.take<double>(counts[3]); st.fun = ar.take<double>(counts[4]); st.jac = ar.take<double>(counts[5]);
  st.hes = ar.take<double>(counts[6]); st.dw = ar.take<double>(counts[7]); st.dy = ar.take<double>(counts[8]);
  st.blk = ar.take<double>(counts[9]); st.vec = ar.take<double>(counts[10]); st.scal = ar.take<double>(counts[11]);
  st.dblk = ar.take<double>(counts[12]); st.eblk = ar.take<double>(counts[13]); st.rhs = ar.take<double>(counts[14]);
  st.filt = ar.take<double>(counts[15]); st.hw = ar.take<double>(counts[16]); st.r1 = ar.take<double>(counts[17]); st.gc = ar.take<double>(counts[18]);
  st.tol = tol;

  double mt_mu0, mt_delta0;
  {
    // Strategy constants of the line search / barrier update.  Measured on 1024 width-perturbed MGKT tracks:
    // (d_up, a_lo, mu_kappa) = (5, 0.3, 10) 96.5 iterations on average, (3, 0.2, 30) 77.8 (round 2).  Round 4: the damping
    // delta came down only after a step longer than a_hi = 0.9 -- but for dozens of iterations per barrier problem the
    // step is cut by the fraction-to-the-boundary rule to 0.3 .. 0.5 and accepted without a halving (iteration history,
    // tools/mintime_history.py), so delta stayed at 0.5 and the iteration crawled.  (a_hi, a_lo) = (0.25, 0.1): 82.1 -> 55.9
    // iterations on the benchmark batch, the same optima (lap times equal to 1e-12), 89.4 -> 66.8 on the 768 instances of
    // tools/mintime_robustness.py, all converged; a plateau: a_hi 0.11 .. 0.25, a_lo 0.05 .. 0.1, d_down 0.2 .. 0.5, d_up 2 .. 3 all
    // give 55 .. 57.  mu_kappa 60 or mu0 0.05 would take off three more; left alone.  They are COMPILE-TIME constants of the
    // product; only a diagnostic build (hipcc -DRL_ABLATION) reads RL_MT_* overrides from the environment.
#ifdef RL_ABLATION
    auto knob = [](const char* name, double dflt) { const char* v = getenv(name); return v ? atof(v) : dflt; };
#else
    auto knob = [](const char*, double dflt) { return dflt; };
#endif
    st.d_down = knob("RL_MT_D_DOWN", 0.4); st.d_up = knob("RL_MT_D_UP", 3.0);
    st.a_hi = knob("RL_MT_A_HI", 0.25); st.a_lo = knob("RL_MT_A_LO", 0.1);
    st.mu_fac = knob("RL_MT_MU_FAC", 0.2); st.mu_pow = knob("RL_MT_MU_POW", 1.5); st.mu_kappa = knob("RL_MT_MU_KAPPA", 30.0);
    // dual step length <= dual_cap x primal step length (0 = uncoupled): with 1 all 768 instances of tools/mintime_robustness.py
    // converge (750 uncoupled: the multipliers ran away while the primal step was cut to a few per cent), at 82 instead of 78
    // iterations on the benchmark batch
    st.dual_cap = knob("RL_MT_DUAL_CAP", 1.0);
    st.th_filter = knob("RL_MT_TH_FILTER", 1e-4);   // 2e-5 .. 1e-3 all converge for mu0 = 0.05, 0.1, 0.2; 1e-2 blocks the early phase
    mt_mu0 = knob("RL_MT_MU0", 1e-1); mt_delta0 = knob("RL_MT_DELTA0", 1e-4);
  }
  RL_HIP(hipMemsetAsync(st.scal, 0, counts[11] * sizeof(double), ctx->stream));
  if (ctx->mt_hes_sweep) RL_HIP(hipMemsetAsync(st.hes, 0, counts[6] * sizeof(double), ctx->stream));
  // A few sub-batches on as many streams: the KKT elimination is one wave per instance and latency bound (its time
  // does not depend on the batch), the derivative kernels are throughput bound -- with the sub-batches offset by the
  // in-order queues one's elimination runs beside another's derivatives.  Instances are independent, so the split
  // changes no result.  The extra streams fork from / join into the context's stream with events:
  // for the caller everything is still "enqueued on the context's stream".
  const int ngrp = B >= 4 * ctx->mt_groups ? ctx->mt_groups : 1;
  for (int g = 0; g + 1 < ngrp; ++g) {
    if (!ctx->aux_stream[g]) {
      RL_HIP(hipStreamCreateWithFlags(&ctx->aux_stream[g], hipStreamNonBlocking));
      RL_HIP(hipEventCreateWithFlags(&ctx->ev_join[g], hipEventDisableTiming));
    }
  }
  if (ngrp > 1 && !ctx->ev_fork) RL_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  struct Grp { int b0, nb; rl::MtProblem P; rl::MtState st; hipStream_t q; double *X, *U, *T, *stats; };
  Grp grp[rl_ctx::kMaxGroups];
  for (int g = 0; g < ngrp; ++g) {
    Grp& G = grp[g];
    G.b0 = (int)((long long)B * g / ngrp);
    G.nb = (int)((long long)B * (g + 1) / ngrp) - G.b0;
    G.q = g == 0 ? ctx->stream : ctx->aux_stream[g - 1];
    const size_t o = (size_t)G.b0 * N;
    G.P = P;
    if (P.bounds_per_instance) { G.P.left = left + o; G.P.right = right + o; }
    G.st = st; G.st.B = G.nb;
    G.st.w = st.w + o * rl::kMtNv; G.st.s = st.s + o * rl::kMtNi; G.st.y = st.y + o * rl::kMtNe; G.st.z = st.z + o * rl::kMtNi;
    G.st.fun = st.fun + o * rl::kMtNf; G.st.jac = st.jac + o * rl::kMtNf * rl::kMtLoc; G.st.hes = st.hes + o * rl::kMtLoc * rl::kMtLoc;
    G.st.dw = st.dw + o * rl::kMtNv; G.st.dy = st.dy + o * rl::kMtNe; G.st.blk = st.blk + o * 3 * 256; G.st.vec = st.vec + o * 16;
    G.st.scal = st.scal + (size_t)G.b0 * 16; G.st.filt = st.filt + (size_t)G.b0 * 2 * rl::kMtFilter; G.st.hw = st.hw + o * rl::kMtHw; G.st.dblk = st.dblk + o * 256; G.st.eblk = st.eblk + o * 256; G.st.rhs = st.rhs + o * 16; G.st.r1 = st.r1 + o * 16; G.st.gc = st.gc + o * rl::kMtGc;
    G.X = X + o * 6; G.U = U + o * 4; G.T = T + o; G.stats = stats + (size_t)G.b0 * 12;
  }
  if (ngrp > 1) {   // fork: the other streams start after everything already enqueued on the first
    RL_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
    for (int g = 1; g < ngrp; ++g) RL_HIP(hipStreamWaitEvent(grp[g].q, ctx->ev_fork, 0));
  }
  auto join = [&]() -> hipError_t {   // the first stream continues after everything enqueued on the others
    for (int g = 1; g < ngrp; ++g) {
      hipError_t e = hipEventRecord(ctx->ev_join[g - 1], grp[g].q);
      if (e != hipSuccess) return e;
      e = hipStreamWaitEvent(ctx->stream, ctx->ev_join[g - 1], 0);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  };
  const dim3 bn64(64);
  for (int g = 0; g < ngrp; ++g) {
    const Grp& G = grp[g];
    const dim3 gn((N + 63) / 64, G.nb);
    hipLaunchKernelGGL(rl::k_mt_pack, gn, bn64, 0, G.q, G.P, G.st, (const double*)G.X, (const double*)G.U, (const double*)G.T);
    hipLaunchKernelGGL(rl::k_mt_derivs<0>, gn, bn64, 0, G.q, G.P, G.st);
    hipLaunchKernelGGL(rl::k_mt_init, dim3((N * rl::kMtNi + 255) / 256, G.nb), dim3(256), 0, G.q, G.P, G.st, mt_mu0, mt_delta0);
  }
  RL_HIP(hipGetLastError());
  if (ctx->mt_poll) {
    if (!ctx->poll_stream) RL_HIP(hipStreamCreateWithFlags(&ctx->poll_stream, hipStreamNonBlocking));
    for (int g = 0; g < ngrp; ++g)
      if (!ctx->ev_chunk[g]) RL_HIP(hipEventCreateWithFlags(&ctx->ev_chunk[g], hipEventDisableTiming));
    for (auto& e : ctx->ev_poll)
      if (!e) RL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (ctx->poll_cap < (size_t)B * 16) {
      if (ctx->poll_host) RL_HIP(hipHostFree(ctx->poll_host));
      ctx->poll_host = nullptr; ctx->poll_cap = 0;
      RL_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->poll_host), 2 * (size_t)B * 16 * sizeof(double), hipHostMallocDefault));
      ctx->poll_cap = (size_t)B * 16;
    }
  }
  for (int it = 0; it < max_iter; ++it) {   // finished instances return at once from every kernel
    for (int g = 0; g < ngrp; ++g) {
      const Grp& G = grp[g];
      const dim3 gn((N + 63) / 64, G.nb);
      if (ctx->mt_hes_sweep) {   // cross-check: forward (over forward) duals through the whole pair function
        hipLaunchKernelGGL(rl::k_mt_derivs<0>, gn, bn64, 0, G.q, G.P, G.st);
        hipLaunchKernelGGL(rl::k_mt_derivs<1>, dim3((N + 63) / 64, G.nb, rl::kMtJacSlices), bn64, 0, G.q, G.P, G.st);
        hipLaunchKernelGGL(rl::k_mt_derivs<2>, dim3((N + 63) / 64, G.nb, rl::kMtHesSlices), bn64, 0, G.q, G.P, G.st);
      } else {
        hipLaunchKernelGGL(rl::k_mt_values, gn, bn64, 0, G.q, G.P, G.st);   // functions; values of the dynamics at the two ends -> midpoint
        hipLaunchKernelGGL(rl::k_mt_jac_dirs, dim3((N + rl::kMtJacNodes - 1) / rl::kMtJacNodes, G.nb, 3), bn64, 0, G.q, G.P, G.st);
        if (ctx->mt_unfused) hipLaunchKernelGGL(rl::k_mt_jac_assemble, dim3(N, G.nb), bn64, 0, G.q, G.P, G.st);
        hipLaunchKernelGGL(rl::k_mt_hes_point<0>, dim3(rl::mt_hes_blocks(N), G.nb, 1), bn64, 0, G.q, G.P, G.st);
        hipLaunchKernelGGL(rl::k_mt_hes_point<1>, dim3(rl::mt_hes_blocks(N), G.nb, 2), bn64, 0, G.q, G.P, G.st);
        if (ctx->mt_unfused) {
          hipLaunchKernelGGL(rl::k_mt_hes_assemble, dim3(N, G.nb), bn64, 0, G.q, G.P, G.st);
        } else {   // Jacobian, Hessian, blocks and right-hand side in one pass over the pairs
          hipLaunchKernelGGL(rl::k_mt_node, dim3(((N + rl::kMtRun - 1) / rl::kMtRun + rl::kMtNodeGroups - 1) / rl::kMtNodeGroups, G.nb), bn64, 0, G.q, G.P, G.st);
          hipLaunchKernelGGL(rl::k_mt_prepare2, dim3(G.nb), dim3(256), 0, G.q, G.P, G.st);
        }
      }
      if (ctx->mt_hes_sweep || ctx->mt_unfused) {
        hipLaunchKernelGGL(rl::k_mt_prepare, dim3(G.nb), dim3(64), 0, G.q, G.P, G.st);
        hipLaunchKernelGGL(rl::k_mt_assemble, dim3(N, G.nb), dim3(64), 0, G.q, G.P, G.st);
        hipLaunchKernelGGL(rl::k_mt_gc_pack, dim3(N, G.nb), dim3(64), 0, G.q, G.P, G.st);
      }
      if (ctx->mt_kkt4 && N >= 64) hipLaunchKernelGGL(rl::k_mt_kkt4, dim3(G.nb), dim3(256), 0, G.q, G.P, G.st);
      else hipLaunchKernelGGL(rl::k_mt_kkt, dim3(G.nb), dim3(128), 0, G.q, G.P, G.st);
      hipLaunchKernelGGL(rl::k_mt_dir, dim3(rl::kMtDirBlocks(N), G.nb), bn64, 0, G.q, G.P, G.st);
      hipLaunchKernelGGL(rl::k_mt_trial, gn, bn64, 0, G.q, G.P, G.st);
      hipLaunchKernelGGL(rl::k_mt_step, dim3(G.nb), dim3(256), 0, G.q, G.P, G.st);
      hipLaunchKernelGGL(rl::k_mt_trial_b, gn, bn64, 0, G.q, G.P, G.st);        // halvings of the instances whose first trial point was rejected
      hipLaunchKernelGGL(rl::k_mt_step_b, dim3(G.nb), dim3(256), 0, G.q, G.P, G.st);
    }
    if (ctx->mt_poll && (it & 7) == 7) {   // host entry point only: stop once every instance has finished
      const int chunk = it >> 3, slot = chunk & 1;
      for (int g = 0; g < ngrp; ++g) {
        RL_HIP(hipEventRecord(ctx->ev_chunk[g], grp[g].q));
        RL_HIP(hipStreamWaitEvent(ctx->poll_stream, ctx->ev_chunk[g], 0));
      }
      // The copy is ordered after THIS chunk's kernels only; the next chunk's kernels, enqueued below, may already be writing
      // `scal` while it runs -- an unsynchronised read, deliberately: the host looks at nothing but word 5 of every instance
      // (status: 0 -> 1 / 2, written once, never back; an aligned 8-byte word is copied whole), so a stale value can only
      // delay the early exit by one chunk and a fresh one cannot end it early: `all` needs EVERY instance finished, and a
      // finished instance's kernels return at their first line.
      RL_HIP(hipMemcpyAsync(ctx->poll_host + (size_t)slot * ctx->poll_cap, st.scal, (size_t)B * 16 * sizeof(double),
                            hipMemcpyDeviceToHost, ctx->poll_stream));
      RL_HIP(hipEventRecord(ctx->ev_poll[slot], ctx->poll_stream));
      if (chunk >= 1) {   // the chunk before this one: its copy is done, or nearly, while this chunk keeps the GPU busy
        RL_HIP(hipEventSynchronize(ctx->ev_poll[slot ^ 1]));
        const double* h = ctx->poll_host + (size_t)(slot ^ 1) * ctx->poll_cap;
        bool all = true;
        for (int b = 0; b < B && all; ++b) all = h[(size_t)b * 16 + 5] != 0.0;
        if (all) break;
      }
    }
  }
  if (ctx->mt_poll) RL_HIP(hipStreamSynchronize(ctx->poll_stream));   // no copy in flight into the pinned buffer when the call returns
  // final residuals of the instances still running (status 0 = iteration limit)
  for (int g = 0; g < ngrp; ++g) {
    const Grp& G = grp[g];
    const dim3 gn((N + 63) / 64, G.nb);
    hipLaunchKernelGGL(rl::k_mt_derivs<0>, gn, bn64, 0, G.q, G.P, G.st);
    if (ctx->mt_hes_sweep) {
      hipLaunchKernelGGL(rl::k_mt_derivs<1>, dim3((N + 63) / 64, G.nb, rl::kMtJacSlices), bn64, 0, G.q, G.P, G.st);
    } else {
      hipLaunchKernelGGL(rl::k_mt_hes_values, gn, bn64, 0, G.q, G.P, G.st);
      hipLaunchKernelGGL(rl::k_mt_jac_dirs, dim3((N + rl::kMtJacNodes - 1) / rl::kMtJacNodes, G.nb, 3), bn64, 0, G.q, G.P, G.st);
      hipLaunchKernelGGL(rl::k_mt_jac_assemble, dim3(N, G.nb), bn64, 0, G.q, G.P, G.st);
    }
    hipLaunchKernelGGL(rl::k_mt_residuals, dim3(G.nb), dim3(64), 0, G.q, G.P, G.st);
    hipLaunchKernelGGL(rl::k_mt_unpack, gn, bn64, 0, G.q, G.P, G.st, G.X, G.U, G.T);
    hipLaunchKernelGGL(rl::k_mt_stats, dim3((G.nb + 63) / 64), bn64, 0, G.q, G.st, G.stats);
  }
  RL_HIP(hipGetLastError());
  RL_HIP(join());
  RL_HIP(ar.end());
  return RL_OK;
}

int rl_mintime_solve_batch(rl_ctx* ctx, const double* model, int B, int N, const double* s, const double* kappa,
                           const double* left, const double* right, int bounds_per_instance, double margin,
                           double track_length, double average_track_width, double speed_cap, double* X, double* U,
                           double* T, int max_iter, double tol, double* stats) {
  if (!ctx || !model || !s || !kappa || !left || !right || !X || !U || !T || !stats) return fail(RL_ERR_ARG, "null argument");
  if (B <= 0 || N < 8) return fail(RL_ERR_ARG, "bad sizes (N >= 8 nodes)");
  RL_HIP(hipSetDevice(ctx->device));
  const size_t bn = (size_t)B * N, nb_ = bounds_per_instance ? bn : (size_t)N;
  for (size_t i = 0; i < nb_; ++i)
    if (!(right[i] + margin < left[i] - margin)) return fail(RL_ERR_ARG, "track narrower than the vehicle plus margins (min_time_optimizer.py:135)");
  PoolBuf<double> ds(ctx), dk(ctx), dl(ctx), dr(ctx), dX(ctx), dU(ctx), dT(ctx), dst(ctx);
  RL_HIP(ds.alloc(N)); RL_HIP(dk.alloc(N)); RL_HIP(dl.alloc(nb_)); RL_HIP(dr.alloc(nb_));
  RL_HIP(dX.alloc(bn * 6)); RL_HIP(dU.alloc(bn * 4)); RL_HIP(dT.alloc(bn)); RL_HIP(dst.alloc((size_t)B * 12));
  auto up = [&](PoolBuf<double>& d, const double* h) { return hipMemcpyAsync(d.p, h, d.n * sizeof(double), hipMemcpyHostToDevice, ctx->stream); };
  RL_HIP(up(ds, s)); RL_HIP(up(dk, kappa)); RL_HIP(up(dl, left)); RL_HIP(up(dr, right));
  RL_HIP(up(dX, X)); RL_HIP(up(dU, U)); RL_HIP(up(dT, T));
  ctx->mt_poll = true;
  const int rc = rl_mintime_solve_batch_dev(ctx, model, B, N, ds.p, dk.p, dl.p, dr.p, bounds_per_instance, margin, track_length,
                                            average_track_width, speed_cap, dX.p, dU.p, dT.p, max_iter, tol, dst.p);
  ctx->mt_poll = false;
  if (rc) return rc;
  RL_HIP(hipMemcpyAsync(X, dX.p, dX.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipMemcpyAsync(U, dU.p, dU.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipMemcpyAsync(T, dT.p, dT.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipMemcpyAsync(stats, dst.p, dst.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipStreamSynchronize(ctx->stream));
  return RL_OK;
}

}  // extern "C"

static int sweep_single(rl_ctx* ctx, rl_track* trk, const int* i_start, int max_iter, double* cx, double* cy,
                        double* points, int* n_success, rl_stats* stats, bool joint) {
  if (!ctx || !trk || !cx || !cy) return fail(RL_ERR_ARG, "null argument");
  RL_HIP(hipSetDevice(ctx->device));
  const int n = trk->n, N = trk->N;
  if (int rc = rl_track_set_control_points(trk, cx, cy)) return rc;
  PoolBuf<double> dctrl(ctx), dpts(ctx);
  PoolBuf<int> dns(ctx), dst(ctx);
  RL_HIP(dctrl.alloc((size_t)n * 2)); RL_HIP(dpts.alloc((size_t)N * RL_NCOL));
  RL_HIP(dns.alloc((size_t)2 * (max_iter > 0 ? max_iter : 1))); RL_HIP(dst.alloc(1));
  RL_HIP(hipMemsetAsync(dpts.p, 0, dpts.n * sizeof(double), ctx->stream));
  RL_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  int rc = solve_batch_common(ctx, trk, RL_BOUNDS_SHARED_RINGS, nullptr, 1, i_start, max_iter,
                              RL_SEARCH_WINDOWED, dctrl.p, nullptr, dpts.p, dns.p, dst.p, stats, nullptr, joint);
  if (rc) return rc;
  RL_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  PoolBuf<double> dseg(ctx), dcx(ctx);
  if (points && trk->length > 0.0) {
    // DIST_TO_SF_BWD / _FWD of the final table (models/trajectory.py:283-289): GK21 segment lengths of the
    // optimised spline, accumulated in the reference's order; DIST_FWD uses the length the spline object
    // was CONSTRUCTED with (set_control_point never updates _length, trajectory.py:296-298)
    RL_HIP(dseg.alloc(N)); RL_HIP(dcx.alloc((size_t)2 * n));
    hipLaunchKernelGGL(rl::k_split_ctrl, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, dctrl.p, n, dcx.p);
    const dim3 grid((N + 127) / 128), block(128);
    if (trk->k == 3)
      hipLaunchKernelGGL(rl::k_sample_geometry<3>, grid, block, 0, ctx->stream, trk->t.p, trk->nt, dcx.p, dcx.p + n, (const double*)nullptr, N, (double*)nullptr, dseg.p);
    else
      hipLaunchKernelGGL(rl::k_sample_geometry<5>, grid, block, 0, ctx->stream, trk->t.p, trk->nt, dcx.p, dcx.p + n, (const double*)nullptr, N, (double*)nullptr, dseg.p);
    hipLaunchKernelGGL(rl::k_sample_cumsum, dim3(1), dim3(64), 0, ctx->stream, dseg.p, N, trk->length, dpts.p);
    RL_HIP(hipGetLastError());
  }
  std::vector<double> ctrl((size_t)2 * n);
  RL_HIP(hipMemcpyAsync(ctrl.data(), dctrl.p, ctrl.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (points) RL_HIP(hipMemcpyAsync(points, dpts.p, dpts.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  std::vector<int> ns_host((size_t)2 * (max_iter > 0 ? max_iter : 1));
  RL_HIP(hipMemcpyAsync(ns_host.data(), dns.p, (size_t)2 * max_iter * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  RL_HIP(hipStreamSynchronize(ctx->stream));
  for (int j = 0; j < n; ++j) { cx[j] = ctrl[2 * j]; cy[j] = ctrl[2 * j + 1]; }
  if (n_success) {  // sweep: [max_iter][fwd,bwd]; sliding window: [max_iter]
    if (joint) for (int j = 0; j < max_iter; ++j) n_success[j] = ns_host[2 * j];
    else for (int j = 0; j < 2 * max_iter; ++j) n_success[j] = ns_host[j];
  }
  if (points) {
    for (int i = 0; i < N; ++i) { points[(size_t)i * RL_NCOL + 17] = (double)i; points[(size_t)i * RL_NCOL + 18] = -1.0; }
  }
  if (stats) {
    float ms = 0.f;
    RL_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    stats->kernel_ms = ms;
  }
  return RL_OK;
}

#ifdef RL_MT_HIST
extern "C" int rl_debug_mt_hist(unsigned long long* out) {   // diagnostic build only
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rl::g_mt_hist), 16 * sizeof(unsigned long long));
}
#endif
