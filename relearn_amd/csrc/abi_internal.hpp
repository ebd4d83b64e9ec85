// abi_internal.hpp — what the translation units of the C ABI (abi*.hip) share: error plumbing, device-memory and
// copy helpers, and the few functions defined in one unit and used by another.
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rl_chacha.h"
#include "../../include/rl_detmath.h"
#include "engine.hpp"
#include "host/cbor.hpp"
#include "kernels.hpp"

// ---------------------------------------------------------------- error plumbing
extern thread_local std::string g_last_error_no_engine;  // abi.hip

// rl_actor_critic_update_begin may have left a critic chain in flight on the auxiliary stream.  Whatever an entry point
// enqueues on the main stream is ordered behind that chain (a device-side wait, no host synchronisation) — except the
// calls that say they handle it themselves (`settle` false: a rollout into another trajectory, _finish).
static inline void engine_settle(rl_engine *e) {
  if (e->pending.active && !e->pending.joined) {
    RL_HIP_CHECK(hipStreamWaitEvent(e->main_stream, e->ev_join, 0));
    e->pending.joined = true;
  }
}

template <typename F>
static inline int32_t guarded(rl_engine *eng, F &&f, bool settle = true) {
  try {
    if (eng != nullptr) eng->call_epoch += 1;
    if (eng != nullptr && settle) engine_settle(eng);
    f();
    if (eng != nullptr) {
      // a kernel that could not be launched (bad configuration, out of resources) leaves only a sticky error behind
      const hipError_t le = hipGetLastError();
      if (le != hipSuccess) throw RlError(RL_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString(le));
    }
    return RL_OK;
  } catch (const RlError &e) {
    if (eng != nullptr) eng->error_epoch += 1;  // (optimisers re-read their step count from the device: rl_adam)
    (eng ? eng->last_error : g_last_error_no_engine) = e.what();
    return e.code;
  } catch (const std::exception &e) {
    if (eng != nullptr) eng->error_epoch += 1;
    (eng ? eng->last_error : g_last_error_no_engine) = e.what();
    return RL_ERR_INVALID_ARGUMENT;
  } catch (...) {
    if (eng != nullptr) eng->error_epoch += 1;
    (eng ? eng->last_error : g_last_error_no_engine) = "unknown error";
    return RL_ERR_INVALID_ARGUMENT;
  }
}

template <typename T>
static inline T *dalloc(size_t count) {
  void *p = nullptr;
  RL_HIP_CHECK(hipMalloc(&p, (count ? count : 1) * sizeof(T)));
  return (T *)p;
}

static inline void dfree(void *p) {
  if (p) (void)hipFree(p);
}

// ---------------------------------------------------------------- helpers
static inline uint64_t b_total(const rl_traj *t) { return t->B * (uint64_t)t->eng->n_ranks; }

static inline void sync(rl_engine *e) { RL_HIP_CHECK(hipStreamSynchronize(e->stream)); }

static inline void h2d(rl_engine *e, void *d, const void *h, size_t bytes) {
  RL_HIP_CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, e->stream));
  sync(e);
}

static inline void d2h(rl_engine *e, void *h, const void *d, size_t bytes) {
  RL_HIP_CHECK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, e->stream));
  sync(e);
}

// The fused kernels' range guard (bf16_tile.hpp range_guard) found weights / observations outside the range in which the
// 2^96-scaled forward is exact: their results are not to be used.  Called after the entry point has synchronised the
// stream its fused launches ran on (every caller has just read results back): the error words live in host memory, one per
// chain (RL_GUARD_POLICY: policy passes and the DQN gradient, RL_GUARD_CRITIC: the critic step), and a caller looks at
// the chains ITS launches ran — under rl_actor_critic_update_begin the critic chain is still in flight on the auxiliary
// stream when the TRPO chain's statistics are read, and its violation is the _finish call's to report.  The violation
// also set the chain's veto word on the device (TrajDev::range_err): the optimiser and line-search kernels behind the
// guarded launch left parameters, moments and step count as they were, so a refused update is refused whole.
constexpr uint32_t RL_GUARD_BOTH = (1u << RL_GUARD_POLICY) | (1u << RL_GUARD_CRITIC);
static inline void range_check(rl_traj *t, uint32_t chains = RL_GUARD_BOTH) {
  // the next entry point's first fused launches check again
  if (chains & (1u << RL_GUARD_POLICY)) t->guard_next_policy = true;
  if (chains & (1u << RL_GUARD_CRITIC)) t->guard_next_critic = true;
  if (t->h_range_err == nullptr) return;
  volatile uint32_t *w = t->h_range_err;
  uint32_t seen = 0u;
  for (int c = 0; c < 2; ++c)
    if ((chains & (1u << c)) != 0u && w[c] != 0u) seen |= 1u << c;
  if (seen == 0u) return;
  for (int c = 0; c < 2; ++c)
    if (seen & (1u << c)) {
      w[c] = 0u;
      RL_HIP_CHECK(hipMemsetAsync(t->d.range + RL_RANGE_WORDS + c, 0, sizeof(uint32_t), t->eng->stream));
    }
  sync(t->eng);
  throw RlError(RL_ERR_UNSUPPORTED,
                std::string("numeric range of the fused update kernels exceeded in the ") +
                    (seen == RL_GUARD_BOTH ? "policy and critic chains" : (seen & 1u) ? "policy chain" : "critic chain") +
                    " (|pre-activation| bound 2^31 or a non-zero pre-activation below 2^-46 possible, or a non-finite "
                    "observation): the result of this call is not valid and the update was not applied; use "
                    "rl_engine_set_kernel_variant(engine, 1) for these magnitudes (include/relearn_hip.h)");
}
// an entry point that gives up for another reason with guarded launches behind it: their words must not outlive the call
static inline void range_discard(rl_traj *t) {
  try {
    range_check(t);
  } catch (const RlError &) {
  }
}

// ---------------------------------------------------------------- shared between the units (C linkage like the entry
// points they sit next to)
extern "C" {
// abi.hip
void engine_release_child(rl_engine *e);
void traj_plan(rl_traj *t, uint64_t B);
rl_traj *traj_alloc(rl_engine *e, uint64_t n_lanes, uint64_t horizon, uint32_t obs_dim, bool resizable);
void seq_ensure(rl_traj *t, const rl_mlp *mod, bool training);
void traj_ensure_pvec(rl_traj *t, uint64_t P);
rl_mlp *seq_exec(const rl_mlp *m);
}

// Running the recurrent kernels for module `m` on trajectory `t`: `x` is the module they run (m itself, or its zero-padded
// twin with the parameter image refreshed: rl_mlp::exec), and the trajectory is presented with the five observation
// planes it physically has (planes past its logical width are zeros).
struct SeqScope {
  rl_traj *t;
  uint32_t logical_D;
  rl_mlp *x;
  SeqScope(rl_traj *traj, const rl_mlp *m) : t(traj), logical_D(traj->d.D), x(nullptr) {
    RL_REQUIRE(m->in_dim == traj->d.D, "module input width does not match the trajectory");
    x = seq_exec(m);
    if (!m->lane_kernels()) t->d.D = 5;  // (the lane-per-thread kernels run at the module's own input width)
  }
  ~SeqScope() { t->d.D = logical_D; }
  SeqScope(const SeqScope &) = delete;
  SeqScope &operator=(const SeqScope &) = delete;
};
