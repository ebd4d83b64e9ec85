// abi.hip — extern "C" entry points of include/relearn_hip.h (host side only; kernels live in kernels_*.hip).
#include <dlfcn.h>

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>

#include "../../include/rl_chacha.h"
#include "../../include/rl_detmath.h"
#include "engine.hpp"
#include "host/cbor.hpp"
#include "kernels.hpp"

// ---------------------------------------------------------------- error plumbing
static thread_local std::string g_last_error_no_engine;

template <typename F>
static int32_t guarded(rl_engine *eng, F &&f) {
  try {
    f();
    if (eng != nullptr) {
      // a kernel that could not be launched (bad configuration, out of resources) leaves only a sticky error behind
      const hipError_t le = hipGetLastError();
      if (le != hipSuccess) throw RlError(RL_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString(le));
    }
    return RL_OK;
  } catch (const RlError &e) {
    (eng ? eng->last_error : g_last_error_no_engine) = e.what();
    return e.code;
  } catch (const std::exception &e) {
    (eng ? eng->last_error : g_last_error_no_engine) = e.what();
    return RL_ERR_INVALID_ARGUMENT;
  } catch (...) {
    (eng ? eng->last_error : g_last_error_no_engine) = "unknown error";
    return RL_ERR_INVALID_ARGUMENT;
  }
}

template <typename T>
static T *dalloc(size_t count) {
  void *p = nullptr;
  RL_HIP_CHECK(hipMalloc(&p, (count ? count : 1) * sizeof(T)));
  return (T *)p;
}

static void dfree(void *p) {
  if (p) (void)hipFree(p);
}

// ---------------------------------------------------------------- profiling scope
ProfScope::ProfScope(rl_engine *eng, int c) : e(eng), cls(c) {
  if (!e->profiling) return;
  auto get = [&]() {
    hipEvent_t ev;
    if (!e->prof_event_pool.empty()) {
      ev = e->prof_event_pool.back();
      e->prof_event_pool.pop_back();
    } else if (hipEventCreate(&ev) != hipSuccess) {
      ev = nullptr;
    }
    return ev;
  };
  a = get();
  b = get();
  if (a) (void)hipEventRecord(a, e->stream);
}

ProfScope::~ProfScope() {
  if (!e->profiling || !a || !b) return;
  (void)hipEventRecord(b, e->stream);
  e->prof_pending.push_back({cls, {a, b}});
}

static void prof_drain(rl_engine *e) {
  for (auto &it : e->prof_pending) {
    float ms = 0.0f;
    (void)hipEventSynchronize(it.second.second);
    if (hipEventElapsedTime(&ms, it.second.first, it.second.second) == hipSuccess) {
      e->prof_ms[it.first] += (double)ms;
      e->prof_launches[it.first] += 1;
    }
    e->prof_event_pool.push_back(it.second.first);
    e->prof_event_pool.push_back(it.second.second);
  }
  e->prof_pending.clear();
}

// ---------------------------------------------------------------- RCCL through dlopen
// The library must load (and export every symbol) on machines without a GPU or without RCCL, and must not
// clash with an RCCL copy a host program (e.g. PyTorch) already mapped; so RCCL is bound at rl_comm_init.
namespace {
struct Rccl {
  void *handle = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, const void * /*ncclUniqueId by value: 128 bytes*/, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
struct UniqueId { char bytes[128]; };
typedef int (*comm_init_rank_t)(void **, int, UniqueId, int);

void rccl_load() {
  if (g_rccl.handle) return;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) {
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) throw RlError(RL_ERR_COMM, std::string("cannot dlopen librccl: ") + dlerror());
  g_rccl.GetUniqueId = (int (*)(void *))dlsym(g_rccl.handle, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(void **, int, const void *, int))dlsym(g_rccl.handle, "ncclCommInitRank");
  g_rccl.CommDestroy = (int (*)(void *))dlsym(g_rccl.handle, "ncclCommDestroy");
  g_rccl.AllReduce =
      (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(g_rccl.handle, "ncclAllReduce");
  g_rccl.GetErrorString = (const char *(*)(int))dlsym(g_rccl.handle, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce)
    throw RlError(RL_ERR_COMM, "librccl is missing expected symbols");
}

void rccl_check(int rc, const char *what) {
  if (rc != 0)
    throw RlError(RL_ERR_COMM, std::string(what) + ": " +
                                   (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : std::to_string(rc).c_str()));
}
}  // namespace

// ---------------------------------------------------------------- in-process loopback collective
// RELEARN_LOOPBACK_COMM=1: rl_comm_init joins the engines of ONE process that present the same unique id into a
// group whose all-reduce is a kernel summing the ranks' buffers in rank order (all engines on the same GPU, each
// driven by its own host thread).  It exists so that the multi-rank arithmetic — lane sharding by global lane id,
// sample-weighted means over all ranks, identical redundant updates — can be exercised on a one-GPU box; the RCCL
// call path itself is exercised with a one-rank communicator (RELEARN_FORCE_RCCL=1).
__global__ void k_loopback_sum(float *const *bufs, int n_ranks, size_t count) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float s = bufs[0][i];
  for (int r = 1; r < n_ranks; ++r) s = s + bufs[r][i];
  for (int r = 0; r < n_ranks; ++r) bufs[r][i] = s;
}

struct LoopbackGroup {
  std::mutex mu;
  std::condition_variable cv;
  int n_ranks = 0, arrived = 0, joined = 0;
  uint64_t generation = 0;
  std::vector<float *> bufs;
  float **d_bufs = nullptr;
  void barrier(std::unique_lock<std::mutex> &lk) {
    const uint64_t gen = generation;
    if (++arrived == n_ranks) {
      arrived = 0;
      generation += 1;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != gen; });
    }
  }
};
static std::mutex g_loopback_mu;
static std::map<std::string, std::shared_ptr<LoopbackGroup>> g_loopback_groups;

static void loopback_allreduce(rl_engine *e, float *d_buf, size_t count) {
  LoopbackGroup *g = e->loopback;
  RL_HIP_CHECK(hipStreamSynchronize(e->stream));  // this rank's contribution is complete
  std::unique_lock<std::mutex> lk(g->mu);
  g->bufs[e->rank] = d_buf;
  g->barrier(lk);
  if (e->rank == 0) {
    RL_HIP_CHECK(hipMemcpyAsync(g->d_bufs, g->bufs.data(), g->n_ranks * sizeof(float *), hipMemcpyHostToDevice,
                                e->stream));
    hipLaunchKernelGGL(k_loopback_sum, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, e->stream, g->d_bufs,
                       g->n_ranks, count);
    RL_HIP_CHECK(hipStreamSynchronize(e->stream));
  }
  g->barrier(lk);
}

void rl_allreduce_sum_f32(rl_engine *e, float *d_buf, size_t count) {
  if (e->loopback) {
    ProfScope ps(e, RL_K_ALLREDUCE);
    loopback_allreduce(e, d_buf, count);
    return;
  }
  if (e->host_allreduce) {
    ProfScope ps(e, RL_K_ALLREDUCE);
    e->host_allreduce_buf.resize(count);
    RL_HIP_CHECK(hipMemcpyAsync(e->host_allreduce_buf.data(), d_buf, count * sizeof(float), hipMemcpyDeviceToHost,
                                e->stream));
    RL_HIP_CHECK(hipStreamSynchronize(e->stream));
    if (e->host_allreduce(e->host_allreduce_ctx, e->host_allreduce_buf.data(), count) != 0)
      throw RlError(RL_ERR_COMM, "the host all-reduce callback failed");
    RL_HIP_CHECK(hipMemcpyAsync(d_buf, e->host_allreduce_buf.data(), count * sizeof(float), hipMemcpyHostToDevice,
                                e->stream));
    RL_HIP_CHECK(hipStreamSynchronize(e->stream));
    return;
  }
  if (!e->comm) return;
  ProfScope ps(e, RL_K_ALLREDUCE);
  // ncclFloat32 = 7, ncclSum = 0
  rccl_check(g_rccl.AllReduce(d_buf, d_buf, count, 7, 0, e->comm, e->stream), "ncclAllReduce");
}

// ---------------------------------------------------------------- helpers
static uint64_t b_total(const rl_traj *t) { return t->B * (uint64_t)t->eng->n_ranks; }

static void sync(rl_engine *e) { RL_HIP_CHECK(hipStreamSynchronize(e->stream)); }

static void h2d(rl_engine *e, void *d, const void *h, size_t bytes) {
  RL_HIP_CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, e->stream));
  sync(e);
}

static void d2h(rl_engine *e, void *h, const void *d, size_t bytes) {
  RL_HIP_CHECK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, e->stream));
  sync(e);
}

extern "C" {

int32_t rl_abi_version(void) { return RL_ABI_VERSION; }

int32_t rl_device_count(int32_t *count) {
  return guarded(nullptr, [&] {
    RL_REQUIRE(count, "count is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
  });
}

int32_t rl_engine_create(int32_t device_ordinal, rl_engine **out) {
  return guarded(nullptr, [&] {
    RL_REQUIRE(out, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
      throw RlError(RL_ERR_NO_DEVICE, "no HIP device visible: this engine has no CPU fallback");
    RL_REQUIRE(device_ordinal >= 0 && device_ordinal < n, "device ordinal out of range");
    std::unique_ptr<rl_engine> e(new rl_engine());
    e->device = device_ordinal;
    RL_HIP_CHECK(hipSetDevice(device_ordinal));
    RL_HIP_CHECK(hipGetDeviceProperties(&e->prop, device_ordinal));
    if (std::strncmp(e->prop.gcnArchName, "gfx950", 6) != 0)
      throw RlError(RL_ERR_NO_DEVICE,
                    std::string("device is ") + e->prop.gcnArchName + ", this library is built for gfx950 only");
    RL_HIP_CHECK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    RL_HIP_CHECK(hipEventCreate(&e->ev_begin));
    RL_HIP_CHECK(hipEventCreate(&e->ev_end));
    e->pinned_bytes = 1 << 16;
    RL_HIP_CHECK(hipHostMalloc(&e->pinned, e->pinned_bytes, hipHostMallocDefault));
    *out = e.release();
  });
}

static void engine_teardown(rl_engine *e);

int32_t rl_engine_destroy(rl_engine *e) {
  if (!e) return RL_OK;
  if (e->live_handles > 0) {
    e->zombie = true;  // torn down when the last child handle goes away
    return RL_OK;
  }
  engine_teardown(e);
  return RL_OK;
}

static void engine_release_child(rl_engine *e) {
  e->live_handles -= 1;
  if (e->zombie && e->live_handles <= 0) engine_teardown(e);
}

static void engine_teardown(rl_engine *e) {
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  if (e->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(e->comm);
  prof_drain(e);
  for (auto ev : e->prof_event_pool) (void)hipEventDestroy(ev);
  if (e->pinned) (void)hipHostFree(e->pinned);
  (void)hipEventDestroy(e->ev_begin);
  (void)hipEventDestroy(e->ev_end);
  (void)hipStreamDestroy(e->stream);
  delete e;
}

int32_t rl_engine_sync(rl_engine *e) {
  return guarded(e, [&] {
    RL_REQUIRE(e, "engine is NULL");
    sync(e);
    // also drain anything a collective library queued on streams of its own
    RL_HIP_CHECK(hipSetDevice(e->device));
    RL_HIP_CHECK(hipDeviceSynchronize());
  });
}

const char *rl_last_error(const rl_engine *e) { return e ? e->last_error.c_str() : g_last_error_no_engine.c_str(); }

int32_t rl_engine_info(const rl_engine *e, char *name_out, size_t name_cap, char *arch_out, size_t arch_cap,
                       int32_t *compute_units) {
  return guarded(const_cast<rl_engine *>(e), [&] {
    RL_REQUIRE(e, "engine is NULL");
    if (name_out && name_cap) std::snprintf(name_out, name_cap, "%s", e->prop.name);
    if (arch_out && arch_cap) std::snprintf(arch_out, arch_cap, "%s", e->prop.gcnArchName);
    if (compute_units) *compute_units = e->prop.multiProcessorCount;
  });
}

int32_t rl_timer_begin(rl_engine *e) {
  return guarded(e, [&] {
    RL_REQUIRE(e, "engine is NULL");
    RL_HIP_CHECK(hipEventRecord(e->ev_begin, e->stream));
  });
}

int32_t rl_timer_end(rl_engine *e, float *elapsed_ms) {
  return guarded(e, [&] {
    RL_REQUIRE(e && elapsed_ms, "NULL argument");
    RL_HIP_CHECK(hipEventRecord(e->ev_end, e->stream));
    RL_HIP_CHECK(hipEventSynchronize(e->ev_end));
    RL_HIP_CHECK(hipEventElapsedTime(elapsed_ms, e->ev_begin, e->ev_end));
  });
}

int32_t rl_engine_set_kernel_variant(rl_engine *e, int32_t variant) {
  return guarded(e, [&] {
    RL_REQUIRE(e, "engine is NULL");
    RL_REQUIRE(variant >= 0 && variant <= 2, "kernel variant must be 0 (best), 1 (v1 reference kernels) or 2");
    e->kernel_variant = variant;
  });
}

int32_t rl_profile_enable(rl_engine *e, int32_t on) {
  return guarded(e, [&] {
    RL_REQUIRE(e, "engine is NULL");
    sync(e);
    prof_drain(e);
    e->profiling = on != 0;
  });
}

int32_t rl_profile_read(rl_engine *e, double *total_ms_out, uint64_t *launches_out, int32_t reset) {
  return guarded(e, [&] {
    RL_REQUIRE(e, "engine is NULL");
    sync(e);
    prof_drain(e);
    for (int i = 0; i < RL_K_CLASS_COUNT; ++i) {
      if (total_ms_out) total_ms_out[i] = e->prof_ms[i];
      if (launches_out) launches_out[i] = e->prof_launches[i];
      if (reset) {
        e->prof_ms[i] = 0.0;
        e->prof_launches[i] = 0;
      }
    }
  });
}

// ---------------------------------------------------------------- comm
int32_t rl_comm_unique_id(uint8_t id_out[128]) {
  return guarded(nullptr, [&] {
    RL_REQUIRE(id_out, "id_out is NULL");
    if (std::getenv("RELEARN_LOOPBACK_COMM")) {
      static std::atomic<uint64_t> counter{0};
      std::memset(id_out, 0, 128);
      uint64_t v = ++counter;
      std::memcpy(id_out, &v, sizeof(v));
      return;
    }
    rccl_load();
    rccl_check(g_rccl.GetUniqueId(id_out), "ncclGetUniqueId");
  });
}

int32_t rl_comm_init(rl_engine *e, int32_t rank, int32_t n_ranks, const uint8_t unique_id[128]) {
  return guarded(e, [&] {
    RL_REQUIRE(e && unique_id, "NULL argument");
    RL_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "bad rank / n_ranks");
    RL_REQUIRE(!e->has_collective(), "communicator already initialised");
    e->rank = rank;
    e->n_ranks = n_ranks;
    // a single rank needs no communicator; RELEARN_FORCE_RCCL=1 creates a 1-rank one anyway so that the whole
    // RCCL call path (dlopen, ncclCommInitRank, ncclAllReduce on the engine stream) can be exercised on one GPU
    if (std::getenv("RELEARN_LOOPBACK_COMM")) {
      RL_REQUIRE(!e->loopback, "communicator already initialised");
      std::lock_guard<std::mutex> lk(g_loopback_mu);
      std::string key((const char *)unique_id, 128);
      auto &grp = g_loopback_groups[key];
      if (!grp) {
        grp = std::make_shared<LoopbackGroup>();
        grp->n_ranks = n_ranks;
        grp->bufs.assign(n_ranks, nullptr);
        RL_HIP_CHECK(hipSetDevice(e->device));
        grp->d_bufs = dalloc<float *>(n_ranks);
      }
      RL_REQUIRE(grp->n_ranks == n_ranks, "loopback group: inconsistent n_ranks");
      grp->joined += 1;
      e->loopback = grp.get();
      return;
    }
    if (n_ranks == 1 && !std::getenv("RELEARN_FORCE_RCCL")) return;
    rccl_load();
    RL_HIP_CHECK(hipSetDevice(e->device));
    UniqueId id;
    std::memcpy(id.bytes, unique_id, 128);
    comm_init_rank_t init = (comm_init_rank_t)(void *)g_rccl.CommInitRank;
    rccl_check(init(&e->comm, n_ranks, id, rank), "ncclCommInitRank");
  });
}

int32_t rl_comm_init_host(rl_engine *e, int32_t rank, int32_t n_ranks, rl_host_allreduce_fn fn, void *ctx) {
  return guarded(e, [&] {
    RL_REQUIRE(e && fn, "NULL argument");
    RL_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "bad rank / n_ranks");
    RL_REQUIRE(!e->has_collective(), "communicator already initialised");
    e->rank = rank;
    e->n_ranks = n_ranks;
    e->host_allreduce = fn;
    e->host_allreduce_ctx = ctx;
  });
}

int32_t rl_comm_destroy(rl_engine *e) {
  return guarded(e, [&] {
    RL_REQUIRE(e, "engine is NULL");
    e->host_allreduce = nullptr;
    e->host_allreduce_ctx = nullptr;
    if (e->comm) {
      sync(e);
      rccl_check(g_rccl.CommDestroy(e->comm), "ncclCommDestroy");
      e->comm = nullptr;
    }
    e->loopback = nullptr;  // groups live for the life of the process (test facility)
    e->rank = 0;
    e->n_ranks = 1;
  });
}

// ---------------------------------------------------------------- env
int32_t rl_cartpole_params_default(rl_cartpole_params *p) {
  return guarded(nullptr, [&] {
    RL_REQUIRE(p, "params is NULL");
    // PhysicalConstants::default / EnvironmentParams::default (reference src/envs/cartpole.rs:178-216)
    p->gravity = 9.8;
    p->mass_cart = 1.0;
    p->mass_pole = 0.1;
    p->length_half_pole = 0.5;
    p->friction_cart = 0.01;
    p->friction_pole = 0.01;
    p->time_step = 0.02;
    p->action_force = 10.0;
    p->max_pos = 2.4;
    p->max_angle = 12.0 * (3.14159265358979323846 / 180.0);  // 12.0f64.to_radians()
    p->discount_factor = 0.99;
  });
}

int32_t rl_env_create(rl_engine *e, const rl_env_config *cfg, rl_env **out) {
  return guarded(e, [&] {
    RL_REQUIRE(e && cfg && out, "NULL argument");
    *out = nullptr;
    if (cfg->kind != RL_ENV_CARTPOLE && cfg->kind != RL_ENV_CHAIN && cfg->kind != RL_ENV_MEMORY)
      throw RlError(RL_ERR_BUILD_ENV, "unknown env kind");
    const uint64_t mem_actions = cfg->memory_num_actions ? cfg->memory_num_actions : 2;
    const uint64_t mem_history = cfg->memory_num_actions || cfg->memory_history_len ? cfg->memory_history_len : 3;
    if (cfg->kind == RL_ENV_MEMORY && !(mem_actions == 2 && mem_history == 3))
      throw RlError(RL_ERR_BUILD_ENV, "the MemoryGame kernels are built for MemoryGame::new(2, 3) (5 observation features)");
    if (cfg->kind == RL_ENV_CHAIN && !(cfg->chain_size == 0 || cfg->chain_size == 5))
      throw RlError(RL_ERR_BUILD_ENV, "the Chain kernels are built for Chain::default (5 states)");
    if (cfg->limit_kind != RL_LIMIT_NONE && (cfg->max_steps == 0 || cfg->max_steps >= (1ull << 32)))
      throw RlError(RL_ERR_BUILD_ENV, "step limit must be positive and < 2^32");  // StepLimit::new asserts > 0
    if (cfg->n_lanes == 0 || cfg->n_lanes >= (1ull << 31)) throw RlError(RL_ERR_BUILD_ENV, "bad n_lanes");
    RL_HIP_CHECK(hipSetDevice(e->device));
    std::unique_ptr<rl_env> env(new rl_env());
    env->eng = e;
    env->cfg = *cfg;
    env->kind = cfg->kind;
    if (cfg->kind == RL_ENV_CHAIN || cfg->kind == RL_ENV_MEMORY)
      env->D = 5 + (cfg->limit_kind == RL_LIMIT_VISIBLE ? 1 : 0);  // one-hot(5) [+ remaining]
    else
      env->D = cfg->limit_kind == RL_LIMIT_VISIBLE ? 5 : 4;
    env->A = 2;
    const rl_cartpole_params &p = cfg->cartpole;
    CartPoleDev &d = env->dev;
    d.gravity = p.gravity;
    d.mass_pole = p.mass_pole;
    d.length_half_pole = p.length_half_pole;
    d.friction_cart = p.friction_cart;
    d.friction_pole = p.friction_pole;
    d.time_step = p.time_step;
    d.action_force = p.action_force;
    d.max_pos = p.max_pos;
    d.max_angle = p.max_angle;
    // From<PhysicalConstants> for InternalPhysicalConstants (cartpole.rs:238-251)
    double total_mass = p.mass_cart + p.mass_pole;
    d.total_weight = p.gravity * total_mass;
    d.inv_total_mass = 1.0 / total_mass;
    d.mass_length_pole = p.mass_pole * p.length_half_pole;
    d.init_low = -0.05;
    d.init_scale = rl_uniform_f64_inclusive_scale(-0.05, 0.05);
    rl_seed_from_u64(cfg->seed_env, d.key_env);
    rl_seed_from_u64(cfg->seed_actor, d.key_actor);
    d.lane_offset = cfg->lane_offset;
    d.max_steps = cfg->max_steps < (1ull << 32) ? (uint32_t)cfg->max_steps : 0u;
    d.limit_kind = cfg->limit_kind;
    d.chain_size = 5;
    d.mem_actions = cfg->kind == RL_ENV_MEMORY ? (uint32_t)mem_actions : 0u;
    size_t n = cfg->n_lanes;
    env->st.x = dalloc<double>(n);
    env->st.xdot = dalloc<double>(n);
    env->st.th = dalloc<double>(n);
    env->st.thdot = dalloc<double>(n);
    env->st.nv_pos = dalloc<uint8_t>(n);
    env->st.steps_remaining = dalloc<uint32_t>(n);
    env->st.reset_count = dalloc<uint32_t>(n);
    env->d_actions = dalloc<uint8_t>(n);
    env->d_flag = dalloc<uint8_t>(n);
    env->d_reward = dalloc<float>(n);
    env->d_obs = dalloc<float>(n * env->D);
    env->d_term_obs = dalloc<float>(n * env->D);
    RL_HIP_CHECK(hipMemsetAsync(env->st.reset_count, 0, n * sizeof(uint32_t), e->stream));
    for (double *p : {env->st.x, env->st.xdot, env->st.th, env->st.thdot})
      RL_HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(double), e->stream));
    RL_HIP_CHECK(hipMemsetAsync(env->st.nv_pos, 0, n, e->stream));
    RL_HIP_CHECK(hipMemsetAsync(env->d_actions, 0, n, e->stream));
    RL_HIP_CHECK(hipMemsetAsync(env->d_term_obs, 0, n * env->D * sizeof(float), e->stream));
    launch_env_reset(env.get());
    sync(e);
    e->live_handles += 1;
    *out = env.release();
  });
}

int32_t rl_env_destroy(rl_env *env) {
  if (!env) return RL_OK;
  (void)hipSetDevice(env->eng->device);
  (void)hipStreamSynchronize(env->eng->stream);
  dfree(env->st.x);
  dfree(env->st.xdot);
  dfree(env->st.th);
  dfree(env->st.thdot);
  dfree(env->st.nv_pos);
  dfree(env->st.steps_remaining);
  dfree(env->st.reset_count);
  dfree(env->d_actions);
  dfree(env->d_flag);
  dfree(env->d_reward);
  dfree(env->d_obs);
  dfree(env->d_term_obs);
  rl_engine *eng = env->eng;
  delete env;
  engine_release_child(eng);
  return RL_OK;
}

int32_t rl_env_dims(const rl_env *env, uint32_t *obs_dim, uint32_t *n_actions) {
  return guarded(env ? env->eng : nullptr, [&] {
    RL_REQUIRE(env, "env is NULL");
    if (obs_dim) *obs_dim = env->D;
    if (n_actions) *n_actions = env->A;
  });
}

int32_t rl_env_reset(rl_env *env) {
  return guarded(env ? env->eng : nullptr, [&] {
    RL_REQUIRE(env, "env is NULL");
    launch_env_reset(env);
  });
}

int32_t rl_env_observe(rl_env *env, float *obs_out) {
  return guarded(env ? env->eng : nullptr, [&] {
    RL_REQUIRE(env && obs_out, "NULL argument");
    launch_env_observe(env, env->d_obs);
    d2h(env->eng, obs_out, env->d_obs, env->cfg.n_lanes * env->D * sizeof(float));
  });
}

int32_t rl_env_upload_actions(rl_env *env, const uint8_t *actions) {
  return guarded(env ? env->eng : nullptr, [&] {
    RL_REQUIRE(env && actions, "NULL argument");
    h2d(env->eng, env->d_actions, actions, env->cfg.n_lanes);
  });
}

int32_t rl_env_step_resident(rl_env *env) {
  return guarded(env ? env->eng : nullptr, [&] {
    RL_REQUIRE(env, "env is NULL");
    launch_env_step(env);
    env->t_global += 1;
  });
}

int32_t rl_env_step(rl_env *env, const uint8_t *actions, float *reward_out, uint8_t *flag_out, float *obs_out,
                    float *term_obs_out) {
  return guarded(env ? env->eng : nullptr, [&] {
    RL_REQUIRE(env && actions, "NULL argument");
    rl_engine *e = env->eng;
    size_t n = env->cfg.n_lanes;
    RL_HIP_CHECK(hipMemcpyAsync(env->d_actions, actions, n, hipMemcpyHostToDevice, e->stream));
    launch_env_step(env);
    env->t_global += 1;
    if (reward_out)
      RL_HIP_CHECK(hipMemcpyAsync(reward_out, env->d_reward, n * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    if (flag_out) RL_HIP_CHECK(hipMemcpyAsync(flag_out, env->d_flag, n, hipMemcpyDeviceToHost, e->stream));
    if (obs_out)
      RL_HIP_CHECK(
          hipMemcpyAsync(obs_out, env->d_obs, n * env->D * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    if (term_obs_out)
      RL_HIP_CHECK(hipMemcpyAsync(term_obs_out, env->d_term_obs, n * env->D * sizeof(float), hipMemcpyDeviceToHost,
                                  e->stream));
    sync(e);
  });
}

int32_t rl_env_get_state(rl_env *env, double *state4, int32_t *nv_pos, uint64_t *steps_remaining,
                         uint64_t *reset_count) {
  return guarded(env ? env->eng : nullptr, [&] {
    RL_REQUIRE(env && state4 && nv_pos && steps_remaining && reset_count, "NULL argument");
    rl_engine *e = env->eng;
    size_t n = env->cfg.n_lanes;
    d2h(e, state4 + 0 * n, env->st.x, n * sizeof(double));
    d2h(e, state4 + 1 * n, env->st.xdot, n * sizeof(double));
    d2h(e, state4 + 2 * n, env->st.th, n * sizeof(double));
    d2h(e, state4 + 3 * n, env->st.thdot, n * sizeof(double));
    std::vector<uint8_t> nv(n);
    std::vector<uint32_t> a(n), b(n);
    d2h(e, nv.data(), env->st.nv_pos, n);
    d2h(e, a.data(), env->st.steps_remaining, n * sizeof(uint32_t));
    d2h(e, b.data(), env->st.reset_count, n * sizeof(uint32_t));
    for (size_t i = 0; i < n; ++i) {
      nv_pos[i] = nv[i];
      steps_remaining[i] = a[i];
      reset_count[i] = b[i];
    }
  });
}

int32_t rl_env_set_state(rl_env *env, const double *state4, const int32_t *nv_pos, const uint64_t *steps_remaining,
                         const uint64_t *reset_count) {
  return guarded(env ? env->eng : nullptr, [&] {
    RL_REQUIRE(env && state4 && nv_pos && steps_remaining && reset_count, "NULL argument");
    rl_engine *e = env->eng;
    size_t n = env->cfg.n_lanes;
    std::vector<uint8_t> nv(n);
    std::vector<uint32_t> a(n), b(n);
    for (size_t i = 0; i < n; ++i) {
      nv[i] = nv_pos[i] ? 1 : 0;
      RL_REQUIRE(steps_remaining[i] < (1ull << 32) && reset_count[i] < (1ull << 32), "state value out of range");
      a[i] = (uint32_t)steps_remaining[i];
      b[i] = (uint32_t)reset_count[i];
    }
    h2d(e, env->st.x, state4 + 0 * n, n * sizeof(double));
    h2d(e, env->st.xdot, state4 + 1 * n, n * sizeof(double));
    h2d(e, env->st.th, state4 + 2 * n, n * sizeof(double));
    h2d(e, env->st.thdot, state4 + 3 * n, n * sizeof(double));
    h2d(e, env->st.nv_pos, nv.data(), n);
    h2d(e, env->st.steps_remaining, a.data(), n * sizeof(uint32_t));
    h2d(e, env->st.reset_count, b.data(), n * sizeof(uint32_t));
  });
}

// ---------------------------------------------------------------- mlp
int32_t rl_mlp_create(rl_engine *e, uint32_t in_dim, uint32_t hidden, uint32_t out_dim, rl_mlp **out) {
  return guarded(e, [&] {
    RL_REQUIRE(e && out, "NULL argument");
    *out = nullptr;
    if (!(in_dim == 4 || in_dim == 5) || !(out_dim == 1 || out_dim == 2) || hidden == 0 || hidden > 128)
      throw RlError(RL_ERR_BUILD_AGENT, "supported MLP shapes: in_dim in {4,5}, 1 <= hidden <= 128, out_dim in {1,2}");
    RL_HIP_CHECK(hipSetDevice(e->device));
    std::unique_ptr<rl_mlp> m(new rl_mlp());
    m->eng = e;
    m->in_dim = in_dim;
    m->hidden = hidden;
    m->out_dim = out_dim;
    m->P = (uint64_t)hidden * in_dim + hidden + (uint64_t)out_dim * hidden + out_dim;
    m->d_params = dalloc<float>(m->P);
    RL_HIP_CHECK(hipMemsetAsync(m->d_params, 0, m->P * sizeof(float), e->stream));
    sync(e);
    e->live_handles += 1;
    *out = m.release();
  });
}

int32_t rl_gru_mlp_create(rl_engine *e, uint32_t in_dim, uint32_t gru_hidden, uint32_t mlp_hidden, uint32_t out_dim,
                          rl_mlp **out) {
  return guarded(e, [&] {
    RL_REQUIRE(e && out, "NULL argument");
    *out = nullptr;
    if (in_dim != 5 || gru_hidden != 128 || mlp_hidden != 128 || !(out_dim == 1 || out_dim == 2))
      throw RlError(RL_ERR_BUILD_AGENT, "supported GRU-MLP shape: in_dim 5, gru_hidden 128, mlp_hidden 128, out_dim in {1,2}");
    RL_HIP_CHECK(hipSetDevice(e->device));
    std::unique_ptr<rl_mlp> m(new rl_mlp());
    m->eng = e;
    m->kind = RL_MODULE_GRU_MLP;
    m->in_dim = in_dim;
    m->gru_hidden = gru_hidden;
    m->hidden = mlp_hidden;
    m->out_dim = out_dim;
    uint64_t H = gru_hidden, D = in_dim, H2 = mlp_hidden, A = out_dim;
    m->P = 3 * H * D + 3 * H * H + 6 * H + H2 * H + H2 + A * H2 + A;
    m->d_params = dalloc<float>(m->P);
    RL_HIP_CHECK(hipMemsetAsync(m->d_params, 0, m->P * sizeof(float), e->stream));
    sync(e);
    e->live_handles += 1;
    *out = m.release();
  });
}

// RnnWeights::new with RnnBaseConfig::default (seq/rnn/mod.rs:36-45,223-257) + the MLP's Linear::new layers.
// Engine-defined stream: ChaCha8(seed), stream 0, one Standard f32 per uniform element; normals for the orthogonal
// matrix by Box-Muller on consecutive draw pairs, QR by modified Gram-Schmidt applied twice in f64 (positive
// diagonal of R, so the sign fold of init_orthogonal, initializers.rs:345-348, is the identity).
static void gru_mlp_init_host(const rl_mlp *m, uint64_t seed, std::vector<float> &h) {
  const uint64_t H = m->gru_hidden, D = m->in_dim, H2 = m->hidden, A = m->out_dim, R = 3 * H;
  h.assign(m->P, 0.0f);
  uint32_t key[8];
  rl_seed_from_u64(seed, key);
  uint32_t words[16];
  uint64_t widx = 0;
  auto next_f32 = [&]() {
    if ((widx & 15) == 0) rl_chacha_block(key, widx >> 4, 0, 4, words);
    float u = rl_u32_to_unit_f32(words[widx & 15]);
    widx += 1;
    return u;
  };
  size_t k = 0;
  float lim = (float)std::sqrt(3.0 * (2.0 / ((double)D + (double)R)));
  for (uint64_t i = 0; i < R * D; ++i) {
    float u = next_f32();
    float t = 2.0f * u;
    t = t - 1.0f;
    h[k++] = t * lim;
  }
  std::vector<double> rowmajor(R * H), a(R * H);
  const double two_pi = 6.283185307179586;
  for (uint64_t i = 0; i < R * H; i += 2) {
    double u1 = (double)next_f32(), u2 = (double)next_f32();
    double rho = std::sqrt(-2.0 * std::log(1.0 - u1)), sn, cs;
    rl_sincos(two_pi * u2, &sn, &cs);
    rowmajor[i] = rho * cs;
    if (i + 1 < R * H) rowmajor[i + 1] = rho * sn;
  }
  for (uint64_t row = 0; row < R; ++row)
    for (uint64_t c = 0; c < H; ++c) a[c * R + row] = rowmajor[row * H + c];
  for (uint64_t c = 0; c < H; ++c) {
    double *v = a.data() + c * R;
    for (int pass = 0; pass < 2; ++pass)
      for (uint64_t q = 0; q < c; ++q) {
        const double *w = a.data() + q * R;
        double dot = 0.0;
        for (uint64_t row = 0; row < R; ++row) dot += w[row] * v[row];
        for (uint64_t row = 0; row < R; ++row) v[row] -= dot * w[row];
      }
    double nrm = 0.0;
    for (uint64_t row = 0; row < R; ++row) nrm += v[row] * v[row];
    nrm = std::sqrt(nrm);
    for (uint64_t row = 0; row < R; ++row) v[row] /= nrm;
  }
  for (uint64_t row = 0; row < R; ++row)
    for (uint64_t c = 0; c < H; ++c) h[k++] = (float)a[c * R + row];
  k += 2 * R;  // biases stay zero
  uint64_t dims[2][2] = {{H, H2}, {H2, A}};
  for (int l = 0; l < 2; ++l) {
    uint64_t in = dims[l][0], out = dims[l][1];
    float lm = (float)std::sqrt(3.0 * (2.0 / ((double)(in + 1) + (double)out)));
    for (uint64_t i = 0; i < in * out + out; ++i) {
      float u = next_f32();
      float t = 2.0f * u;
      t = t - 1.0f;
      h[k++] = t * lm;
    }
  }
}

int32_t rl_mlp_destroy(rl_mlp *m) {
  if (!m) return RL_OK;
  (void)hipSetDevice(m->eng->device);
  (void)hipStreamSynchronize(m->eng->stream);
  dfree(m->d_params);
  rl_engine *eng = m->eng;
  delete m;
  engine_release_child(eng);
  return RL_OK;
}

int32_t rl_mlp_num_params(const rl_mlp *m, uint64_t *n) {
  return guarded(m ? m->eng : nullptr, [&] {
    RL_REQUIRE(m && n, "NULL argument");
    *n = m->P;
  });
}

int32_t rl_mlp_init(rl_mlp *m, uint64_t seed) {
  return guarded(m ? m->eng : nullptr, [&] {
    RL_REQUIRE(m, "mlp is NULL");
    if (m->kind == RL_MODULE_GRU_MLP) {
      std::vector<float> hp;
      gru_mlp_init_host(m, seed, hp);
      h2d(m->eng, m->d_params, hp.data(), m->P * sizeof(float));
      return;
    }
    // Linear::new (reference src/torch/modules/ff/linear.rs:54-68; initializers.rs:31-38,78-108,159-163):
    // Uniform(+-sqrt(3 * 2 / (fan_in + fan_out))) with fan_in = in_dim + 1 for kernel and bias.
    // Stream (engine-defined, libtorch's RNG is unseeded in the reference): ChaCha8(seed), stream 0,
    // one Standard f32 per element in flat parameter order; value = (2u - 1) * lim in f32.
    std::vector<float> h(m->P);
    uint32_t key[8];
    rl_seed_from_u64(seed, key);
    uint32_t words[16];
    uint64_t widx = 0;
    auto next_f32 = [&]() {
      if ((widx & 15) == 0) rl_chacha_block(key, widx >> 4, 0, 4, words);
      float u = rl_u32_to_unit_f32(words[widx & 15]);
      widx += 1;
      return u;
    };
    uint32_t dims[2][2] = {{m->in_dim, m->hidden}, {m->hidden, m->out_dim}};
    size_t k = 0;
    for (int l = 0; l < 2; ++l) {
      uint32_t in = dims[l][0], out = dims[l][1];
      float lim = (float)std::sqrt(3.0 * (2.0 / ((double)(in + 1) + (double)out)));
      size_t cnt = (size_t)in * out + out;
      for (size_t i = 0; i < cnt; ++i) {
        float u = next_f32();
        float t = 2.0f * u;
        t = t - 1.0f;
        h[k++] = t * lim;
      }
    }
    h2d(m->eng, m->d_params, h.data(), m->P * sizeof(float));
  });
}

int32_t rl_params_get(rl_mlp *m, float *host, uint64_t n) {
  return guarded(m ? m->eng : nullptr, [&] {
    RL_REQUIRE(m && host, "NULL argument");
    RL_REQUIRE(n == m->P, "parameter count mismatch");
    d2h(m->eng, host, m->d_params, n * sizeof(float));
  });
}

int32_t rl_params_set(rl_mlp *m, const float *host, uint64_t n) {
  return guarded(m ? m->eng : nullptr, [&] {
    RL_REQUIRE(m && host, "NULL argument");
    RL_REQUIRE(n == m->P, "parameter count mismatch");
    h2d(m->eng, m->d_params, host, n * sizeof(float));
  });
}

int32_t rl_mlp_forward(rl_mlp *m, const float *rows, uint64_t n_rows, float *out) {
  return guarded(m ? m->eng : nullptr, [&] {
    if (m && m->kind != RL_MODULE_MLP)
      throw RlError(RL_ERR_UNSUPPORTED, "row-wise forward is for feed-forward modules; use rl_seq_forward");
    RL_REQUIRE(m && rows && out, "NULL argument");
    if (n_rows == 0) return;
    rl_engine *e = m->eng;
    std::vector<float> soa((size_t)n_rows * m->in_dim), res((size_t)n_rows * m->out_dim);
    for (uint64_t r = 0; r < n_rows; ++r)
      for (uint32_t d = 0; d < m->in_dim; ++d) soa[(size_t)d * n_rows + r] = rows[r * m->in_dim + d];
    float *d_in = dalloc<float>(soa.size()), *d_out = dalloc<float>(res.size());
    try {
      h2d(e, d_in, soa.data(), soa.size() * sizeof(float));
      launch_mlp_forward_host_rows(m, d_in, n_rows, d_out);
      d2h(e, res.data(), d_out, res.size() * sizeof(float));
    } catch (...) {
      dfree(d_in);
      dfree(d_out);
      throw;
    }
    dfree(d_in);
    dfree(d_out);
    for (uint64_t r = 0; r < n_rows; ++r)
      for (uint32_t a = 0; a < m->out_dim; ++a) out[r * m->out_dim + a] = res[(size_t)a * n_rows + r];
  });
}

// ---------------------------------------------------------------- trajectory
static void traj_field(const rl_traj *t, int32_t field, void **ptr, uint64_t *bytes) {
  uint64_t n = t->d.n, T = t->d.T, D = t->d.D;
  switch (field) {
    case RL_TRAJ_OBS: *ptr = t->d.obs; *bytes = D * (T + 1) * n * 4; break;
    case RL_TRAJ_ACTION: *ptr = t->d.action; *bytes = T * n; break;
    case RL_TRAJ_REWARD: *ptr = t->d.reward; *bytes = T * n * 4; break;
    case RL_TRAJ_FLAG: *ptr = t->d.flag; *bytes = T * n; break;
    case RL_TRAJ_TERM_OBS: *ptr = t->d.term_obs; *bytes = D * T * n * 4; break;
    case RL_TRAJ_VALUES: *ptr = t->d.values; *bytes = (T + 1) * n * 4; break;
    case RL_TRAJ_ADVANTAGES: *ptr = t->d.adv; *bytes = T * n * 4; break;
    case RL_TRAJ_RETURNS: *ptr = t->d.rtg; *bytes = T * n * 4; break;
    default: throw RlError(RL_ERR_INVALID_ARGUMENT, "unknown trajectory field");
  }
}

// launch geometry of the update kernels for B samples
static void traj_plan(rl_traj *t, uint64_t B) {
  rl_engine *e = t->eng;
  t->B = B;
  // backward: <= 1024 workgroups of 128 threads, chunk a multiple of 8 samples
  uint64_t chunk = (B + 1023) / 1024;
  chunk = ((chunk + 7) / 8) * 8;
  if (chunk < 64) chunk = 64;
  t->bwd_chunk = (uint32_t)chunk;
  t->nbA = (uint32_t)((B + chunk - 1) / chunk);
  uint64_t nbB = (B + 255) / 256;
  if (nbB > 2048) nbB = 2048;
  t->nbB = (uint32_t)nbB;
  // v2 kernels: persistent grid, one workgroup per CU (fewer, fatter workgroups = fewer slab rows for the reduction
  // that follows every launch), one 32-sample tile per wave and iteration; policy kernels run eight waves per workgroup
  uint64_t n_tiles = (B + 31) / 32;
  uint64_t nbV2 = (n_tiles + 7) / 8;
  uint64_t max_v2 = (uint64_t)e->prop.multiProcessorCount;
  // the critic step runs one workgroup of twelve waves per CU (168 VGPRs: three waves per SIMD; 149 KB LDS)
  uint64_t nbC = (n_tiles + 11) / 12, max_c = (uint64_t)e->prop.multiProcessorCount;
  if (nbC > max_c) nbC = max_c;
  t->nbC = (uint32_t)nbC;
  if (nbV2 > max_v2) nbV2 = max_v2;
  t->nbV2 = (uint32_t)nbV2;
  // pair kernels: 2-wave workgroups, 8 per CU (4 waves per SIMD), a contiguous run of tiles per workgroup
  uint64_t max_pair = 8ull * (uint64_t)e->prop.multiProcessorCount;
  uint64_t tpb = (n_tiles + max_pair - 1) / max_pair;
  if (tpb == 0) tpb = 1;
  t->pair_tiles_per_block = (uint32_t)tpb;
  t->nbPair = (uint32_t)((n_tiles + tpb - 1) / tpb);
}

// `resizable`: the sample count changes between launches (DQN minibatches): slabs are sized for the largest grid
// any B <= n_lanes * horizon can plan
static rl_traj *traj_alloc(rl_engine *e, uint64_t n_lanes, uint64_t horizon, uint32_t obs_dim, bool resizable) {
  RL_REQUIRE(n_lanes > 0 && n_lanes < (1ull << 31), "bad n_lanes");
  RL_REQUIRE(horizon > 0 && horizon < (1ull << 20), "bad horizon");
  RL_REQUIRE(obs_dim == 4 || obs_dim == 5, "obs_dim must be 4 or 5");
  RL_REQUIRE(n_lanes * horizon < (1ull << 32), "T * n must fit 32 bits");
  RL_HIP_CHECK(hipSetDevice(e->device));
  std::unique_ptr<rl_traj> t(new rl_traj());
  t->eng = e;
  t->d.n = (uint32_t)n_lanes;
  t->d.T = (uint32_t)horizon;
  t->d.D = obs_dim;
  uint64_t n = n_lanes, T = horizon, D = obs_dim;
  t->d.obs = dalloc<float>(D * (T + 1) * n);
  t->d.action = dalloc<uint8_t>(T * n);
  t->d.reward = dalloc<float>(T * n);
  t->d.flag = dalloc<uint8_t>(T * n);
  t->d.term_obs = dalloc<float>(D * T * n);
  t->d.values = dalloc<float>((T + 1) * n);
  t->d.adv = dalloc<float>(T * n);
  t->d.rtg = dalloc<float>(T * n);
  t->lp0 = dalloc<float>(2 * n * T);
  t->dz = dalloc<float>(2 * n * T);
  t->Pmax = 128 * 5 + 128 + 2 * 128 + 2;
  traj_plan(t.get(), n * T);
  uint32_t rows = t->nbA;
  if (t->nbV2 > rows) rows = t->nbV2;
  if (t->nbC > rows) rows = t->nbC;
  if (t->nbPair > rows) rows = t->nbPair;
  uint32_t rowsB = t->nbB > rows ? t->nbB : rows;
  if (resizable) {
    uint32_t cap = 8u * (uint32_t)e->prop.multiProcessorCount;
    if (cap < 2048) cap = 2048;
    rows = rowsB = cap;
  }
  t->slabA = dalloc<double>((size_t)rows * t->Pmax);
  t->slabB = dalloc<double>((size_t)rowsB * 4);
  t->vec = dalloc<float>(t->Pmax + 4);
  t->cg_x = dalloc<float>(t->Pmax);
  t->cg_r = dalloc<float>(t->Pmax);
  t->cg_p = dalloc<float>(t->Pmax);
  t->prev_params = dalloc<float>(t->Pmax);
  t->descent = dalloc<float>(t->Pmax);
  t->max_losses = 4096;
  t->losses = dalloc<float>(t->max_losses);
  t->trpo = dalloc<TrpoStateDev>(1);
  RL_HIP_CHECK(hipMemsetAsync(t->d.term_obs, 0, D * T * n * 4, e->stream));
  RL_HIP_CHECK(hipMemsetAsync(t->d.values, 0, (T + 1) * n * 4, e->stream));
  RL_HIP_CHECK(hipMemsetAsync(t->d.adv, 0, T * n * 4, e->stream));
  RL_HIP_CHECK(hipMemsetAsync(t->d.rtg, 0, T * n * 4, e->stream));
  RL_HIP_CHECK(hipMemsetAsync(t->trpo, 0, sizeof(TrpoStateDev), e->stream));
  RL_HIP_CHECK(hipStreamSynchronize(e->stream));
  e->live_handles += 1;
  return t.release();
}

int32_t rl_traj_create(rl_engine *e, uint64_t n_lanes, uint64_t horizon, uint32_t obs_dim, rl_traj **out) {
  return guarded(e, [&] {
    RL_REQUIRE(e && out, "NULL argument");
    *out = nullptr;
    *out = traj_alloc(e, n_lanes, horizon, obs_dim, false);
  });
}

static void seq_free(rl_traj *t);

int32_t rl_traj_destroy(rl_traj *t) {
  if (!t) return RL_OK;
  (void)hipSetDevice(t->eng->device);
  (void)hipStreamSynchronize(t->eng->stream);
  void *ptrs[] = {t->d.obs, t->d.action, t->d.reward, t->d.flag, t->d.term_obs, t->d.values, t->d.adv, t->d.rtg,
                  t->lp0, t->dz, t->slabA, t->slabB, t->vec, t->cg_x, t->cg_r, t->cg_p, t->prev_params, t->descent,
                  t->losses, t->trpo};
  for (void *p : ptrs) dfree(p);
  seq_free(t);
  rl_engine *eng = t->eng;
  delete t;
  engine_release_child(eng);
  return RL_OK;
}

int32_t rl_traj_field_bytes(const rl_traj *t, int32_t field, uint64_t *bytes) {
  return guarded(t ? t->eng : nullptr, [&] {
    RL_REQUIRE(t && bytes, "NULL argument");
    void *p;
    traj_field(t, field, &p, bytes);
  });
}

int32_t rl_traj_read(rl_traj *t, int32_t field, void *host, uint64_t bytes) {
  return guarded(t ? t->eng : nullptr, [&] {
    RL_REQUIRE(t && host, "NULL argument");
    void *p;
    uint64_t need;
    traj_field(t, field, &p, &need);
    RL_REQUIRE(bytes == need, "byte count mismatch for trajectory field");
    d2h(t->eng, host, p, bytes);
  });
}

int32_t rl_traj_write(rl_traj *t, int32_t field, const void *host, uint64_t bytes) {
  return guarded(t ? t->eng : nullptr, [&] {
    RL_REQUIRE(t && host, "NULL argument");
    void *p;
    uint64_t need;
    traj_field(t, field, &p, &need);
    RL_REQUIRE(bytes == need, "byte count mismatch for trajectory field");
    h2d(t->eng, p, host, bytes);
  });
}

// ---------------------------------------------------------------- recurrent workspace
static void seq_ensure(rl_traj *t, const rl_mlp *mod, bool training) {
  RL_REQUIRE(mod->kind == RL_MODULE_GRU_MLP, "not a recurrent module");
  RL_REQUIRE(t->d.n % 32 == 0, "the recurrent kernels work on tiles of 32 lanes: n_lanes must be a multiple of 32");
  RL_REQUIRE(t->d.D == 5 && mod->in_dim == 5, "recurrent path: built for 5 observation features");
  SeqDev &q = t->seq;
  uint64_t n = t->d.n, T = t->d.T;
  if (q.out == nullptr) {
    q.tiles = (uint32_t)(n / 32);
    q.out = dalloc<float>(2 * T * n);
    q.succ = dalloc<float>(2 * T * n);
  }
  if (training && q.act == nullptr) {
    uint64_t blocks = T * q.tiles;
    q.act = dalloc<float>(blocks * 7 * 128 * 32);
    q.dpre = dalloc<float>(blocks * 5 * 128 * 32);
    // weight-gradient partials: contiguous runs of (t, tile) blocks per workgroup, <= 1024 workgroups and at most
    // ~2048 samples accumulated in f32 before the f64 reduction
    uint64_t bpc = (blocks + 1023) / 1024;
    if (bpc < 1) bpc = 1;
    if (bpc > 64) bpc = 64;
    q.blocks_per_chunk = (uint32_t)bpc;
    q.chunks = (uint32_t)((blocks + bpc - 1) / bpc);
  }
  if (training && q.P < mod->P) {
    dfree(q.wg_slab);
    q.wg_slab = nullptr;
    q.wg_slab = dalloc<float>((size_t)q.chunks * mod->P);
    q.P = mod->P;
    // the P-sized vectors of the update workspace grow with the module
    if (t->Pmax < mod->P) {
      for (float **p : {&t->vec, &t->cg_x, &t->cg_r, &t->cg_p, &t->prev_params, &t->descent}) {
        dfree(*p);
        *p = nullptr;
      }
      t->Pmax = (uint32_t)mod->P;
      t->vec = dalloc<float>(t->Pmax + 4);
      t->cg_x = dalloc<float>(t->Pmax);
      t->cg_r = dalloc<float>(t->Pmax);
      t->cg_p = dalloc<float>(t->Pmax);
      t->prev_params = dalloc<float>(t->Pmax);
      t->descent = dalloc<float>(t->Pmax);
    }
  }
}

static void seq_free(rl_traj *t) {
  SeqDev &q = t->seq;
  for (float *p : {q.act, q.dpre, q.out, q.succ, q.wg_slab}) dfree(p);
  q = SeqDev{};
}

int32_t rl_seq_forward(rl_mlp *mod, rl_traj *traj, float *out, float *succ_out) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    RL_REQUIRE(mod && traj && out, "NULL argument");
    RL_REQUIRE(mod->eng == traj->eng, "handles belong to different engines");
    seq_ensure(traj, mod, false);
    launch_gru_seq_forward(traj, mod, traj->seq.out, succ_out ? traj->seq.succ : nullptr, nullptr);
    uint64_t bytes = (uint64_t)mod->out_dim * traj->d.T * traj->d.n * sizeof(float);
    d2h(traj->eng, out, traj->seq.out, bytes);
    if (succ_out) d2h(traj->eng, succ_out, traj->seq.succ, bytes);
  });
}

// ---------------------------------------------------------------- rollout + GAE
int32_t rl_rollout(rl_env *env, const rl_mlp *policy, rl_traj *traj) {
  return guarded(env ? env->eng : nullptr, [&] {
    RL_REQUIRE(env && policy && traj, "NULL argument");
    RL_REQUIRE(env->eng == traj->eng && env->eng == policy->eng, "handles belong to different engines");
    RL_REQUIRE(traj->d.n == env->cfg.n_lanes && traj->d.D == env->D, "trajectory shape does not match the env");
    RL_REQUIRE(policy->in_dim == env->D && policy->out_dim == env->A, "policy shape does not match the env");
    if (policy->kind == RL_MODULE_GRU_MLP) {
      seq_ensure(traj, policy, false);
      launch_rollout_gru(env, policy, traj);
    } else if (env->kind != RL_ENV_CARTPOLE) {
      launch_rollout_chain_mlp(env, policy, traj);
    } else {
      launch_rollout(env, policy, traj);
    }
    env->t_global += traj->d.T;
  });
}

int32_t rl_gae(rl_traj *traj, const rl_mlp *critic, float gamma, float lambda) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    RL_REQUIRE(traj && critic, "NULL argument");
    RL_REQUIRE(critic->in_dim == traj->d.D && critic->out_dim == 1, "critic shape does not match the trajectory");
    if (critic->kind == RL_MODULE_GRU_MLP) {
      seq_ensure(traj, critic, false);
      launch_gru_seq_forward(traj, critic, traj->seq.out, traj->seq.succ, nullptr);
      launch_seq_gae(traj, gamma, lambda);
      return;
    }
    launch_values(traj, critic);
    launch_gae(traj, critic, gamma, lambda);
  });
}

// ---------------------------------------------------------------- recurrent gradient passes
// policy: teacher-forced forward (activation record) -> d loss / d logits -> [backward through time -> weight
// gradients -> reduce] -> traj->vec[0..P) and the per-sample sums in vec[P..P+4)
static void seq_policy_pass(rl_mlp *policy, rl_traj *traj, int mode, bool backward, float lo, float hi) {
  seq_ensure(traj, policy, true);
  uint32_t P = (uint32_t)policy->P;
  launch_gru_seq_forward(traj, policy, traj->seq.out, nullptr, backward ? traj->seq.act : nullptr);
  launch_seq_policy_dlogits(traj, mode, b_total(traj), lo, hi);
  if (backward) launch_gru_backward(traj, policy);
  launch_reduce(traj, P, false, true, 0, traj->nbB);
  if (backward) rl_allreduce_sum_f32(traj->eng, traj->vec, P + 4);
  else rl_allreduce_sum_f32(traj->eng, traj->vec + P, 4);
}

// (loss, KL) of the current parameters against log pi_0: forward without a record -> sums in vec[P..P+4)
static void seq_policy_eval(rl_mlp *policy, rl_traj *traj, const int32_t *d_skip) {
  seq_ensure(traj, policy, true);
  uint32_t P = (uint32_t)policy->P;
  launch_gru_seq_forward(traj, policy, traj->seq.out, nullptr, nullptr, d_skip);
  launch_seq_policy_dlogits(traj, PASS_EVAL, b_total(traj), 0.0f, 0.0f, d_skip);
  launch_reduce(traj, P, false, true, 0, traj->nbB);
  rl_allreduce_sum_f32(traj->eng, traj->vec + P, 4);
}

// Fisher-vector product with the tangent d_v at the parameters whose activation record is in place (the last
// seq_policy_pass with backward = true): vec[0..P) <- J^T (diag(p) - p p^T) J v / B
static void seq_policy_fvp(rl_mlp *policy, rl_traj *traj, const float *d_v, const int32_t *d_skip) {
  seq_ensure(traj, policy, true);
  launch_gru_tangent(traj, policy, d_v, b_total(traj), d_skip);
  launch_gru_backward(traj, policy, d_skip);
  rl_allreduce_sum_f32(traj->eng, traj->vec, (uint32_t)policy->P);
}

static void seq_critic_pass(rl_mlp *critic, rl_traj *traj) {
  seq_ensure(traj, critic, true);
  uint32_t P = (uint32_t)critic->P;
  launch_gru_seq_forward(traj, critic, traj->seq.out, nullptr, traj->seq.act);
  launch_seq_critic_dvalues(traj, b_total(traj));
  launch_gru_backward(traj, critic);
  launch_reduce(traj, P, false, true, 0, traj->nbB);
  rl_allreduce_sum_f32(traj->eng, traj->vec, P + 4);
}

// ---------------------------------------------------------------- TRPO
int32_t rl_trpo_config_default(rl_trpo_config *c) {
  return guarded(nullptr, [&] {
    RL_REQUIRE(c, "cfg is NULL");
    // ConjugateGradientOptimizerConfig::default (conjugate_gradient.rs:55-65), TrpoConfig::default (trpo.rs:29-41)
    c->iterations = 10;
    c->max_backtracks = 15;
    c->backtrack_ratio = 0.8;
    c->hpv_reg_coeff = 1e-5;
    c->max_policy_step_kl = 0.01;
    c->accept_violation = 0;
  });
}

static void check_policy(const rl_mlp *policy, const rl_traj *traj) {
  RL_REQUIRE(policy && traj, "NULL argument");
  RL_REQUIRE(policy->eng == traj->eng, "handles belong to different engines");
  RL_REQUIRE(policy->in_dim == traj->d.D && policy->out_dim == 2, "policy shape does not match the trajectory");
}

// gradient pass: PASS_INIT -> backward -> reduce(A+B) -> allreduce
static void run_policy_gradient(rl_mlp *policy, rl_traj *traj) {
  if (policy->kind == RL_MODULE_GRU_MLP) return seq_policy_pass(policy, traj, PASS_INIT, true, 0.0f, 0.0f);
  uint32_t P = (uint32_t)policy->P;
  if (traj->eng->kernel_variant != 1 && launch_policy_v2(traj, policy, PASS_INIT, nullptr, b_total(traj), nullptr)) {
    launch_reduce(traj, P, true, true, traj->nbV2, traj->nbV2);
  } else {
    launch_policy_pass(traj, policy, PASS_INIT, nullptr, b_total(traj), nullptr);
    launch_mlp_backward(traj, policy, nullptr);
    launch_reduce(traj, P, true, true, traj->nbA, traj->nbB);
  }
  rl_allreduce_sum_f32(traj->eng, traj->vec, P + 4);
}

// (loss, KL) of the current parameters against lp0: PASS_EVAL -> reduce(B) -> allreduce
static void run_policy_eval(rl_mlp *policy, rl_traj *traj, const int32_t *d_skip) {
  if (policy->kind == RL_MODULE_GRU_MLP) return seq_policy_eval(policy, traj, d_skip);
  uint32_t P = (uint32_t)policy->P;
  if (traj->eng->kernel_variant != 1 && launch_policy_v2(traj, policy, PASS_EVAL, nullptr, b_total(traj), d_skip)) {
    launch_reduce(traj, P, false, true, traj->nbV2, traj->nbV2);
  } else {
    launch_policy_pass(traj, policy, PASS_EVAL, nullptr, b_total(traj), d_skip);
    launch_reduce(traj, P, false, true, traj->nbA, traj->nbB);
  }
  rl_allreduce_sum_f32(traj->eng, traj->vec + P, 4);
}

// Fisher/Hessian-vector product pass with tangent d_v: PASS_JVP -> backward -> reduce(A) -> allreduce
static void run_policy_fvp(rl_mlp *policy, rl_traj *traj, const float *d_v, const int32_t *d_skip) {
  if (policy->kind == RL_MODULE_GRU_MLP) return seq_policy_fvp(policy, traj, d_v, d_skip);
  uint32_t P = (uint32_t)policy->P;
  if (traj->eng->kernel_variant != 1 && launch_policy_v2(traj, policy, PASS_JVP, d_v, b_total(traj), d_skip)) {
    launch_reduce(traj, P, true, false, traj->nbV2, traj->nbV2);
  } else {
    launch_policy_pass(traj, policy, PASS_JVP, d_v, b_total(traj), d_skip);
    launch_mlp_backward(traj, policy, d_skip);
    launch_reduce(traj, P, true, false, traj->nbA, traj->nbB);
  }
  rl_allreduce_sum_f32(traj->eng, traj->vec, P);
}

int32_t rl_trpo_update(rl_mlp *policy, rl_traj *traj, const rl_trpo_config *cfg, rl_trpo_stats *stats) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    check_policy(policy, traj);
    RL_REQUIRE(cfg && stats, "NULL argument");
    rl_engine *e = traj->eng;
    uint32_t P = (uint32_t)policy->P;
    uint64_t Bt = b_total(traj);
    float reg = (float)cfg->hpv_reg_coeff;
    // loss gradient at theta0 and CG prologue
    run_policy_gradient(policy, traj);
    launch_trpo_begin(traj, policy, Bt);
    // x = A^-1 g by `iterations` CG steps (early exit handled on the device)
    for (uint64_t it = 0; it < cfg->iterations; ++it) {
      run_policy_fvp(policy, traj, traj->cg_p, &traj->trpo->cg_done);
      launch_cg_step(traj, P, reg, 1e-10f);
    }
    launch_cg_finish(traj, P);
    // step size from x^T A x
    run_policy_fvp(policy, traj, traj->cg_x, nullptr);
    launch_step_size(traj, policy, reg, cfg->max_policy_step_kl);
    // backtracking line search
    double ratio = 1.0;
    for (uint64_t i = 0; i < cfg->max_backtracks; ++i) {
      if (i > 0) ratio *= cfg->backtrack_ratio;  // backtrack_ratio.powi(i)
      launch_ls_set_params(traj, policy, ratio);
      run_policy_eval(policy, traj, &traj->trpo->ls_accepted);
      launch_ls_check(traj, P, Bt, (int)i, ratio, cfg->max_policy_step_kl);
    }
    launch_ls_finalize(traj, policy, cfg->max_policy_step_kl, cfg->accept_violation);
    TrpoStateDev h;
    d2h(e, &h, traj->trpo, sizeof(h));
    stats->entropy = (double)h.entropy;
    stats->step_size = h.step_size;
    stats->loss_initial = (double)h.loss0;
    stats->loss_final = (double)h.ls_loss;
    stats->constraint_val_final = (double)h.ls_kl;
    stats->step_scale = h.ls_accepted ? h.ls_ratio : 0.0;
    stats->num_backtracks = h.ls_accepted ? (int64_t)h.ls_index : -1;
    stats->status = h.status;
    stats->cg_iterations = h.cg_iters;
    if (h.status == RL_OPT_NAN_LOSS || h.status == RL_OPT_NAN_CONSTRAINT)
      throw RlError(RL_ERR_OPT_NAN, h.status == RL_OPT_NAN_LOSS ? "NaN loss in policy optimization"
                                                                : "NaN constraint in policy optimization");
  });
}

int32_t rl_policy_gradient(rl_mlp *policy, rl_traj *traj, float *grad_out, float *loss_out, float *entropy_out) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    check_policy(policy, traj);
    RL_REQUIRE(grad_out, "grad_out is NULL");
    uint32_t P = (uint32_t)policy->P;
    run_policy_gradient(policy, traj);
    std::vector<float> h(P + 4);
    d2h(traj->eng, h.data(), traj->vec, (P + 4) * sizeof(float));
    std::memcpy(grad_out, h.data(), P * sizeof(float));
    double inv_B = 1.0 / (double)b_total(traj);
    if (loss_out) *loss_out = (float)(-((double)h[P] * inv_B));
    if (entropy_out) *entropy_out = (float)((double)h[P + 1] * inv_B);
  });
}

int32_t rl_policy_fvp(rl_mlp *policy, rl_traj *traj, const float *v, float reg, float *out) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    check_policy(policy, traj);
    RL_REQUIRE(v && out, "NULL argument");
    uint32_t P = (uint32_t)policy->P;
    run_policy_gradient(policy, traj);  // the product is taken at the current parameters: refresh log pi_0
    h2d(traj->eng, traj->cg_x, v, P * sizeof(float));  // (the recurrent pass above has grown the workspace)
    run_policy_fvp(policy, traj, traj->cg_x, nullptr);
    std::vector<float> h(P);
    d2h(traj->eng, h.data(), traj->vec, P * sizeof(float));
    for (uint32_t i = 0; i < P; ++i) out[i] = h[i] + reg * v[i];
  });
}

int32_t rl_policy_loss_kl(rl_mlp *policy, rl_traj *traj, const float *params0, float *loss_out, float *kl_out) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    check_policy(policy, traj);
    RL_REQUIRE(params0 && loss_out && kl_out, "NULL argument");
    rl_engine *e = traj->eng;
    uint32_t P = (uint32_t)policy->P;
    uint64_t Bt = b_total(traj);
    // lp0 under params0, then evaluate the current parameters against it
    std::vector<float> cur(P);
    d2h(e, cur.data(), policy->d_params, P * sizeof(float));
    h2d(e, policy->d_params, params0, P * sizeof(float));
    run_policy_gradient(policy, traj);  // fills lp0 under params0
    h2d(e, policy->d_params, cur.data(), P * sizeof(float));
    run_policy_eval(policy, traj, nullptr);
    float h[4];
    d2h(e, h, traj->vec + P, sizeof(h));
    double inv_B = 1.0 / (double)Bt;
    *loss_out = (float)(-((double)h[0] * inv_B));
    *kl_out = (float)((double)h[1] * inv_B);
  });
}

// ---------------------------------------------------------------- critic
int32_t rl_adam_config_default(rl_adam_config *c) {
  return guarded(nullptr, [&] {
    RL_REQUIRE(c, "cfg is NULL");
    c->learning_rate = 1e-3;  // AdamConfig::default (coptimizer.rs:147-156)
    c->beta1 = 0.9;
    c->beta2 = 0.999;
    c->weight_decay = 0.0;
    c->eps = 1e-8;  // libtorch AdamOptions default
  });
}

int32_t rl_adam_create(rl_mlp *module, const rl_adam_config *cfg, rl_adam **out) {
  return guarded(module ? module->eng : nullptr, [&] {
    RL_REQUIRE(module && cfg && out, "NULL argument");
    *out = nullptr;
    rl_engine *e = module->eng;
    RL_HIP_CHECK(hipSetDevice(e->device));
    std::unique_ptr<rl_adam> o(new rl_adam());
    o->eng = e;
    o->mod = module;
    o->cfg = *cfg;
    o->d_m = dalloc<float>(module->P);
    o->d_v = dalloc<float>(module->P);
    o->d_step = dalloc<uint64_t>(1);
    RL_HIP_CHECK(hipMemsetAsync(o->d_m, 0, module->P * sizeof(float), e->stream));
    RL_HIP_CHECK(hipMemsetAsync(o->d_v, 0, module->P * sizeof(float), e->stream));
    RL_HIP_CHECK(hipMemsetAsync(o->d_step, 0, sizeof(uint64_t), e->stream));
    sync(e);
    e->live_handles += 1;
    *out = o.release();
  });
}

int32_t rl_adam_destroy(rl_adam *o) {
  if (!o) return RL_OK;
  (void)hipSetDevice(o->eng->device);
  (void)hipStreamSynchronize(o->eng->stream);
  dfree(o->d_m);
  dfree(o->d_v);
  dfree(o->d_step);
  rl_engine *eng = o->eng;
  delete o;
  engine_release_child(eng);
  return RL_OK;
}

int32_t rl_adam_step_host(rl_adam *o, const float *grad) {
  return guarded(o ? o->mod->eng : nullptr, [&] {
    RL_REQUIRE(o && grad, "NULL argument");
    rl_engine *e = o->mod->eng;
    float *d_g = dalloc<float>(o->mod->P);
    try {
      h2d(e, d_g, grad, o->mod->P * sizeof(float));
      launch_adam_step_vec(o, d_g);
      sync(e);
    } catch (...) {
      dfree(d_g);
      throw;
    }
    dfree(d_g);
  });
}

static void check_critic(const rl_mlp *critic, const rl_traj *traj) {
  RL_REQUIRE(critic && traj, "NULL argument");
  RL_REQUIRE(critic->eng == traj->eng, "handles belong to different engines");
  RL_REQUIRE(critic->in_dim == traj->d.D && critic->out_dim == 1, "critic shape does not match the trajectory");
}

// per-workgroup partial sums of the critic's MSE gradient and loss -> slabA / slabB (feed-forward modules)
static void critic_slabs(rl_mlp *critic, rl_traj *traj, uint32_t *rowsA, uint32_t *rowsB) {
  if (traj->eng->kernel_variant != 1 && launch_critic_step_v2(traj, critic, b_total(traj))) {
    *rowsA = *rowsB = traj->eng->kernel_variant == 2 ? traj->nbPair : traj->nbC;
  } else {
    launch_critic_fwd(traj, critic, b_total(traj));
    launch_mlp_backward(traj, critic, nullptr);
    *rowsA = traj->nbA;
    *rowsB = traj->nbB;
  }
}

static void run_critic_gradient(rl_mlp *critic, rl_traj *traj) {
  if (critic->kind == RL_MODULE_GRU_MLP) return seq_critic_pass(critic, traj);
  uint32_t P = (uint32_t)critic->P, rowsA, rowsB;
  critic_slabs(critic, traj, &rowsA, &rowsB);
  launch_reduce(traj, P, true, true, rowsA, rowsB);
  rl_allreduce_sum_f32(traj->eng, traj->vec, P + 4);
}

int32_t rl_critic_update(rl_mlp *critic, rl_adam *opt, rl_traj *traj, uint64_t opt_steps, rl_critic_stats *stats,
                         float *losses_out) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    check_critic(critic, traj);
    RL_REQUIRE(opt && opt->mod == critic, "optimizer does not belong to this module");
    RL_REQUIRE(opt_steps <= traj->max_losses, "too many optimisation steps per update");
    uint64_t Bt = b_total(traj);
    const bool fused = critic->kind == RL_MODULE_MLP && !traj->eng->has_collective();
    for (uint64_t k = 0; k < opt_steps; ++k) {
      if (fused) {  // no all-reduce between the reduction and the (elementwise) optimiser step: one launch
        uint32_t rowsA, rowsB;
        critic_slabs(critic, traj, &rowsA, &rowsB);
        launch_reduce_adam(traj, opt, rowsA, rowsB, (int)k, Bt);
      } else {
        run_critic_gradient(critic, traj);
        launch_adam_step(traj, opt, (int)k, Bt);
      }
    }
    if (stats || losses_out) {
      std::vector<float> h(opt_steps ? opt_steps : 1);
      if (opt_steps) d2h(traj->eng, h.data(), traj->losses, opt_steps * sizeof(float));
      if (losses_out && opt_steps) std::memcpy(losses_out, h.data(), opt_steps * sizeof(float));
      if (stats) {
        stats->steps = opt_steps;
        stats->loss_first = opt_steps ? (double)h[0] : 0.0;
        stats->loss_last = opt_steps ? (double)h[opt_steps - 1] : 0.0;
      }
    }
  });
}

int32_t rl_critic_gradient(rl_mlp *critic, rl_traj *traj, float *grad_out, float *loss_out) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    check_critic(critic, traj);
    RL_REQUIRE(grad_out, "grad_out is NULL");
    uint32_t P = (uint32_t)critic->P;
    run_critic_gradient(critic, traj);
    std::vector<float> h(P + 4);
    d2h(traj->eng, h.data(), traj->vec, (P + 4) * sizeof(float));
    std::memcpy(grad_out, h.data(), P * sizeof(float));
    if (loss_out) *loss_out = (float)((double)h[P] / (double)b_total(traj));
  });
}

// ---------------------------------------------------------------- PPO / REINFORCE / RewardToGo
int32_t rl_ppo_config_default(rl_ppo_config *c) {
  return guarded(nullptr, [&] {
    RL_REQUIRE(c, "cfg is NULL");
    c->opt_steps_per_update = 10;  // PpoConfig::default (ppo.rs:27-41)
    c->clip_distance = 0.2;
  });
}

// PASS_PPO gradient of the clipped surrogate against lp0 -> vec[0..P), sum of min(...) -> vec[P]
static void run_policy_ppo(rl_mlp *policy, rl_traj *traj, float lo, float hi) {
  if (policy->kind == RL_MODULE_GRU_MLP) return seq_policy_pass(policy, traj, PASS_PPO, true, lo, hi);
  uint32_t P = (uint32_t)policy->P;
  if (traj->eng->kernel_variant != 1 &&
      launch_policy_v2(traj, policy, PASS_PPO, nullptr, b_total(traj), nullptr, lo, hi)) {
    launch_reduce(traj, P, true, true, traj->nbV2, traj->nbV2);
  } else {
    launch_policy_pass(traj, policy, PASS_PPO, nullptr, b_total(traj), nullptr, lo, hi);
    launch_mlp_backward(traj, policy, nullptr);
    launch_reduce(traj, P, true, true, traj->nbA, traj->nbB);
  }
  rl_allreduce_sum_f32(traj->eng, traj->vec, P + 4);
}

int32_t rl_ppo_update(rl_mlp *policy, rl_adam *opt, rl_traj *traj, const rl_ppo_config *cfg,
                      rl_policy_opt_stats *stats, float *losses_out) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    check_policy(policy, traj);
    RL_REQUIRE(opt && opt->mod == policy, "optimizer does not belong to this module");
    RL_REQUIRE(cfg, "cfg is NULL");
    RL_REQUIRE(cfg->opt_steps_per_update <= traj->max_losses, "too many optimisation steps per update");
    rl_engine *e = traj->eng;
    uint32_t P = (uint32_t)policy->P;
    uint64_t Bt = b_total(traj), K = cfg->opt_steps_per_update;
    // initial_log_probs and the logged entropy (ppo.rs:107-118): the PASS_INIT pass stores log pi_0
    if (policy->kind == RL_MODULE_GRU_MLP) seq_policy_pass(policy, traj, PASS_INIT, false, 0.0f, 0.0f);
    else run_policy_gradient(policy, traj);
    float h0[4];
    d2h(e, h0, traj->vec + P, sizeof(h0));
    // clip(1 - d, 1 + d): f64 scalars applied to a Float tensor
    float lo = (float)(1.0 - cfg->clip_distance), hi = (float)(1.0 + cfg->clip_distance);
    for (uint64_t k = 0; k < K; ++k) {
      run_policy_ppo(policy, traj, lo, hi);
      launch_adam_step(traj, opt, (int)k, Bt);
    }
    std::vector<float> h(K ? K : 1, 0.0f);
    if (K) d2h(e, h.data(), traj->losses, K * sizeof(float));
    for (auto &v : h) v = -v;  // loss = -mean(min(...))
    if (losses_out && K) std::memcpy(losses_out, h.data(), K * sizeof(float));
    if (stats) {
      stats->entropy = (double)h0[1] / (double)Bt;
      stats->steps = K;
      stats->loss_first = K ? (double)h[0] : 0.0;
      stats->loss_last = K ? (double)h[K - 1] : 0.0;
    }
  });
}

int32_t rl_reinforce_update(rl_mlp *policy, rl_adam *opt, rl_traj *traj, rl_policy_opt_stats *stats) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    check_policy(policy, traj);
    RL_REQUIRE(opt && opt->mod == policy, "optimizer does not belong to this module");
    rl_engine *e = traj->eng;
    uint32_t P = (uint32_t)policy->P;
    uint64_t Bt = b_total(traj);
    // d(-mean(log pi(a) A))/d theta equals the surrogate gradient at ratio = 1 that PASS_INIT computes
    run_policy_gradient(policy, traj);
    float h0[4];
    d2h(e, h0, traj->vec + P, sizeof(h0));
    launch_adam_step(traj, opt, -1, Bt);
    if (stats) {
      stats->entropy = (double)h0[1] / (double)Bt;
      stats->steps = 1;
      stats->loss_first = stats->loss_last = -((double)h0[2] / (double)Bt);
    }
  });
}

int32_t rl_reward_to_go(rl_traj *traj, float gamma) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    RL_REQUIRE(traj, "NULL argument");
    launch_gae(traj, nullptr, gamma, 0.0f);
  });
}

// ---------------------------------------------------------------- actor serialisation (serde_cbor layout)
static void cbor_interval(cbor::Writer &w, double lo, double hi) {
  w.map(2);  // IntervalSpace { low, high } (spaces/interval.rs:14-18)
  w.key("low");
  w.f64(lo);
  w.key("high");
  w.f64(hi);
}

static void cbor_tensor(cbor::Writer &w, const float *data, std::initializer_list<int64_t> shape) {
  w.map(5);  // TensorDef (torch/serialize.rs:62-81)
  w.key("kind");
  w.text("Float");
  w.key("shape");
  w.array(shape.size());
  size_t count = 1;
  for (int64_t d : shape) {
    w.sint(d);
    count *= (size_t)d;
  }
  w.key("requires_grad");
  w.boolean(true);
  w.key("byte_order");
  w.text("LittleEndian");
  w.key("data");
  w.bytes(data, count * sizeof(float));
}

// Mlp { layers, activation, output_activation } over `p` = [W1, b1, W2, b2] (ff/mlp.rs:45-50, ff/linear.rs:43-50)
static void cbor_mlp(cbor::Writer &w, const float *p, int64_t in, int64_t hid, int64_t out) {
  w.map(3);
  w.key("layers");
  w.array(2);
  const int64_t dims[2][2] = {{in, hid}, {hid, out}};
  for (int l = 0; l < 2; ++l) {
    w.map(2);
    w.key("kernel");
    cbor_tensor(w, p, {dims[l][1], dims[l][0]});
    p += dims[l][0] * dims[l][1];
    w.key("bias");
    cbor_tensor(w, p, {dims[l][1]});
    p += dims[l][1];
  }
  w.key("activation");
  w.text("Relu");
  w.key("output_activation");
  w.text("Identity");
}

static void cbor_module(cbor::Writer &w, const rl_mlp *m, const std::vector<float> &p) {
  if (m->kind == RL_MODULE_MLP) {
    cbor_mlp(w, p.data(), m->in_dim, m->hidden, m->out_dim);
    return;
  }
  const int64_t H = m->gru_hidden, D = m->in_dim;
  w.map(3);  // Chain { first, second, activation } (modules/chain.rs:58-63)
  w.key("first");
  w.map(4);  // RnnBase { weights, hidden_size, dropout, type_ } (`device` is #[serde(skip)], seq/rnn/mod.rs:90-99)
  w.key("weights");
  w.map(2);  // RnnWeights { flat_weights, has_biases } (seq/rnn/mod.rs:186-191)
  w.key("flat_weights");
  w.array(4);
  const float *q = p.data();
  cbor_tensor(w, q, {3 * H, D});
  q += 3 * H * D;
  cbor_tensor(w, q, {3 * H, H});
  q += 3 * H * H;
  cbor_tensor(w, q, {3 * H});
  q += 3 * H;
  cbor_tensor(w, q, {3 * H});
  q += 3 * H;
  w.key("has_biases");
  w.boolean(true);
  w.key("hidden_size");
  w.uint((uint64_t)H);
  w.key("dropout");
  w.f64(0.0);
  w.key("type_");
  w.null();  // PhantomData
  w.key("second");
  cbor_mlp(w, q, H, m->hidden, m->out_dim);
  w.key("activation");
  w.text("Relu");
}

static void cbor_observation_space(cbor::Writer &w, const rl_env *env) {
  w.map(1);  // NonEmptyFeatures { inner } (spaces/nonempty_features.rs:20-25)
  w.key("inner");
  auto inner = [&]() {
    if (env->kind != RL_ENV_CARTPOLE) {
      w.map(1);  // IndexSpace { size } (spaces/index.rs:19-22)
      w.key("size");
      w.uint(env->dev.chain_size);
      return;
    }
    // CartPolePhysicalStateSpace (envs/cartpole.rs:73-82, 273-284); default intervals = [f64::MIN, f64::MAX]
    const double lo = -1.7976931348623157e308, hi = 1.7976931348623157e308;
    w.map(4);
    w.key("cart_position");
    cbor_interval(w, -env->cfg.cartpole.max_pos, env->cfg.cartpole.max_pos);
    w.key("cart_velocity");
    cbor_interval(w, lo, hi);
    w.key("pole_angle");
    cbor_interval(w, -env->cfg.cartpole.max_angle, env->cfg.cartpole.max_angle);
    w.key("pole_angular_velocity");
    cbor_interval(w, lo, hi);
  };
  if (env->cfg.limit_kind == RL_LIMIT_VISIBLE) {
    w.map(2);  // StepLimitObsSpace { inner, remaining } (wrappers/step_limit.rs:133-138)
    w.key("inner");
    inner();
    w.key("remaining");
    cbor_interval(w, 0.0, 1.0);
  } else {
    inner();  // the latent limit and the bare env keep the env's own observation space
  }
}

int32_t rl_actor_to_cbor(rl_env *env, rl_mlp *module, int32_t actor_kind, double exploration_rate, uint8_t *buf,
                         uint64_t cap, uint64_t *len_out) {
  return guarded(env ? env->eng : nullptr, [&] {
    RL_REQUIRE(env && module && len_out, "NULL argument");
    RL_REQUIRE(actor_kind == RL_ACTOR_POLICY || actor_kind == RL_ACTOR_DQN, "unknown actor kind");
    RL_REQUIRE(module->eng == env->eng, "handles belong to different engines");
    RL_REQUIRE(module->in_dim == env->D && module->out_dim == env->A, "module shape does not match the env");
    std::vector<float> p(module->P);
    d2h(module->eng, p.data(), module->d_params, module->P * sizeof(float));
    cbor::Writer w;
    w.map(actor_kind == RL_ACTOR_DQN ? 4 : 3);
    w.key("observation_space");
    cbor_observation_space(w, env);
    w.key("action_space");
    w.map(0);  // IndexedTypeSpace<T>: its only field is #[serde(skip)] (spaces/indexed_type.rs:57-64)
    w.key(actor_kind == RL_ACTOR_DQN ? "action_value_fn" : "policy_module");
    cbor_module(w, module, p);
    if (actor_kind == RL_ACTOR_DQN) {
      w.key("exploration_rate");
      w.f64(exploration_rate);
    }
    *len_out = w.out.size();
    if (buf != nullptr) {
      RL_REQUIRE(cap >= w.out.size(), "buffer too small for the CBOR document");
      std::memcpy(buf, w.out.data(), w.out.size());
    }
  });
}

static void cbor_read_tensor(const cbor::Value &t, std::initializer_list<int64_t> shape, float *dst) {
  RL_REQUIRE(t.at("kind").s == "Float", "CBOR tensor: kind must be Float");
  RL_REQUIRE(t.at("byte_order").s == "LittleEndian", "CBOR tensor: data has non-native byte order");
  const cbor::Value &sh = t.at("shape");
  RL_REQUIRE(sh.kind == cbor::Value::ARRAY && sh.items.size() == shape.size(), "CBOR tensor: unexpected rank");
  size_t count = 1, i = 0;
  for (int64_t d : shape) {
    RL_REQUIRE(sh.items[i++]->as_int() == d, "CBOR tensor: unexpected shape");
    count *= (size_t)d;
  }
  const cbor::Value &data = t.at("data");
  RL_REQUIRE(data.kind == cbor::Value::BYTES && data.s.size() == count * sizeof(float), "CBOR tensor: bad data length");
  std::memcpy(dst, data.s.data(), data.s.size());
}

static float *cbor_read_mlp(const cbor::Value &m, int64_t in, int64_t hid, int64_t out, float *dst) {
  RL_REQUIRE(m.at("activation").s == "Relu" && m.at("output_activation").s == "Identity",
             "CBOR module: only Relu hidden / Identity output activations are built");
  const cbor::Value &layers = m.at("layers");
  RL_REQUIRE(layers.kind == cbor::Value::ARRAY && layers.items.size() == 2, "CBOR module: expected one hidden layer");
  const int64_t dims[2][2] = {{in, hid}, {hid, out}};
  for (int l = 0; l < 2; ++l) {
    const cbor::Value &lin = *layers.items[l];
    cbor_read_tensor(lin.at("kernel"), {dims[l][1], dims[l][0]}, dst);
    dst += dims[l][0] * dims[l][1];
    RL_REQUIRE(lin.at("bias").kind == cbor::Value::MAP, "CBOR module: layers without bias are not built");
    cbor_read_tensor(lin.at("bias"), {dims[l][1]}, dst);
    dst += dims[l][1];
  }
  return dst;
}

int32_t rl_module_from_cbor(rl_mlp *module, const uint8_t *buf, uint64_t len) {
  return guarded(module ? module->eng : nullptr, [&] {
    RL_REQUIRE(module && buf, "NULL argument");
    cbor::ValuePtr doc = cbor::Reader(buf, (size_t)len).parse();
    const cbor::Value &mod = doc->has("policy_module") ? doc->at("policy_module") : doc->at("action_value_fn");
    std::vector<float> p(module->P);
    float *end;
    if (module->kind == RL_MODULE_MLP) {
      end = cbor_read_mlp(mod, module->in_dim, module->hidden, module->out_dim, p.data());
    } else {
      const int64_t H = module->gru_hidden, D = module->in_dim;
      RL_REQUIRE(mod.at("activation").s == "Relu", "CBOR module: Chain activation must be Relu");
      const cbor::Value &rnn = mod.at("first");
      RL_REQUIRE(rnn.at("hidden_size").as_int() == H, "CBOR module: GRU hidden size mismatch");
      RL_REQUIRE(rnn.at("dropout").as_float() == 0.0, "CBOR module: dropout is not built");
      const cbor::Value &wts = rnn.at("weights");
      RL_REQUIRE(wts.at("has_biases").kind == cbor::Value::BOOL && wts.at("has_biases").b, "CBOR module: GRU biases required");
      const cbor::Value &fw = wts.at("flat_weights");
      RL_REQUIRE(fw.kind == cbor::Value::ARRAY && fw.items.size() == 4, "CBOR module: expected a one-layer GRU");
      float *q = p.data();
      cbor_read_tensor(*fw.items[0], {3 * H, D}, q);
      q += 3 * H * D;
      cbor_read_tensor(*fw.items[1], {3 * H, H}, q);
      q += 3 * H * H;
      cbor_read_tensor(*fw.items[2], {3 * H}, q);
      q += 3 * H;
      cbor_read_tensor(*fw.items[3], {3 * H}, q);
      q += 3 * H;
      end = cbor_read_mlp(mod.at("second"), H, module->hidden, module->out_dim, q);
    }
    RL_REQUIRE((uint64_t)(end - p.data()) == module->P, "CBOR module: parameter count mismatch");
    h2d(module->eng, module->d_params, p.data(), module->P * sizeof(float));
  });
}

// ---------------------------------------------------------------- DQN (src/torch/agents/dqn.rs)
int32_t rl_dqn_config_default(rl_dqn_config *c) {
  return guarded(nullptr, [&] {
    RL_REQUIRE(c, "cfg is NULL");
    std::memset(c, 0, sizeof(*c));
    c->target = RL_DQN_TARGET_REWARD_TO_GO;               // StepValueTarget::default (critics/mod.rs:211-215)
    c->exploration_kind = RL_SCHEDULE_LINEAR_ANNEALED;    // schedules.rs:23-31
    c->exploration_start = 1.0;
    c->exploration_end = 0.1;
    c->exploration_period = 10000000;
    c->minibatch_steps = 100000;                          // dqn.rs:63-70
    c->opt_steps_per_update = 50;
    c->buffer_capacity = 0;
    c->episode_capacity = 0;
    c->update_kind = RL_COLLECT_FIRST_REST;
    c->update_first = 1000000;
    c->update_rest = 100000;
    c->discount_factor = 0.99f;
  });
}

static double dqn_exploration_rate(const rl_dqn *q, bool training) {
  if (!training) return 0.0;  // schedules.rs:38
  if (q->cfg.exploration_kind == RL_SCHEDULE_CONSTANT) return q->cfg.exploration_start;
  double frac = (double)q->global_steps / (double)q->cfg.exploration_period;
  if (!(frac < 1.0)) frac = 1.0;  // f64::min(1.0)
  return frac * (q->cfg.exploration_end - q->cfg.exploration_start) + q->cfg.exploration_start;
}

static ReplayDev replay_alloc(rl_engine *e, uint32_t N, uint32_t C, uint32_t E, uint32_t D) {
  ReplayDev r{};
  r.N = N;
  r.C = C;
  r.E = E;
  r.D = D;
  size_t cn = (size_t)C * N;
  r.obs = dalloc<float>(cn * D);
  r.next_obs = dalloc<float>(cn * D);
  r.action = dalloc<uint8_t>(cn);
  r.reward = dalloc<float>(cn);
  r.flag = dalloc<uint8_t>(cn);
  r.head = dalloc<uint32_t>(N);
  r.count = dalloc<uint32_t>(N);
  r.ep_head = dalloc<uint32_t>(N);
  r.ep_count = dalloc<uint32_t>(N);
  r.total = dalloc<uint32_t>(N);
  r.ep_end = dalloc<uint32_t>((size_t)E * N);
  r.actor_pos = dalloc<uint64_t>(N);
  r.error = dalloc<int32_t>(1);
  uint32_t *zero_u32[] = {r.head, r.count, r.ep_head, r.ep_count, r.total};
  for (uint32_t *p : zero_u32) RL_HIP_CHECK(hipMemsetAsync(p, 0, (size_t)N * 4, e->stream));
  RL_HIP_CHECK(hipMemsetAsync(r.actor_pos, 0, (size_t)N * 8, e->stream));
  RL_HIP_CHECK(hipMemsetAsync(r.error, 0, 4, e->stream));
  RL_HIP_CHECK(hipMemsetAsync(r.next_obs, 0, cn * D * 4, e->stream));
  return r;
}

static void replay_free(ReplayDev &r) {
  void *ptrs[] = {r.obs,      r.next_obs, r.action, r.reward, r.flag,      r.head, r.count,
                  r.ep_head,  r.ep_count, r.total,  r.ep_end, r.actor_pos, r.error};
  for (void *p : ptrs) dfree(p);
  r = ReplayDev{};
}

int32_t rl_dqn_create(rl_env *env, rl_mlp *qnet, rl_adam *opt, const rl_dqn_config *cfg, rl_dqn **out) {
  return guarded(env ? env->eng : nullptr, [&] {
    RL_REQUIRE(env && qnet && opt && cfg && out, "NULL argument");
    *out = nullptr;
    rl_engine *e = env->eng;
    RL_REQUIRE(qnet->eng == e && opt->eng == e, "handles belong to different engines");
    RL_REQUIRE(opt->mod == qnet, "optimizer does not belong to the action-value module");
    RL_REQUIRE(qnet->in_dim == env->D && qnet->out_dim == env->A, "action-value module does not match the env");
    RL_REQUIRE(env->A == 2, "DQN kernels are built for 2-action envs");
    RL_REQUIRE(cfg->target == RL_DQN_TARGET_REWARD_TO_GO || cfg->target == RL_DQN_TARGET_ONE_STEP_TD, "bad target");
    RL_REQUIRE(cfg->minibatch_steps > 0 && cfg->minibatch_steps < (1ull << 30), "bad minibatch_steps");
    RL_REQUIRE(cfg->buffer_capacity > 0 && cfg->buffer_capacity < (1ull << 31), "bad buffer_capacity");
    uint64_t E = cfg->episode_capacity ? cfg->episode_capacity : cfg->buffer_capacity;
    RL_REQUIRE(E <= cfg->buffer_capacity, "episode_capacity exceeds buffer_capacity");
    RL_REQUIRE(cfg->opt_steps_per_update <= 4096, "too many optimisation steps per update");
    if (cfg->exploration_kind == RL_SCHEDULE_LINEAR_ANNEALED)
      RL_REQUIRE(cfg->exploration_period > 0, "exploration_period must be positive");
    uint64_t N = env->cfg.n_lanes;
    RL_REQUIRE(cfg->buffer_capacity * N * 46 < (200ull << 30), "replay store would not fit in HBM");
    RL_HIP_CHECK(hipSetDevice(e->device));
    std::unique_ptr<rl_dqn> q(new rl_dqn());
    q->eng = e;
    q->env = env;
    q->qnet = qnet;
    q->opt = opt;
    q->cfg = *cfg;
    q->cfg.episode_capacity = E;
    q->rp = replay_alloc(e, (uint32_t)N, (uint32_t)cfg->buffer_capacity, (uint32_t)E, env->D);
    q->d_agent_pos = dalloc<uint64_t>(1);
    RL_HIP_CHECK(hipMemsetAsync(q->d_agent_pos, 0, 8, e->stream));
    // take_while accepts episodes while total < minibatch_steps and every episode has >= 1 step
    q->max_eps = (uint32_t)cfg->minibatch_steps;
    q->max_steps_mb = cfg->minibatch_steps - 1 + cfg->buffer_capacity;
    // the episode lists of all opt_steps_per_update minibatches of an update are drawn in one launch
    const size_t nb = cfg->opt_steps_per_update ? cfg->opt_steps_per_update : 1;
    q->d_ep_lane = dalloc<uint32_t>(nb * q->max_eps);
    q->d_ep_start = dalloc<uint32_t>(nb * q->max_eps);
    q->d_ep_len = dalloc<uint32_t>(nb * q->max_eps);
    q->d_ep_off = dalloc<uint32_t>(nb * q->max_eps);
    q->d_counts = dalloc<DqnCountsDev>(nb);
    RL_HIP_CHECK(hipMemsetAsync(q->d_counts, 0, nb * sizeof(DqnCountsDev), e->stream));
    q->mb = traj_alloc(e, q->max_steps_mb, 1, env->D, true);
    sync(e);
    e->live_handles += 1;
    *out = q.release();
  });
}

int32_t rl_dqn_destroy(rl_dqn *q) {
  if (!q) return RL_OK;
  (void)hipSetDevice(q->eng->device);
  (void)hipStreamSynchronize(q->eng->stream);
  replay_free(q->rp);
  void *ptrs[] = {q->d_agent_pos, q->d_ep_lane, q->d_ep_start, q->d_ep_len, q->d_ep_off, q->d_counts, q->d_flags};
  for (void *p : ptrs) dfree(p);
  rl_traj_destroy(q->mb);
  rl_engine *eng = q->eng;
  delete q;
  engine_release_child(eng);
  return RL_OK;
}

int32_t rl_dqn_exploration_rate(const rl_dqn *q, int32_t training, double *rate_out) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q && rate_out, "NULL argument");
    *rate_out = dqn_exploration_rate(q, training != 0);
  });
}

int32_t rl_dqn_min_update_size(const rl_dqn *q, uint64_t *min_steps_out, uint64_t *slack_steps_out) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q && min_steps_out && slack_steps_out, "NULL argument");
    // DataCollectionSchedule::update_size (schedules.rs:58-68)
    uint64_t min_steps;
    if (q->cfg.update_kind == RL_COLLECT_CONSTANT) min_steps = q->cfg.update_first;
    else min_steps = q->global_steps < q->cfg.update_first ? q->cfg.update_first : q->cfg.update_rest;
    *min_steps_out = min_steps;
    // HistoryDataBound::with_default_slack (src/agents/buffers/mod.rs:54-63): 1 % of min_steps, between 5 and 1000
    uint64_t slack = min_steps / 100;
    slack = slack < 5 ? 5 : (slack > 1000 ? 1000 : slack);
    *slack_steps_out = slack;
  });
}

int32_t rl_dqn_collect(rl_dqn *q, uint64_t horizon, rl_dqn_collect_stats *stats) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q, "NULL argument");
    RL_REQUIRE(horizon > 0 && horizon < (1ull << 31), "bad horizon");
    rl_engine *e = q->eng;
    uint64_t N = q->rp.N;
    if (q->flags_cap < horizon * N) {
      dfree(q->d_flags);
      q->d_flags = nullptr;
      q->flags_cap = 0;
      q->d_flags = dalloc<uint8_t>(horizon * N);
      q->flags_cap = horizon * N;
    }
    // DqnAgent::actor(Training) (dqn.rs:200-211) + Bernoulli::new(p) of rand 0.8.5: p_int = (p * 2^64) as u64,
    // p == 1.0 always true without a draw
    double eps = dqn_exploration_rate(q, true);
    RL_REQUIRE(eps >= 0.0 && eps <= 1.0, "exploration rate outside [0, 1]");
    int always = eps == 1.0 ? 1 : 0;
    uint64_t p_int = always ? ~0ull : (uint64_t)(eps * 18446744073709551616.0);
    launch_rollout_dqn(q->env, q->qnet, q->rp, (uint32_t)horizon, p_int, always, q->d_flags);
    q->env->t_global += horizon;
    q->steps_per_lane += horizon;
    q->last_horizon = horizon;
    int32_t err = 0;
    d2h(e, &err, q->rp.error, sizeof(err));
    if (err != 0) throw RlError(RL_ERR_BUFFER_FULL, "replay buffer full: an episode outgrew the lane capacity");
    if (stats) {
      std::vector<uint8_t> fl(horizon * N);
      d2h(e, fl.data(), q->d_flags, fl.size());
      uint64_t ended = 0;
      for (uint8_t f : fl) ended += f != RL_SUCC_CONTINUE;
      stats->exploration_rate = eps;
      stats->steps = horizon * N;
      stats->episodes_ended = ended;
    }
  });
}

static AgentKey dqn_key(const rl_dqn *q) {
  AgentKey k;
  std::memcpy(k.w, q->cfg.agent_key, sizeof(k.w));
  return k;
}

// one sample_minibatch (dqn.rs:279-314): draw episodes, gather them, compute targets
// draw the episode lists of `n_batches` consecutive minibatches (dqn.rs:280-291) in one launch and read back their
// sizes: the draws do not depend on the network, so the whole update needs this one host round trip
static void dqn_draw_minibatches(rl_dqn *q, int sequential, uint32_t n_batches, std::vector<DqnCountsDev> &counts,
                                 std::vector<uint64_t> &totals) {
  rl_engine *e = q->eng;
  launch_dqn_sample(e, q->rp, dqn_key(q), q->d_agent_pos, (uint32_t)q->cfg.minibatch_steps, q->max_eps,
                    q->d_ep_lane, q->d_ep_start, q->d_ep_len, q->d_ep_off, q->d_counts, sequential, n_batches);
  counts.resize(n_batches);
  d2h(e, counts.data(), q->d_counts, n_batches * sizeof(DqnCountsDev));
  for (const DqnCountsDev &c : counts) {
    if (c.error == 2)
      throw RlError(RL_ERR_INVALID_ARGUMENT, "minibatch sampling from a lane without a complete episode");
    if (c.error != 0) throw RlError(RL_ERR_BUFFER_FULL, "replay buffer full");
    RL_REQUIRE(c.n_eps <= q->max_eps && c.n_steps <= q->max_steps_mb, "minibatch exceeds its workspace");
    RL_REQUIRE(c.n_steps > 0, "empty minibatch");
  }
  // the loss is a mean over all ranks' samples: sum the per-rank counts (two 16-bit halves each, exact in f32)
  totals.resize(n_batches);
  for (uint32_t k = 0; k < n_batches; ++k) totals[k] = counts[k].n_steps;
  if (e->n_ranks > 1) {
    std::vector<float> halves(2 * n_batches);
    for (uint32_t k = 0; k < n_batches; ++k) {
      halves[2 * k] = (float)(counts[k].n_steps & 0xffffu);
      halves[2 * k + 1] = (float)(counts[k].n_steps >> 16);
    }
    RL_REQUIRE(2 * n_batches <= q->mb->Pmax, "too many minibatches for the exchange buffer");
    h2d(e, q->mb->vec, halves.data(), halves.size() * sizeof(float));
    rl_allreduce_sum_f32(e, q->mb->vec, halves.size());
    d2h(e, halves.data(), q->mb->vec, halves.size() * sizeof(float));
    for (uint32_t k = 0; k < n_batches; ++k) totals[k] = (uint64_t)halves[2 * k] + ((uint64_t)halves[2 * k + 1] << 16);
  }
}

// gather minibatch `k` of the last draw and compute its targets (dqn.rs:293-314)
static void dqn_build_minibatch(rl_dqn *q, uint32_t k, const DqnCountsDev &c, uint64_t total) {
  q->last_n_eps = c.n_eps;
  q->last_n_steps = c.n_steps;
  q->last_total_steps = total;
  q->last_batch_index = k;
  rl_traj *mb = q->mb;
  mb->d.n = c.n_steps;
  mb->d.T = 1;
  traj_plan(mb, c.n_steps);
  const size_t o = (size_t)k * q->max_eps;
  launch_dqn_build_minibatch(q->eng, q->rp, c.n_eps, q->d_ep_lane + o, q->d_ep_start + o, q->d_ep_len + o,
                             q->d_ep_off + o, mb->d.obs, (size_t)2 * c.n_steps, mb->d.action, mb->d.adv,
                             q->cfg.discount_factor, q->cfg.target == RL_DQN_TARGET_ONE_STEP_TD ? 1 : 0, q->qnet);
}

static void dqn_sample_minibatch(rl_dqn *q, int sequential) {
  std::vector<DqnCountsDev> counts;
  std::vector<uint64_t> totals;
  dqn_draw_minibatches(q, sequential, 1, counts, totals);
  dqn_build_minibatch(q, 0, counts[0], totals[0]);
}

// gradient of mean((Q(s)[a] - target)^2) over the current minibatch -> mb->vec[0..P), loss sum -> mb->vec[P]
// `step_opt` != nullptr: also take the optimiser step, recording the loss in slot `loss_slot`; without an all-reduce
// between them the reduction and the (elementwise) step are one launch
static void dqn_gradient(rl_dqn *q, rl_adam *step_opt = nullptr, int loss_slot = -1) {
  rl_traj *mb = q->mb;
  uint32_t P = (uint32_t)q->qnet->P;
  uint32_t rowsA, rowsB;
  if (q->eng->kernel_variant != 1 && launch_policy_v2(mb, q->qnet, PASS_DQN, nullptr, q->last_total_steps, nullptr)) {
    rowsA = rowsB = mb->nbV2;
  } else {
    launch_policy_pass(mb, q->qnet, PASS_DQN, nullptr, q->last_total_steps, nullptr);
    launch_mlp_backward(mb, q->qnet, nullptr);
    rowsA = mb->nbA;
    rowsB = mb->nbB;
  }
  if (step_opt && !q->eng->has_collective()) {
    launch_reduce_adam(mb, step_opt, rowsA, rowsB, loss_slot, q->last_total_steps);
    return;
  }
  launch_reduce(mb, P, true, true, rowsA, rowsB);
  rl_allreduce_sum_f32(q->eng, mb->vec, P + 4);
  if (step_opt) launch_adam_step(mb, step_opt, loss_slot, q->last_total_steps);
}

int32_t rl_dqn_update(rl_dqn *q, rl_dqn_update_stats *stats, float *losses_out) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q, "NULL argument");
    rl_engine *e = q->eng;
    // self.global_steps = sum of total_step_count over the buffers (dqn.rs:276); every lane of every rank has
    // taken the same number of steps and the horizon rule drops none
    q->global_steps = q->steps_per_lane * (uint64_t)q->rp.N * (uint64_t)e->n_ranks;
    uint64_t K = q->cfg.opt_steps_per_update;
    std::vector<DqnCountsDev> counts;
    std::vector<uint64_t> totals;
    if (K) dqn_draw_minibatches(q, 0, (uint32_t)K, counts, totals);
    for (uint64_t k = 0; k < K; ++k) {
      dqn_build_minibatch(q, (uint32_t)k, counts[k], totals[k]);
      dqn_gradient(q, q->opt, (int)k);
    }
    std::vector<float> h(K ? K : 1, 0.0f);
    if (K) d2h(e, h.data(), q->mb->losses, K * sizeof(float));
    if (losses_out && K) std::memcpy(losses_out, h.data(), K * sizeof(float));
    if (stats) {
      stats->opt_steps = K;
      stats->loss_first = K ? (double)h[0] : 0.0;
      stats->loss_last = K ? (double)h[K - 1] : 0.0;
      stats->global_steps = q->global_steps;
      stats->last_minibatch_steps = q->last_n_steps;
      stats->last_minibatch_episodes = q->last_n_eps;
    }
  });
}

static void replay_field(const rl_dqn *q, int32_t field, void **ptr, uint64_t *bytes) {
  const ReplayDev &r = q->rp;
  uint64_t N = r.N, C = r.C, E = r.E, D = r.D;
  switch (field) {
    case RL_REPLAY_HEAD: *ptr = r.head; *bytes = N * 4; break;
    case RL_REPLAY_COUNT: *ptr = r.count; *bytes = N * 4; break;
    case RL_REPLAY_EP_HEAD: *ptr = r.ep_head; *bytes = N * 4; break;
    case RL_REPLAY_EP_COUNT: *ptr = r.ep_count; *bytes = N * 4; break;
    case RL_REPLAY_TOTAL: *ptr = r.total; *bytes = N * 4; break;
    case RL_REPLAY_EP_END: *ptr = r.ep_end; *bytes = E * N * 4; break;
    case RL_REPLAY_OBS: *ptr = r.obs; *bytes = D * C * N * 4; break;
    case RL_REPLAY_NEXT_OBS: *ptr = r.next_obs; *bytes = D * C * N * 4; break;
    case RL_REPLAY_ACTION: *ptr = r.action; *bytes = C * N; break;
    case RL_REPLAY_REWARD: *ptr = r.reward; *bytes = C * N * 4; break;
    case RL_REPLAY_FLAG: *ptr = r.flag; *bytes = C * N; break;
    case RL_REPLAY_ACTOR_POS: *ptr = r.actor_pos; *bytes = N * 8; break;
    case RL_REPLAY_LAST_FLAGS: *ptr = q->d_flags; *bytes = q->last_horizon * N; break;
    default: throw RlError(RL_ERR_INVALID_ARGUMENT, "unknown replay field");
  }
}

int32_t rl_dqn_replay_field_bytes(const rl_dqn *q, int32_t field, uint64_t *bytes) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q && bytes, "NULL argument");
    void *p;
    replay_field(q, field, &p, bytes);
  });
}

int32_t rl_dqn_replay_read(rl_dqn *q, int32_t field, void *host, uint64_t bytes) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q && host, "NULL argument");
    void *p;
    uint64_t need;
    replay_field(q, field, &p, &need);
    RL_REQUIRE(bytes == need, "byte count mismatch for replay field");
    if (bytes) d2h(q->eng, host, p, bytes);
  });
}

int32_t rl_dqn_minibatch_sample(rl_dqn *q, int32_t sequential, uint64_t *n_episodes_out, uint64_t *n_steps_out) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q, "NULL argument");
    dqn_sample_minibatch(q, sequential);
    if (n_episodes_out) *n_episodes_out = q->last_n_eps;
    if (n_steps_out) *n_steps_out = q->last_n_steps;
  });
}

int32_t rl_dqn_minibatch_read(rl_dqn *q, int32_t field, void *host, uint64_t bytes) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q && host, "NULL argument");
    uint64_t ne = q->last_n_eps, ns = q->last_n_steps, D = q->rp.D;
    RL_REQUIRE(ns > 0, "no minibatch has been sampled");
    rl_engine *e = q->eng;
    switch (field) {
      case RL_MB_EP_LANE: RL_REQUIRE(bytes == ne * 4, "byte count mismatch"); d2h(e, host, q->d_ep_lane + (size_t)q->last_batch_index * q->max_eps, bytes); break;
      case RL_MB_EP_START: RL_REQUIRE(bytes == ne * 4, "byte count mismatch"); d2h(e, host, q->d_ep_start + (size_t)q->last_batch_index * q->max_eps, bytes); break;
      case RL_MB_EP_LEN: RL_REQUIRE(bytes == ne * 4, "byte count mismatch"); d2h(e, host, q->d_ep_len + (size_t)q->last_batch_index * q->max_eps, bytes); break;
      case RL_MB_EP_OFFSET: RL_REQUIRE(bytes == ne * 4, "byte count mismatch"); d2h(e, host, q->d_ep_off + (size_t)q->last_batch_index * q->max_eps, bytes); break;
      case RL_MB_OBS: {
        RL_REQUIRE(bytes == D * ns * 4, "byte count mismatch");
        // feature planes are 2 * n_steps apart in the workspace (T = 1 trajectory layout)
        for (uint64_t d = 0; d < D; ++d)
          d2h(e, (char *)host + d * ns * 4, q->mb->d.obs + d * 2 * ns, ns * 4);
        break;
      }
      case RL_MB_ACTION: RL_REQUIRE(bytes == ns, "byte count mismatch"); d2h(e, host, q->mb->d.action, bytes); break;
      case RL_MB_TARGET: RL_REQUIRE(bytes == ns * 4, "byte count mismatch"); d2h(e, host, q->mb->d.adv, bytes); break;
      default: throw RlError(RL_ERR_INVALID_ARGUMENT, "unknown minibatch field");
    }
  });
}

int32_t rl_dqn_minibatch_gradient(rl_dqn *q, float *grad_out, float *loss_out) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q && grad_out, "NULL argument");
    RL_REQUIRE(q->last_n_steps > 0, "no minibatch has been sampled");
    uint32_t P = (uint32_t)q->qnet->P;
    dqn_gradient(q);
    std::vector<float> h(P + 4);
    d2h(q->eng, h.data(), q->mb->vec, (P + 4) * sizeof(float));
    std::memcpy(grad_out, h.data(), P * sizeof(float));
    if (loss_out) *loss_out = (float)((double)h[P] / (double)q->last_total_steps);
  });
}

int32_t rl_dqn_agent_rng_pos(rl_dqn *q, uint64_t *pos_out) {
  return guarded(q ? q->eng : nullptr, [&] {
    RL_REQUIRE(q && pos_out, "NULL argument");
    d2h(q->eng, pos_out, q->d_agent_pos, sizeof(uint64_t));
  });
}

}  // extern "C"
