// abi.hip — extern "C" entry points of include/relearn_hip.h, part: engine, collectives, environments, modules,
// trajectories, rollout and advantage estimation (host side only; kernels live in kernels_*.hip).
#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>

#include "abi_internal.hpp"

thread_local std::string g_last_error_no_engine;

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
  // RCCL must run on the SAME HIP runtime as this library: a host program may have mapped a second copy of the ROCm
  // libraries (PyTorch bundles libamdhip64 / librccl under torch/lib), and a communicator created by the other copy's
  // RCCL cannot use this runtime's streams ("unhandled cuda error").  So: first the librccl that sits next to the
  // libamdhip64 this library is bound to, then the usual names.
  std::vector<std::string> names;
  Dl_info info;
  if (dladdr((void *)&hipGetDeviceCount, &info) != 0 && info.dli_fname != nullptr) {
    std::string dir(info.dli_fname);
    const size_t slash = dir.rfind('/');
    if (slash != std::string::npos) {
      dir.resize(slash);
      names.push_back(dir + "/librccl.so.1");
      names.push_back(dir + "/librccl.so");
    }
  }
  names.push_back("librccl.so.1");
  names.push_back("librccl.so");
  names.push_back("/opt/rocm/lib/librccl.so.1");
  for (const std::string &n : names) {
    g_rccl.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
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
// (`bad_rank` >= 0: a test of the tests — that rank's vector enters the sum multiplied by `bad_weight`, what a rank that
// weighs its samples by a wrong B_local / B_total contributes; every rank still receives the same sums, so the job runs
// to its end and must FAIL the sharded parity bars: tests/test_gpu_multirank.py, RELEARN_LOOPBACK_TEST_WEIGHT)
__global__ void k_loopback_sum(float *const *bufs, int n_ranks, size_t count, int bad_rank, float bad_weight) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float s = bad_rank == 0 ? bad_weight * bufs[0][i] : bufs[0][i];
  for (int r = 1; r < n_ranks; ++r) s = s + (r == bad_rank ? bad_weight * bufs[r][i] : bufs[r][i]);
  for (int r = 0; r < n_ranks; ++r) bufs[r][i] = s;
}

struct LoopbackGroup {
  std::mutex mu;
  std::condition_variable cv;
  int n_ranks = 0, arrived = 0, joined = 0;
  uint64_t generation = 0;
  std::vector<float *> bufs;
  float **d_bufs = nullptr;
  int bad_rank = -1;        // RELEARN_LOOPBACK_TEST_WEIGHT="rank:weight" when the group was made (k_loopback_sum)
  float bad_weight = 1.0f;  // ... applied to the gradient-sized vectors only (the set-up's rank count stays a count)
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
                       g->n_ranks, count, count > 64 ? g->bad_rank : -1, g->bad_weight);
    RL_HIP_CHECK(hipStreamSynchronize(e->stream));
  }
  g->barrier(lk);
}

void rl_allreduce_sum_f32(rl_engine *e, float *d_buf, size_t count) {
  if (e->ipc_active) {
    if (!ipc_allreduce_fits(e, count))
      throw RlError(RL_ERR_UNSUPPORTED, "the peer-mailbox collective carries vectors of <= 2048 floats (feed-forward "
                                        "modules); use the RCCL transport for larger ones");
    ProfScope ps(e, RL_K_ALLREDUCE);
    ipc_allreduce(e, d_buf, count);
    return;
  }
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
  if (!e->comm) {
    // several ranks and nothing to exchange with: every update would scale by the global sample count while summing
    // local samples only, and the replicas would drift apart silently
    if (e->n_ranks > 1) throw RlError(RL_ERR_COMM, "engine has n_ranks > 1 but no collective (rl_comm_init failed?)");
    return;
  }
  ProfScope ps(e, RL_K_ALLREDUCE);
  // ncclFloat32 = 7, ncclSum = 0
  // (the auxiliary chain has a communicator of its own: collectives of two chains in flight on two streams must not
  // share one — rl_actor_critic_update runs the chains one after the other when there is no second communicator)
  void *comm = e->chan == 1 && e->comm_aux ? e->comm_aux : e->comm;
  rccl_check(g_rccl.AllReduce(d_buf, d_buf, count, 7, 0, comm, e->stream), "ncclAllReduce");
}

// Settings every rank has to take the same way, decided through the collective that was just installed: an all-reduce
// that also counts the ranks (a transport that does not reach everybody fails here, not in the first update).
//   RELEARN_SERIAL_UPDATE in ANY rank's environment -> the update chains run in turn on EVERY rank (a rank running them
//   side by side would put the critic's collectives on the auxiliary channel, its peers on the main one).
// rl_engine_set_serial_update is an API call, not an environment: the host program makes it on every rank or on none.
void comm_agree(rl_engine *e) {
  const bool mine = std::getenv("RELEARN_SERIAL_UPDATE") != nullptr;
  if (!e->has_collective()) {
    e->agreed_serial_env = mine;
    return;
  }
  float h[2] = {mine ? 1.0f : 0.0f, 1.0f};
  float *d = dalloc<float>(2);
  try {
    h2d(e, d, h, sizeof(h));
    rl_allreduce_sum_f32(e, d, 2);
    d2h(e, h, d, sizeof(h));
    ipc_check(e);
  } catch (...) {
    dfree(d);
    throw;
  }
  dfree(d);
  if (h[1] != (float)e->n_ranks)
    throw RlError(RL_ERR_COMM, "the collective's first all-reduce counted " + std::to_string((int)h[1]) + " of " +
                                   std::to_string(e->n_ranks) + " ranks");
  e->agreed_serial_env = h[0] != 0.0f;
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
    RL_HIP_CHECK(hipStreamCreateWithFlags(&e->main_stream, hipStreamNonBlocking));
    RL_HIP_CHECK(hipStreamCreateWithFlags(&e->aux_stream, hipStreamNonBlocking));
    e->stream = e->main_stream;
    RL_HIP_CHECK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    RL_HIP_CHECK(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
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

void engine_release_child(rl_engine *e) {
  e->live_handles -= 1;
  if (e->zombie && e->live_handles <= 0) engine_teardown(e);
}

static void engine_teardown(rl_engine *e) {
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->main_stream);
  (void)hipStreamSynchronize(e->aux_stream);
  if (e->comm_aux && g_rccl.CommDestroy) g_rccl.CommDestroy(e->comm_aux);
  if (e->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(e->comm);
  ipc_teardown(e);
  prof_drain(e);
  for (auto ev : e->prof_event_pool) (void)hipEventDestroy(ev);
  if (e->pinned) (void)hipHostFree(e->pinned);
  (void)hipEventDestroy(e->ev_begin);
  (void)hipEventDestroy(e->ev_end);
  (void)hipEventDestroy(e->ev_fork);
  (void)hipEventDestroy(e->ev_join);
  (void)hipStreamDestroy(e->aux_stream);
  (void)hipStreamDestroy(e->main_stream);
  delete e;
}

int32_t rl_engine_sync(rl_engine *e) {
  return guarded(e, [&] {
    RL_REQUIRE(e, "engine is NULL");
    sync(e);
    // also drain anything a collective library queued on streams of its own
    RL_HIP_CHECK(hipSetDevice(e->device));
    RL_HIP_CHECK(hipDeviceSynchronize());
    ipc_check(e);
  });
}

const char *rl_last_error(const rl_engine *e) { return e ? e->last_error.c_str() : g_last_error_no_engine.c_str(); }

int32_t rl_engine_info(const rl_engine *e, char *name_out, size_t name_cap, char *arch_out, size_t arch_cap,
                       int32_t *compute_units) {
  return guarded(const_cast<rl_engine *>(e), [&] {
    RL_REQUIRE(e, "engine is NULL");
    // (some driver stacks report an empty marketing name: the architecture then stands in for it)
    if (name_out && name_cap) std::snprintf(name_out, name_cap, "%s", e->prop.name[0] ? e->prop.name : e->prop.gcnArchName);
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
    RL_REQUIRE(variant == 0 || variant == 1, "kernel variant must be 0 (best) or 1 (v1 reference kernels)");
    e->kernel_variant = variant;
  });
}

int32_t rl_engine_set_serial_update(rl_engine *e, int32_t serial) {
  return guarded(e, [&] {
    RL_REQUIRE(e, "engine is NULL");
    e->serial_update = serial != 0;
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
int32_t rl_comm_available(void) {
  return guarded(nullptr, [&] {
    if (std::getenv("RELEARN_LOOPBACK_COMM")) return;
    rccl_load();
  });
}

int32_t rl_comm_library_paths(char *rccl_out, size_t rccl_cap, char *hip_out, size_t hip_cap) {
  return guarded(nullptr, [&] {
    Dl_info info;
    if (rccl_out && rccl_cap) {
      rccl_out[0] = 0;
      if (g_rccl.AllReduce && dladdr((void *)g_rccl.AllReduce, &info) != 0 && info.dli_fname)
        std::snprintf(rccl_out, rccl_cap, "%s", info.dli_fname);
    }
    if (hip_out && hip_cap) {
      hip_out[0] = 0;
      if (dladdr((void *)&hipGetDeviceCount, &info) != 0 && info.dli_fname)
        std::snprintf(hip_out, hip_cap, "%s", info.dli_fname);
    }
  });
}

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
    // rank / n_ranks are set only once the collective exists: a failed initialisation leaves a one-rank engine
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
        if (const char *tw = std::getenv("RELEARN_LOOPBACK_TEST_WEIGHT")) {
          int r = -1;
          float w = 1.0f;
          if (std::sscanf(tw, "%d:%f", &r, &w) == 2 && r >= 0 && r < n_ranks) grp->bad_rank = r, grp->bad_weight = w;
        }
      }
      RL_REQUIRE(grp->n_ranks == n_ranks, "loopback group: inconsistent n_ranks");
      grp->joined += 1;
      e->loopback = grp.get();
      e->rank = rank;
      e->n_ranks = n_ranks;
      e->agreed_serial_env = std::getenv("RELEARN_SERIAL_UPDATE") != nullptr;  // (one process: one environment)
      return;
    }
    if (n_ranks == 1 && !std::getenv("RELEARN_FORCE_RCCL")) return;
    rccl_load();
    RL_HIP_CHECK(hipSetDevice(e->device));
    UniqueId id;
    std::memcpy(id.bytes, unique_id, 128);
    comm_init_rank_t init = (comm_init_rank_t)(void *)g_rccl.CommInitRank;
    void *comm = nullptr;
    rccl_check(init(&comm, n_ranks, id, rank), "ncclCommInitRank");
    e->comm = comm;
    e->rank = rank;
    e->n_ranks = n_ranks;
    // A second communicator over the same ranks for the auxiliary update chain (rl_actor_critic_update runs the policy
    // and the critic chain on two streams, each with its collectives).  Its unique id is made on rank 0 and handed out
    // through the first communicator — an all-reduce in which only rank 0 contributes (bytes as small integers: exact in
    // f32) — so the caller's bootstrap stays one id.  Whether the job HAS the second communicator is decided
    // collectively: rank 0 joins the hand-out even when it could not make an id (zeros plus a failure flag), a rank with
    // RELEARN_NO_AUX_COMM=1 in its environment vetoes for all, and after the attempt the ranks all-reduce their outcome
    // over the first communicator — one rank without it and every rank drops it (the two chains then run in turn
    // everywhere: ranks that disagreed would issue the critic's collectives on different communicators and hang).
    try {
      const bool want_aux = std::getenv("RELEARN_NO_AUX_COMM") == nullptr;
      UniqueId id2;
      std::memset(id2.bytes, 0, sizeof(id2.bytes));
      bool id_ok = false;
      if (rank == 0 && want_aux) id_ok = g_rccl.GetUniqueId(id2.bytes) == 0;
      float h_id[130];
      for (int i = 0; i < 128; ++i) h_id[i] = rank == 0 && id_ok ? (float)(unsigned char)id2.bytes[i] : 0.0f;
      h_id[128] = rank == 0 && !id_ok ? 1.0f : 0.0f;  // rank 0 has no id to offer
      h_id[129] = want_aux ? 0.0f : 1.0f;             // this rank declines
      float *d_id = dalloc<float>(130);
      void *comm2 = nullptr;
      try {
        h2d(e, d_id, h_id, sizeof(h_id));
        rccl_check(g_rccl.AllReduce(d_id, d_id, 130, 7, 0, e->comm, e->stream), "ncclAllReduce (second unique id)");
        d2h(e, h_id, d_id, sizeof(h_id));
        const bool attempt = h_id[128] == 0.0f && h_id[129] == 0.0f;
        float outcome = 0.0f;
        if (attempt) {
          for (int i = 0; i < 128; ++i) id2.bytes[i] = (char)(unsigned char)h_id[i];
          if (init(&comm2, n_ranks, id2, rank) == 0 && comm2 != nullptr) outcome = 1.0f;
          else comm2 = nullptr;
        }
        h2d(e, d_id, &outcome, sizeof(outcome));
        rccl_check(g_rccl.AllReduce(d_id, d_id, 1, 7, 0, e->comm, e->stream), "ncclAllReduce (second communicator: outcome)");
        d2h(e, &outcome, d_id, sizeof(outcome));
        if (outcome == (float)n_ranks) {
          e->comm_aux = comm2;
          comm2 = nullptr;
        } else {
          if (rank == 0)
            std::fprintf(stderr, "relearn_hip: no second communicator (%s): update chains will run one after the other "
                                 "on every rank\n",
                         !attempt ? (h_id[129] != 0.0f ? "RELEARN_NO_AUX_COMM on some rank" : "rank 0 could not make an id")
                                  : "ncclCommInitRank failed on some rank");
        }
      } catch (...) {
        if (comm2) g_rccl.CommDestroy(comm2);
        dfree(d_id);
        throw;
      }
      if (comm2) g_rccl.CommDestroy(comm2);  // created here, but not everywhere
      dfree(d_id);
      comm_agree(e);
    } catch (...) {  // the first communicator could not even carry the agreement: no collective at all
      if (e->comm_aux) g_rccl.CommDestroy(e->comm_aux);
      g_rccl.CommDestroy(e->comm);
      e->comm = e->comm_aux = nullptr;
      e->rank = 0;
      e->n_ranks = 1;
      throw;
    }
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
    try {
      comm_agree(e);  // (every rank installs its callback at the same point of the job: the first all-reduce runs here)
    } catch (...) {
      e->host_allreduce = nullptr;
      e->host_allreduce_ctx = nullptr;
      e->rank = 0;
      e->n_ranks = 1;
      throw;
    }
  });
}

// Every rank sends pseudo-random payloads that every rank can recompute: small integers, so that the expected sum is
// exact in f32 whatever order a transport adds in.  240 rounds x {2048, 1030, 901, 64, 4} elements: on the mailbox
// transport that is every one of the 32 chunks, both slots many times over, sequence numbers running — a torn or stale
// 64-bit word, or a lost store, shows up as a wrong element.
static inline int32_t selftest_payload(uint32_t round, uint32_t rank, uint32_t i) {
  uint64_t z = ((uint64_t)round << 40) ^ ((uint64_t)rank << 32) ^ (uint64_t)i;
  z += 0x9E3779B97F4A7C15ull;  // splitmix64
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (int32_t)(z & 0xFFFF) - 32768;
}

int32_t rl_comm_selftest(rl_engine *e) {
  return guarded(e, [&] {
    RL_REQUIRE(e, "engine is NULL");
    const uint32_t sizes[5] = {2048, 1030, 901, 64, 4};  // IPC_CAP; policy gradient + scalars; critic; one chunk; scalars
    std::vector<float> h(2048);
    float *d = dalloc<float>(2048);
    try {
      for (uint32_t round = 0; round < 240; ++round) {
        const uint32_t count = sizes[round % 5];
        for (uint32_t i = 0; i < count; ++i) h[i] = (float)selftest_payload(round, (uint32_t)e->rank, i);
        h2d(e, d, h.data(), count * sizeof(float));
        rl_allreduce_sum_f32(e, d, count);
        if (round % 16 == 15 || round == 239) {  // (between checks the collectives run back to back, as in an update)
          d2h(e, h.data(), d, count * sizeof(float));
          ipc_check(e);
          for (uint32_t i = 0; i < count; ++i) {
            int32_t want = 0;
            for (int r = 0; r < e->n_ranks; ++r) want += selftest_payload(round, (uint32_t)r, i);
            if (h[i] != (float)want)
              throw RlError(RL_ERR_COMM, "collective self-test: wrong sum at element " + std::to_string(i) + " of round " +
                                             std::to_string(round));
          }
        }
      }
    } catch (...) {
      dfree(d);
      throw;
    }
    dfree(d);
  });
}

int32_t rl_comm_destroy(rl_engine *e) {
  return guarded(e, [&] {
    RL_REQUIRE(e, "engine is NULL");
    e->host_allreduce = nullptr;
    e->host_allreduce_ctx = nullptr;
    if (e->comm_aux) {
      RL_HIP_CHECK(hipStreamSynchronize(e->aux_stream));
      rccl_check(g_rccl.CommDestroy(e->comm_aux), "ncclCommDestroy");
      e->comm_aux = nullptr;
    }
    if (e->comm) {
      sync(e);
      rccl_check(g_rccl.CommDestroy(e->comm), "ncclCommDestroy");
      e->comm = nullptr;
    }
    e->loopback = nullptr;  // groups live for the life of the process (test facility)
    if (e->ipc_box) {
      sync(e);
      ipc_teardown(e);
    }
    e->ipc_active = false;
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

// every device allocation of an env handle (also the clean-up of a failed rl_env_create)
static void env_release_device(rl_env *env) {
  void *ptrs[] = {env->st.x, env->st.xdot, env->st.th, env->st.thdot, env->st.nv_pos, env->st.steps_remaining,
                  env->st.reset_count, env->d_actions, env->d_flag, env->d_reward, env->d_obs, env->d_term_obs};
  for (void *p : ptrs) dfree(p);
  env->st = EnvStateDev{};
  env->d_actions = env->d_flag = nullptr;
  env->d_reward = env->d_obs = env->d_term_obs = nullptr;
}

int32_t rl_env_create(rl_engine *e, const rl_env_config *cfg, rl_env **out) {
  return guarded(e, [&] {
    RL_REQUIRE(e && cfg && out, "NULL argument");
    *out = nullptr;
    if (cfg->kind != RL_ENV_CARTPOLE && cfg->kind != RL_ENV_CHAIN && cfg->kind != RL_ENV_MEMORY &&
        cfg->kind != RL_ENV_BANDIT)
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
    if (cfg->kind == RL_ENV_BANDIT && cfg->limit_kind != RL_LIMIT_NONE)
      throw RlError(RL_ERR_BUILD_ENV, "bandit lanes take no step limit (every step ends the episode)");
    if (cfg->kind == RL_ENV_CHAIN || cfg->kind == RL_ENV_MEMORY || cfg->kind == RL_ENV_BANDIT)
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
    d.bandit = cfg->kind == RL_ENV_BANDIT ? 1u : 0u;
    d.bandit_r[0] = (float)cfg->bandit_values[0];  // Reward -> f32 feedback, as every env's reward record
    d.bandit_r[1] = (float)cfg->bandit_values[1];
    size_t n = cfg->n_lanes;
    try {
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
    } catch (...) {  // (unique_ptr frees the host struct only)
      env_release_device(env.get());
      throw;
    }
    e->live_handles += 1;
    *out = env.release();
  });
}

int32_t rl_env_destroy(rl_env *env) {
  if (!env) return RL_OK;
  (void)hipSetDevice(env->eng->device);
  (void)hipStreamSynchronize(env->eng->stream);
  (void)hipStreamSynchronize(env->eng->aux_stream);
  env_release_device(env);
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

// one thread per word: block = word / 16 of the stream, the word's position in it
__global__ void k_debug_stream_words(AgentKey key, uint64_t stream, uint64_t first_word, uint32_t n_words,
                                     uint32_t *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_words) return;
  const uint64_t w = first_word + i;
  uint32_t words[16];
  rl_chacha_block(key.w, w >> 4, stream, 4, words);
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) v = (uint32_t)(w & 15) == (uint32_t)k ? words[k] : v;
  out[i] = v;
}

int32_t rl_debug_stream_words(rl_engine *engine, uint64_t seed, uint64_t stream, uint64_t first_word, uint32_t n_words,
                              uint32_t *words_out) {
  return guarded(engine, [&] {
    RL_REQUIRE(engine && words_out, "NULL argument");
    RL_REQUIRE(n_words > 0 && n_words <= (1u << 20), "n_words must be in [1, 2^20]");
    AgentKey key;
    rl_seed_from_u64(seed, key.w);
    uint32_t *d = dalloc<uint32_t>(n_words);
    try {
      hipLaunchKernelGGL(k_debug_stream_words, dim3((n_words + 255) / 256), dim3(256), 0, engine->stream, key, stream,
                         first_word, n_words, d);
      RL_HIP_CHECK(hipGetLastError());
      d2h(engine, words_out, d, (size_t)n_words * sizeof(uint32_t));
    } catch (...) {
      dfree(d);
      throw;
    }
    dfree(d);
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
    m->widths[0] = hidden;
    m->P = (uint64_t)hidden * in_dim + hidden + (uint64_t)out_dim * hidden + out_dim;
    m->d_params = dalloc<float>(m->P);
    RL_HIP_CHECK(hipMemsetAsync(m->d_params, 0, m->P * sizeof(float), e->stream));
    sync(e);
    e->live_handles += 1;
    *out = m.release();
  });
}

static int32_t mlp_create_config(rl_engine *e, uint32_t in_dim, const uint32_t *hidden_sizes, uint32_t n_hidden,
                                 uint32_t out_dim, int32_t activation, int32_t output_activation, bool has_bias,
                                 rl_mlp **out) {
  // one hidden layer of at most 128 units, Relu inside and Identity on the output, with biases: the fused kernels
  if (has_bias && e && out && hidden_sizes && n_hidden == 1 && hidden_sizes[0] <= 128 && activation == RL_ACT_RELU &&
      output_activation == RL_ACT_IDENTITY && (in_dim == 4 || in_dim == 5))
    return rl_mlp_create(e, in_dim, hidden_sizes[0], out_dim, out);
  return guarded(e, [&] {
    RL_REQUIRE(e && out && (hidden_sizes || n_hidden == 0), "NULL argument");
    *out = nullptr;
    // (the envs of this library have 4 or 5 features; other widths serve host-made histories: rl_traj_write)
    bool ok = in_dim >= 1 && in_dim <= RL_TRAJ_MAX_OBS_DIM && (out_dim == 1 || out_dim == 2) && n_hidden <= RL_MLP_MAX_HIDDEN;
    for (uint32_t l = 0; ok && l < n_hidden; ++l) ok = hidden_sizes[l] >= 1 && hidden_sizes[l] <= RL_MLP_MAX_WIDTH;
    if (!ok)
      throw RlError(RL_ERR_BUILD_AGENT, "supported MLP shapes: in_dim 1..8, at most 4 hidden layers of 1..256 units, "
                                        "out_dim in {1,2}");
    if (activation < RL_ACT_IDENTITY || activation > RL_ACT_TANH || output_activation < RL_ACT_IDENTITY ||
        output_activation > RL_ACT_TANH)
      throw RlError(RL_ERR_BUILD_AGENT, "activation / output_activation must be one of rl_activation "
                                        "(Identity, Relu, Sigmoid, Tanh)");
    RL_HIP_CHECK(hipSetDevice(e->device));
    std::unique_ptr<rl_mlp> m(new rl_mlp());
    m->eng = e;
    m->in_dim = in_dim;
    m->hidden = 0;  // no fused kernel takes this module
    m->out_dim = out_dim;
    m->n_hidden = n_hidden;
    for (uint32_t l = 0; l < n_hidden; ++l) m->widths[l] = hidden_sizes[l];
    m->general = true;
    m->act = activation;
    m->out_act = output_activation;
    m->has_bias = has_bias;
    m->P = m->layer_offset(m->n_layers());
    // (bias-less layers read their dot products' starting value from zeros behind the parameters: rl_mlp::bias_offset)
    const size_t alloc = m->P + (has_bias ? 0 : RL_MLP_MAX_WIDTH);
    m->d_params = dalloc<float>(alloc);
    RL_HIP_CHECK(hipMemsetAsync(m->d_params, 0, alloc * sizeof(float), e->stream));
    sync(e);
    e->live_handles += 1;
    *out = m.release();
  });
}

int32_t rl_mlp_create_layers(rl_engine *e, uint32_t in_dim, const uint32_t *hidden_sizes, uint32_t n_hidden,
                             uint32_t out_dim, int32_t activation, int32_t output_activation, rl_mlp **out) {
  return mlp_create_config(e, in_dim, hidden_sizes, n_hidden, out_dim, activation, output_activation, true, out);
}

int32_t rl_mlp_create_config(rl_engine *e, uint32_t in_dim, const uint32_t *hidden_sizes, uint32_t n_hidden,
                             uint32_t out_dim, int32_t activation, int32_t output_activation, int32_t bias, rl_mlp **out) {
  return mlp_create_config(e, in_dim, hidden_sizes, n_hidden, out_dim, activation, output_activation, bias != 0, out);
}

// ChainConfig<GruConfig | LstmConfig, MlpConfig>::build_module (modules/chain.rs:19-56, seq/rnn/mod.rs:20-45,223-281):
// RnnBaseConfig { hidden_size, num_layers, .. } -> ReLU -> Mlp with one hidden layer.  The kernels are built for
// 5 -> 128 -> 128 -> {1, 2}; narrower shapes (in_dim <= 5, widths <= 128) run on them embedded with zero padding
// (rl_mlp::exec).  num_layers in 2..4 (stacked layers): the lane-per-thread kernels of kernels_seq_stack.hip at the
// module's own widths.
static void seq_module_create(rl_engine *e, int kind, uint32_t in_dim, uint32_t rnn_hidden, uint32_t num_layers,
                              uint32_t mlp_hidden, uint32_t out_dim, rl_mlp **out, bool rnn_bias = true) {
  RL_REQUIRE(e && out, "NULL argument");
  *out = nullptr;
  if (num_layers == 0) throw RlError(RL_ERR_BUILD_AGENT, "RnnBaseConfig::num_layers must be at least 1");
  if (num_layers > RL_RNN_MAX_LAYERS)
    throw RlError(RL_ERR_UNSUPPORTED, "recurrent chains are built for RnnBaseConfig::num_layers <= 4");
  if (in_dim < 1 || in_dim > RL_TRAJ_MAX_OBS_DIM || rnn_hidden < 1 || rnn_hidden > RL_MLP_MAX_WIDTH || mlp_hidden < 1 ||
      mlp_hidden > RL_MLP_MAX_WIDTH || !(out_dim == 1 || out_dim == 2))
    throw RlError(RL_ERR_BUILD_AGENT,
                  "supported recurrent chain shapes: in_dim 1..8, recurrent hidden 1..256, mlp_hidden 1..256, out_dim in {1,2}");
  RL_HIP_CHECK(hipSetDevice(e->device));
  auto make = [&](uint32_t D, uint32_t H, uint32_t H2, uint32_t layers, bool bias) {
    std::unique_ptr<rl_mlp> m(new rl_mlp());
    m->eng = e;
    m->kind = kind;
    m->in_dim = D;
    m->gru_hidden = H;
    m->hidden = H2;
    m->out_dim = out_dim;
    const uint64_t A = out_dim;
    m->rnn_layers = layers;
    m->has_bias = bias;  // of the recurrent weights (RnnBaseConfig::bias_init); the chain's MLP always has its own
    m->P = m->rnn_layer_offset(m->rnn_layers) + (uint64_t)H2 * H + H2 + A * H2 + A;
    // (bias-less recurrent weights: the gate rows start from zeros kept behind the parameters)
    const size_t alloc = m->P + (m->has_bias ? 0 : 4 * RL_MLP_MAX_WIDTH);
    m->d_params = dalloc<float>(alloc);
    RL_HIP_CHECK(hipMemsetAsync(m->d_params, 0, alloc * sizeof(float), e->stream));
    return m;
  };
  std::unique_ptr<rl_mlp> m = make(in_dim, rnn_hidden, mlp_hidden, num_layers, rnn_bias);
  if (!m->lane_kernels() && (in_dim != 5 || rnn_hidden != 128 || mlp_hidden != 128)) {
    std::unique_ptr<rl_mlp> x;
    try {
      x = make(5, 128, 128, 1, true);  // every padding entry stays 0 for the life of the module
      m->x_tmp = dalloc<float>(x->P);
      m->x_tan = dalloc<float>(x->P);
      RL_HIP_CHECK(hipMemsetAsync(m->x_tan, 0, x->P * sizeof(float), e->stream));
    } catch (...) {
      if (x) dfree(x->d_params);
      dfree(m->x_tmp);
      dfree(m->x_tan);
      dfree(m->d_params);
      throw;
    }
    m->exec = x.release();
  }
  sync(e);
  e->live_handles += 1;
  *out = m.release();
}

// the module the recurrent kernels run (see rl_mlp::exec), its parameter image refreshed from the module's flat vector
rl_mlp *seq_exec(const rl_mlp *m) {
  if (m->exec == nullptr) return const_cast<rl_mlp *>(m);
  launch_seq_pad(m, m->exec->d_params, m->d_params);
  return m->exec;
}

int32_t rl_gru_mlp_create(rl_engine *e, uint32_t in_dim, uint32_t gru_hidden, uint32_t mlp_hidden, uint32_t out_dim,
                          rl_mlp **out) {
  return guarded(e, [&] { seq_module_create(e, RL_MODULE_GRU_MLP, in_dim, gru_hidden, 1, mlp_hidden, out_dim, out); });
}

int32_t rl_lstm_mlp_create(rl_engine *e, uint32_t in_dim, uint32_t lstm_hidden, uint32_t mlp_hidden, uint32_t out_dim,
                           rl_mlp **out) {
  return guarded(e, [&] { seq_module_create(e, RL_MODULE_LSTM_MLP, in_dim, lstm_hidden, 1, mlp_hidden, out_dim, out); });
}

int32_t rl_rnn_mlp_create(rl_engine *e, int32_t cell, uint32_t in_dim, uint32_t hidden_size, uint32_t num_layers,
                          uint32_t mlp_hidden, uint32_t out_dim, rl_mlp **out) {
  return guarded(e, [&] {
    if (cell != RL_CELL_GRU && cell != RL_CELL_LSTM) throw RlError(RL_ERR_BUILD_AGENT, "unknown recurrent cell");
    seq_module_create(e, cell == RL_CELL_GRU ? RL_MODULE_GRU_MLP : RL_MODULE_LSTM_MLP, in_dim, hidden_size, num_layers,
                      mlp_hidden, out_dim, out);
  });
}

int32_t rl_rnn_mlp_create_config(rl_engine *e, int32_t cell, uint32_t in_dim, uint32_t hidden_size, uint32_t num_layers,
                                 int32_t bias, uint32_t mlp_hidden, uint32_t out_dim, rl_mlp **out) {
  return guarded(e, [&] {
    if (cell != RL_CELL_GRU && cell != RL_CELL_LSTM) throw RlError(RL_ERR_BUILD_AGENT, "unknown recurrent cell");
    seq_module_create(e, cell == RL_CELL_GRU ? RL_MODULE_GRU_MLP : RL_MODULE_LSTM_MLP, in_dim, hidden_size, num_layers,
                      mlp_hidden, out_dim, out, bias != 0);
  });
}

// The engine's initialisation stream and TensorBuilder::build on it (initializers.rs:8-64,67-83,152-176,328-364):
// ChaCha8(seed), stream 0; one Standard f32 per uniform element, value = (2u - 1) * lim in f32; normal elements by
// Box-Muller on consecutive draw pairs (an odd count leaves the pair's second value unused); orthogonal: QR of the normal
// matrix by modified Gram-Schmidt applied twice in f64 — positive diagonal of R, so the sign fold of init_orthogonal is the
// identity.  Zeros / Constant draw nothing.  (libtorch's RNG is unseeded in the reference: the stream is engine-defined.)
struct InitStream {
  uint32_t key[8];
  uint32_t words[16];
  uint64_t widx = 0;
  explicit InitStream(uint64_t seed) { rl_seed_from_u64(seed, key); }
  float next_f32() {
    if ((widx & 15) == 0) rl_chacha_block(key, widx >> 4, 0, 4, words);
    float u = rl_u32_to_unit_f32(words[widx & 15]);
    widx += 1;
    return u;
  }
  void normals(size_t count, std::vector<double> &z) {
    z.assign(count, 0.0);
    const double two_pi = 6.283185307179586;
    for (size_t i = 0; i < count; i += 2) {
      const double u1 = (double)next_f32(), u2 = (double)next_f32();
      double rho = std::sqrt(-2.0 * std::log(1.0 - u1)), sn, cs;
      rl_sincos(two_pi * u2, &sn, &cs);
      z[i] = rho * cs;
      if (i + 1 < count) z[i + 1] = rho * sn;
    }
  }
  static double variance(const rl_initializer &it, double fan_in, double fan_out) {
    switch (it.scale) {
      case RL_SCALE_CONSTANT: return it.value;
      case RL_SCALE_FAN_IN: return 1.0 / fan_in;
      case RL_SCALE_FAN_OUT: return 1.0 / fan_out;
      default: return 2.0 / (fan_in + fan_out);
    }
  }
  // one tensor of `rows` x `cols` elements (a 1-D tensor: rows = its length, cols = 1; fan_out = shape[0] either way,
  // calculate_fan_in_and_fan_out, initializers.rs:90-103)
  void fill(const rl_initializer &it, float *dst, uint64_t rows, uint64_t cols, double fan_in) {
    const size_t count = (size_t)rows * cols;
    const double fan_out = (double)rows;
    if (it.kind == RL_INIT_ZEROS) {
      for (size_t i = 0; i < count; ++i) dst[i] = 0.0f;
    } else if (it.kind == RL_INIT_CONSTANT) {
      for (size_t i = 0; i < count; ++i) dst[i] = (float)it.value;
    } else if (it.kind == RL_INIT_UNIFORM) {
      const float lim = (float)std::sqrt(3.0 * variance(it, fan_in, fan_out));
      for (size_t i = 0; i < count; ++i) {
        float t = 2.0f * next_f32();
        t = t - 1.0f;
        dst[i] = t * lim;
      }
    } else if (it.kind == RL_INIT_NORMAL) {
      const double sd = std::sqrt(variance(it, fan_in, fan_out));
      std::vector<double> z;
      normals(count, z);
      for (size_t i = 0; i < count; ++i) dst[i] = (float)(sd * z[i]);
    } else {  // orthogonal: tall = the [rows, cols] matrix, transposed when it is wide
      std::vector<double> z;
      normals(count, z);
      const bool wide = rows < cols;
      const uint64_t R = wide ? cols : rows, Cn = wide ? rows : cols;  // tall matrix R x Cn, columns orthonormalised
      std::vector<double> a((size_t)R * Cn);                           // column-major: a[c * R + r]
      for (uint64_t r = 0; r < rows; ++r)
        for (uint64_t c = 0; c < cols; ++c) {
          const double v = z[(size_t)r * cols + c];
          if (wide) a[(size_t)r * R + c] = v;  // tall[c][r] = flat[r][c]
          else a[(size_t)c * R + r] = v;
        }
      for (uint64_t c = 0; c < Cn; ++c) {
        double *v = a.data() + (size_t)c * R;
        for (int pass = 0; pass < 2; ++pass)
          for (uint64_t q = 0; q < c; ++q) {
            const double *w = a.data() + (size_t)q * R;
            double dot = 0.0;
            for (uint64_t r = 0; r < R; ++r) dot += w[r] * v[r];
            for (uint64_t r = 0; r < R; ++r) v[r] -= dot * w[r];
          }
        double nrm = 0.0;
        for (uint64_t r = 0; r < R; ++r) nrm += v[r] * v[r];
        nrm = std::sqrt(nrm);
        for (uint64_t r = 0; r < R; ++r) v[r] /= nrm;
      }
      for (uint64_t r = 0; r < rows; ++r)
        for (uint64_t c = 0; c < cols; ++c) dst[(size_t)r * cols + c] = (float)(wide ? a[(size_t)r * R + c] : a[(size_t)c * R + r]);
    }
  }
};

// RnnWeights::new (seq/rnn/mod.rs:223-257: per layer W_ih [G H, layer input] from input_weights_init, W_hh [G H, H] from
// hidden_weights_init, b_ih and b_hh [G H] from bias_init — 1-D tensors: fan_in 1, fan_out G H) + the MLP's Linear::new
// layers (ff/linear.rs:54-68: kernel and bias with fan_in = in + 1), drawn in flat order from one stream.
struct RnnInits {
  rl_initializer input, hidden, bias, mlp_kernel, mlp_bias;
};
static RnnInits rnn_default_inits() {  // RnnBaseConfig::default (seq/rnn/mod.rs:36-45), LinearConfig::default
  const rl_initializer glorot{RL_INIT_UNIFORM, RL_SCALE_FAN_AVG, 0.0}, ortho{RL_INIT_ORTHOGONAL, RL_SCALE_FAN_AVG, 0.0},
      zeros{RL_INIT_ZEROS, RL_SCALE_FAN_AVG, 0.0};
  return RnnInits{glorot, ortho, zeros, glorot, glorot};
}
static void rnn_mlp_init_host(const rl_mlp *m, uint64_t seed, const RnnInits &in, std::vector<float> &h) {
  const uint64_t H = m->gru_hidden, D = m->in_dim, H2 = m->hidden, A = m->out_dim, R = rl_module_gates(m->kind) * H;
  h.assign(m->P, 0.0f);
  InitStream st(seed);
  size_t k = 0;
  for (uint32_t layer = 0; layer < m->rnn_layers; ++layer) {
    const uint64_t K = layer == 0 ? D : H;
    st.fill(in.input, h.data() + k, R, K, (double)K);
    k += R * K;
    st.fill(in.hidden, h.data() + k, R, H, (double)H);
    k += R * H;
    if (m->has_bias) {  // (RnnBaseConfig::bias_init = None: no tensors, no draws — seq/rnn/mod.rs:246-251)
      st.fill(in.bias, h.data() + k, R, 1, 1.0);
      k += R;
      st.fill(in.bias, h.data() + k, R, 1, 1.0);
      k += R;
    }
  }
  const uint64_t dims[2][2] = {{H, H2}, {H2, A}};
  for (int l = 0; l < 2; ++l) {
    const uint64_t fin = dims[l][0], fout = dims[l][1];
    st.fill(in.mlp_kernel, h.data() + k, fout, fin, (double)(fin + 1));
    k += fin * fout;
    st.fill(in.mlp_bias, h.data() + k, fout, 1, (double)(fin + 1));
    k += fout;
  }
}

int32_t rl_mlp_destroy(rl_mlp *m) {
  if (!m) return RL_OK;
  (void)hipSetDevice(m->eng->device);
  (void)hipStreamSynchronize(m->eng->stream);
  (void)hipStreamSynchronize(m->eng->aux_stream);  // (a critic chain left in flight by rl_actor_critic_update_begin)
  dfree(m->d_params);
  dfree(m->d_wimg);
  if (m->exec) {
    dfree(m->exec->d_params);
    delete m->exec;
  }
  dfree(m->x_tmp);
  dfree(m->x_tan);
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

// Linear::new for every layer of a feed-forward module (reference src/torch/modules/ff/linear.rs:54-68) with the given
// initializers, drawn from the engine's stream (InitStream above) in flat parameter order (kernel then bias per layer)
static void mlp_init_host(rl_mlp *m, uint64_t seed, const rl_initializer &kinit, const rl_initializer &binit) {
  std::vector<float> h(m->P);
  InitStream st(seed);
  auto fill = [&](const rl_initializer &it, float *dst, uint64_t rows, uint64_t cols, double fan_in) {
    st.fill(it, dst, rows, cols, fan_in);
  };
  size_t k = 0;
  for (uint32_t l = 0; l < m->n_layers(); ++l) {
    const uint64_t in = m->fan_in(l), out = m->fan_out(l);
    fill(kinit, h.data() + k, out, in, (double)(in + 1));  // kernel [out][in]
    k += (size_t)in * out;
    if (m->has_bias) {  // (LinearConfig::bias_init = None: no tensor, no draws — linear.rs:54-68)
      fill(binit, h.data() + k, out, 1, (double)(in + 1));  // bias [out]
      k += out;
    }
  }
  h2d(m->eng, m->d_params, h.data(), m->P * sizeof(float));
  wimg_invalidate(m);  // (the parameters changed under the module's weight image, bf16_tile.hpp)
}

int32_t rl_mlp_init(rl_mlp *m, uint64_t seed) {
  return guarded(m ? m->eng : nullptr, [&] {
    RL_REQUIRE(m, "mlp is NULL");
    if (rl_module_is_recurrent(m->kind)) {
      std::vector<float> hp;
      rnn_mlp_init_host(m, seed, rnn_default_inits(), hp);
      h2d(m->eng, m->d_params, hp.data(), m->P * sizeof(float));
      wimg_invalidate(m);  // (the parameters changed under the module's weight image, bf16_tile.hpp)
      return;
    }
    const rl_initializer glorot{RL_INIT_UNIFORM, RL_SCALE_FAN_AVG, 0.0};
    mlp_init_host(m, seed, glorot, glorot);
  });
}

int32_t rl_mlp_init_with(rl_mlp *m, uint64_t seed, const rl_initializer *kernel_init, const rl_initializer *bias_init) {
  return guarded(m ? m->eng : nullptr, [&] {
    RL_REQUIRE(m && kernel_init, "NULL argument");
    if (rl_module_is_recurrent(m->kind))
      throw RlError(RL_ERR_UNSUPPORTED, "rl_mlp_init_with: feed-forward modules only (the recurrent chains use "
                                        "RnnBaseConfig::default's initializers)");
    // LinearConfig::bias_init = None <=> the module was built without bias vectors (rl_mlp_create_config, bias = 0)
    if ((bias_init == nullptr) != !m->has_bias)
      throw RlError(RL_ERR_INVALID_ARGUMENT, m->has_bias ? "this module has bias vectors: bias_init must be given (build it "
                                                           "with rl_mlp_create_config(..., bias = 0) for bias_init = None)"
                                                         : "this module has no bias vectors: bias_init must be NULL");
    for (const rl_initializer *i : {kernel_init, bias_init}) {
      if (i == nullptr) continue;
      RL_REQUIRE(i->kind >= RL_INIT_ZEROS && i->kind <= RL_INIT_ORTHOGONAL, "unknown initializer kind");
      if (i->kind == RL_INIT_UNIFORM || i->kind == RL_INIT_NORMAL) {
        RL_REQUIRE(i->scale >= RL_SCALE_CONSTANT && i->scale <= RL_SCALE_FAN_AVG, "unknown variance scale");
        RL_REQUIRE(i->scale != RL_SCALE_CONSTANT || i->value >= 0.0, "a variance must not be negative");
      }
    }
    if (bias_init && bias_init->kind == RL_INIT_ORTHOGONAL)  // init_orthogonal asserts shape.len() >= 2 (initializers.rs:331-334)
      throw RlError(RL_ERR_INVALID_ARGUMENT, "tensor for orthogonal init must be at least 2D: not a bias initializer");
    mlp_init_host(m, seed, *kernel_init, bias_init ? *bias_init : *kernel_init);
  });
}

int32_t rl_rnn_mlp_init_with(rl_mlp *m, uint64_t seed, const rl_initializer *input_weights_init,
                             const rl_initializer *hidden_weights_init, const rl_initializer *bias_init,
                             const rl_initializer *mlp_kernel_init, const rl_initializer *mlp_bias_init) {
  return guarded(m ? m->eng : nullptr, [&] {
    RL_REQUIRE(m, "module is NULL");
    if (!rl_module_is_recurrent(m->kind))
      throw RlError(RL_ERR_INVALID_ARGUMENT, "rl_rnn_mlp_init_with: recurrent chains only (rl_mlp_init_with for MLPs)");
    // RnnBaseConfig::bias_init = None <=> the module was built without recurrent bias vectors (rl_rnn_mlp_create_config)
    if ((bias_init == nullptr) != !m->has_bias)
      throw RlError(RL_ERR_INVALID_ARGUMENT, m->has_bias ? "this module has recurrent bias vectors: bias_init must be given "
                                                           "(rl_rnn_mlp_create_config(..., bias = 0, ...) builds it without)"
                                                         : "this module has no recurrent bias vectors: bias_init must be NULL");
    if (mlp_bias_init == nullptr)
      throw RlError(RL_ERR_UNSUPPORTED, "the chain's MLP is built with bias vectors: its bias_init = None is not");
    RL_REQUIRE(input_weights_init && hidden_weights_init && mlp_kernel_init, "NULL initializer");
    for (const rl_initializer *i : {input_weights_init, hidden_weights_init, bias_init, mlp_kernel_init, mlp_bias_init}) {
      if (i == nullptr) continue;
      RL_REQUIRE(i->kind >= RL_INIT_ZEROS && i->kind <= RL_INIT_ORTHOGONAL, "unknown initializer kind");
      if (i->kind == RL_INIT_UNIFORM || i->kind == RL_INIT_NORMAL) {
        RL_REQUIRE(i->scale >= RL_SCALE_CONSTANT && i->scale <= RL_SCALE_FAN_AVG, "unknown variance scale");
        RL_REQUIRE(i->scale != RL_SCALE_CONSTANT || i->value >= 0.0, "a variance must not be negative");
      }
    }
    if ((bias_init && bias_init->kind == RL_INIT_ORTHOGONAL) || mlp_bias_init->kind == RL_INIT_ORTHOGONAL)  // initializers.rs:331-334
      throw RlError(RL_ERR_INVALID_ARGUMENT, "tensor for orthogonal init must be at least 2D: not a bias initializer");
    std::vector<float> hp;
    rnn_mlp_init_host(m, seed, RnnInits{*input_weights_init, *hidden_weights_init, bias_init ? *bias_init : *mlp_bias_init,
                                        *mlp_kernel_init, *mlp_bias_init},
                      hp);
    h2d(m->eng, m->d_params, hp.data(), m->P * sizeof(float));
    wimg_invalidate(m);  // (the parameters changed under the module's weight image, bf16_tile.hpp)
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
    wimg_invalidate(m);  // (the parameters changed under the module's weight image, bf16_tile.hpp)
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
      if (m->general) {
        rl_traj scratch{};  // (only the workspace of the per-layer kernels is used)
        scratch.eng = e;
        try {
          launch_gen_forward(&scratch, m, d_in, n_rows, n_rows, d_out);
          sync(e);
        } catch (...) {
          gen_free(&scratch);
          throw;
        }
        gen_free(&scratch);
      } else {
        launch_mlp_forward_host_rows(m, d_in, n_rows, d_out);
      }
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
    case RL_TRAJ_TARGETS:
      RL_REQUIRE(t->last_targets != nullptr, "no value targets yet: they are written by rl_values_opt_update");
      *ptr = const_cast<float *>(t->last_targets);
      *bytes = T * n * 4;
      break;
    default: throw RlError(RL_ERR_INVALID_ARGUMENT, "unknown trajectory field");
  }
}

// launch geometry of the update kernels for B samples
void traj_plan(rl_traj *t, uint64_t B) {
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
}

// `resizable`: the sample count changes between launches (DQN minibatches): slabs are sized for the largest grid
// any B <= n_lanes * horizon can plan
rl_traj *traj_alloc(rl_engine *e, uint64_t n_lanes, uint64_t horizon, uint32_t obs_dim, bool resizable) {
  RL_REQUIRE(n_lanes > 0 && n_lanes < (1ull << 31), "bad n_lanes");
  RL_REQUIRE(horizon > 0 && horizon < (1ull << 20), "bad horizon");
  RL_REQUIRE(obs_dim >= 1 && obs_dim <= RL_TRAJ_MAX_OBS_DIM, "obs_dim must be in 1..8");
  RL_REQUIRE(n_lanes * horizon < (1ull << 32), "T * n must fit 32 bits");
  RL_HIP_CHECK(hipSetDevice(e->device));
  std::unique_ptr<rl_traj> t(new rl_traj());
  t->eng = e;
  t->d.n = (uint32_t)n_lanes;
  t->d.T = (uint32_t)horizon;
  t->d.D = obs_dim;
  // (at least five observation planes whatever the logical width: the fused recurrent kernels are built for five features
  // and read the planes past the module's in_dim as zeros)
  uint64_t n = n_lanes, T = horizon, D = obs_dim > 5 ? obs_dim : 5;
  t->d.obs = dalloc<float>(D * (T + 1) * n);
  RL_HIP_CHECK(hipMemsetAsync(t->d.obs, 0, D * (T + 1) * n * 4, e->stream));
  t->d.action = dalloc<uint8_t>(T * n);
  t->d.reward = dalloc<float>(T * n);
  t->d.flag = dalloc<uint8_t>(T * n);
  t->d.term_obs = dalloc<float>(D * T * n);
  t->d.values = dalloc<float>((T + 1) * n);
  t->d.adv = dalloc<float>(T * n);
  t->d.rtg = dalloc<float>(T * n);
  t->d.tgt = t->d.rtg;  // the critic regresses on the returns unless rl_values_opt_update selects other targets
  t->d.range = dalloc<uint32_t>(RL_RANGE_ALLOC_WORDS);
  RL_HIP_CHECK(hipMemsetAsync(t->d.range, 0, RL_RANGE_ALLOC_WORDS * sizeof(uint32_t), e->stream));
  {
    void *hp = nullptr;
    RL_HIP_CHECK(hipHostMalloc(&hp, 64, hipHostMallocMapped));
    static_cast<volatile uint32_t *>(hp)[RL_GUARD_POLICY] = 0u;
    static_cast<volatile uint32_t *>(hp)[RL_GUARD_CRITIC] = 0u;
    void *dp = nullptr;
    RL_HIP_CHECK(hipHostGetDevicePointer(&dp, hp, 0));
    t->h_range_err = static_cast<uint32_t *>(hp);
    t->d.range_err = static_cast<uint32_t *>(dp);
  }
  t->lp0 = dalloc<float>(2 * n * T);
  t->dz = dalloc<float>(2 * n * T);
  t->Pmax = 128 * 5 + 128 + 2 * 128 + 2;
  traj_plan(t.get(), n * T);
  uint32_t rows = t->nbA;
  if (t->nbV2 > rows) rows = t->nbV2;
  if (t->nbC > rows) rows = t->nbC;
  uint32_t rowsB = t->nbB > rows ? t->nbB : rows;
  if (resizable) {
    uint32_t cap = 8u * (uint32_t)e->prop.multiProcessorCount;
    if (cap < 2048) cap = 2048;
    rows = rowsB = cap;
  }
  t->slabA = dalloc<double>((size_t)rows * t->Pmax);
  t->slabB = dalloc<double>((size_t)rowsB * 4);
  t->cap_slabA = (uint64_t)rows * t->Pmax;
  t->cap_slabB = (uint64_t)rowsB * 4;
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
  (void)hipStreamSynchronize(t->eng->main_stream);
  (void)hipStreamSynchronize(t->eng->aux_stream);
  if (t->eng->pending.traj == t) t->eng->pending.active = false;  // (its pending update dies with it)
  void *ptrs[] = {t->d.obs, t->d.action, t->d.reward, t->d.flag, t->d.term_obs, t->d.values, t->d.adv, t->d.rtg,
                  t->lp0, t->dz, t->slabA, t->slabB, t->vec, t->cg_x, t->cg_r, t->cg_p, t->prev_params, t->descent,
                  t->losses, t->trpo, t->td, t->aux_slabA, t->aux_slabB, t->aux_vec, t->d.range};
  for (void *p : ptrs) dfree(p);
  if (t->h_range_err) (void)hipHostFree(t->h_range_err);
  seq_free(t);
  gen_free(t);
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
    if (field == RL_TRAJ_OBS) t->range_valid = t->range_reset = false;  // (the range guard reads the planes' magnitudes)
    t->rtg_scan_valid = false;  // (whatever was written, the return plane is no longer known to match the rewards)
  });
}

// ---------------------------------------------------------------- recurrent workspace
// the P-sized vectors of the update workspace grow with the module
void traj_ensure_pvec(rl_traj *t, uint64_t P) {
  if (t->Pmax >= P) return;
  for (float **p : {&t->vec, &t->cg_x, &t->cg_r, &t->cg_p, &t->prev_params, &t->descent}) {
    dfree(*p);
    *p = nullptr;
  }
  t->Pmax = (uint32_t)P;
  t->vec = dalloc<float>(t->Pmax + 4);
  t->cg_x = dalloc<float>(t->Pmax);
  t->cg_r = dalloc<float>(t->Pmax);
  t->cg_p = dalloc<float>(t->Pmax);
  t->prev_params = dalloc<float>(t->Pmax);
  t->descent = dalloc<float>(t->Pmax);
}

void seq_ensure(rl_traj *t, const rl_mlp *mod, bool training) {
  RL_REQUIRE(rl_module_is_recurrent(mod->kind), "not a recurrent module");
  if (mod->lane_kernels()) return stack_ensure(t, mod, training);  // lane-per-thread kernels: no tile or width condition
  RL_REQUIRE(t->d.n % 32 == 0, "the recurrent kernels work on tiles of 32 lanes: n_lanes must be a multiple of 32");
  RL_REQUIRE(t->d.D == 5 && mod->in_dim == 5, "recurrent path: built for 5 observation features");
  SeqDev &q = t->seq;
  uint64_t n = t->d.n, T = t->d.T;
  q.tiles = (uint32_t)(n / 32);  // (the output planes may already exist: the general-MLP path shares them)
  if (q.out == nullptr) {
    q.out = dalloc<float>(2 * T * n);
    q.succ = dalloc<float>(2 * T * n);
  }
  if (training && q.act == nullptr) {
    uint64_t blocks = T * q.tiles;
    q.act = dalloc<float>(blocks * RL_SEQ_ACT_ARRAYS * 128 * 32);
    q.dpre = dalloc<float>(blocks * RL_SEQ_DPRE_ARRAYS * 128 * 32);
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
    // + the rows of the head kernel (head columns) and of the backward recurrence (one per tile, input-side columns)
    q.wg_slab = dalloc<float>((size_t)(q.chunks + (q.tiles > RL_SEQ_HEAD_ROWS ? q.tiles : RL_SEQ_HEAD_ROWS)) * mod->P);
    q.P = mod->P;
    traj_ensure_pvec(t, mod->P);
  }
}

static void seq_free(rl_traj *t) {
  SeqDev &q = t->seq;
  for (float *p : {q.act, q.dpre, q.out, q.succ, q.wg_slab}) dfree(p);
  stack_free(t);
  q = SeqDev{};
}

int32_t rl_seq_forward(rl_mlp *mod, rl_traj *traj, float *out, float *succ_out) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    RL_REQUIRE(mod && traj && out, "NULL argument");
    RL_REQUIRE(mod->eng == traj->eng, "handles belong to different engines");
    RL_REQUIRE(rl_module_is_recurrent(mod->kind), "not a recurrent module");
    {
      SeqScope sc(traj, mod);
      seq_ensure(traj, sc.x, false);
      launch_gru_seq_forward(traj, sc.x, traj->seq.out, succ_out ? traj->seq.succ : nullptr, nullptr);
    }
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
    // A critic chain left in flight by rl_actor_critic_update_begin reads its own trajectory and the critic: a rollout
    // into ANOTHER trajectory touches neither (it reads the policy the TRPO chain has already finished with) and goes
    // ahead beside it on the main stream; anything else waits for the chain like every other call.
    if (traj == env->eng->pending.traj || policy == env->eng->pending.critic) engine_settle(env->eng);
    traj->range_valid = traj->range_reset = false;  // new observations: the range guard has them measured again
    traj->rtg_scan_valid = false;                   // new rewards: the return plane belongs to the old ones
    RL_REQUIRE(traj->d.n == env->cfg.n_lanes && traj->d.D == env->D, "trajectory shape does not match the env");
    RL_REQUIRE(policy->in_dim == env->D && policy->out_dim == env->A, "policy shape does not match the env");
    if (policy->general) {  // any hidden_sizes: one launch sequence per step, either env family (advances t_global)
      launch_gen_rollout(env, policy, traj);
      return;
    }
    if (rl_module_is_recurrent(policy->kind) && policy->lane_kernels()) {  // a launch sequence per step
      seq_ensure(traj, policy, false);
      launch_stack_rollout(env, policy, traj);  // (advances t_global)
      return;
    }
    if (rl_module_is_recurrent(policy->kind)) {
      SeqScope sc(traj, policy);  // (in_dim == env->D == 5 here: the rollout kernels compute five features)
      RL_REQUIRE(env->D == 5, "recurrent rollouts need an env with five observation features");
      seq_ensure(traj, sc.x, false);
      launch_rollout_gru(env, sc.x, traj);
    } else if (env->kind != RL_ENV_CARTPOLE) {
      launch_rollout_chain_mlp(env, policy, traj);
    } else {
      launch_rollout(env, policy, traj);
    }
    env->t_global += traj->d.T;
  }, false);
}

int32_t rl_gae(rl_traj *traj, const rl_mlp *critic, float gamma, float lambda) {
  return guarded(traj ? traj->eng : nullptr, [&] {
    RL_REQUIRE(traj && critic, "NULL argument");
    RL_REQUIRE(critic->in_dim == traj->d.D && critic->out_dim == 1, "critic shape does not match the trajectory");
    if (rl_module_is_recurrent(critic->kind)) {
      SeqScope sc(traj, critic);
      seq_ensure(traj, sc.x, false);
      launch_gru_seq_forward(traj, sc.x, traj->seq.out, traj->seq.succ, nullptr);
      launch_seq_gae(traj, gamma, lambda);
      return;
    }
    if (critic->general) {  // values and successor values as arrays, then the array-fed scan of the recurrent path
      launch_gen_values(traj, critic);
      launch_seq_gae(traj, gamma, lambda);
      return;
    }
    launch_values(traj, critic);
    launch_gae(traj, critic, gamma, lambda);
  });
}

}  // extern "C"
