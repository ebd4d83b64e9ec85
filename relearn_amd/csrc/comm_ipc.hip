// comm_ipc.hip — a single-launch all-reduce for the <= 8 KiB vectors of the update (gradients, Fisher-vector products,
// line-search scalars) over peer-mapped mailboxes: every rank owns a mailbox in its own HBM, maps the mailboxes of its
// peers (hipIpcOpenMemHandle: the ranks are processes of one node; over xGMI the mapping is a peer-to-peer window), and
// an all-reduce is ONE kernel per rank —
//   publish   my 64-column chunk of the vector goes into slot `seq & 1` of MY row in EVERY rank's mailbox, every element
//             as one 64-bit word {sequence number, float bits} (relaxed system-scope atomic stores into the peer windows);
//   gather    every lane polls ITS column in the n_ranks rows of its own mailbox until the word carries this collective's
//             sequence number (bounded spin), and adds the values in RANK ORDER — every rank adds the same numbers in the
//             same order, so the replicas stay bit-identical without a broadcast.
// No flag and no fence: a word is complete or absent.  Two slots are enough: a rank cannot start collective k + 2 (which
// reuses the slot of k) before it has read every peer's words of k + 1, and a peer writes those only after it has
// finished reading k.
// This replaces the `Vec<buffer>` hand-off of the reference's threads (src/simulation/train.rs:180) for the multi-GPU
// configuration; RCCL (abi.hip) stays the default transport — its small-message latency is what this path avoids.
// Every wait is bounded in wall-clock time (30 s by default, RELEARN_IPC_TIMEOUT_MS; rank skew — a first-launch code
// object load, a host stall, a profiler — is far below that).  A peer that never arrives sets the engine's STICKY error
// word; the exchange is FAIL-STOP: the wave that timed out returns before it stores anything, every later collective
// fails fast without publishing, every kernel that writes parameters or optimiser state from a reduced vector tests the
// word first (k_reduce_adam's fused exchange; k_adam_step, k_ls_set_params and k_ls_finalize behind the stand-alone
// k_ipc_allreduce, which leaves LOCAL sums in the vector when it fails — kernels_update.hip: a TRPO update ends on the
// parameters it started from), and the host raises RL_ERR_COMM at its next synchronising call.  The decision is taken
// per 64-element chunk: when a peer never publishes, nothing is stepped anywhere (tests/test_gpu_multirank.py, both
// update kinds); a peer that arrives within microseconds of the bound can pass some chunks of a fused reduce + Adam
// launch and fail others, and sees all of this rank's words itself — after RL_ERR_COMM the replicas are not defined
// and the job must stop (or restore a checkpoint): the word stays set so that nothing continues by accident.
// EXPERIMENTAL: the protocol has run between processes sharing one GPU (IPC handles of one device map like peer windows)
// and never across xGMI; rl_comm_init_ipc checks what it can for mailboxes on other devices (peer access, native
// atomics), the self-test hammers both slots and every chunk with random payloads, but 8-byte store atomicity and
// visibility through a real peer window are unverified on this pool.  RCCL is the default transport.
#include <cstdlib>
#include <map>
#include <mutex>

#include "abi_internal.hpp"
#include "comm_ipc.hpp"

namespace {

// Mailboxes of THIS process by their handle bytes.  A host that drives several engines from one process (one thread per
// GPU — how a Rust host would replace train_parallel's threads, src/simulation/train.rs:98-151) hands every engine the
// same list of handles; an engine recognises the ones made in its own address space and takes their mailboxes by
// address (hipIpcOpenMemHandle refuses a handle of the calling process), enabling peer access when the mailbox lives on
// another device.
struct LocalBox {
  float *box;
  int device;
};
std::mutex g_local_mu;
std::map<std::string, LocalBox> g_local_boxes;

// one workgroup (one wave) per 64-column chunk
__global__ void __launch_bounds__(IPC_CHUNK) k_ipc_allreduce(float *__restrict__ vec, uint32_t count, IpcPeers peers) {
  const uint32_t chunk = blockIdx.x, p = chunk * IPC_CHUNK + threadIdx.x;
  const float mine = p < count ? vec[p] : 0.0f;
  float s;
  if (!ipc_exchange_chunk(peers, chunk, threadIdx.x, mine, s)) return;  // failed: vec keeps the local values
  if (p < count) vec[p] = s;
}

}  // namespace

bool ipc_allreduce_fits(const rl_engine *e, size_t count) { return e->ipc_box != nullptr && count <= IPC_CAP; }

void ipc_allreduce(rl_engine *e, float *d_buf, size_t count) {
  const IpcPeers peers = ipc_peers_next(e);
  hipLaunchKernelGGL(k_ipc_allreduce, dim3((unsigned)((count + IPC_CHUNK - 1) / IPC_CHUNK)), dim3(IPC_CHUNK), 0,
                     e->stream, d_buf, (uint32_t)count, peers);
}

void ipc_check(rl_engine *e) {
  if (e->ipc_box == nullptr) return;
  int32_t err = 0;
  d2h(e, &err, e->ipc_err, sizeof(err));
  if (err != 0)
    throw RlError(RL_ERR_COMM, "peer-mailbox all-reduce: rank " + std::to_string(err - 1) + " never arrived");
}

void ipc_teardown(rl_engine *e) {
  for (int r = 0; r < RL_IPC_MAX_RANKS; ++r) {
    if (e->ipc_peer[r] != nullptr && r != e->ipc_rank_of_box && !e->ipc_peer_local[r])
      (void)hipIpcCloseMemHandle(e->ipc_peer[r]);
    e->ipc_peer[r] = nullptr;
    e->ipc_peer_local[r] = false;
  }
  if (e->ipc_box) {
    std::lock_guard<std::mutex> lk(g_local_mu);
    for (auto it = g_local_boxes.begin(); it != g_local_boxes.end();)
      it = it->second.box == e->ipc_box ? g_local_boxes.erase(it) : std::next(it);
  }
  if (e->ipc_box) (void)hipFree(e->ipc_box);
  if (e->ipc_err) (void)hipFree(e->ipc_err);
  e->ipc_box = nullptr;
  e->ipc_err = nullptr;
  e->ipc_seq[0] = e->ipc_seq[1] = 0;
}

extern "C" {

int32_t rl_comm_ipc_handle(rl_engine *e, int32_t n_ranks, uint8_t handle_out[64]) {
  return guarded(e, [&] {
    RL_REQUIRE(e && handle_out, "NULL argument");
    RL_REQUIRE(n_ranks >= 1 && n_ranks <= RL_IPC_MAX_RANKS, "n_ranks out of range for the mailbox collective");
    RL_REQUIRE(!e->has_collective(), "communicator already initialised");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    RL_HIP_CHECK(hipSetDevice(e->device));
    if (e->ipc_box == nullptr) try {
      const size_t bytes = ipc_box_bytes((uint32_t)n_ranks);
      void *p = nullptr;
      // fine-grained: peers' stores and this device's loads of the mailbox are coherent without cache maintenance
      RL_HIP_CHECK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained));
      e->ipc_box = (float *)p;
      e->ipc_box_ranks = n_ranks;
      RL_HIP_CHECK(hipMemsetAsync(p, 0, bytes, e->stream));
      e->ipc_err = dalloc<int32_t>(1);
      RL_HIP_CHECK(hipMemsetAsync(e->ipc_err, 0, sizeof(int32_t), e->stream));
      sync(e);
      uint64_t ms = IPC_TIMEOUT_MS_DEFAULT;
      if (const char *env = std::getenv("RELEARN_IPC_TIMEOUT_MS")) {
        const long long v = std::atoll(env);
        if (v > 0) ms = (uint64_t)v;
      }
      e->ipc_timeout_ticks = ms * 100000ull;  // s_memrealtime counts at 100 MHz
    } catch (...) {  // all or nothing: a retry must not find a mailbox without its error word
      ipc_teardown(e);
      throw;
    }
    RL_REQUIRE(e->ipc_box_ranks == n_ranks, "the mailbox was sized for another number of ranks");
    hipIpcMemHandle_t h;
    RL_HIP_CHECK(hipIpcGetMemHandle(&h, e->ipc_box));
    std::memcpy(handle_out, &h, 64);
    std::lock_guard<std::mutex> lk(g_local_mu);
    g_local_boxes[std::string((const char *)handle_out, 64)] = LocalBox{e->ipc_box, e->device};
  });
}

int32_t rl_comm_init_ipc(rl_engine *e, int32_t rank, int32_t n_ranks, const uint8_t *handles) {
  return guarded(e, [&] {
    RL_REQUIRE(e && handles, "NULL argument");
    RL_REQUIRE(n_ranks >= 1 && n_ranks <= RL_IPC_MAX_RANKS && rank >= 0 && rank < n_ranks, "bad rank / n_ranks");
    RL_REQUIRE(!e->has_collective(), "communicator already initialised");
    RL_REQUIRE(e->ipc_box != nullptr && e->ipc_box_ranks == n_ranks, "call rl_comm_ipc_handle first (same n_ranks)");
    RL_HIP_CHECK(hipSetDevice(e->device));
    try {
      for (int r = 0; r < n_ranks; ++r) {
        if (r == rank) {
          e->ipc_peer[r] = e->ipc_box;
          continue;
        }
        hipIpcMemHandle_t h;
        std::memcpy(&h, handles + (size_t)r * 64, 64);
        void *p = nullptr;
        LocalBox local{nullptr, -1};
        {
          std::lock_guard<std::mutex> lk(g_local_mu);
          auto it = g_local_boxes.find(std::string((const char *)handles + (size_t)r * 64, 64));
          if (it != g_local_boxes.end()) local = it->second;
        }
        if (local.box != nullptr) {  // an engine of this process: no IPC mapping, its mailbox by address
          if (local.device != e->device) {
            const hipError_t pe = hipDeviceEnablePeerAccess(local.device, 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
              (void)hipGetLastError();
              throw RlError(RL_ERR_COMM, "peer-mailbox transport: device " + std::to_string(e->device) +
                                             " cannot enable peer access to device " + std::to_string(local.device));
            }
            (void)hipGetLastError();
          }
          p = local.box;
          e->ipc_peer_local[r] = true;
        } else {
          RL_HIP_CHECK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        }
        e->ipc_peer[r] = (float *)p;
        // a mailbox on ANOTHER device is reached through a peer window: the protocol needs peer access and 64-bit
        // stores that land as one unit (native atomics over the link); refuse the transport otherwise
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, p) == hipSuccess && attr.device != e->device) {
          int can = 0, atomics = 0;
          RL_HIP_CHECK(hipDeviceCanAccessPeer(&can, e->device, attr.device));
          if (can) (void)hipDeviceGetP2PAttribute(&atomics, hipDevP2PAttrNativeAtomicSupported, e->device, attr.device);
          if (!can || !atomics)
            throw RlError(RL_ERR_COMM, "peer-mailbox transport: device " + std::to_string(e->device) +
                                           (can ? " has no native atomics to device " : " cannot access device ") +
                                           std::to_string(attr.device) + " (use RCCL)");
        }
      }
    } catch (...) {
      for (int r = 0; r < n_ranks; ++r) {
        if (r != rank && e->ipc_peer[r] && !e->ipc_peer_local[r]) (void)hipIpcCloseMemHandle(e->ipc_peer[r]);
        e->ipc_peer[r] = nullptr;
        e->ipc_peer_local[r] = false;
      }
      throw;
    }
    e->ipc_rank_of_box = rank;
    e->ipc_active = true;
    e->rank = rank;
    e->n_ranks = n_ranks;
    try {
      comm_agree(e);  // (also the first exchange with every peer: a mailbox that cannot be reached fails here)
    } catch (...) {
      e->ipc_active = false;
      e->rank = 0;
      e->n_ranks = 1;
      for (int r = 0; r < n_ranks; ++r) {
        if (r != rank && e->ipc_peer[r] && !e->ipc_peer_local[r]) (void)hipIpcCloseMemHandle(e->ipc_peer[r]);
        e->ipc_peer[r] = nullptr;
        e->ipc_peer_local[r] = false;
      }
      throw;
    }
  });
}

}  // extern "C"
