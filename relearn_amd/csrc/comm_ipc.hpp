// comm_ipc.hpp — device side of the peer-mailbox all-reduce (protocol and host side: comm_ipc.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "engine.hpp"

constexpr uint32_t IPC_CAP = 2048;             // floats per vector (the MLP updates need <= 1030)
constexpr uint32_t IPC_CHUNK = 64;             // columns per exchanging wave
constexpr uint32_t IPC_CHUNKS = IPC_CAP / IPC_CHUNK;
constexpr uint32_t IPC_SPIN_LIMIT = 1u << 24;  // polls of ~1 us before a wait gives up

// layout of one mailbox (receiver-owned): data[sender][slot][IPC_CAP] floats, then flags[sender][slot][IPC_CHUNKS]
__host__ __device__ inline size_t ipc_data_off(uint32_t sender, uint32_t slot) {
  return ((size_t)sender * 2 + slot) * IPC_CAP;
}
__host__ __device__ inline size_t ipc_flag_off(uint32_t n_ranks, uint32_t sender, uint32_t slot, uint32_t chunk) {
  return (size_t)n_ranks * 2 * IPC_CAP + ((size_t)sender * 2 + slot) * IPC_CHUNKS + chunk;
}
inline size_t ipc_box_words(uint32_t n_ranks) { return (size_t)n_ranks * 2 * (IPC_CAP + IPC_CHUNKS); }

struct IpcPeers {
  float *box[RL_IPC_MAX_RANKS];  // box[r] = rank r's mailbox as seen from this process (box[rank] = the own one)
  uint32_t rank, n_ranks, seq;
  int32_t *err;
};

inline IpcPeers ipc_peers_next(rl_engine *e) {  // the descriptor of the engine's NEXT collective
  IpcPeers p;
  for (int r = 0; r < RL_IPC_MAX_RANKS; ++r) p.box[r] = r < e->n_ranks ? e->ipc_peer[r] : nullptr;
  e->ipc_seq += 1;
  p.rank = (uint32_t)e->rank;
  p.n_ranks = (uint32_t)e->n_ranks;
  p.seq = e->ipc_seq;
  p.err = e->ipc_err;
  return p;
}

// One wave exchanges one 64-column chunk: lane `l` (0..63) contributes `mine` for column 64 chunk + l and gets the sum
// over all ranks, added in rank order.  All 64 lanes of the wave must call it (wave-level barriers inside).
__device__ __forceinline__ float ipc_exchange_chunk(const IpcPeers &pe, uint32_t chunk, uint32_t l, float mine) {
  const uint32_t slot = pe.seq & 1u, p = chunk * IPC_CHUNK + l;
  // publish: my row in every mailbox (the own one included), then the flags
  for (uint32_t r = 0; r < pe.n_ranks; ++r) __builtin_nontemporal_store(mine, pe.box[r] + ipc_data_off(pe.rank, slot) + p);
  __threadfence_system();
  __builtin_amdgcn_wave_barrier();
  if (l == 0)
    for (uint32_t r = 0; r < pe.n_ranks; ++r)
      __hip_atomic_store(reinterpret_cast<uint32_t *>(pe.box[r]) + ipc_flag_off(pe.n_ranks, pe.rank, slot, chunk),
                         pe.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  // wait for every rank's row of this chunk: lane r polls rank r's flag
  float *own = pe.box[pe.rank];
  if (l < pe.n_ranks) {
    const uint32_t *flag = reinterpret_cast<const uint32_t *>(own) + ipc_flag_off(pe.n_ranks, l, slot, chunk);
    uint32_t spins = 0;
    // sequence numbers only grow: `>= seq` also accepts a peer that is already one collective ahead on the other slot
    while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - pe.seq) < 0) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > IPC_SPIN_LIMIT) {
        atomicExch(pe.err, (int32_t)(1 + l));  // which rank never arrived (1-based)
        break;
      }
    }
  }
  __threadfence_system();
  __builtin_amdgcn_wave_barrier();
  float s = __builtin_nontemporal_load(own + ipc_data_off(0, slot) + p);
  for (uint32_t r = 1; r < pe.n_ranks; ++r) s = s + __builtin_nontemporal_load(own + ipc_data_off(r, slot) + p);
  return s;
}
