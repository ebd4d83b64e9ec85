// comm_ipc.hpp — device side of the peer-mailbox all-reduce (protocol and host side: comm_ipc.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "engine.hpp"

constexpr uint32_t IPC_CAP = 2048;             // floats per vector (the MLP updates need <= 1030)
constexpr uint32_t IPC_CHUNK = 64;             // columns per exchanging wave
constexpr uint32_t IPC_CHUNKS = IPC_CAP / IPC_CHUNK;
constexpr uint32_t IPC_SPIN_LIMIT = 1u << 24;  // polls of ~1 us before a wait gives up

// layout of one mailbox (receiver-owned): words[sender][slot][IPC_CAP], one 64-bit word per vector element =
// {sequence number of the collective : 32 | the float's bits : 32}
__host__ __device__ inline size_t ipc_word_off(uint32_t sender, uint32_t slot) {
  return ((size_t)sender * 2 + slot) * IPC_CAP;
}
inline size_t ipc_box_bytes(uint32_t n_ranks) { return (size_t)n_ranks * 2 * IPC_CAP * sizeof(uint64_t); }

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
// over all ranks, added in rank order.  Value and sequence number travel in ONE 64-bit word, written and polled with
// relaxed system-scope atomics (the mailboxes are fine-grained memory): no flag, no fence.  (The first version — plain
// stores, a system-scope release fence, one flag per chunk, an acquire fence — cost 26 us per collective with two
// processes on one GPU: on this part a release / acquire fence at device or system scope writes back and invalidates
// the L2 of the XCD)
__device__ __forceinline__ float ipc_exchange_chunk(const IpcPeers &pe, uint32_t chunk, uint32_t l, float mine) {
  const uint32_t slot = pe.seq & 1u, p = chunk * IPC_CHUNK + l;
  const uint64_t word = ((uint64_t)pe.seq << 32) | (uint64_t)__builtin_bit_cast(uint32_t, mine);
  // publish: my row in every mailbox (the own one included)
  for (uint32_t r = 0; r < pe.n_ranks; ++r)
    __hip_atomic_store(reinterpret_cast<uint64_t *>(pe.box[r]) + ipc_word_off(pe.rank, slot) + p, word, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  // gather: this column from every rank's row of the own mailbox, in rank order
  const uint64_t *own = reinterpret_cast<const uint64_t *>(pe.box[pe.rank]);
  float s = 0.0f;
  for (uint32_t r = 0; r < pe.n_ranks; ++r) {
    const uint64_t *q = own + ipc_word_off(r, slot) + p;
    uint64_t w = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t spins = 0;
    while ((uint32_t)(w >> 32) != pe.seq) {  // (a slot is rewritten two collectives later, after this read: header of comm_ipc.hip)
      __builtin_amdgcn_s_sleep(2);
      if (++spins > IPC_SPIN_LIMIT) {
        atomicExch(pe.err, (int32_t)(1 + r));  // which rank never arrived (1-based)
        break;
      }
      w = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const float v = __builtin_bit_cast(float, (uint32_t)w);
    s = r == 0 ? v : s + v;
  }
  return s;
}
