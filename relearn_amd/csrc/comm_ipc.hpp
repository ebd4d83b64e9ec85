// comm_ipc.hpp — device side of the peer-mailbox all-reduce (protocol and host side: comm_ipc.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "engine.hpp"

constexpr uint32_t IPC_CAP = 2048;             // floats per vector (the MLP updates need <= 1030)
constexpr uint32_t IPC_CHUNK = 64;             // columns per exchanging wave
constexpr uint32_t IPC_CHUNKS = IPC_CAP / IPC_CHUNK;
constexpr uint64_t IPC_TIMEOUT_MS_DEFAULT = 30000;  // wall-clock bound of one wait (RELEARN_IPC_TIMEOUT_MS overrides it)

// layout of one mailbox (receiver-owned): words[sender][slot][IPC_CAP], one 64-bit word per vector element =
// {sequence number of the collective : 32 | the float's bits : 32}
__host__ __device__ inline size_t ipc_word_off(uint32_t sender, uint32_t slot) {
  return ((size_t)sender * 2 + slot) * IPC_CAP;
}
// a mailbox holds two channels (rl_engine::chan: the main and the auxiliary update chain may both have a collective in
// flight), each with its own rows and its own sequence numbers
inline size_t ipc_chan_words(uint32_t n_ranks) { return (size_t)n_ranks * 2 * IPC_CAP; }
inline size_t ipc_box_bytes(uint32_t n_ranks) { return 2 * ipc_chan_words(n_ranks) * sizeof(uint64_t); }

struct IpcPeers {
  float *box[RL_IPC_MAX_RANKS];  // box[r] = rank r's mailbox as seen from this process (box[rank] = the own one)
  uint32_t rank, n_ranks, seq;
  int32_t *err;                  // sticky error word of the engine: 0, or 1 + the rank that never arrived
  uint64_t timeout_ticks;        // bound of one wait in ticks of the constant 100 MHz counter (s_memrealtime)
};

inline IpcPeers ipc_peers_next(rl_engine *e) {  // the descriptor of the NEXT collective of the engine's current channel
  IpcPeers p;
  const size_t chan_floats = 2 * ipc_chan_words((uint32_t)e->n_ranks) * (size_t)e->chan;  // (64-bit words)
  for (int r = 0; r < RL_IPC_MAX_RANKS; ++r) p.box[r] = r < e->n_ranks ? e->ipc_peer[r] + chan_floats : nullptr;
  e->ipc_seq[e->chan] += 1;
  p.rank = (uint32_t)e->rank;
  p.n_ranks = (uint32_t)e->n_ranks;
  p.seq = e->ipc_seq[e->chan];
  p.err = e->ipc_err;
  p.timeout_ticks = e->ipc_timeout_ticks;
  return p;
}

// One wave exchanges one 64-column chunk: lane `l` (0..63) contributes `mine` for column 64 chunk + l and gets the sum
// over all ranks, added in rank order.  Value and sequence number travel in ONE 64-bit word, written and polled with
// relaxed system-scope atomics (the mailboxes are fine-grained memory): no flag, no fence.  (The first version — plain
// stores, a system-scope release fence, one flag per chunk, an acquire fence — cost 26 us per collective with two
// processes on one GPU: on this part a release / acquire fence at device or system scope writes back and invalidates
// the L2 of the XCD)
// Returns false — for the WHOLE wave, and then `sum` is not to be used: the caller returns before it stores anything —
// when the engine's error word is already set (an earlier collective failed: every later one fails fast, nothing is
// published) or when a peer's word has not arrived within the wall-clock bound.  A timeout sets the error word and
// stays set; the host raises RL_ERR_COMM at its next synchronising call (ipc_check).
__device__ __forceinline__ bool ipc_exchange_chunk(const IpcPeers &pe, uint32_t chunk, uint32_t l, float mine,
                                                   float &sum) {
  if (__hip_atomic_load(pe.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;  // wave-uniform
  const uint32_t slot = pe.seq & 1u, p = chunk * IPC_CHUNK + l;
  const uint64_t word = ((uint64_t)pe.seq << 32) | (uint64_t)__builtin_bit_cast(uint32_t, mine);
  // publish: my row in every mailbox (the own one included)
  for (uint32_t r = 0; r < pe.n_ranks; ++r)
    __hip_atomic_store(reinterpret_cast<uint64_t *>(pe.box[r]) + ipc_word_off(pe.rank, slot) + p, word, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  // gather: this column from every rank's row of the own mailbox, in rank order
  const uint64_t *own = reinterpret_cast<const uint64_t *>(pe.box[pe.rank]);
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  float s = 0.0f;
  int32_t missing = 0;  // 1 + the first rank whose word this lane gave up on
  for (uint32_t r = 0; r < pe.n_ranks; ++r) {
    const uint64_t *q = own + ipc_word_off(r, slot) + p;
    uint64_t w = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t polls = 0;
    while ((uint32_t)(w >> 32) != pe.seq) {  // (a slot is rewritten two collectives later, after this read: header of comm_ipc.hip)
      __builtin_amdgcn_s_sleep(2);
      if ((++polls & 255u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > pe.timeout_ticks) {
        missing = (int32_t)(1 + r);
        break;
      }
      w = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (missing != 0) break;
    const float v = __builtin_bit_cast(float, (uint32_t)w);
    s = r == 0 ? v : s + v;
  }
  sum = s;
  // the decision is the wave's, not the lane's: one late word fails the whole chunk, nobody stores a partial sum
  const bool ok = __builtin_amdgcn_ballot_w64(missing != 0) == 0;
  if (!ok && missing != 0) atomicCAS(pe.err, 0, missing);  // first failure wins and stays
  return ok;
}
