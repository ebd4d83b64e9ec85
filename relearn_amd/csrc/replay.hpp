// replay.hpp — the per-lane replay ring shared by the collection kernel (device) and its host-side index mirror.
//
// Reference: ReplayBuffer (src/agents/buffers/replay.rs:11-127) = a bounded deque of steps plus a deque of episode
// ends; when a write finds the deque full the WHOLE OLDEST EPISODE is dropped first (replay.rs:89-105), an
// episode that cannot fit at all is `WriteExperienceError::Full`.  In the engine every lane is one such buffer
// (the reference has one per worker thread): the bulk step data is a time ring in HBM, `[C][N]` with the lane
// fastest, and the bookkeeping below is run identically by the collecting kernel (which needs it to know where
// to write) and by the host (which samples episodes with the agent's Prng, dqn.rs:280-291).
#pragma once
#include <cstdint>

#if defined(__HIPCC__) || defined(__HIP__)
#define RL_RING_HD __host__ __device__ inline
#else
#define RL_RING_HD inline
#endif

struct LaneRing {
  uint32_t head;      // absolute index of the oldest stored step
  uint32_t count;     // stored steps
  uint32_t ep_head;   // ring index of the oldest stored episode end
  uint32_t ep_count;  // stored (complete) episodes
  uint32_t total;     // total_step_count (replay.rs:21-26)
};

// One `write_step` (replay.rs:89-115).  `EpEnds` provides get(i)/set(i, v) over the lane's ring of E episode ends
// (absolute one-past-the-end step indices).  Returns the ring slot (absolute index) the step goes to, or
// 0xffffffff when the buffer is full of a single unfinished episode (WriteExperienceError::Full).
template <typename EpEnds>
RL_RING_HD uint32_t ring_write_step(LaneRing &r, uint32_t C, uint32_t E, EpEnds &ep, bool episode_done) {
  if (r.count == C) {
    if (r.ep_count == 0) return 0xffffffffu;
    const uint32_t end = ep.get(r.ep_head % E);
    r.count -= end - r.head;
    r.head = end;
    r.ep_head += 1;
    r.ep_count -= 1;
  }
  const uint32_t slot = r.head + r.count;
  r.count += 1;
  r.total += 1;
  if (episode_done) {
    if (r.ep_count == E) {  // episode table full (more than E episodes in C steps): drop the oldest episode
      const uint32_t end = ep.get(r.ep_head % E);
      r.count -= end - r.head;
      r.head = end;
      r.ep_head += 1;
      r.ep_count -= 1;
    }
    ep.set((r.ep_head + r.ep_count) % E, r.head + r.count);
    r.ep_count += 1;
  }
  return slot;
}
