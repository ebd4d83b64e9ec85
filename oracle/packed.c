/* packed.c — oracle restatement of PackedTensor structure ops and LazyHistoryFeatures.
 * TEST INFRASTRUCTURE (see oracle.h).
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

/* PackedStructure::from_sorted_sequence_lengths (src/torch/packed.rs:346-456):
 * batch_sizes[t] = number of sequences longer than t; lengths must be non-increasing. */
int64_t oracle_packed_batch_sizes(const uint64_t *sorted_lengths, uint64_t n_seq, uint64_t *batch_sizes_out,
                                  uint64_t cap) {
  for (uint64_t i = 1; i < n_seq; ++i)
    if (sorted_lengths[i] > sorted_lengths[i - 1]) return -1; /* PackingError::Increasing */
  uint64_t max_len = n_seq ? sorted_lengths[0] : 0;
  if (max_len > cap) return -2;
  uint64_t alive = n_seq;
  for (uint64_t t = 0; t < max_len; ++t) {
    while (alive > 0 && sorted_lengths[alive - 1] <= t) alive -= 1;
    batch_sizes_out[t] = alive;
  }
  return (int64_t)max_len;
}

/* PackedSeqIter::from_sorted (packed.rs:753-832): time-major interleave */
void oracle_packed_order(const uint64_t *sorted_lengths, uint64_t n_seq, uint64_t *seq_out, uint64_t *off_out) {
  uint64_t max_len = n_seq ? sorted_lengths[0] : 0;
  uint64_t k = 0;
  for (uint64_t t = 0; t < max_len; ++t)
    for (uint64_t s = 0; s < n_seq && sorted_lengths[s] > t; ++s) {
      seq_out[k] = s;
      off_out[k] = t;
      k += 1;
    }
}

/* inplace_discounted_cumsum_from_end (packed.rs:312-342): walk the batches from the last time slice
 * to the first; `*a += *b * discount` — an f32 multiply rounded, then an f32 add (not fused). */
void oracle_discounted_cumsum_from_end_f32(float *data, uint64_t n, float discount, const uint64_t *batch_sizes,
                                           uint64_t n_batches) {
  uint64_t offset = n;
  uint64_t prev_offset = n, prev_size = 0;
  for (uint64_t bi = n_batches; bi-- > 0;) {
    uint64_t batch_size = batch_sizes[bi];
    offset -= batch_size;
    for (uint64_t i = 0; i < prev_size; ++i) {
      float prod = data[prev_offset + i] * discount;
      data[offset + i] = data[offset + i] + prod;
    }
    prev_offset = offset;
    prev_size = batch_size;
  }
}

/* BatchSizes::trim(n) (packed.rs:604-625): remove n steps from every sequence == drop the first n
 * batch sizes when trimming the start; for trim_end the new sizes are the old sizes shifted by n. */
uint64_t oracle_packed_trim_batch_sizes(const uint64_t *batch_sizes, uint64_t n, uint64_t trim, uint64_t *out) {
  if (trim >= n) return 0;
  for (uint64_t i = 0; i + trim < n; ++i) out[i] = batch_sizes[i + trim];
  return n - trim;
}

/* PackedTensor::trim_end, Ragged arm (packed.rs:237-265): time slice t keeps only the first
 * new_batch_sizes[t] = batch_sizes[t + trim] rows (those sequences that still have `trim` more). */
void oracle_packed_trim_end_f32(const float *in, const uint64_t *batch_sizes, uint64_t n_batches, uint64_t trim,
                                float *out) {
  uint64_t src = 0, dst = 0;
  for (uint64_t t = 0; t < n_batches; ++t) {
    uint64_t keep = (t + trim < n_batches) ? batch_sizes[t + trim] : 0;
    for (uint64_t i = 0; i < keep; ++i) out[dst++] = in[src + i];
    src += batch_sizes[t];
  }
}

/* ---------------------------------------------------------------- LazyHistoryFeatures
 * src/torch/agents/features.rs:70-213 */
typedef struct {
  const oracle_vecbuffer *buf;
  uint64_t start, len, order;
} episode_ref;

static int cmp_episode_len_desc(const void *a, const void *b) {
  const episode_ref *x = (const episode_ref *)a, *y = (const episode_ref *)b;
  if (x->len != y->len) return x->len > y->len ? -1 : 1;
  /* the reference uses sort_unstable (order among equal lengths unspecified); keep source order */
  return x->order < y->order ? -1 : (x->order > y->order ? 1 : 0);
}

oracle_features *oracle_features_from_buffers(oracle_vecbuffer *const *buffers, uint64_t n_buffers) {
  uint64_t n_eps = 0;
  for (uint64_t b = 0; b < n_buffers; ++b) n_eps += buffers[b]->n_episode_ends;
  episode_ref *eps = (episode_ref *)malloc((n_eps ? n_eps : 1) * sizeof(episode_ref));
  uint64_t *src_base = (uint64_t *)malloc((n_buffers ? n_buffers : 1) * sizeof(uint64_t));
  uint64_t k = 0, base = 0;
  for (uint64_t b = 0; b < n_buffers; ++b) {
    src_base[b] = base;
    uint64_t start = 0;
    for (uint64_t e = 0; e < buffers[b]->n_episode_ends; ++e) {
      uint64_t end = buffers[b]->episode_ends[e];
      eps[k].buf = buffers[b];
      eps[k].start = start;
      eps[k].len = end - start;
      eps[k].order = k;
      k += 1;
      start = end;
    }
    base += buffers[b]->len;
  }
  qsort(eps, n_eps, sizeof(episode_ref), cmp_episode_len_desc);

  oracle_features *f = (oracle_features *)calloc(1, sizeof(*f));
  uint32_t D = n_buffers ? buffers[0]->obs_dim : 0;
  f->obs_dim = D;
  f->n_episodes = n_eps;
  uint64_t max_len = n_eps ? eps[0].len : 0;
  uint64_t *lens = (uint64_t *)malloc((n_eps ? n_eps : 1) * sizeof(uint64_t));
  uint64_t *ext_lens = (uint64_t *)malloc((n_eps ? n_eps : 1) * sizeof(uint64_t));
  for (uint64_t e = 0; e < n_eps; ++e) {
    lens[e] = eps[e].len;
    ext_lens[e] = eps[e].len + 1;
    f->n_steps += eps[e].len;
  }
  f->n_ext = f->n_steps + n_eps;
  f->batch_sizes = (uint64_t *)malloc((max_len + 1) * sizeof(uint64_t));
  f->ext_batch_sizes = (uint64_t *)malloc((max_len + 2) * sizeof(uint64_t));
  f->n_batches = (uint64_t)oracle_packed_batch_sizes(lens, n_eps, f->batch_sizes, max_len + 1);
  f->n_ext_batches = (uint64_t)oracle_packed_batch_sizes(ext_lens, n_eps, f->ext_batch_sizes, max_len + 2);

  f->obs = (float *)calloc((f->n_steps ? f->n_steps : 1) * (D ? D : 1), sizeof(float));
  f->ext_obs = (float *)calloc((f->n_ext ? f->n_ext : 1) * (D ? D : 1), sizeof(float));
  f->is_invalid = (uint8_t *)calloc(f->n_ext ? f->n_ext : 1, 1);
  f->actions = (int64_t *)calloc(f->n_steps ? f->n_steps : 1, sizeof(int64_t));
  f->rewards = (float *)calloc(f->n_steps ? f->n_steps : 1, sizeof(float));
  f->src_index = (uint64_t *)calloc(f->n_steps ? f->n_steps : 1, sizeof(uint64_t));

  /* observation / action / reward packing (features.rs:127-137, 187-207) */
  uint64_t p = 0;
  for (uint64_t t = 0; t < max_len; ++t)
    for (uint64_t e = 0; e < n_eps && eps[e].len > t; ++e) {
      const oracle_vecbuffer *vb = eps[e].buf;
      uint64_t i = eps[e].start + t;
      memcpy(f->obs + p * D, vb->obs + i * D, D * sizeof(float));
      f->actions[p] = vb->action[i];
      f->rewards[p] = (float)vb->reward[i]; /* f64::from(feedback) as f32 (features.rs:202) */
      uint64_t bidx = 0;
      for (uint64_t b = 0; b < n_buffers; ++b)
        if (buffers[b] == vb) bidx = b;
      f->src_index[p] = src_base[bidx] + i;
      p += 1;
    }
  /* extended observations (features.rs:139-185, 216-261): per episode its observations followed by
   * the Interrupt successor observation, or an all-zero row flagged invalid */
  p = 0;
  for (uint64_t t = 0; t < max_len + 1; ++t)
    for (uint64_t e = 0; e < n_eps && eps[e].len + 1 > t; ++e) {
      const oracle_vecbuffer *vb = eps[e].buf;
      if (t < eps[e].len) {
        memcpy(f->ext_obs + p * D, vb->obs + (eps[e].start + t) * D, D * sizeof(float));
      } else {
        uint64_t last = eps[e].start + eps[e].len - 1;
        if (eps[e].len > 0 && vb->next[last] == ORACLE_INTERRUPT)
          memcpy(f->ext_obs + p * D, vb->next_obs + last * D, D * sizeof(float));
        else
          f->is_invalid[p] = 1;
      }
      p += 1;
    }
  free(lens);
  free(ext_lens);
  free(eps);
  free(src_base);
  return f;
}

void oracle_features_free(oracle_features *f) {
  if (!f) return;
  free(f->batch_sizes);
  free(f->ext_batch_sizes);
  free(f->obs);
  free(f->ext_obs);
  free(f->is_invalid);
  free(f->actions);
  free(f->rewards);
  free(f->src_index);
  free(f);
}
