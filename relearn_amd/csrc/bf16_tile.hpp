// bf16_tile.hpp — the tile machinery shared by the fused update kernels (kernels_critic.hip, kernels_mfma.hip):
// layer 1 of the 5-128-A MLP and the masked-sum backward on the bf16 matrix pipe, every product exact.
//
// What the hardware dictates (measured, scripts/probe/pipe_overlap.hip): v_mfma_f32_32x32x2_f32 runs at the f32 vector
// rate AND occupies the vector ALU — its time adds to every other wave's VALU time on the SIMD — while the bf16 matrix
// pipe (v_mfma_f32_32x32x16_bf16, 36 cycles) runs beside vector work at the price of 8 issue cycles.  So the
// GEMM-shaped parts run on the bf16 pipe:
//   an f32 value splits EXACTLY into three bf16 pieces, v = p0 + p1 + p2 (8 + 8 + 8 significand bits; each residual is
//   representable), and a product of two bf16 numbers is exact in the f32 accumulator.  The only roundings left are the
//   f32 accumulations inside the instruction — measured below the error of a sequential f32 fma chain over the same
//   terms (scripts/probe/mfma_bf16_mask.hip).  Nothing is computed at reduced precision.
//
// One wavefront owns a tile of 32 samples at a time.
//   forward   pre[s][j] = sum_k x~[s][k] W~1[j][k]  (k = 5: bias, x~ = 1) as the sum over the 9 piece pairs of every k:
//             48 contraction slots = 3 issues per 32-unit hidden tile, oriented with the HIDDEN UNIT on the lane
//             (col = lane & 31) and the SAMPLE in the accumulator registers (row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).
//   backward  M[j][k] = sum_s [pre_sj > 0] . u_sk  with  u_sk = (dL/dy)_s . x~_sk — a masked sum: the mask is 0 or 1, u
//             splits into three pieces, M = G^T [p0 | p1 | p2].  The forward accumulator tile is already laid out as this
//             instruction's A operand (sum over its row index: "X^T . B", cdna_hip_programming.md §3), so the mask goes
//             from registers to the matrix pipe without lane movement: per (sample, hidden unit) the VALU spends one
//             multiply-clamp (relu') and half a convert instead of the seven operations of a 6-column fma backward.
//             The 18 piece columns occupy 18 of the 32 output columns; the three pieces of a column are added when the
//             f32 accumulators are flushed into the f64 level of the two-level accumulation (kernels_update.hip says why
//             TRPO / Adam want that level).
//
// Contraction slots of the forward (48 = 3 issues x 16; lane half h of issue i holds slots 16 i + 8 h + 0..7, i.e. the
// half's own list u = 8 i + j, 24 entries).  An entry pairs piece a of an input with piece b of the matching weight:
//   u = 0..8    input 2h,     (a, b) = (0,0) (0,1) (0,2) (1,0) (1,1) (1,2) (2,0) (2,1) (2,2)
//   u = 9..17   input 2h + 1, the same nine pairs
//   u = 18..23  h = 0: input 4, (0,0) (0,1) (0,2) (1,0) (1,1) (1,2)
//               h = 1: input 4, (2,0) (2,1) (2,2); then the bias input 1.0 (one piece) with the bias's three pieces
// so every lane needs three observation features of its sample: 2h, 2h + 1 and 4.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace bt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int NT = 4;     // 32-unit hidden tiles (H = 128)
// How the fused kernels with two waves per SIMD deal their tiles (round 6).  The older wave of a SIMD (waves 0-3 of an
// eight-wave workgroup) wins the issue arbitration: dealt evenly, it finished its tiles at 0.71 of the launch and the
// younger wave ran the rest alone, with nothing to overlap its matrix instructions with.  An older wave now plays
// SHARE_OLD virtual waves, a younger one SHARE_YOUNG; 5 : 3 lets both finish within 8 us of each other at the headline
// size and keeps the tile counts integral at every power-of-two lane count (9:7, 12:7, 7:4, 16:9, 9:5, 11:6, 2:1, 7:3, 3:1
// measured: none better; profiles/r06_critic_step_timeline.txt).  RL_CRITIC_SHARES=a:b overrides at run time.
constexpr uint32_t SHARE_OLD = 5, SHARE_YOUNG = 3;
constexpr int COLS = 19;  // piece columns of the backward: pcol(k) + p, k = input feature (5 = bias), p = piece; the three
                          // pieces of a feature stay inside one 16-lane row (column 15 is unused), so the flush adds them
                          // with row shifts
constexpr int UROW = 36;  // halfwords per row of the piece image [piece column][sample] (72-byte rows: conflict-free)

__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wavefront execute in program order; what is needed is that the compiler keeps that
  // order across lanes it cannot see a dependence between.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// two f32 -> packed bf16 (element 0 in the low half), round to nearest even: v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2 x = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2));
}
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t bits16) { return __builtin_bit_cast(float, bits16 << 16); }
// v = p0 + p1 + p2 exactly, each piece a bf16 bit pattern in the low half of a register
__device__ __forceinline__ void split3(float v, uint32_t &p0, uint32_t &p1, uint32_t &p2) {
  p0 = pack_bf16(v, 0.0f);
  const float r1 = v - bf16_bits_to_f32(p0);
  p1 = pack_bf16(r1, 0.0f);
  const float r2 = r1 - bf16_bits_to_f32(p1);
  p2 = pack_bf16(r2, 0.0f);
}
__device__ __forceinline__ uint32_t pk(uint32_t lo, uint32_t hi) { return lo | (hi << 16); }
// The same split by TRUNCATION, for the per-tile paths (round 3): p0 = the top 16 bits of v, the residual is exact and
// has at most 16 significant bits, p1 = its top 16 bits, what is left has at most 8 and IS the third piece — two ands and
// two subtractions instead of three conversions, two shifts and two subtractions.  The pieces come out as f32 bit
// patterns (bf16 bits in the HIGH half); pkh packs two of them with one byte permute.  (v = p0 + p1 + p2 exactly, as
// with rounding; |p1| <= 2^-7 |v| here, so this form is for the kernels that multiply ALL nine piece pairs.)
__device__ __forceinline__ void split3t(float v, uint32_t &p0, uint32_t &p1, uint32_t &p2) {
  p0 = __builtin_bit_cast(uint32_t, v) & 0xFFFF0000u;
  const float r1 = v - __builtin_bit_cast(float, p0);
  p1 = __builtin_bit_cast(uint32_t, r1) & 0xFFFF0000u;
  p2 = __builtin_bit_cast(uint32_t, r1 - __builtin_bit_cast(float, p1));
}
__device__ __forceinline__ uint32_t pkh(uint32_t lo, uint32_t hi) {  // element 0 = lo's piece, element 1 = hi's
  return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
}

// the value of lane (l & 31) and of lane (l & 31) + 32 in every lane: one v_permlane32_swap (VALU) instead of a
// ds_bpermute round trip
__device__ __forceinline__ void both_halves(float v, float &lo, float &hi) {
  // the instruction swaps the upper half of its first operand with the lower half of its second: fed the same value
  // twice (in two registers), the first ends up holding the lower half's values in both halves, the second the upper
  // half's.  (Inline asm: the builtin's second result is mis-modelled by this compiler when both inputs are one value;
  // the s_nop covers the VALU-write -> permlane-read wait state the compiler would insert.)
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  lo = a;
  hi = b;
}

// sum of 16 consecutive floats of a transposed LDS row, as four interleaved chains joined pairwise (a short dependent
// chain instead of a 16-long one)
__device__ __forceinline__ float row_sum16(const float *row) {
  float a[4] = {row[0], row[1], row[2], row[3]};
#pragma unroll
  for (int c = 4; c < 16; ++c) a[c & 3] = a[c & 3] + row[c];
  return (a[0] + a[1]) + (a[2] + a[3]);
}
// the same from a 16-byte aligned row (rows of YROW floats: four ds_read_b128 instead of eight ds_read2_b32, and the
// 144-byte row stride keeps the four lane groups of a b128 read on distinct banks: scripts/lds_conflicts.py)
constexpr int YROW = 36;
__device__ __forceinline__ float row_sum16v(const float *row) {
  const float4 *v = reinterpret_cast<const float4 *>(row);
  const float4 q0 = v[0], q1 = v[1], q2 = v[2], q3 = v[3];
  const float a0 = ((q0.x + q1.x) + q2.x) + q3.x, a1 = ((q0.y + q1.y) + q2.y) + q3.y;
  const float a2 = ((q0.z + q1.z) + q2.z) + q3.z, a3 = ((q0.w + q1.w) + q2.w) + q3.w;
  return (a0 + a1) + (a2 + a3);
}

// Sums over the 32 unit lanes of a half WITHOUT LDS (round 5).  An accumulator tile has the hidden unit on the lane and
// the sample in the registers; the output layer needs, for each of the 16 rows, the sum over the half's 32 lanes.  Through
// LDS that is a transpose: 8 ds_write2 + 4 ds_read_b128 per tile, and an LDS instruction holds the SIMD's issue port for
// ~18 cycles (profiles/r03_slot_cost.txt).  Here the rows are folded across the lanes by a halving butterfly in registers:
//   level 1  partner lane ^ 8 (row_ror:8): lanes with bit 3 clear keep rows 0..7, the others rows 8..15 — one
//            v_add_f32_dpp per row and lane group, the groups separated by the instruction's bank mask (16 issues -> 8 rows)
//   level 2  partner lane ^ 7 (row_half_mirror; it preserves bit 3): groups by bit 2, again a bank mask (8 issues -> 4)
//   level 3  partner lane ^ 2 (quad_perm), groups by bit 1: inside a bank, so two sums and a select (6 -> 2 rows)
//   level 4  partner lane ^ 1, groups by bit 0 (3 -> 1 row)
//   level 5  the other 16-lane row of the half (v_permlane16_swap, 3)
// after which lane L of a half holds the half-wide sum of ITS row L & 15 (36 vector instructions, no memory).  The sums
// reach the sample-owning lanes through one ds_bpermute_b32 (row_sums_to_samples).  Every level adds each pair in the
// same order on both partners, so the result does not depend on which lane ends up holding it.
// The DPP instructions are inline asm (the bank-masked second write into the same destination cannot be expressed with
// the builtins); a DPP read of a register needs two wait states after the vector instruction that wrote it, which the
// compiler does not track into asm statements: every statement starts with s_nop 1.
__device__ __forceinline__ float fold_rows16(const float (&v)[16], int lane) {
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc"
        : "=&v"(a[i])
        : "v"(v[i]), "v"(v[i + 8]));
  float b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa"
        : "=&v"(b[i])
        : "v"(a[i]), "v"(a[i + 4]));
  const bool bit1 = (lane & 2) != 0, bit0 = (lane & 1) != 0;
  float c[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float lo, hi;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "=&v"(lo), "=&v"(hi)
        : "v"(b[i]), "v"(b[i + 2]));
    c[i] = bit1 ? hi : lo;
  }
  float lo, hi;
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
      : "=&v"(lo), "=&v"(hi)
      : "v"(c[0]), "v"(c[1]));
  float d = bit0 ? hi : lo;
  // the two 16-lane rows of the half: the instruction swaps the odd rows of its first operand with the even rows of its
  // second (fed one value twice: the first then holds the even rows' values everywhere, the second the odd rows')
  float e = d;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(d), "+v"(e));
  return d + e;
}
// byte address (for ds_bpermute_b32) of the lane that holds, after fold_rows16, the sum of sample n's row: the half that
// owns the sample (accumulator rows of half h are the samples with bit 2 == h) and its row index there
__device__ __forceinline__ int row_sum_source(int n) { return 4 * (32 * ((n >> 2) & 1) + ((n & 3) | ((n >> 3) << 2))); }
__device__ __forceinline__ float row_sums_to_samples(float folded, int src_addr) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_addr, __builtin_bit_cast(int, folded)));
}

// ---- tile operands through buffer loads: a raw buffer resource over a device array (loads past `bytes` return 0), a
// per-lane byte offset that never changes and the tile's byte offset in an SGPR — no vector address arithmetic per tile
// (64-bit global addresses cost ~15 vector instructions per tile in these kernels' loops)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_f32(rsrc_t r, uint32_t lane_off, uint32_t tile_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_off, (int)tile_off, 0));
}
__device__ __forceinline__ uint32_t buf_u8(rsrc_t r, uint32_t lane_off, uint32_t tile_off) {
  return (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(r, (int)lane_off, (int)tile_off, 0);
}

union Frag {
  bf16x8 v;
  uint32_t u[4];
  uint64_t q[2];
  uint4 x;
};

// B operands of the forward for one hidden unit: its weights for inputs 2 hf, 2 hf + 1, 4 and its bias
__device__ __forceinline__ void weight_frags(float wa, float wb, float w4, float bias, int hf, Frag (&f)[3]) {
  uint32_t a0, a1, a2, c0, c1, c2, e0, e1, e2, g0, g1, g2;
  split3(wa, a0, a1, a2);
  split3(wb, c0, c1, c2);
  split3(w4, e0, e1, e2);
  split3(bias, g0, g1, g2);
  f[0].u[0] = pk(a0, a1);
  f[0].u[1] = pk(a2, a0);
  f[0].u[2] = pk(a1, a2);
  f[0].u[3] = pk(a0, a1);
  f[1].u[0] = pk(a2, c0);
  f[1].u[1] = pk(c1, c2);
  f[1].u[2] = pk(c0, c1);
  f[1].u[3] = pk(c2, c0);
  f[2].u[0] = pk(c1, c2);
  f[2].u[1] = pk(e0, e1);
  f[2].u[2] = hf == 0 ? pk(e2, e0) : pk(e2, g0);
  f[2].u[3] = hf == 0 ? pk(e1, e2) : pk(g1, g2);
}

constexpr float FWD_SCALE = 0x1p96f, FWD_UNSCALE = 0x1p-96f;  // ("relu' in half an instruction per value", below)

// ---- the weight image (round 6).  Every fused launch used to rebuild its weight-piece fragments from the flat parameter
// vector in every wave: 20 strided loads and ~330 vector instructions (twelve three-piece splits and their packing) per
// wave before the first tile — a third of a wave's instructions when a rank holds 8,192 lanes (16 tiles per wave), most
// of them for a DQN minibatch (3 tiles per wave).  The kernels that WRITE the parameters (k_reduce_adam, k_adam_step,
// k_ls_set_params: one lane owns one parameter) now also store that parameter's three 2^96-scaled pieces where the
// consumers' lanes will load them, so a consumer's prologue is 12 x 16-byte loads per lane plus 5 raw floats per hidden
// tile.  Same numbers as weight_frags(2^96 w ...): the consumers compute bit for bit what they computed before.
//   frag part   uint4 [NT * 3][64]: entry (3 t + i, lane) = fragment i of hidden tile t for that lane, i.e. the halfword
//               sequence h = 0..23 of weight_frags — pieces of input 2 hf (h 0..8: p0 p1 p2 three times), of input 2 hf + 1
//               (h 9..17), then for half 0 input 4 twice (h 18..23), for half 1 input 4 once and the bias (h 18..23)
//   raw part    float [NT][WIMG_RAW][64]: the unscaled weights of the lane's unit — inputs 2 hf, 2 hf + 1, then input 4
//               (half 0) or the bias (half 1), then the unit's output weights W2[0][j] (and W2[1][j] for two outputs)
// Who may trust it: the image belongs to a module and is valid for one C-ABI call at a time (rl_mlp::wimg_epoch against
// rl_engine::call_epoch, engine.hpp): the first fused launch of a call builds it with k_wimg_build, the parameter-writing
// kernels of the same call keep it current, and whatever else touches the parameters makes the next call rebuild it.
constexpr int WIMG_RAW = 5;
constexpr int WIMG_FRAG_WORDS = NT * 3 * 64 * 4;
constexpr int WIMG_WORDS = WIMG_FRAG_WORDS + NT * WIMG_RAW * 64;
__device__ __forceinline__ uint32_t wimg_half_index(int t, int lane, int h) {
  return (uint32_t)(((t * 3 + (h >> 3)) * 64 + lane) * 8 + (h & 7));
}
__device__ __forceinline__ uint32_t wimg_raw_index(int t, int lane, int c) {
  return (uint32_t)(WIMG_FRAG_WORDS + (t * WIMG_RAW + c) * 64 + lane);
}
// parameter p of a flat [W1 (H x D), b1 (H), W2 (A x H), b2 (A)] vector with H = 128, D = 5 has the new value w
__device__ __forceinline__ void wimg_store_param(uint32_t *__restrict__ img, uint32_t p, float w, int A) {
  constexpr int H = 128, D = 5;
  unsigned short *h16 = reinterpret_cast<unsigned short *>(img);
  float *f32 = reinterpret_cast<float *>(img);
  if (p >= (uint32_t)(H * D + H + A * H)) return;  // the output biases are read from the vector itself
  if (p >= (uint32_t)(H * D + H)) {                // W2[a][j]: both halves of unit j
    const int q = (int)p - (H * D + H), a = q / H, j = q % H;
    f32[wimg_raw_index(j >> 5, j & 31, 3 + a)] = w;
    f32[wimg_raw_index(j >> 5, 32 + (j & 31), 3 + a)] = w;
    return;
  }
  uint32_t p0, p1, p2;
  split3(FWD_SCALE * w, p0, p1, p2);
  const unsigned short q0 = (unsigned short)p0, q1 = (unsigned short)p1, q2 = (unsigned short)p2;
  auto put = [&](int t, int lane, int h) {
    h16[wimg_half_index(t, lane, h)] = q0;
    h16[wimg_half_index(t, lane, h + 1)] = q1;
    h16[wimg_half_index(t, lane, h + 2)] = q2;
  };
  if (p >= (uint32_t)(H * D)) {  // b1[j]: the last three slots of half 1
    const int j = (int)p - H * D, t = j >> 5, n = j & 31;
    put(t, 32 + n, 21);
    f32[wimg_raw_index(t, 32 + n, 2)] = w;
    return;
  }
  const int j = (int)p / D, k = (int)p % D, t = j >> 5, n = j & 31;
  if (k < 4) {
    const int lane = n + 32 * (k >> 1), base = 9 * (k & 1);
    put(t, lane, base);
    put(t, lane, base + 3);
    put(t, lane, base + 6);
    f32[wimg_raw_index(t, lane, k & 1)] = w;
  } else {
    put(t, n, 18);
    put(t, n, 21);
    put(t, 32 + n, 18);
    f32[wimg_raw_index(t, n, 2)] = w;
  }
}
// a consumer's prologue: this lane's fragments of hidden tile t and the raw weights of its unit
struct WRaw {
  float wa, wb, wc, w2[2];  // wc: input 4 (half 0) or the bias (half 1)
};
__device__ __forceinline__ void wimg_load(const uint32_t *__restrict__ img, int t, int lane, Frag (&f)[3], WRaw &r, int A);

// A operands of the forward for one sample: its features 2 hf, 2 hf + 1, 4 and the bias input (1, or 0 for a padding
// sample)
__device__ __forceinline__ void input_frags(float xa, float xb, float xc, bool valid, int hf, Frag (&f)[3]) {
  uint32_t a0, a1, a2, c0, c1, c2, e0, e1, e2;
  split3t(xa, a0, a1, a2);
  split3t(xb, c0, c1, c2);
  split3t(xc, e0, e1, e2);
  const uint32_t one = valid ? 0x3F800000u : 0u;
  const uint32_t x0 = hf == 0 ? e0 : e2, x1 = hf == 0 ? e1 : one;
  f[0].u[0] = pkh(a0, a0);
  f[0].u[1] = pkh(a0, a1);
  f[0].u[2] = pkh(a1, a1);
  f[0].u[3] = pkh(a2, a2);
  f[1].u[0] = pkh(a2, c0);
  f[1].u[1] = pkh(c0, c0);
  f[1].u[2] = pkh(c1, c1);
  f[1].u[3] = pkh(c1, c2);
  f[2].u[0] = pkh(c2, c2);
  f[2].u[1] = pkh(x0, x0);
  f[2].u[2] = pkh(x0, x1);
  f[2].u[3] = pkh(x1, x1);
}

__device__ __forceinline__ void wimg_load(const uint32_t *__restrict__ img, int t, int lane, Frag (&f)[3], WRaw &r, int A) {
  const uint4 *__restrict__ fi = reinterpret_cast<const uint4 *>(img);
  const float *__restrict__ rw = reinterpret_cast<const float *>(img);
#pragma unroll
  for (int i = 0; i < 3; ++i) f[i].x = fi[(t * 3 + i) * 64 + lane];
  r.wa = rw[wimg_raw_index(t, lane, 0)];
  r.wb = rw[wimg_raw_index(t, lane, 1)];
  r.wc = rw[wimg_raw_index(t, lane, 2)];
  r.w2[0] = rw[wimg_raw_index(t, lane, 3)];
  r.w2[1] = A > 1 ? rw[wimg_raw_index(t, lane, 4)] : 0.0f;
}
// pre-activations of one 32-unit hidden tile for the wave's 32 samples
__device__ __forceinline__ f32x16 layer1(const Frag (&fa)[3], const Frag (&fw)[3]) {
  f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 3; ++i) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].v, fw[i].v, c, 0, 0, 0);
  return c;
}

// relu'(pre) of a tile's 16 registers, packed as the two A operands (sample groups s = 0, 1) of the backward
__device__ __forceinline__ void pack_mask(const float (&gm)[16], Frag (&ga)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) ga[s].u[i] = pack_bf16(gm[8 * s + 2 * i], gm[8 * s + 2 * i + 1]);
}
// the same, and keep the packed form (8 registers) live instead of the tile's 16 pre-activations: for kernels whose
// register budget is the tighter constraint
__device__ __forceinline__ void pack_mask_now(const float (&gm)[16], Frag (&ga)[2]) {
  pack_mask(gm, ga);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(ga[s].u[i]));
}

// relu' in half an instruction per value (round 3).  The forward weights are scaled by 2^96 (exact: a power of two), so
// the accumulator holds 2^96 pre, and v_cvt_pk_bf16_f32 with the VOP3 clamp bit turns two of them into packed 0 / 1
// masks in ONE instruction: 1 for pre >= 2^-96, 0 for pre <= 0 (NaN -> 0; scripts/probe/cvt_clamp_tr.hip).  A non-zero
// sum of exact products below 2^-96 would give a fractional mask; the f32 accumulation of terms of ordinary size cannot
// produce one.  Whatever else reads the accumulator multiplies by 2^-96 (folded into a constant operand: exact).
// The compiler's hazard recogniser does not look into inline asm, and a matrix instruction's result registers must not
// be read (or overwritten) by a vector instruction before the required wait states have passed (no hardware
// interlock): the first reader of the tile is therefore a real instruction — the clamp of element 0, which the
// conversion's own clamp leaves unchanged — and every asm statement takes its result as an extra operand, so all of
// them follow it in program order, behind the wait states the compiler inserted for it.

// The numeric range in which the scaled forward is exact (include/relearn_hip.h, "Numeric range of the fused kernels"),
// checked once per call from the weights the first fused launch has just loaded (hidden unit j: its weights for inputs
// 2 hf, 2 hf + 1, input 4 and its bias) and the magnitude range of the trajectory's observations (xmin = the smallest
// non-zero |x|, xmax = the largest |x|, from the range words: range_bounds; a NaN / Inf observation shows up as a
// non-finite xmax):
//   overflow   2^96 pre must stay finite with room for the partial sums:  sum_k |W1[j][k]| max|x| + |b1[j]| < 2^31;
//   relu'      a non-zero pre must reach 2^-96, or the clamped conversion returns a FRACTIONAL mask.  pre is a sum of
//              exact products whose f32 accumulation can cancel down to 2^-48 of its largest term, so the largest term of
//              a unit that is not identically zero must be able to reach 2^-46: max(max_k |W1[j][k]| min_nz|x|, |b1[j]|).
// A violation sets the chain's word err[chain] (host memory mapped into the device) and its veto word (device memory, after
// the range words: range_veto); the launch goes on (its numbers are then not to be used), the optimiser / line-search
// kernels that follow it in the same call read the veto word and leave parameters and optimiser state as they are, and
// the host raises RL_ERR_UNSUPPORTED after its next synchronisation (kernel variant 1 has no such bound).  Two chains,
// because the two run at once under rl_actor_critic_update: GUARD_POLICY (policy passes, the DQN gradient) and
// GUARD_CRITIC (the critic step).  ~60 instructions for one wave of the launch.
// The range words: RANGE_SLOTS minima and RANGE_SLOTS maxima, one 128-byte line each (writers fold into slot
// workgroup % RANGE_SLOTS: tens of thousands of atomics on ONE address resolve one after the other in the L2 and cost
// a 4,096-lane period 0.1 ms; the reader is one wave, one slot per lane).
constexpr int RANGE_SLOTS = 64, RANGE_STRIDE = 32;  // (words)
constexpr int RANGE_WORDS = 2 * RANGE_SLOTS * RANGE_STRIDE;
constexpr int GUARD_POLICY = 0, GUARD_CRITIC = 1;
constexpr int RANGE_ALLOC_WORDS = RANGE_WORDS + RANGE_STRIDE;  // + one line: the veto words of the two chains
__device__ __forceinline__ uint32_t *range_veto(uint32_t *range, int chain) { return range + RANGE_WORDS + chain; }
__device__ __forceinline__ uint32_t *range_lo_slot(uint32_t *range, int s) { return range + s * RANGE_STRIDE; }
__device__ __forceinline__ uint32_t *range_hi_slot(uint32_t *range, int s) { return range + (RANGE_SLOTS + s) * RANGE_STRIDE; }
__device__ __forceinline__ void range_fold(uint32_t *range, uint32_t workgroup, uint32_t lo, uint32_t hi) {
  const int s = (int)(workgroup % RANGE_SLOTS);
  if (lo < __hip_atomic_load(range_lo_slot(range, s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(range_lo_slot(range, s), lo);
  if (hi > __hip_atomic_load(range_hi_slot(range, s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(range_hi_slot(range, s), hi);
}
__device__ __forceinline__ void range_reset(uint32_t *range, int tid, int nthreads) {
  for (int s = tid; s < RANGE_SLOTS; s += nthreads) {
    *range_lo_slot(range, s) = 0x7F7FFFFFu;
    *range_hi_slot(range, s) = 0u;
  }
}
// a producer's side of the words: every lane accumulates the magnitudes it writes, the wave folds once at its end
__device__ __forceinline__ void range_accumulate(float x, uint32_t &lo, uint32_t &hi) {
  const uint32_t a = __builtin_bit_cast(uint32_t, x) & 0x7FFFFFFFu;
  hi = a > hi ? a : hi;
  lo = a != 0u && a < lo ? a : lo;
}
__device__ __forceinline__ void range_fold_wave(uint32_t *range, uint32_t key, uint32_t lo, uint32_t hi) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const uint32_t ol = (uint32_t)__shfl_xor((int)lo, m, 64), oh = (uint32_t)__shfl_xor((int)hi, m, 64);
    lo = ol < lo ? ol : lo;
    hi = oh > hi ? oh : hi;
  }
  if ((threadIdx.x & 63) == 0) range_fold(range, key, lo, hi);
}
// the whole wave: lane l reads slot l; every lane returns the bounds
__device__ __forceinline__ void range_bounds(uint32_t *range, int lane, float &xmin, float &xmax) {
  uint32_t lo = *range_lo_slot(range, lane % RANGE_SLOTS), hi = *range_hi_slot(range, lane % RANGE_SLOTS);
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const uint32_t ol = (uint32_t)__shfl_xor((int)lo, m, 64), oh = (uint32_t)__shfl_xor((int)hi, m, 64);
    lo = ol < lo ? ol : lo;
    hi = oh > hi ? oh : hi;
  }
  xmax = __builtin_bit_cast(float, hi);
  xmin = hi == 0u ? 0.0f : __builtin_bit_cast(float, lo);  // (all observations zero: pre = bias)
}
__device__ __forceinline__ void range_guard(float wa, float wb, float w4, float bj, int hf, float xmin, float xmax,
                                            uint32_t *err, uint32_t *veto) {
  const float aa = __builtin_fabsf(wa), ab = __builtin_fabsf(wb);
  const float own = hf == 0 ? __builtin_fabsf(w4) : 0.0f, bias = __builtin_fabsf(bj);
  float hi = (aa + ab + own) * xmax, lo = __builtin_fmaxf(__builtin_fmaxf(aa, ab), own) * xmin, nz = aa + ab + own;
  float hi0, hi1, lo0, lo1, nz0, nz1;
  both_halves(hi, hi0, hi1);
  both_halves(lo, lo0, lo1);
  both_halves(nz, nz0, nz1);
  const float upper = hi0 + hi1 + bias, largest = __builtin_fmaxf(__builtin_fmaxf(lo0, lo1), bias);
  const bool zero_unit = nz0 + nz1 + bias == 0.0f;
  const bool bad = !(upper < 0x1p31f) || (!zero_unit && !(largest >= 0x1p-46f));
  if (bad) {
    __hip_atomic_store(veto, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// the range guard's view of a unit from the image: input 4's weight lives in half 0, the bias in half 1
__device__ __forceinline__ void range_guard_img(const WRaw &r, int hf, float xmin, float xmax, uint32_t *err,
                                                uint32_t *veto) {
  float w4, bj;
  both_halves(r.wc, w4, bj);
  range_guard(r.wa, r.wb, w4, bj, hf, xmin, xmax, err, veto);
}

__device__ __forceinline__ uint32_t mask_pair(float lo, float hi, float after) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(lo), "v"(hi), "v"(after));
  return r;
}
// relu' of a scaled forward tile, packed as the two A operands of the backward
__device__ __forceinline__ void mask_tile(const f32x16 &c, Frag (&ga)[2]) {
  const float first = __builtin_amdgcn_fmed3f(c[0], 0.0f, 1.0f);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      ga[s].u[i] = mask_pair(s + i == 0 ? first : c[8 * s + 2 * i], c[8 * s + 2 * i + 1], first);
}

// A sum over the hidden units weighted by relu' as a masked sum on the matrix pipe (round 3).  With g_sj = relu'(pre_sj),
//   sum_j g_sj (x~_s . Z_j) = sum_k x~_sk q_sk,   q_sk = sum_j g_sj Z_jk
// for any matrix Z that is linear in the inputs — the Fisher-vector product's tangent logit (kernels_mfma.hip) — is the
// same kind of product as the backward, contracted over the hidden unit instead of the sample: 18 piece columns (three
// exact bf16 pieces of each Z_jk), 8 issues per tile.  The contraction runs over the forward tile's LANE index, so the
// packed mask is transposed first — by the matrix pipe itself: mask^T = mask^T . I with the packed mask as A operand (a
// sum over its row index: no lane movement) and an identity selection as B, 2 issues per hidden tile, exact 0 / 1
// results with the SAMPLE on the lane, packed by 8 conversions into the B operands of q^T[c][s] = sum_j Zp^T[c][j]
// mask^T[j][s].  (The same transpose through LDS, ds_write_b64 + ds_read_b64_tr_b16, is 32 LDS instructions per tile,
// ~690 issue cycles against ~190: scripts/probe/cvt_clamp_tr.hip, slot_cost.hip; measured in the critic step, where the
// whole scheme came out even with the vector-unit output layer and was not kept: 208-214 us against 195.)
constexpr int L2_KS = 8;  // k-steps of the product (128 hidden units / 16)
// An accumulator tile X (rows in the registers, column on the lane) packed as the operand of a product that sums over
// X's ROW index: element e of lane half h of k-step s is row acc_row(s, h, e) of X — the order every operand that meets
// such a fragment in a product has to follow.
__device__ __forceinline__ constexpr int acc_row(int s, int h, int e) { return 16 * s + 8 * (e >> 2) + 4 * h + (e & 3); }
// B operand of  X^T = X^T . I  for a packed 32 x 32 tile X given as A operand: I[k][n] = [row of slot k == n]
__device__ __forceinline__ void ident_frags(int lane, Frag (&id)[2]) {
  const int n = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      id[s].u[i] = pk(acc_row(s, hh, 2 * i) == n ? 0x3F80u : 0u, acc_row(s, hh, 2 * i + 1) == n ? 0x3F80u : 0u);
}
// Cooperative build of the A-operand fragments of Zp[j][k] = val(j, k), k = 0..5, in LDS, lane-linear per k-step
// (img[ks][lane]: every ds_read_b128 of a wave is 1 KB contiguous).  Row c = lane & 31 of the operand (= accumulator
// row of q^T) holds piece p of input k for c = row(3 i + p) of the lane half hf_out that owns input k (the slots of
// input_frags: 2 hf, 2 hf + 1, 4 + hf), nothing otherwise; element e of lane (c, hh) of k-step ks = 2 t + u is that
// piece of Zp[32 t + acc_row(u, hh, e)][k] — the hidden-unit order of the transposed masks.  Every thread of the
// workgroup calls it; the caller synchronises before the first l2_frag.
template <typename F>
__device__ __forceinline__ void l2_build(uint4 (*img)[64], int tid, int nthreads, F val) {
  for (int idx = tid; idx < L2_KS * 64; idx += nthreads) {
    const int ks = idx >> 6, ln = idx & 63, m = ln & 31, hh = ln >> 5;
    const int hf_out = (m >> 2) & 1, r = (m & 3) + 4 * (m >> 3), i = r / 3, p = r % 3;
    const int k = i < 2 ? 2 * hf_out + i : 4 + hf_out;
    uint32_t h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      uint32_t q0, q1, q2;
      split3(val(32 * (ks >> 1) + acc_row(ks & 1, hh, e), k), q0, q1, q2);
      h[e] = r < 9 ? (p == 0 ? q0 : (p == 1 ? q1 : q2)) : 0u;
    }
    img[ks][ln] = make_uint4(pk(h[0], h[1]), pk(h[2], h[3]), pk(h[4], h[5]), pk(h[6], h[7]));
  }
}
// q^T += Zp^T mask^T for one hidden tile: transpose the packed mask (2 issues), pack it (8 conversions), 2 product issues
__device__ __forceinline__ f32x16 masked_sum_tile(const Frag (&ga)[2], const Frag (&idb)[2], const uint4 (*img)[64],
                                                  int t, int lane, f32x16 q) {
  f32x16 gt = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  gt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[0].v, idb[0].v, gt, 0, 0, 0);
  gt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[1].v, idb[1].v, gt, 0, 0, 0);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    Frag mt, fz;
#pragma unroll
    for (int i = 0; i < 4; ++i) mt.u[i] = pack_bf16(gt[8 * s + 2 * i], gt[8 * s + 2 * i + 1]);
    fz.x = img[2 * t + s][lane];
    q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fz.v, mt.v, q, 0, 0, 0);
  }
  return q;
}
// sum_k x~_k q_k over this half's three inputs (a, b and c = x_4 or the bias input) from the q^T accumulator; the
// other half's part comes through both_halves
__device__ __forceinline__ float l2_dot(const f32x16 &q, float xa, float xb, float xc) {
  const float sa = (q[0] + q[1]) + q[2], sb = (q[3] + q[4]) + q[5], sc = (q[6] + q[7]) + q[8];
  return __builtin_fmaf(xc, sc, __builtin_fmaf(xb, sb, xa * sa));
}

// The pieces of u = g x~ from the sample lanes to the backward's B operand (sample in the registers, piece column on
// the lane) WITHOUT LDS (round 3; an LDS instruction costs 14-25 cycles of the SIMD's issue port and the write -> read
// round trip sits in every tile's dependency chain: scripts/probe/slot_cost.hip): each sample lane packs its nine pieces
// as an A operand — slots 0..7 of k-step 0 and slot 0 of k-step 1 of its half — and a selection matrix (B operand, built
// once per launch) routes slot (k-step, half, e) to piece column pcol(k) + p: two matrix instructions give the piece
// image U[sample][column] as an accumulator tile (every entry one exact piece times 1), and eight conversions pack it
// in the row order of relu' tiles (pack_mask).
struct Pieces3 {
  uint32_t p[3];
};
__device__ __forceinline__ Pieces3 split3v(float v) {  // (truncating split: pieces in the high halves)
  Pieces3 r;
  split3t(v, r.p[0], r.p[1], r.p[2]);
  return r;
}
__device__ __forceinline__ void piece_operand(const Pieces3 &a, const Pieces3 &b, const Pieces3 &c, Frag (&pa)[2]) {
  pa[0].u[0] = pkh(a.p[0], a.p[1]);
  pa[0].u[1] = pkh(a.p[2], b.p[0]);
  pa[0].u[2] = pkh(b.p[1], b.p[2]);
  pa[0].u[3] = pkh(c.p[0], c.p[1]);
  pa[1].u[0] = c.p[2] >> 16;
  pa[1].u[1] = pa[1].u[2] = pa[1].u[3] = 0u;
}
__device__ __forceinline__ void sel_frags(int lane, Frag (&sel)[2]) {
  const int n = lane & 31, hh = lane >> 5;
  auto col = [&](int i) {  // piece i = 3 q + p of this half's input slot q (inputs 2 hh, 2 hh + 1, 4 + hh)
    const int q = i / 3, p = i % 3;
    return (q == 0 ? 6 * hh : (q == 1 ? 6 * hh + 3 : (hh == 0 ? 12 : 16))) + p;
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) sel[0].u[i] = pk(col(2 * i) == n ? 0x3F80u : 0u, col(2 * i + 1) == n ? 0x3F80u : 0u);
  sel[1].u[0] = col(8) == n ? 0x3F80u : 0u;
  sel[1].u[1] = sel[1].u[2] = sel[1].u[3] = 0u;
}
// B operands of the backward from g and this lane's inputs (xa, xb: features 2 hf, 2 hf + 1; xc: feature 4)
__device__ __forceinline__ void piece_frags_mfma(float g, float xa, float xb, float xc, int hf, const Frag (&sel)[2],
                                                 Frag (&ub)[2]) {
  Frag pa[2];
  piece_operand(split3v(g * xa), split3v(g * xb), split3v(hf == 0 ? g * xc : g), pa);
  f32x16 ut = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  ut = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[0].v, sel[0].v, ut, 0, 0, 0);
  ut = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[1].v, sel[1].v, ut, 0, 0, 0);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) ub[s].u[i] = pack_bf16(ut[8 * s + 2 * i], ut[8 * s + 2 * i + 1]);
}

__device__ __forceinline__ constexpr int pcol(int k) { return k < 5 ? 3 * k : 16; }

// publish u[sample n][k] = g * x~_k for this lane's three k (half 0: k = 0, 1, 4; half 1: k = 2, 3 and 5, where
// x~_5 = 1) as exact bf16 pieces in the wave's piece image
__device__ __forceinline__ void publish_pieces(unsigned short (*ubf)[UROW], float g, float xa, float xb, float xc, int n,
                                               int hf) {
  const float uv[3] = {g * xa, g * xb, hf == 0 ? g * xc : g};
  const int cc[3] = {6 * hf, 6 * hf + 3, hf == 0 ? 12 : 16};  // pcol of k = 2 hf, 2 hf + 1, 4 + hf
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    uint32_t q0, q1, q2;
    split3(uv[q], q0, q1, q2);
    ubf[cc[q] + 0][n] = (unsigned short)q0;
    ubf[cc[q] + 1][n] = (unsigned short)q1;
    ubf[cc[q] + 2][n] = (unsigned short)q2;
  }
}

// B operands of the backward: element j of lane half hf = piece[column n][sample 16 s + 8 (j >> 2) + 4 hf + (j & 3)]
__device__ __forceinline__ void piece_frags(const unsigned short (*ubf)[UROW], int n, int hf, Frag (&ub)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    ub[s].q[0] = 0;
    ub[s].q[1] = 0;
    if (n < COLS && n != 15) {  // column 15 is unused (never written)
      ub[s].q[0] = *reinterpret_cast<const uint64_t *>(&ubf[n][16 * s + 4 * hf]);
      ub[s].q[1] = *reinterpret_cast<const uint64_t *>(&ubf[n][16 * s + 8 + 4 * hf]);
    }
  }
}

__device__ __forceinline__ void backward(const Frag (&ga)[NT][2], const Frag (&ub)[2], f32x16 (&dm)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int s = 0; s < 2; ++s) dm[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[t][s].v, ub[s].v, dm[t], 0, 0, 0);
}

// f32 -> f64 flush of the backward accumulators into the wave's image acc64[j * stride + k] (LDS): the three pieces of
// an input column sit in neighbouring lanes of one 16-lane row; add them (p0 + p1) + p2 with row shifts, then one
// ds_add_f64 per value on the lane that holds piece 0 (every address belongs to exactly one lane: no contention, and
// the order of the additions into an address is the program order of the flushes)
// `first` (wave-uniform): this is the wave's first flush — its image holds nothing yet, the values are STORED (a wave that
// passes `true` for its first flush needs no zeroed image: every slot the epilogues read is written here)
__device__ __forceinline__ void flush(f32x16 (&dm)[NT], double *acc64, int stride, int n, int hf, bool first = false) {
  const bool owner = (n < 15 && (n % 3) == 0) || n == 16;
  const int k = n == 16 ? 5 : n / 3;
  // -DRL_FLUSH_ADD_DPP (A/B build, round 6): the row shifts ride on the additions themselves (v_add_f32_dpp; the compiler
  // keeps a v_mov_b32_dpp per shift when they are written with the builtin — 128 more vector instructions per flush, a
  // quarter of it).  Inline asm, so the matrix instructions' results get their wait states by hand (no interlock, and the
  // hazard recogniser does not look into asm): one statement that names all accumulator tiles and idles 20 cycles comes
  // first, every addition after it in program order through its operand.  Measured: the flush of a wave goes from 1.94
  // to 1.75 us (profiles/r06_critic_step_timeline.txt), the critic chain per step stays where it was at 4,096 / 8,192 /
  // 65,536 lanes (three runs each on one box, to the microsecond) — the default keeps the compiler-scheduled form.
#ifdef RL_FLUSH_ADD_DPP
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(dm[0]), "+v"(dm[1]), "+v"(dm[2]), "+v"(dm[3]));
#endif
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float tot[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // row_shl:1 / row_shl:2: lane i reads lane i + 1 / i + 2 of its 16-lane row (0 past the end of the row);
      // tot = (v + v1) + v2 in this order (float addition commutes bit for bit: the shifted operand is source 0)
#ifdef RL_FLUSH_ADD_DPP
      float sum;
      asm("v_add_f32_dpp %0, %1, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_add_f32_dpp %0, %1, %0 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1"
          : "=&v"(sum)
          : "v"(dm[t][r]));
      tot[r] = sum;
#else  // (default: the shifts as v_mov_b32_dpp)
      const float v = dm[t][r];
      const float v1 = __builtin_bit_cast(
          float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
      const float v2 = __builtin_bit_cast(
          float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x102, 0xf, 0xf, true));
      tot[r] = (v + v1) + v2;
#endif
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) dm[t][r] = 0.0f;
    if (owner && first) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
        acc64[j * stride + k] = (double)tot[r];
      }
    } else if (owner) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
        __hip_atomic_fetch_add(&acc64[j * stride + k], (double)tot[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    }
  }
}

}  // namespace bt
