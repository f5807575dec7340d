// kernels_decode.h -- parameter blocks of the fused batch-1 decode kernels (k_decode.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/uzu_hip.h"
#include "../../include/uzu_model_desc.h"

namespace uzu {
namespace k {

struct DecNorm {
    uint32_t present;
    uint32_t full_layer;
    float eps, offset;
    const float* scales;
};

// Per-workgroup phase timestamps (100 MHz wall clock, s_memrealtime) for tools/timeline.py.  Compiled in only with
// -DUZU_TIMELINE (a separate library variant: the production kernels carry no trace of it).  The stamps live in
// registers and are stored by thread 0 at the very end of the kernel (a store in front of the first loads would sit in
// the same in-order VMEM queue and delay what it is meant to measure).
#ifdef UZU_TIMELINE
#define UZU_TL_FIELD unsigned long long* tl;
#define UZU_TL_SLOTS 8
#define UZU_TL_DECL                                                        \
    unsigned long long tl_t[UZU_TL_SLOTS] = {0, 0, 0, 0, 0, 0, 0, 0};      \
    __shared__ unsigned long long tl_last_;                                \
    __shared__ unsigned tl_cnt_;                                           \
    if (threadIdx.x == 0) tl_last_ = 0ull, tl_cnt_ = 0u
#define UZU_TL_STAMP(slot) (tl_t[slot] = __builtin_amdgcn_s_memrealtime())
// slot 7 = when the LAST wave of the workgroup got here (thread 0's stamps say nothing about the other waves: with batches handed out
// dynamically, or simply uneven rows, wave 0 can be done a microsecond before its workgroup is)
#define UZU_TL_FLUSH(p)                                                                              \
    do {                                                                                             \
        if ((p).tl) {                                                                                \
            const unsigned wg_ = blockIdx.x + gridDim.x * blockIdx.y;                                \
            if (threadIdx.x == 0 && wg_ < 1024)                                                      \
                for (int i_ = 0; i_ < UZU_TL_SLOTS - 1; ++i_) (p).tl[wg_ * UZU_TL_SLOTS + i_] = tl_t[i_]; \
            if ((threadIdx.x & 63) == 0) {                                                           \
                const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();                    \
                atomicMax(&tl_last_, now_);                                                          \
                const unsigned n_ = atomicAdd(&tl_cnt_, 1u);                                         \
                if (n_ + 1 == (blockDim.x * blockDim.y + 63) / 64 && wg_ < 1024) (p).tl[wg_ * UZU_TL_SLOTS + 7] = atomicMax(&tl_last_, 0ull); \
            }                                                                                        \
        }                                                                                            \
    } while (0)
unsigned long long* timeline_next_slot(); // next per-launch block of 1024 x UZU_TL_SLOTS stamps, or null
extern "C" void uzu_hip_debug_set_timeline(unsigned long long* base, uint32_t max_launches);
#else
#define UZU_TL_FIELD
#define UZU_TL_DECL do { } while (0)
#define UZU_TL_STAMP(slot) do { } while (0)
#define UZU_TL_FLUSH(p) do { } while (0)
#endif

struct DecGemvParams {
    UZU_TL_FIELD
    // one or two weight matrices with the same k / quantisation (e.g. qkv_projection || gate_projection)
    const uint8_t* w[2];
    const uint16_t* scales[2];
    const uint16_t* biases[2];
    const uint8_t* zp[2];
    const uint16_t* out_bias[2];
    uint16_t* out[2];
    uint32_t n[2];
    uint32_t k, bits, group_size, b_kind;
    const uint16_t* x; // input row (bf16 [k])
    // Normalization prologue (RMS; ShortcutMode Copy/Add)
    uint32_t norm_plain;          // RMSNorm without scales
    const float* norm_scales;     // non-null => normalise x first
    float norm_eps, norm_offset;
    uint32_t norm_full_layer, residual_add;
    const uint16_t* shortcut_in;
    uint16_t* shortcut_out;
    uint16_t* normed_out;         // optional copy of the normalised row (debug / taps)
    // epilogues
    uint32_t act_mul, act_type;   // out[0][j] = up_j * act(gate_j), n[0] = 2h
    float* part_val;              // arg-max partials, one per workgroup
    uint32_t* part_idx;
    uint32_t part_capacity;       // entries in part_val / part_idx: the grid is clamped to it (0 = unchecked, 8192 in tools)
    float* out_f32;               // tensor parallel: matrix 0 writes f32 partial sums here instead of bf16 into out[0]
    uint32_t wg_batches;          // set by the launcher: batches a wide workgroup owns per round (0 = one per wave)
    // DeltaNetConvUpdate epilogue (in-proj): rows < conv_dim of matrix 0 go through the causal conv + SiLU of their
    // channel (one lane owns a channel: taps read, shifted and written by that lane only) -- conv_update.rs:17-55
    const float* conv_w;          // [conv_dim, ks]
    const float* conv_b;          // [conv_dim] or null
    float* conv_state;            // [conv_dim, ks - 1]
    uint32_t conv_dim, conv_ks;
    // DeltaNet norm-gate prologue (out-proj): x[e] = bf16(o[e] * inv_rms(head) * w[e % dv] * sz[e]) from the f32
    // outputs of delta_dec -- the tail of update.rs:30-143
    const float* dg_o;            // [k] raw delta-rule outputs
    const float* dg_sz;           // [k] SiLU(z)
    const float* dg_w;            // [dv] norm weight
    uint32_t dg_dv;
    float dg_eps;
    // Randomised Hadamard transforms around the Normalization prologue (RHTLinearWrapper, linear/rht_wrapper.rs:215-298; the PRO == 3
    // instances).  Sign factors as ONE BIT per element (bit i of word s: the factor of element 32 s + i is -1), packed at load.
    //   x_rht_bits / x_rht_bias: the row in `x` is the RAW output of an RHT linear -- its OutputRht (butterfly, 1/sqrt(32), factors, rounded
    //     to bf16) and then its bias (rounded again) are applied before anything else (MatmulDOps::rht_factors, kernel.rs:296-303)
    //   in_rht_bits: InputRht of the normalised row (factors, butterfly, 1/sqrt(32), rounded to bf16) = this linear's own input transform
    const uint32_t* x_rht_bits;   // [k / 32] or null
    const uint16_t* x_rht_bias;   // bf16 [k] or null
    const uint32_t* in_rht_bits;  // [k / 32] or null
    // The STRIPE epilogue (PRO == 5, round 5): THIS linear is an RHT linear whose consumer needs its OutputRht first.  A workgroup owns whole 32-row
    // blocks, parks a block's raw rows in LDS, and one wave applies the OutputRht (ep_out_bits: one sign bit per output row, [n / 32] words;
    // act_mul: up blocks then gate blocks) + ep_bias (bf16 [n] or null) and then
    //   act_mul: GatedActMul of the block's (up, gate) pairs + the NEXT linear's InputRht (ep_next_in_bits: [n / 64] words or null) -> out[0][n / 2]
    //   else   : the DeltaNet conv of rows < conv_dim (conv_w / conv_b / conv_state / conv_ks as above, any kernel size) -> out[0][n]
    // -- rht_mlp_join's / rht_out_rows's arithmetic (k_elementwise.hip) without their launches.  gemv_dec_stripe_supported says where it applies.
    const uint32_t* ep_out_bits;
    const uint16_t* ep_bias;
    const uint32_t* ep_next_in_bits;
};
bool gemv_dec_stripe_supported(const DecGemvParams& p, int num_cus);
uint32_t gemv_dec_grid(const DecGemvParams& p, int num_cus, int* lpr_log2, int* R);
uzu_status gemv_dec(hipStream_t s, const DecGemvParams& p, int num_cus, uint32_t* grid_out);
bool gemv_dec_plain_in_rht_supported(uint32_t k, uint32_t bits); // in_rht_bits on a plain (not normalised, not norm-gated) input row
// The weight-streaming engine (k_stream.hip): LDS-DMA loader wave + consumer waves per CU; same arithmetic as gemv_dec.  gemv_dec
// routes to it when gemv_stream_wanted (bandwidth regime, int4 ScaleBias, UZU_DEC_STREAM / uzu_hip_debug_set_decode_stream).
bool gemv_stream_supported(const DecGemvParams& p);
bool gemv_stream_wanted(const DecGemvParams& p);
uzu_status gemv_stream(hipStream_t s, const DecGemvParams& p, int num_cus, uint32_t* grid_out);
// host synchronisation points: UZU_ERR_HIP if a bounded spin of a stream kernel gave up since the last check (its outputs are then
// garbage); costs nothing until a stream kernel has been launched in this process
uzu_status gemv_stream_check();
// Embedding row of the token the commit kernel has just sampled (the next decode step's input row): the lookup of
// quant_embedding.rs:36-116 / full_precision_embedding.rs:17-31 for one token, done by the committing workgroup instead
// of a launch of its own at the head of the next step.  `output` null => plain commit.
struct CommitEmbed {
    const uint8_t* weights;      // packed codes [vocab, dim / pack], or bf16 [vocab, dim] when method == UZU_QUANT_NONE
    const uint16_t* scales;      // bf16 [vocab, groups]
    const uint8_t* zero_points;
    const uint16_t* biases;
    uint16_t* output;            // bf16 [dim]
    uint32_t vocab_size, model_dim, group_size, bits, method;
    float input_scale;
    const uint32_t* token_in;    // non-null: the token was sampled by another kernel (stochastic UnifiedSampling); the partials are ignored
};
uzu_status argmax_commit(hipStream_t s, const float* pv, const uint32_t* pi, uint32_t parts, uint32_t* ctx_len, uint32_t* tokens,
                         uint32_t* out_token, uint32_t* sampled, const CommitEmbed* embed = nullptr);

struct DeltaDecParams {
    UZU_TL_FIELD
    const uint16_t* in_proj;  // post-conv row: [q | k | v | z | beta | a] (the conv ran in the in-proj epilogue)
    const float* a_log;
    const float* dt_bias;
    float* state;             // [Hv, Dv, Dk] f32
    float* o;                 // [Hv * Dv] f32 raw outputs (normalised + gated by the out-proj prologue)
    float* sz;                // [Hv * Dv] f32 SiLU(z)
    uint32_t num_v_heads, num_k_heads, head_v_dim, key_dim, value_dim;
};
struct DecGemvPlan {
    int lpr_log2;                  // lanes per row = 1 << lpr_log2
    int rows_per_lane_group;       // R
    uint32_t steps_per_lane;       // 32-element steps of the row a lane owns
    uint32_t waves;                // waves per workgroup
    uint32_t wave_batches;         // batches (rows_per_lane_group x (64 >> lpr_log2) rows each), rounded up to a multiple of four
    uint32_t wg_batches, workgroups; // a partial round spread over every CU (0, 0: the persistent grid)
};
void gemv_dec_plan_query(const DecGemvParams& p, int num_cus, DecGemvPlan* out); // host arithmetic only
uzu_status delta_dec(hipStream_t s, const DeltaDecParams& p);

struct AttnDecParams {
    UZU_TL_FIELD
    const uint16_t* qkv;   // packed [q heads | k heads | v heads] x head_dim of the new token
    uint16_t* keys;        // cache [tokens, kv_heads, hd]
    uint16_t* values;
    const float* cosines;  // [positions, rope_dim]
    const float* sines;
    const uint32_t* ctx_len;
    DecNorm q_norm, k_norm;
    uint32_t num_heads, gqa_factor, head_dim, rope_dim;
    float scale;
    float* partials;       // [heads, splits, hd]
    float* sums;           // [heads, splits]
    float* maxs;
    uint32_t cache_rows;   // rows the K / V caches are allocated for (> context length): the first loads are clamped to it, not to the context
    // One-launch SDPA decode (round 5): pass 2 (attn_merge_kernel's arithmetic, bit for bit) + SigmoidGate inside the same launch.  Every
    // workgroup publishes its split's partials with write-through stores, draws a ticket of its KV-head group, waits until the group's
    // `splits` workgroups have all drawn theirs, and then merges ITS slice -- gqa-group x head_dim / splits output elements -- of the
    // group's attention rows.  tickets: one monotonic u32 per (kv head, head sub-group), zeroed once at allocation; out != null selects it.
    uint32_t* tickets;
    const uint16_t* gate;  // [heads, hd] or null (SigmoidGate)
    uint16_t* out;         // bf16 [heads, hd]: the attention rows
};
uzu_status attn_dec(hipStream_t s, const AttnDecParams& p, uint32_t splits);
// can attn_dec run pass 2 inside the launch for this geometry?  (every workgroup of a KV-head group must be resident at once: grid <= 2 x CUs;
// the group's gqa x head_dim outputs must split evenly over the workgroups)
bool attn_dec_fused_supported(uint32_t num_heads, uint32_t gqa_factor, uint32_t head_dim, uint32_t splits, int num_cus);
// host sync points: UZU_ERR_HIP if a bounded wait of a fused attn_dec gave up since the last check (its rows are then garbage)
uzu_status attn_dec_check();
// query heads of one KV head a workgroup of attn_dec serves together (the K / V rows are read once for all of them):
// the largest divisor of the GQA factor <= 6 (LDS: 8 KB of merge state per head)
inline uint32_t attn_dec_group_size(uint32_t gqa_factor) {
    for (uint32_t c = 6; c > 1; --c)
        if (gqa_factor % c == 0) return c;
    return 1;
}
uzu_status attn_merge(hipStream_t s, const float* partials, const float* sums, const float* maxs, const uint16_t* gate, uint16_t* out,
                      uint32_t num_heads, uint32_t head_dim, uint32_t splits);

} // namespace k
} // namespace uzu
