// kernels.h -- launch interface of the gfx950 kernels (raw device pointers + stream).
// Used by the C-ABI wrappers (capi_kernels.cpp) and by the model driver (engine.cpp).
//
// `dyn` arguments: optional device pointer to the sequence's current context length (u32).  When
// non-null the kernel adds *dyn to the position-like scalar it documents.  This is what lets one
// captured hipGraph be replayed for every decode step (SURVEY.md build-plan step 6): the only thing
// that changes between steps lives in device memory.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/uzu_hip.h"

namespace uzu {
namespace k {

enum DT : uint32_t { BF16 = UZU_BF16, F32 = UZU_F32 };

struct NormParams;
// ---------------------------------------------------------------- matmul (quantised / full-precision B)
struct MatmulParams {
    const void* a;      // [m,k] input dtype
    const void* b;      // codes [n, k*bits/8] or full precision [n,k]
    const void* scales; // weights dtype [n, groups]
    const void* biases; // weights dtype [n, groups] (ScaleBias)
    const uint8_t* zero_points;
    void* d;            // [m,n] output dtype
    const void* bias;   // weights dtype [n]
    const uint32_t* gather; // [m,n] or null
    uint32_t w_dt, a_dt, d_dt;
    uint32_t b_kind;    // uzu_matmul_b_kind
    uint32_t bits;      // 4 / 8 (quantised)
    uint32_t group_size;
    uint32_t signed_codes;
    float ab_scale;
    uint32_t accumulate;
    uint32_t has_soft_cap;
    float soft_cap;
    uint32_t m, n, k;
    // ---- fused prologue / epilogue options used by the engine (all 0 through the C ABI) ----
    uint32_t act_mul;   // epilogue: rows [0,n/2) = up, [n/2,n) = gate; writes d[m, n/2] = up * act(gate)
    uint32_t act_type;
    // Offset-term tables of the large-tile prefill GEMM (k_gemm128.hip) supplied by the caller instead of its pre-pass launch:
    // pre_coef [groups][n] f32 depends on the weights only (gemm_coef_table, once at load); pre_rowsum [groups][Mp] f32, Mp = m rounded up
    // to 4, pad rows zero = the group row sums of A, written by the kernel that produced A (NormParams::rowsum_out).  Both given: no pre-pass.
    const float* pre_rowsum;
    const float* pre_coef;
    // pre_rowsum may hold 2^x partial sums per quant group, [groups << x][Mp] (a producer whose workgroups own less than a group of a row: the gated
    // epilogue's 64 columns, a DeltaNet head narrower than the group): the offset term then walks the parts against the group's coefficient row
    uint32_t rowsum_parts_log2;
    // act_mul: the epilogue also files the f32 sums of the 64 gated columns it writes per row, [n / 2 / 64][Mp] (Mp = m rounded up to 4, pad rows zero) --
    // pre_rowsum (parts of 64 columns) of the GEMM that reads D next, which then needs no pre-pass launch.  The large-tile kernel only; null = not filed.
    float* gated_rowsum_out;
    // The Normalization that reads D next (engine prefill: out-projection -> pre-MLP norm, down-projection -> the next layer's pre-mixer norm; its
    // `input` is D).  A split-K GEMM then finishes with ONE launch that adds the partial tiles, applies the epilogue, stores D and normalises the rows
    // (normalization_from_partials) and sets *post_norm_done; every other kernel choice ignores it and the caller runs the normalisation itself.
    const NormParams* post_norm;
    uint32_t* post_norm_done;
};
bool gemm_coef_table_supported(const MatmulParams& p);                          // the large-tile kernel's quantisation family
uzu_status gemm_coef_table(hipStream_t s, const MatmulParams& p, float* coef); // coef[g][n]; p needs b / scales / biases / zero_points, n, k, bits, group_size, b_kind
// runtime.hip: grow-only per-stream scratch block; nullptr while `s` is being captured (or when the allocation fails)
void* stream_workspace(hipStream_t s, size_t bytes);
void stream_workspace_release(hipStream_t s);
// `variant` (optional) receives a short label of the kernel instance chosen (for per-kernel profiles)
uzu_status matmul(hipStream_t s, const MatmulParams& p, int num_cus, const char** variant = nullptr);
// true when matmul() can run `p` with the fused GatedActMul epilogue (act_mul = 1: D = [m, n/2]); otherwise run the matmul and
// gated_act_mul separately
bool matmul_act_mul_supported(hipStream_t s, const MatmulParams& p, int num_cus);
// k_gemv_rows.hip: 2 <= M <= 16 activation rows against int4 codes (group % 128 == 0) as ONE pass over the weights on the matrix cores
bool gemv_rows_mfma_supported(const MatmulParams& p);
// Normalization (RMS; ShortcutMode Copy / Add; normalization.rs:56-125) of the m <= 16 activation rows as the PROLOGUE of the few-rows kernel
// (speculative verify passes, prefill tails: round 5): every workgroup normalises the rows while it stages them, with the element mapping and
// reduction order of normalization_kernel (bit-identical rows); workgroup 0 also writes the residual rows and, if asked, the normalised rows.
struct RowsNorm {
    const float* scales;         // f32 [k] or null (plain RMS norm)
    float eps, offset;
    uint32_t full_layer, residual_add;
    const uint16_t* shortcut_in; // residual_add: bf16 [m, k], added to the rows first
    uint16_t* shortcut_out;      // bf16 [m, k] or null: the (summed) rows, i.e. the new residual -- must not alias shortcut_in (other workgroups still read it)
    uint16_t* normed_out;        // bf16 [m, k] or null: the normalised rows (for a second linear that reads the same rows)
};
bool gemv_rows_norm_supported(const MatmulParams& p);
uzu_status gemv_rows_mfma(hipStream_t s, const MatmulParams& p, const RowsNorm* norm = nullptr);
bool gemm_q_mfma_supported(const MatmulParams& p);   // k_gemm.hip: M >= 20, bf16 activations, int4/int8 codes, group % 64 == 0
uzu_status gemm_q_mfma(hipStream_t s, const MatmulParams& p, int num_cus);
// k_gemm128.hip: 128 x 128 tiles, M >= 128, group 64 / 128 / 256; `workspace` holds the activation row-sum pieces (+ split-K partials)
bool gemm_q_mfma128_supported(const MatmulParams& p, int num_cus);
size_t gemm_q_mfma128_workspace_bytes(const MatmulParams& p, int num_cus);
uzu_status gemm_q_mfma128(hipStream_t s, const MatmulParams& p, int num_cus, void* workspace);
void gemm_q_mfma128_plan_query(const MatmulParams& p, int num_cus, uint32_t* large_tile, uint32_t* form, uint32_t* splits, uint32_t* workgroups);
extern unsigned long long* g_gemm128_dbg; // profiling aid (tools/kbench KB_GEMM_DBG)
size_t matmul_algorithmic_bytes(const MatmulParams& p); // codes + scales + correction + A + D, SURVEY.md §8d

// ---------------------------------------------------------------- normalization
struct NormParams {
    const void* input; // null => in place
    const void* scales;
    const void* biases;
    void* output;
    void* shortcut;
    uint32_t io_dt, affine_dt;
    uint32_t batch_size, element_count;
    float epsilon, scale_offset, post_layer_scalar;
    uint32_t subtract_mean, full_layer, copy_to_shortcut, residual_add, scale_residual_sum, scale_output;
    // optional: the sums of the OUTPUT row over groups of `rowsum_group` elements, f32 [element_count / rowsum_group][Mp], Mp = batch_size
    // rounded up to 4 (pad rows zeroed) -- MatmulParams::pre_rowsum of the GEMM that consumes the rows (see normalization_rowsum_supported)
    float* rowsum_out;
    uint32_t rowsum_group;
};
uzu_status normalization(hipStream_t s, const NormParams& p);
bool normalization_rowsum_supported(uint32_t element_count, uint32_t group);
// rows that are still split-K partial tiles [splits][total] f32 of the GEMM producing them (`d` = that GEMM's bf16 output, `bias` its bf16 epilogue bias or null)
struct NormPartials {
    const float* partials;
    uint32_t splits;
    size_t total;
    const uint16_t* bias;
    uint16_t* d;
};
bool normalization_from_partials_supported(const NormParams& p, const NormPartials& sp);
uzu_status normalization_from_partials(hipStream_t s, const NormParams& p, const NormPartials& sp);

uzu_status qkv_norm(hipStream_t s, void* qkv, uint32_t dt, const float* scales, uint32_t batch_size,
                    uint32_t total_heads, uint32_t head_dim, float epsilon, float scale_offset, uint32_t head_offset,
                    uint32_t head_count, uint32_t full_layer);

// dyn: kv_token_offset += *dyn; cos/sin row index += *dyn (tables indexed by absolute position)
// kv_rows_fixed: the K / V rows are kv_token_offset + batch_idx whatever *dyn says (ring state: the suffix region sits behind the ring);
// the cos / sin row index still moves with *dyn (absolute token positions)
uzu_status attention_prepare(hipStream_t s, const uint16_t* qkv, uint16_t* queries, uint16_t* keys, uint16_t* values,
                             const float* cosines, const float* sines, uint32_t num_q_heads, uint32_t num_kv_heads,
                             uint32_t head_dim, uint32_t rope_dim, uint32_t kv_token_offset, uint32_t batch_dim,
                             uint32_t has_kv, const uint32_t* dyn, uint32_t kv_rows_fixed = 0, const uint32_t* trie = nullptr);
// QKVNorm of the query / key / value heads + AttentionPrepare in one launch (k_elementwise.hip::attention_prepare_normed_kernel; bit-identical to the separate launches)
struct PrepNorm {
    const float* scales;
    float epsilon, scale_offset;
    uint32_t full_layer, present;
};
bool attention_prepare_normed_supported(uint32_t head_dim);
uzu_status attention_prepare_normed(hipStream_t s, const uint16_t* qkv, uint16_t* queries, uint16_t* keys, uint16_t* values, const float* cosines, const float* sines,
                                    const PrepNorm& qn, const PrepNorm& kn, const PrepNorm& vn, uint32_t num_q_heads, uint32_t num_kv_heads, uint32_t head_dim, uint32_t rope_dim,
                                    uint32_t kv_token_offset, uint32_t batch_dim, uint32_t has_kv, const uint32_t* dyn, uint32_t kv_rows_fixed = 0, const uint32_t* trie = nullptr);
// trie (speculated tree, {trie_start, trie_end, height} per row): the RoPE position of row i is the base + height_i (transformer.rs:247)
// instead of base + i; the K / V rows still go to consecutive cache rows (DFS order)
// AttentionState::encode_accept on a Ring (state.rs:200-219) for a flat full accept of `batch` suffix rows, driven by the device-resident
// count n = *accepted of tokens accepted before: suffix row idx (at ring_window + idx) -> ring slot (n + idx) % ring_window; when batch >
// ring_window only the last ring_window rows survive (the reference's sequential copies overwrite the earlier ones)
uzu_status kv_ring_insert(hipStream_t s, void* keys, void* values, uint32_t dt, const uint32_t* accepted, uint32_t batch, uint32_t ring_window,
                          uint32_t element_dim);

struct AttentionParams {
    const void* queries;
    const void* keys;
    const void* values;
    uint32_t dt;
    uint32_t head_dim, gqa_factor, sequence_length; // dyn: sequence_length += *dyn
    uint32_t k_head_stride, k_seq_stride, v_head_stride, v_seq_stride;
    uint32_t is_kv_cache_ring, ring_offset, ring_length;
    float scale;
    uint32_t is_sliding_window, sliding_window_size;
    const void* sinks;
    uint32_t num_heads, suffix_length, is_causal;
    const uint32_t* dyn;
    // Ring KV state driven from the device-resident context length (engine; AttentionStateType::Ring, state.rs:16-55): when non-zero (with
    // `dyn`), the prefix is a ring of `ring_window` rows and n = *dyn tokens have been accepted so far, so sequence_length = ring_window +
    // suffix_length, ring_length = min(n, W), ring_offset = n > W ? (n - W) % W : 0 (what encode_accept's offset / length bookkeeping
    // amounts to, state.rs:200-219) -- a replayed graph cannot carry host-side ring parameters.
    uint32_t ring_window;
    // Speculative tree (is_trie): one {trie_start, trie_end, height} node per suffix token (gpu_types/trie.rs); a suffix key sits at position
    // suffix_position + height and is visible to the queries trie_start .. trie_end (mask.rs:21-29); null = linear suffix
    const uint32_t* trie;
};
// the (sequence_length, ring_offset, ring_length) a kernel works with: `dyn` applied (host code only reads the fields)
__host__ __device__ inline void attention_resolve_dyn(AttentionParams& a) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (a.dyn) {
        const uint32_t n = *a.dyn;
        if (a.ring_window) {
            const uint32_t W = a.ring_window;
            a.sequence_length = W + a.suffix_length;
            a.is_kv_cache_ring = 1;
            a.ring_length = n < W ? n : W;
            a.ring_offset = n > W ? (n - W) % W : 0u;
        } else {
            a.sequence_length += n;
        }
        a.dyn = nullptr;
    }
#endif
}
uzu_status attention_single_pass(hipStream_t s, const AttentionParams& p, void* out);
bool attention_prefill_mfma_supported(const AttentionParams& p); // k_attention_mfma.hip: causal bf16 prefill tiles on the matrix cores
// gate / gate_done (optional): when the pass ends in the key-split merge, SigmoidGate of `gate` (bf16, the layout of `out`) is applied there and *gate_done = 1
// (bit-identical to the separate launch); otherwise the caller runs sigmoid_gate itself
uzu_status attention_prefill_mfma(hipStream_t s, const AttentionParams& p, void* out, const void* gate = nullptr, uint32_t* gate_done = nullptr);
uzu_status attention_two_pass1(hipStream_t s, const AttentionParams& p, float* partials, float* sums, float* maxs);
uzu_status attention_two_pass2(hipStream_t s, const float* partials, const float* sums, const float* maxs, void* out,
                               uint32_t dt, uint32_t head_dim, uint32_t num_heads, uint32_t suffix_length);

uzu_status kv_cache_update(hipStream_t s, void* keys, void* values, uint32_t dt, const uzu_kv_copy* copies_host,
                           uint32_t copy_count, uint32_t element_dim);
uzu_status sigmoid_gate(hipStream_t s, const void* gate, void* output, uint32_t dt, uint32_t total);
uzu_status gated_act_mul(hipStream_t s, const void* act_operand, const void* value_operand, void* fp_out, uint32_t dt,
                         uint32_t gated_dim, uint32_t batch_dim, uint32_t value_offset, uint32_t value_row_stride,
                         uint32_t act_type, uint32_t interleaved);
// one decode row of an RHT MLP: OutputRht of the up projection's halves (+ bias), GatedActMul, InputRht for the down projection (engine.hip)
uzu_status rht_mlp_join(hipStream_t s, const uint16_t* up_row, const uint32_t* up_out_bits, const uint16_t* up_bias, const uint32_t* down_in_bits, uint16_t* out,
                        uint32_t hidden, uint32_t act_type);
// OutputRht (+ bias) of one or two raw output rows in place; row 0's first conv_dim channels go on through DeltaNetConvUpdate (conv_w non-null)
uzu_status rht_out_rows(hipStream_t s, uint16_t* row0, const uint32_t* bits0, const uint16_t* bias0, uint32_t n0, uint16_t* row1, const uint32_t* bits1, const uint16_t* bias1,
                        uint32_t n1, const float* conv_w, const float* conv_b, float* conv_state, uint32_t kernel_size, uint32_t conv_dim);
uzu_status quantized_embedding_lookup(hipStream_t s, const uint32_t* token_ids, const uint8_t* weights,
                                      const void* scales, const uint8_t* zero_points, const void* biases, void* output,
                                      uint32_t dt, uint32_t batch_size, uint32_t vocab_size, uint32_t model_dim,
                                      float input_scale, uint32_t group_size, uint32_t bits, uint32_t method);
uzu_status full_precision_embedding_lookup(hipStream_t s, const uint32_t* token_ids, const void* weights, void* output,
                                           uint32_t dt, uint32_t batch_size, uint32_t vocab_size, uint32_t model_dim,
                                           float input_scale);
uzu_status logit_transform(hipStream_t s, void* logits, uint32_t dt, uint32_t length, float scale, float soft_cap,
                           uint32_t has_soft_cap);
uzu_status tensor_add_bias(hipStream_t s, const void* input, const void* bias, void* output, uint32_t dt,
                           uint32_t bias_dt, uint32_t num_cols, uint32_t length);
uzu_status tensor_add_scale(hipStream_t s, const void* input, const void* bias, void* output, uint32_t dt,
                            uint32_t num_cols, uint32_t length, float scale);
uzu_status tensor_add_swap(hipStream_t s, void* skip, void* main_buf, uint32_t dt, uint32_t length);
uzu_status tensor_copy(hipStream_t s, const void* src, void* dst, uint32_t dt, uint32_t length);
// greedy UnifiedSampling; scratch: >= 2 * batch * 1024 u32 words (device)
uzu_status argmax(hipStream_t s, const void* logits, uint32_t dt, uint32_t* output, uint32_t vocab_size,
                  uint32_t batch_size, void* scratch);
size_t argmax_scratch_bytes(uint32_t batch_size);
// ActivationTransform (k_activation_transform.hip): op 0 InputRht, 1 OutputRht, 2 Quantize, 3 QuantizeWithGroupSums; input null = in place
uzu_status activation_transform(hipStream_t s, const void* input, void* fp_out, int8_t* q_out, float* scales_out, int32_t* group_sums_out,
                                const int32_t* rht_factors, uint32_t dt, uint32_t batch_size, uint32_t element_count, uint32_t op,
                                uint32_t activation_scale_group_size, uint32_t sum_group_size);
// MatmulA::Int8Symmetric: p.a is ignored, the activations are a_q [m,k] int8 with a_scales [m, k / a_group_size]
uzu_status matmul_a8(hipStream_t s, const MatmulParams& p, const int8_t* a_q, const float* a_scales, uint32_t a_group_size);
// UnifiedSampling with any of: grammar bitmask, temperature, top-k / top-p / min-p, Gumbel-max noise (k_sampling.hip).
// `scratch` (unified_sampling_scratch_bytes) is only used when no filter is set (two-level arg-max over 256 workgroups).
struct UnifiedSamplingParams {
    const void* logits;      // [batch, vocab] of dt
    uint32_t dt;
    uint32_t* output;        // [batch]
    const uint64_t* seeds;   // [batch] or null (greedy)
    const uint32_t* bitmask; // [batch, ceil(vocab / 32)] or null
    uint32_t has_temperature, has_top_k, has_top_p, has_min_p;
    float temperature;
    uint32_t top_k;
    float top_p, min_p;
    uint32_t vocab_size, batch_size;
};
uzu_status unified_sampling(hipStream_t s, const UnifiedSamplingParams& p, void* scratch);
// PRng::derive (sampling/prng.rs): *out = finaliser(base + *position + offset)
uzu_status derive_seed(hipStream_t s, uint64_t base, const uint32_t* position, uint32_t offset, uint64_t* out);
// seeds of the nodes of a speculated tree: out[i] = PRng(base).derive(*position + height_i)
uzu_status derive_tree_seeds(hipStream_t s, uint64_t base, const uint32_t* position, const uint32_t* trie, uint32_t nodes, uint64_t* out);
size_t unified_sampling_scratch_bytes(uint32_t batch_size);

// ---------------------------------------------------------------- gated delta net
uzu_status delta_net_conv_update(hipStream_t s, const float* conv_weight, const float* bias, uint16_t* in_out,
                                 float* state, uint32_t kernel_size, uint32_t conv_dim, uint32_t state_stride);
uzu_status delta_net_update(hipStream_t s, const uint16_t* in_proj, const float* a_log, const float* dt_bias,
                            const float* norm_weight, float* state, uint16_t* out, uint32_t num_v_heads,
                            uint32_t num_k_heads, uint32_t head_k_dim, uint32_t head_v_dim, uint32_t key_dim,
                            uint32_t value_dim, float norm_epsilon);
uzu_status conv1d_pack(hipStream_t s, const float* state_in, const uint16_t* x, float* padded, uint32_t state_stride,
                       uint32_t row_stride, uint32_t suffix_len, uint32_t num_channels);
uzu_status delta_net_conv_scan(hipStream_t s, const float* conv_padded, const float* conv_weight, const float* bias,
                               uint16_t* in_proj, float* state_out, uint32_t suffix_len, uint32_t kernel_size,
                               uint32_t row_stride, uint32_t state_stride, uint32_t conv_dim, uint32_t out_stride);
uzu_status delta_net_prefill_prep(hipStream_t s, const uint16_t* in_proj, const float* a_log, const float* dt_bias,
                                  float* q_norm_out, float* k_norm_out, float* beta_out, float* decay_out,
                                  uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_k_dim, uint32_t key_dim,
                                  uint32_t value_dim, uint32_t suffix_len);
uzu_status delta_net_prefill(hipStream_t s, const float* q_norm, const float* k_norm, const float* beta,
                             const float* decay, const uint16_t* in_proj, float* state, uint16_t* out,
                             uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_k_dim, uint32_t head_v_dim,
                             uint32_t key_dim, uint32_t value_dim, uint32_t suffix_len);
// Conv1dPack + DeltaNetConvScan without the packed f32 buffer (engine prefill); `halo` holds delta_net_conv_fused_workspace_floats floats
size_t delta_net_conv_fused_workspace_floats(uint32_t suffix_len, uint32_t kernel_size, uint32_t conv_dim);
uzu_status delta_net_conv_fused(hipStream_t s, uint16_t* in_proj, const float* conv_weight, const float* bias, float* state, float* halo, uint32_t suffix_len,
                                uint32_t kernel_size, uint32_t conv_dim, uint32_t out_stride);
// out-of-place form (round 6; kernel size 4): the conv'd channels go to conv_out [suffix_len][conv_dim] bf16, the in-projection rows stay raw, no halo launch; the carried
// state is only READ -- delta_net_prefill_chunked_fused (its conv_state argument) writes the next one.  Bit-identical rows.
bool delta_net_conv_out_of_place_supported(const uint16_t* in_proj, const float* conv_weight, const float* bias, const uint16_t* conv_out, uint32_t kernel_size, uint32_t conv_dim,
                                           uint32_t in_stride);
uzu_status delta_net_conv_out_of_place(hipStream_t s, const uint16_t* in_proj, const float* conv_weight, const float* bias, const float* state, uint16_t* conv_out,
                                       uint32_t suffix_len, uint32_t conv_dim, uint32_t in_stride);
// chunked form (k_deltanet_chunk.hip): 32-token chunks, T / P matrices built in parallel, four dense products per chunk
bool delta_net_prefill_chunked_supported(uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_k_dim, uint32_t head_v_dim, uint32_t suffix_len);
size_t delta_net_chunk_workspace_bytes(uint32_t num_v_heads, uint32_t value_dim, uint32_t suffix_len); // T / P matrices + the pieces of a split scan
uzu_status delta_net_prefill_chunked(hipStream_t s, const float* q_norm, const float* k_norm, const float* beta, const float* decay, const uint16_t* in_proj,
                                     float* state, uint16_t* out, float* workspace, uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_v_dim,
                                     uint32_t key_dim, uint32_t value_dim, uint32_t suffix_len);
// the same with DeltaNetPrefillPrep inside the chunk preparation (round 6): q_norm_out / k_norm_out f32 [suffix_len, key_dim] are written for the scan, beta / decay stay
// in the kernel; bit-identical to delta_net_prefill_prep + delta_net_prefill_chunked
bool delta_net_prefill_prep_fused_enabled(); // UZU_HIP_TUNE=prep_fused=0: the separate launch
// conv_rows != null: the conv'd q | k | v channels live there ([suffix_len][2 key_dim + value_dim] bf16: delta_net_conv_out_of_place) instead of in the in-projection rows,
// and conv_state f32 [conv_dim][3] receives the next pass's carried state X[T - 3 ..] from the raw rows
uzu_status delta_net_prefill_chunked_fused(hipStream_t s, const uint16_t* in_proj, const float* a_log, const float* dt_bias, float* q_norm_out, float* k_norm_out, float* state,
                                           uint16_t* out, float* workspace, uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_v_dim, uint32_t key_dim, uint32_t value_dim,
                                           uint32_t suffix_len, const uint16_t* conv_rows = nullptr, float* conv_state = nullptr);
// rowsum_out (optional; production kernel only): the f32 sum of every (token, head)'s rounded outputs at [head][rowsum_row0 + token], row stride rowsum_stride --
// MatmulParams::pre_rowsum of the out-projection in parts of head_v_dim columns; rows [rowsum_row0 + suffix_len, rowsum_pad_to) are zeroed (the pad rows)
uzu_status delta_net_norm_gate(hipStream_t s, uint16_t* in_out, const uint16_t* in_proj, const float* norm_weight,
                               uint32_t num_v_heads, uint32_t head_v_dim, uint32_t value_dim, uint32_t conv_dim,
                               uint32_t total_proj_dim, float norm_epsilon, uint32_t suffix_len, float* rowsum_out = nullptr, uint32_t rowsum_stride = 0,
                               uint32_t rowsum_row0 = 0, uint32_t rowsum_pad_to = 0);

// ---------------------------------------------------------------- reference-order mode (k_exact.hip, k_matmul.hip::matmul_ref_kernel)
// UZU_HIP_EXACT=1 / uzu_hip_set_exact(1): every reduction kernel (matmul, Normalization, QKVNorm, attention, DeltaNet) runs one thread
// per reduction in the reference's loop order => a forward pass reproduces the CPU backend bit for bit.  The launchers above route to
// these themselves; the fused / matrix-core / streaming variants report "not supported" while the mode is on.
bool exact_mode();
void set_exact_matmul(bool enabled);
uzu_status normalization_exact(hipStream_t s, const NormParams& p);
uzu_status qkv_norm_exact(hipStream_t s, void* qkv, uint32_t dt, const float* scales, uint32_t batch_size, uint32_t total_heads, uint32_t head_dim, float epsilon,
                          float scale_offset, uint32_t head_offset, uint32_t head_count, uint32_t full_layer);
uzu_status attention_single_pass_exact(hipStream_t s, const AttentionParams& a, void* out);
uzu_status attention_two_pass1_exact(hipStream_t s, const AttentionParams& a, float* partials, float* sums, float* maxs);
uzu_status attention_two_pass2_exact(hipStream_t s, const float* partials, const float* sums, const float* maxs, void* out, uint32_t dt, uint32_t head_dim, uint32_t num_heads,
                                     uint32_t suffix_length);
uzu_status delta_net_update_exact(hipStream_t s, const uint16_t* in_proj, const float* a_log, const float* dt_bias, const float* norm_weight, float* state, uint16_t* out,
                                  uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_k_dim, uint32_t head_v_dim, uint32_t key_dim, uint32_t value_dim, float norm_epsilon);
// ---- the tree speculators' kernels (k_speculator.hip; cpu/kernel/attention/ancestor_attention.rs, cpu/kernel/weaver/*.rs) ----
uzu_status ancestor_attention(hipStream_t s, const uint16_t* prefix_kv, uint16_t* node_kv, const uint16_t* current_qkv, const float* cosines, const float* sines,
                              const uint32_t* node_metadata, const uint32_t* ancestor_indices, const uint32_t* ancestor_counts, const uint32_t* node_indices,
                              uint16_t* output, uint32_t rows, uint32_t prefix_length, uint32_t ancestor_stride, uint32_t node_capacity, uint32_t max_depth, float scale,
                              uint32_t num_heads, uint32_t head_dim);
uzu_status weaver_frontier_select(hipStream_t s, uint32_t* frontier, uint32_t* packed_tree, uint32_t* slot_ancestors, uint32_t* node_token_ids, uint32_t* node_metadata,
                                  uint32_t* node_ancestor_indices, uint32_t* node_valid, const uint32_t* candidate_pool_ids, const float* candidate_pool_logits,
                                  uint32_t* node_candidate_ids, float* node_candidate_logits, uint32_t frontier_capacity, uint32_t tree_slot_count, uint32_t node_count,
                                  uint32_t batch_start_slot, uint32_t ancestor_stride, uint32_t max_depth, uint32_t lookahead_count, uint32_t candidate_depth_count,
                                  uint32_t candidates_per_depth);
uzu_status weaver_frontier_insert_children(hipStream_t s, const uint32_t* packed_tree, const uint32_t* node_metadata, const uint32_t* node_valid, const uint32_t* child_ids,
                                           const float* child_logprobs, uint32_t* frontier, uint32_t frontier_capacity, uint32_t tree_slot_count, uint32_t node_count,
                                           uint32_t expand_width);
uzu_status weaver_top_children(hipStream_t s, const uint16_t* residual_logits, const float* candidate_logits, const uint32_t* candidate_ids, const uint64_t* depth_seeds,
                               const uint32_t* node_metadata, uint32_t* output_token_ids, float* output_model_logprobs, uint32_t rows, uint32_t candidates,
                               uint32_t expand_width, uint32_t vocab_size);
// RadixTopKSmall (cpu/kernel/radix_top_k_small.rs:25-79): per row the k <= 512 best columns by (value descending under total_cmp, column ascending); exact
uzu_status radix_top_k_small(hipStream_t s, const float* input, uint32_t* output_ids, float* output_scores, uint32_t rows, uint32_t columns, uint32_t k);
// ---- Gated DeltaNet over a speculated token tree (k_deltanet_tree.hip; cpu/kernel/gdn/tree_verify/*.rs, delta_net.rs:334-437) ----
constexpr uint32_t kDnTreeMaxNodes = 32; // nodes of one verify pass (the reference's stream speculates <= 16: stream.rs:550-554)
// ConvTreeScan (do_conv) and / or DeltaNetPrefillPrep in its tree instantiation (do_prep: QKT = bf16, log decays, compact v).  in_proj
// [n, total_proj_dim] bf16; base_state f32 [conv_dim, k-1]; parents i32 [n]; out_proj (optional) = ConvTreeScan's full output rows;
// suffix_state f32 [n, conv_dim, k-1]; q_out / k_out bf16 [n, key_dim]; v_out bf16 [n, value_dim]; beta_out / log_decay_out f32 [n, Hv].
uzu_status delta_net_tree_prep(hipStream_t s, const uint16_t* in_proj, const float* conv_w, const float* conv_b, const float* base_state, const int32_t* parents,
                               uint16_t* out_proj, float* suffix_state, const float* a_log, const float* dt_bias, uint16_t* q_out, uint16_t* k_out, uint16_t* v_out,
                               float* beta_out, float* log_decay_out, uint32_t n, uint32_t kernel_size, uint32_t Hk, uint32_t Hv, uint32_t Dk, uint32_t Dv, bool do_conv,
                               bool do_prep);
// DeltaNetTreeVerify::encode (metal/kernel/gdn/tree_verify.rs:92-187: prefix -> Gram -> solve -> out, batch 1, scale 1, h0 = the SSM state):
// out bf16 [n, Hv, Dv], bit-identical to the CPU kernels' composition
uzu_status delta_net_tree_verify(hipStream_t s, const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint32_t* trie, const float* log_decay, const float* beta,
                                 const float* h0, uint16_t* out, uint32_t n, uint32_t Hk, uint32_t Hv, uint32_t Dk, uint32_t Dv);
// StateAdvance (state_advance.rs): the delta rule over the accepted path on the SSM state in place
uzu_status delta_net_state_advance(hipStream_t s, const uint16_t* k_norm, const uint16_t* v, const float* log_decay, const float* beta, const uint32_t* accepted_indices,
                                   float* state, uint32_t accepted_len, uint32_t Hv, uint32_t Hk, uint32_t Dk);
uzu_status delta_net_prefill_prep_exact(hipStream_t s, const uint16_t* in_proj, const float* a_log, const float* dt_bias, float* q_norm_out, float* k_norm_out, float* beta_out,
                                        float* decay_out, uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_k_dim, uint32_t key_dim, uint32_t value_dim, uint32_t suffix_len);
uzu_status delta_net_prefill_exact(hipStream_t s, const float* q_norm, const float* k_norm, const float* beta, const float* decay, const uint16_t* in_proj, float* state, uint16_t* out,
                                   uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_k_dim, uint32_t head_v_dim, uint32_t key_dim, uint32_t value_dim, uint32_t suffix_len);
uzu_status delta_net_norm_gate_exact(hipStream_t s, uint16_t* in_out, const uint16_t* in_proj, const float* norm_weight, uint32_t num_v_heads, uint32_t head_v_dim, uint32_t value_dim,
                                     uint32_t conv_dim, uint32_t total_proj_dim, float norm_epsilon, uint32_t suffix_len);

// ---------------------------------------------------------------- engine helpers
uzu_status advance_u32(hipStream_t s, uint32_t* counter, uint32_t amount); // *counter += amount
uzu_status fill_u32(hipStream_t s, uint32_t* dst, uint32_t value, uint32_t count);

// ---- Mixture of experts (k_moe.hip; MoeBlock::encode, encodable_block/mlp/moe/mod.rs:204-350); bf16 tensors
struct MoeExpertParams {
    uint32_t d_model, d_ff, gating_sel; // 0 GELU(up), 1 SiLU(up), 2 SwiGLU, 3 GEGLU
    float gate_clip_min, gate_clip_max, up_clip_min, up_clip_max, silu_alpha;
};
uzu_status moe_router_topk(hipStream_t s, const uint16_t* input, const uint16_t* weight, const uint16_t* bias, int32_t* topk_ids, uint16_t* topk_probs, uint32_t t, uint32_t d_model, uint32_t e,
                           uint32_t k, uint32_t renorm);
uzu_status moe_counts_offsets(hipStream_t s, const int32_t* topk_ids, uint32_t* offsets, uint32_t* sum_k_out, uint32_t* partials, uint32_t t, uint32_t e, uint32_t k);
uzu_status moe_scatter_buckets(hipStream_t s, const int32_t* topk_ids, const uint16_t* topk_probs, const uint32_t* offsets, int32_t* bucketed_ids, uint16_t* bucketed_probs, int32_t* tok2row,
                               uint32_t* row_expert_map, uint32_t t, uint32_t e, uint32_t k);
uzu_status moe_gather(hipStream_t s, const uint16_t* x, const int32_t* bucketed_ids, uint16_t* x_perm, const uint32_t* sumk, uint32_t d_model, uint32_t t, uint32_t k);
// `capacity` = rows the launch covers (tokens x active experts); rows past *sumk return at once
uzu_status moe_experts_pass_a(hipStream_t s, const uint16_t* x_perm, const uint32_t* row_expert_map, const uint32_t* sumk, const uint16_t* w13_all, const uint16_t* up_biases, float* hidden_out,
                              const MoeExpertParams& q, uint32_t capacity);
uzu_status moe_experts_down(hipStream_t s, const float* hidden, const uint32_t* row_expert_map, const uint32_t* sumk, const uint16_t* w2_all, const uint16_t* down_biases, uint16_t* y_out,
                            uint32_t d_model, uint32_t d_ff, uint32_t capacity);
uzu_status moe_finalize(hipStream_t s, const int32_t* tok2row, const uint16_t* probs, const uint16_t* y_partial, uint16_t* y, uint32_t t_count, uint32_t d_model, uint32_t k);

} // namespace k
} // namespace uzu
