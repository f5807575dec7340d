/*
 * uzu_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT).
 *
 * A plain-C restatement of the reference's CPU backend kernels on the transformer forward path
 * (crates/backend-uzu/src/backends/cpu/kernel/**) and of the op order of
 * crates/backend-uzu/src/encodable_block/**.  Same loops, same f32 accumulation order, bf16
 * round-to-nearest-even at every store (`half` 2.7 `bf16::from_f32`), glibc libm for
 * exp/log/tanh/sin/cos/pow (what Rust's `f32::*` lowers to on Linux).
 *
 * PARITY STATUS: the reference is Rust 1.94 (no toolchain in this environment, no network) and
 * ships no golden vectors for this path, so this oracle cannot be checked against outputs of the
 * reference itself: **parity unpinned** except for (a) the literal known answers the reference tree
 * holds (gated_act_mul_test.rs:139-160; gumbel_test.rs: the uniform mapping of the sampler) and the
 * expectations its tests compute in-test, independently of any kernel, and hold the CPU backend to
 * (tensor_add_bias / add_swap / copy, full-precision embedding, the KV-cache copy patterns, gather == dense,
 * DeltaNet prep's compact V, tree prefix sums: tests/test_reference_vectors.py replays each on the
 * reference's own procedural inputs), (b) the published Random123 known-answer vectors of Philox4x32-10,
 * the sampler's generator, and (c) independent float64 NumPy references of the same math, mirroring the
 * reference's own `reference_attention` style checks (tests/test_oracle_*.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 * The product (uzu_amd/) never does.
 *
 * Build: `make -C oracle` -> oracle/_build/liboracle.so  (gcc -O2 -ffp-contract=off: Rust never
 * contracts a*b+c into an FMA, so neither may we).
 */
#ifndef UZU_ORACLE_H
#define UZU_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../include/uzu_model_desc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* element types a kernel can be instantiated with (data_type.rs:5-35, subset used by the LM path) */
typedef enum { ORC_BF16 = 0, ORC_F32 = 2 } orc_dtype;

void orc_set_threads(int n); /* OpenMP threads over independent outputs; results are bit-identical for any n */
int orc_get_max_threads(void);

/* half::bf16 conversions */
uint16_t orc_f32_to_bf16(float v);
float orc_bf16_to_f32(uint16_t v);
void orc_f32_to_bf16_array(const float* src, uint16_t* dst, size_t n);
void orc_bf16_to_f32_array(const uint16_t* src, float* dst, size_t n);

/* ---- MatmulKernel (cpu/kernel/matmul/kernel.rs:56-307) ---- */
typedef struct {
    /* A: MatmulA::FullPrecision { values, offset } */
    const void* a;
    uint32_t a_dtype; /* input_data_type */
    /* B */
    const void* b;             /* codes (u8 / u32 words) or full-precision [n,k] */
    const void* scales;        /* weights_data_type [n, groups] */
    const void* biases;        /* weights_data_type [n, groups]  (ScaleBias) */
    const uint8_t* zero_points; /* (ScaleZeroPoint) */
    uint32_t w_dtype;          /* weights_data_type */
    uint32_t method;           /* uzu_quant_method */
    uint32_t bits;             /* 4 / 8 (quantized) */
    uint32_t group_size;
    uint32_t signed_codes;
    uint32_t b_transpose;      /* full precision only */
    uint32_t b_leading_dimension; /* 0 => default */
    /* D + MatmulDOps */
    void* d;
    uint32_t d_dtype; /* output_data_type */
    float ab_scale;
    uint32_t accumulate;
    const void* bias; /* weights_data_type [n] or NULL */
    uint32_t has_soft_cap;
    float soft_cap;
    const uint32_t* gather_indices; /* [m,n] or NULL */
    uint32_t m, n, k;
    /* A: MatmulA::Int8Symmetric { values, scales, group_sums, group_size } (matmul_a.rs:9-14) when a == NULL */
    const int8_t* a_q;      /* [m, k] */
    const float* a_scales;  /* [m, ceil(k / a_group_size)] */
    uint32_t a_group_size;  /* 32 / 64 / 128 */
    /* MatmulDOps::rht_factors (d_ops.rs:3-9): output RHT in place on D after the store, THEN the bias (kernel.rs:296-303) */
    const int32_t* rht_factors; /* [n] or NULL */
} orc_matmul_args;
void orc_matmul(const orc_matmul_args* args);

/* ---- ActivationTransform (cpu/kernel/activation_transform/activation_transform.rs:43-136, mod.rs:9-47) ----
 * Randomised Hadamard transform over stripes of 32 elements (+-1 factors before the butterfly for InputRht / Quantize*, after it
 * for OutputRht), optionally followed by symmetric int8 quantisation per `activation_scale_group_size` elements with i32 code
 * sums per `sum_group_size`.  op: 0 InputRht, 1 OutputRht, 2 Quantize, 3 QuantizeWithGroupSums (gpu_types/activation_transform.rs).
 * input == NULL <=> in place on fp_out. */
void orc_activation_transform(const void* input, void* fp_out, int8_t* q_out, float* scales_out, int32_t* group_sums_out,
                              const int32_t* rht_factors, uint32_t dtype, uint32_t batch_size, uint32_t element_count, uint32_t op,
                              uint32_t activation_scale_group_size, uint32_t sum_group_size);

/* ---- Normalization (cpu/kernel/normalization/normalization.rs:7-126) ---- */
typedef struct {
    const void* input; /* NULL => in place (reads output) */
    const void* scales; /* affine dtype */
    const void* biases;
    void* output;
    void* shortcut; /* copy_to_shortcut */
    uint32_t io_dtype;     /* InputT == OutputT */
    uint32_t affine_dtype; /* AffineT */
    uint32_t batch_size, element_count;
    float epsilon, scale_offset, post_layer_scalar;
    uint32_t subtract_mean, full_layer, copy_to_shortcut, residual_add;
    uint32_t scale_residual_sum, scale_output;
} orc_norm_args;
void orc_normalization(const orc_norm_args* args);

/* ---- QKVNorm (cpu/kernel/attention/qkv_norm.rs:7-77), in place ---- */
void orc_qkv_norm(void* qkv, uint32_t dtype, const float* scales /* may be NULL */, uint32_t batch_size,
                  uint32_t total_heads, uint32_t head_dim, float epsilon, float scale_offset, uint32_t head_offset,
                  uint32_t head_count, uint32_t full_layer);

/* ---- host RoPE table (encodable_block/mixer/attention/rope.rs:13-114); out f32 [n_pos, head_dim] ---- */
void orc_rope_tables(const uzu_rope_desc* rope, const uint32_t* token_positions, uint32_t n_pos, float* cosines,
                     float* sines);

/* ---- AttentionPrepare (cpu/kernel/attention/attention_prepare.rs:7-126), ElementT = bf16 ---- */
void orc_attention_prepare(const uint16_t* qkv, uint16_t* queries, uint16_t* keys, uint16_t* values,
                           const float* cosines, const float* sines, uint32_t num_q_heads, uint32_t num_kv_heads,
                           uint32_t head_dim, uint32_t rope_dim /* 0 => no rope */, uint32_t kv_token_offset,
                           uint32_t batch_dim, uint32_t has_kv);

/* ---- attention cores (attention_single_pass.rs / attention_two_pass.rs / mask.rs) ---- */
typedef struct {
    const void* queries; /* [heads, suffix, hd] */
    const void* keys;
    const void* values;
    uint32_t dtype;
    uint32_t head_dim, gqa_factor, sequence_length;
    uint32_t k_head_stride, k_seq_stride, v_head_stride, v_seq_stride;
    uint32_t is_kv_cache_ring, ring_offset, ring_length;
    float scale;
    uint32_t is_sliding_window, sliding_window_size;
    const void* sinks; /* dtype [heads] or NULL */
    uint32_t num_heads, suffix_length, is_causal;
    const uint32_t* trie; /* is_trie: {trie_start, trie_end, height} per suffix token (gpu_types/trie.rs), or NULL */
} orc_attention_args;
void orc_attention_single_pass(const orc_attention_args* a, void* out /* [suffix, heads, hd] */);
void orc_attention_two_pass1(const orc_attention_args* a, float* partials, float* sums, float* maxs);
void orc_attention_two_pass2(const float* partials, const float* sums, const float* maxs, void* out, uint32_t dtype,
                             uint32_t head_dim, uint32_t num_heads, uint32_t suffix_length);

/* ---- small ops ---- */
typedef struct { uint32_t source, destination; } orc_copy;
void orc_kv_cache_update(void* keys, void* values, uint32_t dtype, const orc_copy* copies, uint32_t copy_count,
                         uint32_t element_dim);                                             /* kv_cache_update.rs */
void orc_sigmoid_gate(const void* gate, void* output, uint32_t dtype, uint32_t total);     /* sigmoid_gate.rs */
float orc_activate(uint32_t act_type, float x, uint32_t dtype);                             /* activation_type.rs */
void orc_gated_act_mul_rht(const void* act_operand, const void* value_operand, void* fp_out, int8_t* q_out, float* scales_out, int32_t* group_sums_out,
                           const int32_t* hadamard_factors, uint32_t dtype, uint32_t gated_dim, uint32_t batch_dim, uint32_t value_offset,
                           uint32_t value_row_stride, uint32_t act_type, uint32_t interleaved, uint32_t ops, uint32_t activation_scale_group_size,
                           uint32_t sum_group_size);                                                 /* gated_act_mul.rs:47-118 (use_hadamard) */
void orc_gated_act_mul(const void* act_operand, const void* value_operand, void* fp_out, uint32_t dtype,
                       uint32_t gated_dim, uint32_t batch_dim, uint32_t value_offset, uint32_t value_row_stride,
                       uint32_t act_type, uint32_t interleaved);                            /* gated_act_mul.rs */
void orc_quantized_embedding_lookup(const uint32_t* token_ids, const uint8_t* weights, const void* scales,
                                    const uint8_t* zero_points, const void* biases, void* output, uint32_t dtype,
                                    uint32_t batch_size, uint32_t vocab_size, uint32_t model_dim, float input_scale,
                                    uint32_t group_size, uint32_t bits, uint32_t method); /* quant_embedding.rs */
void orc_full_precision_embedding_lookup(const uint32_t* token_ids, const void* weights, void* output, uint32_t dtype,
                                         uint32_t batch_size, uint32_t vocab_size, uint32_t model_dim,
                                         float input_scale);
void orc_logit_transform(void* logits, uint32_t dtype, uint32_t length, float scale, float soft_cap,
                         uint32_t has_soft_cap);
void orc_tensor_add_bias(const void* input, const void* bias, void* output, uint32_t dtype, uint32_t bias_dtype,
                         uint32_t num_cols, uint32_t length);
void orc_tensor_add_scale(const void* input, const void* bias, void* output, uint32_t dtype, uint32_t num_cols,
                          uint32_t length, float scale);
void orc_tensor_add_swap(void* skip, void* main_buf, uint32_t dtype, uint32_t length);
void orc_tensor_copy(const void* src, void* dst, uint32_t dtype, uint32_t length);
void orc_argmax(const void* logits, uint32_t dtype, uint32_t* output, uint32_t vocab_size,
                uint32_t batch_size);
/* sampler RNG (gumbel.rs) and the full UnifiedSampling kernel (unified_sampling.rs:13-99); seeds / bitmask may be NULL */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
float orc_unit_interval(uint32_t word);
float orc_uniform_float(uint64_t key, uint32_t offset, uint32_t word);
float orc_gumbel_float(uint64_t key, uint32_t offset, uint32_t word);
void orc_revidx(uint32_t logit_idx, uint32_t vocab_size, uint32_t* offset, uint32_t* word);
void orc_unified_sampling(const void* logits, uint32_t dtype, uint32_t* output, const uint64_t* seeds, const uint32_t* bitmask,
                          uint32_t has_temperature, float temperature, uint32_t has_top_k, uint32_t top_k, uint32_t has_top_p,
                          float top_p, uint32_t has_min_p, float min_p, uint32_t vocab_size, uint32_t batch_size); /* unified_sampling.rs greedy: ties -> lowest index */

/* ---- Gated DeltaNet (cpu/kernel/gdn/{conv_update,update,conv_scan,prefill_prep,prefill,norm_gate}.rs,
 *      ssm/conv1d.rs Conv1dPack) with the Metal kernels' buffer types: T = bf16 activations,
 *      f32 a_log / dt_bias / norm_weight / conv + ssm state (metal/kernel/gdn/update.metal:19-31;
 *      SURVEY.md F5) ---- */
void orc_delta_net_conv_update(const float* conv_weight, const float* bias, uint16_t* in_out, float* state,
                               uint32_t kernel_size, uint32_t conv_dim, uint32_t state_stride);
void orc_delta_net_update(const uint16_t* in_proj, const float* a_log, const float* dt_bias,
                          const float* norm_weight, float* state, uint16_t* out, uint32_t num_v_heads,
                          uint32_t num_k_heads, uint32_t head_k_dim, uint32_t head_v_dim, uint32_t key_dim,
                          uint32_t value_dim, float norm_epsilon);
void orc_conv1d_pack(const float* state_in, const uint16_t* x, float* padded, uint32_t state_stride,
                     uint32_t row_stride, uint32_t suffix_len, uint32_t num_channels);
void orc_delta_net_conv_scan(const float* conv_padded, const float* conv_weight, const float* bias, uint16_t* in_proj,
                             float* state_out, uint32_t suffix_len, uint32_t kernel_size, uint32_t row_stride,
                             uint32_t state_stride, uint32_t conv_dim, uint32_t out_stride);
void orc_delta_net_prefill_prep(const uint16_t* in_proj, const float* a_log, const float* dt_bias, float* q_norm_out,
                                float* k_norm_out, float* beta_out, float* decay_out, uint32_t num_v_heads,
                                uint32_t num_k_heads, uint32_t head_k_dim, uint32_t key_dim, uint32_t value_dim,
                                uint32_t suffix_len);
void orc_delta_net_prefill(const float* q_norm, const float* k_norm, const float* beta_buf, const float* decay_buf,
                           const uint16_t* in_proj, float* state, uint16_t* out, uint32_t num_v_heads,
                           uint32_t num_k_heads, uint32_t head_k_dim, uint32_t head_v_dim, uint32_t key_dim,
                           uint32_t value_dim, uint32_t suffix_len);
void orc_delta_net_norm_gate(uint16_t* in_out, const uint16_t* in_proj, const float* norm_weight,
                             uint32_t num_v_heads, uint32_t head_v_dim, uint32_t value_dim, uint32_t conv_dim,
                             uint32_t total_proj_dim, float norm_epsilon, uint32_t suffix_len);

/* ---- Gated DeltaNet over a speculated token tree (cpu/kernel/gdn/tree_verify/*.rs; uzu_oracle_tree_verify.c).  `dt` = element type of the
 *      activation tensors (ORC_BF16 in the model; the reference's kernel tests also run f32); trie = 3 u32 per node {start, end, height} ---- */
void orc_conv_tree_scan(const void* in_proj, const float* conv_weight, const float* bias, const float* base_state, const int32_t* parents,
                        void* out_proj, float* suffix_state, uint32_t dt, uint32_t suffix_len, uint32_t kernel_size, uint32_t total_proj_dim,
                        uint32_t conv_dim);
void orc_delta_net_tree_prep(const void* in_proj, const float* a_log, const float* dt_bias, void* q_norm_out, void* k_norm_out,
                             void* compact_v_out, float* beta_out, float* log_decay_out, uint32_t dt, uint32_t num_v_heads,
                             uint32_t num_k_heads, uint32_t head_k_dim, uint32_t key_dim, uint32_t value_dim, uint32_t suffix_len);
void orc_build_tree_prefix(const uint32_t* trie, const float* log_decay, float* prefix, uint32_t batch_size, uint32_t tree_size,
                           uint32_t value_heads);
void orc_build_tree_gram(const void* q, const void* k, uint32_t dt, const uint32_t* trie, const float* prefix, const float* beta, const float* h0,
                         const int32_t* h0_idx, float* a_packed, float* qkd, float* a_inv, float* kh0, float scale, uint32_t batch_size,
                         uint32_t tree_size, uint32_t k_heads, uint32_t value_heads, uint32_t head_k_dim, uint32_t head_v_dim);
void orc_tree_update_solve(const float* kh0, const void* v, uint32_t dt, const float* prefix, const float* beta, const float* a_packed,
                           const float* a_inv, const int32_t* h0_idx, float* u, uint32_t batch_size, uint32_t tree_size, uint32_t value_heads,
                           uint32_t head_v_dim);
void orc_build_tree_out(const void* q, uint32_t qk_dt, const float* prefix, const float* qkd, const float* u, const float* h0, const int32_t* h0_indices,
                        void* o, uint32_t out_dt, float scale, uint32_t batch_size, uint32_t tree_size, uint32_t qk_heads, uint32_t value_heads,
                        uint32_t head_k_dim, uint32_t head_v_dim);
void orc_state_advance(const void* k_norm, const void* v, uint32_t dt, const float* log_decay_buf, const float* beta_buf, const uint32_t* accepted_indices,
                       float* state, uint32_t accepted_len, uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_k_dim);

/* ---- the tree speculators' kernels (cpu/kernel/attention/ancestor_attention.rs, cpu/kernel/weaver/*.rs; uzu_oracle_speculator.c);
 *      structure-of-arrays layouts of backends/common/gpu_types/weaver.rs ---- */
void orc_ancestor_attention(const uint16_t* prefix_kv, uint16_t* node_kv, const uint16_t* current_qkv, const float* cosines, const float* sines,
                            const uint32_t* node_metadata, const uint32_t* ancestor_indices, const uint32_t* ancestor_counts, const uint32_t* node_indices,
                            uint16_t* output, uint32_t rows, uint32_t prefix_length, uint32_t ancestor_stride, uint32_t node_capacity, uint32_t max_depth,
                            float scale, uint32_t num_heads, uint32_t head_dim);
void orc_weaver_frontier_select(uint32_t* frontier, uint32_t* packed_tree, uint32_t* slot_ancestors, uint32_t* node_token_ids, uint32_t* node_metadata,
                                uint32_t* node_ancestor_indices, uint32_t* node_valid, const uint32_t* candidate_pool_ids, const float* candidate_pool_logits,
                                uint32_t* node_candidate_ids, float* node_candidate_logits, uint32_t frontier_capacity, uint32_t tree_slot_count,
                                uint32_t node_count, uint32_t batch_start_slot, uint32_t ancestor_stride, uint32_t max_depth, uint32_t lookahead_count,
                                uint32_t candidate_depth_count, uint32_t candidates_per_depth);
void orc_weaver_frontier_insert_children(const uint32_t* packed_tree, const uint32_t* node_metadata, const uint32_t* node_valid, const uint32_t* child_ids,
                                         const float* child_logprobs, uint32_t* frontier, uint32_t frontier_capacity, uint32_t tree_slot_count,
                                         uint32_t node_count, uint32_t expand_width);
void orc_weaver_top_children(const uint16_t* residual_logits, const float* candidate_logits, const uint32_t* candidate_ids, const uint64_t* depth_seeds,
                             const uint32_t* node_metadata, uint32_t* output_token_ids, float* output_model_logprobs, uint32_t rows, uint32_t candidates,
                             uint32_t expand_width, uint32_t vocab_size);

/* ---- model driver: Decoder::encode + greedy Sampling + encode_accept
 *      (decoder.rs:138-203, transformer.rs:226-329, transformer_layer.rs:194-238,
 *       engine/language_model/stream/stream.rs:190-345,593-751) ---- */
typedef struct orc_model orc_model;
orc_model* orc_model_create(const uzu_model_desc* desc); /* keeps the desc's pointers: caller keeps them alive */
void orc_model_destroy(orc_model* m);
void orc_model_reset(orc_model* m);
void orc_model_fill_synthetic_context(orc_model* m, uint32_t n); /* timing aid (bench.py cpu_baseline): see the .c file */
uint32_t orc_model_context_length(const orc_model* m);
/* One forward pass over `count` tokens of one sequence (count <= 1024) appended at the current context
 * length; writes logits (bf16 [vocab]) of the LAST row if logits_out != NULL, returns greedy token of the
 * last row.  This is one prefill chunk (count > 1) or one decode step (count == 1). */
uint32_t orc_model_forward(orc_model* m, const uint32_t* token_ids, uint32_t count, uint16_t* logits_out);
/* One forward pass over a speculated TREE of `tree_size` tokens (DFS order; trie = {start, end, height} per node) hanging off the current
 * context, NOT accepted (stream.rs:556-628 with full_accept = false): greedy token of every node into sampled_out [tree_size], logits of
 * every node (bf16 [tree_size, vocab]) if logits_out != NULL.  Follow with orc_model_accept. */
void orc_model_verify_tree(orc_model* m, const uint32_t* token_ids, const uint32_t* trie, uint32_t tree_size, uint32_t* sampled_out, uint16_t* logits_out);
/* TransformerState::encode_accept (stream.rs:441-444): `accepted` = a root path of the pending tree (FlatTrie::accept, trie.rs:271-305) */
void orc_model_accept(orc_model* m, const uint32_t* accepted, uint32_t n);
/* Debug taps: copy the last forward's per-layer outputs (bf16 [count, model_dim]) */
const uint16_t* orc_model_layer_output(const orc_model* m, uint32_t layer, uint32_t* rows);
const uint16_t* orc_model_final_hidden(const orc_model* m); /* output_norm of last row, bf16 [model_dim] */
/* speculator taps (stream.rs:213-214,632-633): with capture on, every forward keeps capture_residual(shortcut, hidden) of every layer (bf16 [rows, d],
 * transformer.rs:160-171,285-293) and the output norm of every output row (DecoderEncodeOutput::final_hidden, decoder.rs:190-194) */
void orc_model_capture_features(orc_model* m, uint32_t on);
const uint16_t* orc_model_hidden_feature(const orc_model* m, uint32_t layer, uint32_t* rows);
const uint16_t* orc_model_final_hidden_rows(const orc_model* m, uint32_t* rows);
const uzu_model_desc* orc_model_desc(const orc_model* m);

/* ---- Weaver tree constructor (encodable_block/weaver.rs:166-676; uzu_oracle_weaver.c) ---- */
void orc_radix_top_k_small(const float* input, uint32_t* output_ids, float* output_scores, uint32_t rows, uint32_t columns, uint32_t k); /* cpu/kernel/radix_top_k_small.rs */
typedef struct orc_weaver orc_weaver;
orc_weaver* orc_weaver_create(const uzu_weaver_desc* desc); /* keeps the desc's tensor pointers */
void orc_weaver_destroy(orc_weaver* w);
/* -> 0, or 1 = WeaverEncodeError::InvalidTreeInput.  packed_tree u32 [6, slots], frontier u32 [7, slots * expand_width], slots = 1 + (rounds - 1) * expand_per_round */
int orc_weaver_encode_tree(const orc_weaver* w, const orc_model* target, const uint16_t* target_hidden, const uint16_t* draft_hidden, const float* logits, const uint64_t* depth_seeds,
                           uint32_t depth_seed_count, uint32_t root_token_id, const uzu_weaver_tree_shape* shape, uint32_t* packed_tree, uint32_t* frontier);

/* ---- Mixture of experts (encodable_block/mlp/moe/mod.rs; uzu_oracle_moe.c: which kernels follow a CPU body and which the Metal shader is said there) ---- */
void orc_moe_router_topk(const void* input, const void* weight, const void* bias, int32_t* topk_ids, void* topk_probs, uint32_t dt, uint32_t t, uint32_t d_model, uint32_t e,
                         uint32_t k, uint32_t renorm);
void orc_moe_counts_offsets_fused(const int32_t* topk_ids, uint32_t* offsets, uint32_t* sum_k_out, uint32_t* partials, uint32_t t, uint32_t e, uint32_t k);
void orc_moe_scatter_buckets(const int32_t* topk_ids, const void* topk_probs, const uint32_t* offsets, int32_t* bucketed_ids, void* bucketed_probs, int32_t* tok2row, uint32_t dt,
                             uint32_t t, uint32_t e, uint32_t k);
void orc_moe_gather(const void* x, const int32_t* bucketed_ids, void* x_perm, const uint32_t* sumk_buf, uint32_t dt, uint32_t d_model, uint32_t t, uint32_t k);
void orc_moe_experts_pass_a(const void* x_perm, const uint32_t* expert_offsets, const void* w13_all, const void* up_biases, float* hidden_out, uint32_t dt, uint32_t d_model,
                            uint32_t d_ff, uint32_t e, float gate_clip_min, float gate_clip_max, float up_clip_min, float up_clip_max, float alpha, uint32_t gating_sel);
void orc_moe_experts_down(const float* hidden, const uint32_t* row_expert_map, const void* w2_all, const void* down_biases, void* y_out, uint32_t dt, uint32_t total_rows,
                          uint32_t d_model, uint32_t d_ff, uint32_t e);
void orc_moe_finalize(const int32_t* tok2row, const void* probs, const void* y_partial, void* y, uint32_t dt, uint32_t t_count, uint32_t d_model, uint32_t k);

/* ---- DFlash draft model (encodable_block/dflash.rs:41-346; uzu_oracle_dflash.c) ---- */
typedef struct orc_dflash orc_dflash;
orc_dflash* orc_dflash_create(const uzu_dflash_desc* desc); /* keeps the desc's tensor pointers */
void orc_dflash_destroy(orc_dflash* f);
void orc_dflash_reset(orc_dflash* f);
uint32_t orc_dflash_context_length(const orc_dflash* f);
/* encode_accept: target_features[i] = bf16 [rows, model_dim] of target layer target_layer_ids[i] */
void orc_dflash_accept(orc_dflash* f, const uint16_t* const* target_features, const uint32_t* accepted_indices, uint32_t num_tokens);
/* encode_draft + the greedy tokens of the Argmax construction: draft_hidden bf16 [batch, d], logits f32 [batch - 1, vocab], tokens [batch - 1] (each optional) */
void orc_dflash_draft(orc_dflash* f, const orc_model* target, uint32_t target_output_token, uint32_t batch_size, uint16_t* draft_hidden_out, float* logits_out,
                      uint32_t* tokens_out);

#ifdef __cplusplus
}
#endif
#endif
