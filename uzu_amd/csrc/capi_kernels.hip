// capi_kernels.hip -- C-ABI `_create` / `_encode` entry points for every kernel on the forward path
// (include/uzu_hip.h).  `_create` mirrors `XxxKernel::new` (validates the type / specialisation
// parameters and remembers them), `_encode` mirrors `XxxKernel::encode` (argument checks that are
// `assert!`s in the reference become UZU_ERR_INVALID_ARGUMENT) and enqueues the gfx950 kernel on the
// command buffer's stream.
#include "internal.h"
#include "kernels.h"
#include "uzu_math.h"

using namespace uzu;

namespace {

enum KernelKind : uint32_t {
    KK_MATMUL = 1, KK_NORMALIZATION, KK_QKV_NORM, KK_ATTENTION_PREPARE, KK_ATTENTION_SINGLE_PASS, KK_ATTENTION_TWO_PASS1,
    KK_ATTENTION_TWO_PASS2, KK_ATTENTION_GEMM, KK_ACTIVATION_TRANSFORM, KK_KV_CACHE_UPDATE, KK_SIGMOID_GATE, KK_GATED_ACT_MUL, KK_QUANT_EMBEDDING, KK_FP_EMBEDDING,
    KK_LOGIT_TRANSFORM, KK_TENSOR_ADD_BIAS, KK_TENSOR_ADD_SCALE, KK_TENSOR_ADD_SWAP, KK_TENSOR_COPY, KK_UNIFIED_SAMPLING,
    KK_DN_CONV_UPDATE, KK_DN_UPDATE, KK_CONV1D_PACK, KK_DN_CONV_SCAN, KK_DN_PREFILL_PREP, KK_DN_PREFILL, KK_DN_NORM_GATE,
    KK_CONV_TREE_SCAN, KK_DN_TREE_VERIFY, KK_STATE_ADVANCE, KK_ANCESTOR_ATTENTION, KK_WEAVER_SELECT, KK_WEAVER_INSERT, KK_WEAVER_TOP_CHILDREN,
    KK_MOE_ROUTER_TOPK, KK_MOE_COUNTS_OFFSETS, KK_MOE_SCATTER, KK_MOE_GATHER, KK_MOE_PASS_A, KK_MOE_DOWN, KK_MOE_FINALIZE,
};

bool is_float_dt(uint32_t dt) { return dt == UZU_BF16 || dt == UZU_F32; }

uzu_status make_kernel(uzu_hip_context* ctx, uint32_t kind, uzu_hip_kernel** out, uzu_hip_kernel** created) {
    if (!ctx || !out) {
        set_error("kernel create: null argument");
        return UZU_ERR_INVALID_ARGUMENT;
    }
    auto* k = new uzu_hip_kernel();
    k->ctx = ctx;
    k->kind = kind;
    *out = k;
    *created = k;
    return UZU_OK;
}

uzu_status check(uzu_hip_kernel* k, uint32_t kind, uzu_hip_cmdbuf* cb) {
    if (!k || k->kind != kind) {
        set_error("encode: kernel handle is null or of the wrong kind");
        return UZU_ERR_INVALID_ARGUMENT;
    }
    return cmdbuf_check_encoding(cb);
}

#define REQ_DT(dt, what) UZU_UNSUPPORTED(!is_float_dt(dt), what ": unsupported data type %u (BF16 / F32 only)", (unsigned)(dt))

} // namespace

namespace uzu { namespace k {
uzu_status matmul_full_precision(hipStream_t s, const MatmulParams& p, uint32_t b_transpose, uint32_t ld);
} }

extern "C" {

// ------------------------------------------------------------------------------------- Matmul
uzu_status uzu_hip_matmul_create(uzu_hip_context* ctx, uint32_t weights_dt, uint32_t input_dt, uint32_t output_dt, uzu_hip_kernel** out) {
    // MatmulError::UnsupportedDataType (cpu/kernel/matmul/kernel.rs:36-40; LinearMatmul allows BF16 / F32)
    REQ_DT(weights_dt, "matmul");
    REQ_DT(input_dt, "matmul");
    REQ_DT(output_dt, "matmul");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_MATMUL, out, &k));
    k->t[0] = weights_dt, k->t[1] = input_dt, k->t[2] = output_dt;
    return UZU_OK;
}

// MatmulKernel::a8_activation_plan (kernel.rs:28-34; Metal's policy: metal/kernel/matmul/mod.rs:132-166): whether THIS backend can run the
// shape with symmetric int8 activations, and with which quantisation groups.  The HIP matmul takes MatmulA::Int8Symmetric for every
// quantised B (int4 / int8 codes, signed or not, all three prologues, groups of 32 / 64 / 128) with bf16 tables (k_activation_transform.hip:
// matmul_a8); the activation group is the reference's ACTIVATION_SCALE_GROUP_SIZE, the sum group min(weight group, activation group) for the
// prologues that carry an offset term.
uzu_status uzu_hip_matmul_a8_activation_plan(uzu_hip_kernel* k, const uzu_matmul_shape* shape, uint32_t* has_plan, uzu_a8_activation_plan* out) {
    UZU_REQUIRE(k && k->kind == KK_MATMUL && shape && has_plan && out, "matmul_a8_activation_plan: bad argument");
    constexpr uint32_t kActivationGroup = 128; // ACTIVATION_SCALE_GROUP_SIZE (backends/common/kernel/activation_transform.rs:17)
    *has_plan = 0;
    out->activation_group_size = 0, out->has_sum_group_size = 0, out->sum_group_size = 0;
    const bool quant = shape->b_kind != UZU_MATMUL_B_FULL_PRECISION;
    if (k->t[1] != UZU_BF16 || k->t[2] != UZU_BF16 || k->t[0] != UZU_BF16 || shape->a_full_precision || !quant || !shape->b_transpose || shape->has_b_leading_dimension ||
        shape->gathered || !(shape->b_bits == 4 || shape->b_bits == 8) || !(shape->b_group_size == 32 || shape->b_group_size == 64 || shape->b_group_size == 128) ||
        shape->k % kActivationGroup || shape->k % shape->b_group_size)
        return UZU_OK;
    *has_plan = 1;
    out->activation_group_size = kActivationGroup;
    if (shape->b_kind != UZU_MATMUL_B_SCALE_SYMMETRIC) {
        out->has_sum_group_size = 1;
        out->sum_group_size = shape->b_group_size < kActivationGroup ? shape->b_group_size : kActivationGroup;
    }
    return UZU_OK;
}
// MatmulKernel::select_activation_format (kernel.rs:36-42; Metal: mod.rs:168-183 answers Int8 only where its GEMM runs on the matrix unit
// AND is faster there).  MI355X, measured (tests/test_gpu_kernels.py::test_matmul_int8_activations_throughput_report): the
// int8-activation GEMM on v_mfma_i32_32x32x32_i8 runs at the bf16-activation GEMM's rate (721 vs 746 T(FL)OP/s at 4096x14336x4096 on int4
// weights, 501 vs 612 on int8 weights) -- both are bound by the per-group scale fold / int4 conversion on the vector unit, not by the matrix
// pipe -- and decode GEMVs gain nothing from quantised activations.  With no speed to buy for the activation-quantisation error the answer
// is Bf16 for every shape; the entry exists so that a Rust shim forwards the trait method instead of hard-coding the default.
uzu_status uzu_hip_matmul_select_activation_format(uzu_hip_kernel* k, const uzu_matmul_shape* bf16_shape, uint32_t* format_out) {
    UZU_REQUIRE(k && k->kind == KK_MATMUL && bf16_shape && format_out, "matmul_select_activation_format: bad argument");
    *format_out = UZU_ACTIVATION_FORMAT_BF16;
    return UZU_OK;
}

uzu_status uzu_hip_matmul_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, const uzu_matmul_arguments* a) {
    UZU_PROPAGATE(check(k, KK_MATMUL, cb));
    UZU_REQUIRE(a && a->a.buffer && a->b.buffer && a->d.buffer, "matmul: a, b and d are required");
    UZU_REQUIRE(a->a_kind <= 1u, "matmul: unknown MatmulA kind %u", a->a_kind);
    const bool post_rht = a->rht_factors.buffer != nullptr;
    UZU_REQUIRE(!post_rht || a->n % 32u == 0, "matmul: output RHT needs n %% 32 == 0 (Hadamard block), got %u", a->n);
    k::MatmulParams p{};
    const size_t in_sz = k->t[1] == UZU_F32 ? 4 : 2;
    p.a = (const char*)bptr(a->a) + a->a_offset_elements * in_sz;
    p.b = bptr(a->b);
    p.scales = bptr(a->scales);
    p.biases = bptr(a->biases);
    p.zero_points = (const uint8_t*)bptr(a->zero_points);
    p.d = bptr(a->d);
    p.bias = post_rht ? nullptr : bptr(a->bias); // bias_after_rht (kernel.rs:167)
    p.gather = (const uint32_t*)bptr(a->gather_indices);
    p.w_dt = k->t[0], p.a_dt = k->t[1], p.d_dt = k->t[2];
    p.b_kind = a->b_kind;
    p.group_size = a->group_size;
    p.signed_codes = a->signed_codes;
    p.ab_scale = a->ab_scale;
    p.accumulate = a->accumulate;
    p.has_soft_cap = a->has_soft_cap;
    p.soft_cap = a->soft_cap;
    p.m = a->m, p.n = a->n, p.k = a->k;
    if (a->b_kind == UZU_MATMUL_B_FULL_PRECISION) {
        const uint32_t ld = a->has_b_leading_dimension ? a->b_leading_dimension : (a->b_transpose ? a->k : a->n);
        return k::matmul_full_precision(cb_stream(cb), p, a->b_transpose, ld);
    }
    UZU_REQUIRE(a->b_transpose, "matmul: quantized B requires b_transpose (MatmulError::UnsupportedLayout)");
    UZU_REQUIRE(a->scales.buffer, "matmul: quantized B requires scales");
    UZU_REQUIRE(a->b_kind != UZU_MATMUL_B_SCALE_BIAS || a->biases.buffer, "matmul: ScaleBias requires biases");
    UZU_REQUIRE(a->b_kind != UZU_MATMUL_B_SCALE_ZERO_POINT || a->zero_points.buffer, "matmul: ScaleZeroPoint requires zero_points");
    UZU_UNSUPPORTED(a->mode == UZU_QMODE_I8, "matmul: I8 quantization mode is not a weight-matrix mode (weight_matrix.rs:60-68)");
    p.bits = a->mode == UZU_QMODE_U4 ? 4 : 8;
    UZU_UNSUPPORTED(a->group_size == 0, "matmul: group size must be non-zero (MatmulError::UnsupportedGroupSize)");
    if (a->a_kind == 1u) { // MatmulA::Int8Symmetric
        UZU_REQUIRE(a->a_scales.buffer, "matmul: Int8Symmetric activations need their scales");
        UZU_REQUIRE(a->a_offset_elements == 0, "matmul: Int8Symmetric activations carry no element offset");
        UZU_PROPAGATE(k::matmul_a8(cb_stream(cb), p, (const int8_t*)bptr(a->a), (const float*)bptr(a->a_scales), a->a_group_size));
    } else {
        UZU_PROPAGATE(k::matmul(cb_stream(cb), p, cb->ctx->num_cus));
    }
    if (post_rht) { // kernel.rs:296-303: OutputRht in place on D, then TensorAddBias
        UZU_PROPAGATE(k::activation_transform(cb_stream(cb), nullptr, p.d, nullptr, nullptr, nullptr, (const int32_t*)bptr(a->rht_factors), k->t[2], a->m, a->n, 1u, 0u, 0u));
        if (a->bias.buffer) UZU_PROPAGATE(k::tensor_add_bias(cb_stream(cb), p.d, bptr(a->bias), p.d, k->t[2], k->t[0], a->n, a->m * a->n));
    }
    return UZU_OK;
}

// ------------------------------------------------------------------------------------- ActivationTransform
uzu_status uzu_hip_activation_transform_create(uzu_hip_context* ctx, uint32_t t, uint32_t ops, uint32_t in_place, uint32_t activation_scale_group_size,
                                               uint32_t sum_group_size, uzu_hip_kernel** out) {
    REQ_DT(t, "activation_transform");
    UZU_REQUIRE(ops <= UZU_ACTIVATION_TRANSFORM_QUANTIZE_WITH_GROUP_SUMS, "activation_transform: unknown op %u", ops);
    UZU_REQUIRE(!in_place || ops <= UZU_ACTIVATION_TRANSFORM_OUTPUT_RHT, "activation_transform: a Quantize op cannot be in place (it has no fp_out)");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_ACTIVATION_TRANSFORM, out, &k));
    k->t[0] = t;
    k->f[0] = ops, k->f[1] = in_place, k->f[2] = activation_scale_group_size, k->f[3] = sum_group_size;
    return UZU_OK;
}
uzu_status uzu_hip_activation_transform_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf input, uzu_buf fp_out, uzu_buf q_out, uzu_buf scales_out,
                                               uzu_buf group_sums_out, uzu_buf rht_factors, uint32_t batch_size, uint32_t element_count) {
    UZU_PROPAGATE(check(k, KK_ACTIVATION_TRANSFORM, cb));
    const uint32_t ops = k->f[0];
    const bool in_place = k->f[1] != 0, rht = ops <= UZU_ACTIVATION_TRANSFORM_OUTPUT_RHT, sums = ops == UZU_ACTIVATION_TRANSFORM_QUANTIZE_WITH_GROUP_SUMS;
    UZU_REQUIRE(rht_factors.buffer, "activation_transform: rht_factors are required");
    UZU_REQUIRE((input.buffer == nullptr) == in_place, "activation_transform: input presence must equal !in_place");
    UZU_REQUIRE((fp_out.buffer != nullptr) == rht, "activation_transform: fp_out presence must equal (ops is an RHT op)");
    UZU_REQUIRE((q_out.buffer != nullptr) == !rht && (scales_out.buffer != nullptr) == !rht, "activation_transform: q_out / scales_out presence must equal (ops is a Quantize op)");
    UZU_REQUIRE((group_sums_out.buffer != nullptr) == sums, "activation_transform: group_sums_out presence must equal QuantizeWithGroupSums");
    return k::activation_transform(cb_stream(cb), bptr(input), bptr(fp_out), (int8_t*)bptr(q_out), (float*)bptr(scales_out), (int32_t*)bptr(group_sums_out),
                                   (const int32_t*)bptr(rht_factors), k->t[0], batch_size, element_count, ops, k->f[2], k->f[3]);
}

// ------------------------------------------------------------------------------------- Normalization
uzu_status uzu_hip_normalization_create(uzu_hip_context* ctx, uint32_t input_t, uint32_t affine_t, uint32_t output_t, uint32_t accum_t,
                                        uint32_t in_place, uint32_t subtract_mean, uint32_t full_layer, uint32_t copy_to_shortcut,
                                        uint32_t residual_add, uint32_t use_hadamard, uint32_t scale_residual_sum, uint32_t scale_output,
                                        uint32_t has_biases, uint32_t has_scales, uzu_hip_kernel** out) {
    REQ_DT(input_t, "normalization");
    REQ_DT(affine_t, "normalization");
    UZU_UNSUPPORTED(input_t != output_t, "normalization: InputT != OutputT is not instantiated");
    UZU_UNSUPPORTED(accum_t != UZU_F32, "normalization: AccumT must be F32");
    // use_hadamard (normalization.metal:134-140; `unimplemented!` in the reference's own CPU kernel, normalization.rs:46-48): the input RHT of
    // the linear behind the norm, hoisted into it -- OutputT(hadamard32(float(val) * factor)) on the rounded result, i.e. exactly
    // ActivationTransform::InputRht on the norm's output: run as those two kernels (bit-identical to the fused Metal form).  The Metal kernel
    // applies scale_output AFTER the transform: that combination is not instantiated.
    UZU_UNSUPPORTED(use_hadamard && scale_output, "normalization: use_hadamard with scale_output is not instantiated");
    UZU_REQUIRE(copy_to_shortcut || !residual_add, "normalization: residual_add requires copy_to_shortcut");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_NORMALIZATION, out, &k));
    k->t[0] = input_t, k->t[1] = affine_t;
    const uint32_t f[] = {in_place, subtract_mean, full_layer, copy_to_shortcut, residual_add, scale_residual_sum, scale_output, has_biases, has_scales, use_hadamard};
    for (int i = 0; i < 10; ++i) k->f[i] = f[i];
    return UZU_OK;
}

uzu_status uzu_hip_normalization_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf input, uzu_buf scales, uzu_buf biases, uzu_buf output,
                                        uzu_buf shortcut, uzu_buf hadamard_factors, uint32_t batch_size, uint32_t element_count, float epsilon,
                                        float scale_offset, float post_layer_scalar) {
    UZU_PROPAGATE(check(k, KK_NORMALIZATION, cb));
    const bool in_place = k->f[0], copy_to_shortcut = k->f[3], has_biases = k->f[7], has_scales = k->f[8];
    UZU_REQUIRE(output.buffer, "normalization: output is required");
    UZU_REQUIRE((input.buffer == nullptr) == in_place, "normalization: input presence must equal !in_place");
    UZU_REQUIRE((shortcut.buffer != nullptr) == copy_to_shortcut, "normalization: shortcut presence must equal copy_to_shortcut");
    UZU_REQUIRE((biases.buffer != nullptr) == has_biases, "normalization: biases presence must equal has_biases");
    UZU_REQUIRE((scales.buffer != nullptr) == has_scales, "normalization: scales presence must equal has_scales");
    const bool use_hadamard = k->f[9];
    UZU_REQUIRE((hadamard_factors.buffer != nullptr) == use_hadamard, "normalization: hadamard_factors presence must equal use_hadamard");
    UZU_REQUIRE(!use_hadamard || element_count % 32u == 0, "normalization: use_hadamard needs element_count %% 32 == 0 (got %u)", element_count);
    k::NormParams p{};
    p.input = bptr(input), p.scales = bptr(scales), p.biases = bptr(biases), p.output = bptr(output), p.shortcut = bptr(shortcut);
    p.io_dt = k->t[0], p.affine_dt = k->t[1];
    p.batch_size = batch_size, p.element_count = element_count;
    p.epsilon = epsilon, p.scale_offset = scale_offset, p.post_layer_scalar = post_layer_scalar;
    p.subtract_mean = k->f[1], p.full_layer = k->f[2], p.copy_to_shortcut = k->f[3], p.residual_add = k->f[4];
    p.scale_residual_sum = k->f[5], p.scale_output = k->f[6];
    UZU_PROPAGATE(k::normalization(cb_stream(cb), p));
    if (!use_hadamard) return UZU_OK;
    return k::activation_transform(cb_stream(cb), nullptr, p.output, nullptr, nullptr, nullptr, (const int32_t*)bptr(hadamard_factors), k->t[0], batch_size, element_count,
                                   UZU_ACTIVATION_TRANSFORM_INPUT_RHT, 0, 0);
}

// ------------------------------------------------------------------------------------- QKVNorm
uzu_status uzu_hip_qkv_norm_create(uzu_hip_context* ctx, uint32_t input_t, uint32_t scale_t, uint32_t output_t, uint32_t accum_t,
                                   uint32_t in_place, uint32_t has_scales, uzu_hip_kernel** out) {
    REQ_DT(input_t, "qkv_norm");
    UZU_UNSUPPORTED(input_t != output_t || scale_t != UZU_F32 || accum_t != UZU_F32 || !in_place,
                    "qkv_norm: only in-place (T, F32 scales, F32 accum) is instantiated (the LM path, qkv_norm.rs:113-121)");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_QKV_NORM, out, &k));
    k->t[0] = input_t;
    k->f[0] = has_scales;
    return UZU_OK;
}

uzu_status uzu_hip_qkv_norm_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf qkv_input, uzu_buf scales, uzu_buf qkv_output,
                                   uint32_t batch_size, uint32_t total_heads, uint32_t head_dim, float epsilon, float scale_offset,
                                   uint32_t head_offset, uint32_t head_count, uint32_t full_layer) {
    UZU_PROPAGATE(check(k, KK_QKV_NORM, cb));
    UZU_REQUIRE(qkv_input.buffer == nullptr && qkv_output.buffer, "qkv_norm: in-place kernel takes qkv_output only");
    UZU_REQUIRE((scales.buffer != nullptr) == (k->f[0] != 0), "qkv_norm: scales presence must equal has_scales");
    UZU_REQUIRE(head_offset + head_count <= total_heads, "qkv_norm: head range exceeds total_heads");
    return k::qkv_norm(cb_stream(cb), bptr(qkv_output), k->t[0], (const float*)bptr(scales), batch_size, total_heads, head_dim, epsilon,
                       scale_offset, head_offset, head_count, full_layer);
}

// ------------------------------------------------------------------------------------- AttentionPrepare
uzu_status uzu_hip_attention_prepare_create(uzu_hip_context* ctx, uint32_t element_t, uint32_t rope_t, uint32_t has_kv, uint32_t has_rope,
                                            uzu_hip_kernel** out) {
    UZU_UNSUPPORTED(element_t != UZU_BF16 || rope_t != UZU_F32, "attention_prepare: variants are (ElementT = BF16, RopeT = F32) only (attention_prepare.rs:31-33)");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_ATTENTION_PREPARE, out, &k));
    k->f[0] = has_kv, k->f[1] = has_rope;
    return UZU_OK;
}

uzu_status uzu_hip_attention_prepare_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf qkv, uzu_buf queries, uzu_buf keys, uzu_buf values,
                                            uzu_buf cosines, uzu_buf sines, uint32_t num_q_heads, uint32_t num_kv_heads, uint32_t head_dim,
                                            uint32_t rope_dim, uint32_t kv_token_offset, uint32_t batch_dim) {
    UZU_PROPAGATE(check(k, KK_ATTENTION_PREPARE, cb));
    const bool has_kv = k->f[0], has_rope = k->f[1];
    UZU_REQUIRE(qkv.buffer && queries.buffer, "attention_prepare: qkv and queries are required");
    UZU_REQUIRE(num_q_heads > 0 || has_kv, "attention prepare without KV requires at least one query head");
    UZU_REQUIRE(head_dim > 0, "attention prepare requires nonzero head_dim");
    UZU_REQUIRE((keys.buffer != nullptr) == has_kv && (values.buffer != nullptr) == has_kv, "attention prepare keys/values presence mismatch");
    UZU_REQUIRE(!has_kv || num_kv_heads > 0, "attention prepare has_kv requires nonzero num_kv_heads");
    UZU_REQUIRE((cosines.buffer != nullptr) == has_rope && (sines.buffer != nullptr) == has_rope, "attention prepare cosines/sines presence mismatch");
    if (has_rope) {
        UZU_REQUIRE(rope_dim > 0 && rope_dim <= head_dim && rope_dim % 2 == 0, "attention prepare: rope_dim must be even, nonzero and <= head_dim");
    }
    return k::attention_prepare(cb_stream(cb), (const uint16_t*)bptr(qkv), (uint16_t*)bptr(queries), (uint16_t*)bptr(keys), (uint16_t*)bptr(values),
                                (const float*)bptr(cosines), (const float*)bptr(sines), num_q_heads, num_kv_heads, head_dim,
                                has_rope ? rope_dim : 0, kv_token_offset, batch_dim, has_kv, nullptr);
}

// ------------------------------------------------------------------------------------- Attention cores
static uzu_status attention_core_create(uzu_hip_context* ctx, uint32_t kind, uint32_t t, uint32_t head_dim, uint32_t has_sinks,
                                        uint32_t is_kv_cache_ring, uint32_t is_causal, uint32_t is_trie, uint32_t is_sliding_window,
                                        uzu_hip_kernel** out) {
    REQ_DT(t, "attention");
    UZU_UNSUPPORTED(!(head_dim == 64 || head_dim == 128 || head_dim == 256 || head_dim == 512), "attention: HEAD_DIM variants are 64/128/256/512");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, kind, out, &k));
    k->t[0] = t;
    k->f[0] = head_dim, k->f[1] = has_sinks, k->f[2] = is_kv_cache_ring, k->f[3] = is_causal, k->f[4] = is_sliding_window, k->f[5] = is_trie;
    return UZU_OK;
}

static k::AttentionParams attention_params(uzu_hip_kernel* k, uzu_buf queries, uzu_buf keys, uzu_buf values, uint32_t gqa_factor,
                                           uint32_t sequence_length, uint32_t k_head_stride, uint32_t k_seq_stride, uint32_t v_head_stride,
                                           uint32_t v_seq_stride, uzu_ring_params ring, float scale, uint32_t sliding_window_size, uzu_buf sinks,
                                           uint32_t num_heads, uint32_t suffix_length, uzu_buf trie) {
    k::AttentionParams a{};
    a.queries = bptr(queries), a.keys = bptr(keys), a.values = bptr(values);
    a.dt = k->t[0];
    a.head_dim = k->f[0];
    a.gqa_factor = gqa_factor, a.sequence_length = sequence_length;
    a.k_head_stride = k_head_stride, a.k_seq_stride = k_seq_stride, a.v_head_stride = v_head_stride, a.v_seq_stride = v_seq_stride;
    a.is_kv_cache_ring = k->f[2], a.ring_offset = ring.ring_offset, a.ring_length = ring.ring_length;
    a.scale = scale;
    a.is_sliding_window = k->f[4], a.sliding_window_size = sliding_window_size;
    a.sinks = bptr(sinks);
    a.num_heads = num_heads, a.suffix_length = suffix_length, a.is_causal = k->f[3];
    a.dyn = nullptr;
    a.trie = (const uint32_t*)bptr(trie);
    return a;
}

uzu_status uzu_hip_attention_single_pass_create(uzu_hip_context* ctx, uint32_t t, uint32_t head_dim, uint32_t has_sinks, uint32_t is_kv_cache_ring,
                                                uint32_t is_causal, uint32_t is_trie, uint32_t is_sliding_window, uzu_hip_kernel** out) {
    return attention_core_create(ctx, KK_ATTENTION_SINGLE_PASS, t, head_dim, has_sinks, is_kv_cache_ring, is_causal, is_trie, is_sliding_window, out);
}

uzu_status uzu_hip_attention_single_pass_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf queries, uzu_buf keys, uzu_buf values, uzu_buf out,
                                                uint32_t gqa_factor, uint32_t sequence_length, uint32_t k_head_stride, uint32_t k_seq_stride,
                                                uint32_t v_head_stride, uint32_t v_seq_stride, uzu_ring_params ring_params, float scale, uzu_buf trie,
                                                uint32_t sliding_window_size, uzu_buf sinks, uint32_t num_heads, uint32_t suffix_length) {
    UZU_PROPAGATE(check(k, KK_ATTENTION_SINGLE_PASS, cb));
    UZU_REQUIRE(queries.buffer && keys.buffer && values.buffer && out.buffer, "attention_single_pass: queries/keys/values/out are required");
    UZU_REQUIRE((trie.buffer != nullptr) == (k->f[5] != 0), "attention_single_pass: trie presence must equal is_trie");
    UZU_REQUIRE((sinks.buffer != nullptr) == (k->f[1] != 0), "attention_single_pass: sinks presence must equal has_sinks");
    UZU_REQUIRE(sequence_length >= suffix_length, "attention_single_pass: sequence_length < suffix_length");
    const k::AttentionParams a = attention_params(k, queries, keys, values, gqa_factor, sequence_length, k_head_stride, k_seq_stride, v_head_stride,
                                                  v_seq_stride, ring_params, scale, sliding_window_size, sinks, num_heads, suffix_length, trie);
    return k::attention_single_pass(cb_stream(cb), a, bptr(out));
}

uzu_status uzu_hip_attention_two_pass1_create(uzu_hip_context* ctx, uint32_t t, uint32_t head_dim, uint32_t has_sinks, uint32_t is_kv_cache_ring,
                                              uint32_t is_causal, uint32_t is_trie, uint32_t is_sliding_window, uzu_hip_kernel** out) {
    return attention_core_create(ctx, KK_ATTENTION_TWO_PASS1, t, head_dim, has_sinks, is_kv_cache_ring, is_causal, is_trie, is_sliding_window, out);
}

uzu_status uzu_hip_attention_two_pass1_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf queries, uzu_buf keys, uzu_buf values, uzu_buf out_partials,
                                              uzu_buf sums, uzu_buf maxs, uint32_t gqa_factor, uint32_t sequence_length, uint32_t k_head_stride,
                                              uint32_t k_seq_stride, uint32_t v_head_stride, uint32_t v_seq_stride, uzu_ring_params ring_params,
                                              float scale, uint32_t num_heads, uint32_t suffix_length, uzu_buf trie, uint32_t sliding_window_size,
                                              uzu_buf sinks) {
    UZU_PROPAGATE(check(k, KK_ATTENTION_TWO_PASS1, cb));
    UZU_REQUIRE(queries.buffer && keys.buffer && values.buffer && out_partials.buffer && sums.buffer && maxs.buffer,
                "attention_two_pass1: queries/keys/values/out/sums/maxs are required");
    UZU_REQUIRE((trie.buffer != nullptr) == (k->f[5] != 0), "attention_two_pass1: trie presence must equal is_trie");
    UZU_REQUIRE((sinks.buffer != nullptr) == (k->f[1] != 0), "attention_two_pass1: sinks presence must equal has_sinks");
    UZU_REQUIRE(sequence_length >= suffix_length, "attention_two_pass1: sequence_length < suffix_length");
    const k::AttentionParams a = attention_params(k, queries, keys, values, gqa_factor, sequence_length, k_head_stride, k_seq_stride, v_head_stride,
                                                  v_seq_stride, ring_params, scale, sliding_window_size, sinks, num_heads, suffix_length, trie);
    return k::attention_two_pass1(cb_stream(cb), a, (float*)bptr(out_partials), (float*)bptr(sums), (float*)bptr(maxs));
}

// AttentionGemmCore (BU/backends/common/kernel/attention_gemm/kernel.rs:8-24; arguments: encodable_block/mixer/attention/
// core/mod.rs:17-38): the prefill attention on the matrix cores (k_attention_mfma.hip).  is_supported mirrors the trait's
// static query; a backend that answers false is never constructed by AttentionCores::new (core/mod.rs:53-61).
static bool attention_gemm_args_supported(const uzu_attention_core_arguments* a) {
    // causal bf16 with or without sinks, a sliding window and a ring KV prefix (a ring always comes with its window, state.rs:69-136);
    // the speculated-tree mask stays on the single- / two-pass cores
    if (a->data_type != UZU_BF16 || !a->is_causal || a->is_trie || (a->is_kv_cache_ring && !a->has_sliding_window)) return false;
    if (a->has_sliding_window && !a->sliding_window_size) return false;
    if (!(a->head_dim == 64 || a->head_dim == 128 || a->head_dim == 256)) return false;
    if (a->num_groups == 0 || a->num_q_heads % a->num_groups) return false;
    const uint32_t gqa = a->num_q_heads / a->num_groups;
    return gqa > 4 ? gqa % 4 == 0 : 4 % gqa == 0;
}
uzu_status uzu_hip_attention_gemm_is_supported(uzu_hip_context* ctx, const uzu_attention_core_arguments* arguments, uint32_t* out) {
    UZU_REQUIRE(ctx && arguments && out, "attention_gemm_is_supported: null argument");
    *out = attention_gemm_args_supported(arguments) ? 1u : 0u;
    return UZU_OK;
}
uzu_status uzu_hip_attention_gemm_create(uzu_hip_context* ctx, const uzu_attention_core_arguments* arguments, uzu_hip_kernel** out) {
    UZU_REQUIRE(ctx && arguments && out, "attention_gemm_create: null argument");
    UZU_UNSUPPORTED(!attention_gemm_args_supported(arguments), "attention_gemm: unsupported configuration (query uzu_hip_attention_gemm_is_supported first)");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_ATTENTION_GEMM, out, &k));
    k->t[0] = arguments->data_type;
    k->f[0] = arguments->head_dim, k->f[1] = arguments->num_groups, k->f[2] = arguments->num_q_heads, k->f[3] = arguments->has_scale;
    k->f[4] = uzu::f32_to_bits(arguments->scale);
    k->f[5] = arguments->has_sinks, k->f[6] = arguments->is_kv_cache_ring, k->f[7] = arguments->has_sliding_window ? arguments->sliding_window_size : 0u;
    return UZU_OK;
}
// AttentionCoreEncodeArguments (core/mod.rs:30-38) with the state type spelled out: Full { length } (is_ring = 0, `length` = the prefix
// length) or Ring { offset, length, max_length } (state.rs:16-55: the prefix region is `ring_max_length` slots, `length` of them live).
uzu_status uzu_hip_attention_gemm_encode_state(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf queries, uzu_buf keys, uzu_buf values, uzu_buf sinks, uzu_buf out,
                                               uint32_t is_ring, uint32_t length, uint32_t ring_offset, uint32_t ring_max_length, uint32_t suffix_length) {
    UZU_PROPAGATE(check(k, KK_ATTENTION_GEMM, cb));
    UZU_REQUIRE(queries.buffer && keys.buffer && values.buffer && out.buffer, "attention_gemm: queries/keys/values/out are required");
    UZU_REQUIRE(suffix_length > 0, "attention_gemm: empty suffix");
    UZU_REQUIRE((sinks.buffer != nullptr) == (k->f[5] != 0), "attention_gemm: sinks presence must equal has_sinks");
    UZU_REQUIRE((is_ring != 0) == (k->f[6] != 0), "attention_gemm: the state type must match is_kv_cache_ring");
    UZU_REQUIRE(!is_ring || (ring_max_length > 0 && length <= ring_max_length && ring_offset < ring_max_length), "attention_gemm: bad ring parameters");
    const uint32_t hd = k->f[0], nkv = k->f[1], nq = k->f[2];
    k::AttentionParams a{};
    a.queries = bptr(queries), a.keys = bptr(keys), a.values = bptr(values), a.sinks = bptr(sinks);
    a.dt = k->t[0], a.head_dim = hd, a.gqa_factor = nq / nkv;
    a.sequence_length = (is_ring ? ring_max_length : length) + suffix_length; // physical_prefix_length + the suffix being attended (state.rs:26-37)
    if (is_ring) a.is_kv_cache_ring = 1, a.ring_offset = ring_offset, a.ring_length = length;
    if (k->f[7]) a.is_sliding_window = 1, a.sliding_window_size = k->f[7];
    a.k_head_stride = hd, a.k_seq_stride = nkv * hd, a.v_head_stride = hd, a.v_seq_stride = nkv * hd;
    if (k->f[3]) a.scale = uzu::bits_to_f32(k->f[4]);
    else a.scale = 1.0f / sqrtf((float)hd);
    a.num_heads = nq, a.suffix_length = suffix_length, a.is_causal = 1;
    if (!k::attention_prefill_mfma_supported(a)) return k::attention_single_pass(cb_stream(cb), a, bptr(out)); // suffixes too short for a tile
    return k::attention_prefill_mfma(cb_stream(cb), a, bptr(out));
}
// the plain Full-state form (no sinks)
uzu_status uzu_hip_attention_gemm_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf queries, uzu_buf keys, uzu_buf values, uzu_buf out,
                                         uint32_t prefix_length, uint32_t suffix_length) {
    UZU_PROPAGATE(check(k, KK_ATTENTION_GEMM, cb));
    UZU_REQUIRE(queries.buffer && keys.buffer && values.buffer && out.buffer, "attention_gemm: queries/keys/values/out are required");
    UZU_REQUIRE(suffix_length > 0, "attention_gemm: empty suffix");
    UZU_REQUIRE(!k->f[5] && !k->f[6], "attention_gemm: a kernel with sinks or a ring state is encoded with uzu_hip_attention_gemm_encode_state");
    const uint32_t hd = k->f[0], nkv = k->f[1], nq = k->f[2];
    k::AttentionParams a{};
    a.queries = bptr(queries), a.keys = bptr(keys), a.values = bptr(values);
    a.dt = k->t[0], a.head_dim = hd, a.gqa_factor = nq / nkv;
    if (k->f[7]) a.is_sliding_window = 1, a.sliding_window_size = k->f[7];
    a.sequence_length = prefix_length + suffix_length; // AttentionStateType::Full { length } + the suffix being attended
    // strides of the KV cache layout [tokens, kv_heads, head_dim] (core/single_pass.rs:44-73)
    a.k_head_stride = hd, a.k_seq_stride = nkv * hd, a.v_head_stride = hd, a.v_seq_stride = nkv * hd;
    if (k->f[3]) a.scale = uzu::bits_to_f32(k->f[4]);
    else a.scale = 1.0f / sqrtf((float)hd);
    a.num_heads = nq, a.suffix_length = suffix_length, a.is_causal = 1;
    if (!k::attention_prefill_mfma_supported(a)) return k::attention_single_pass(cb_stream(cb), a, bptr(out)); // suffixes too short for a tile
    return k::attention_prefill_mfma(cb_stream(cb), a, bptr(out));
}

uzu_status uzu_hip_attention_two_pass2_create(uzu_hip_context* ctx, uint32_t t, uint32_t head_dim, uzu_hip_kernel** out) {
    REQ_DT(t, "attention_two_pass2");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_ATTENTION_TWO_PASS2, out, &k));
    k->t[0] = t;
    k->f[0] = head_dim;
    return UZU_OK;
}

uzu_status uzu_hip_attention_two_pass2_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf partials, uzu_buf sums, uzu_buf maxs, uzu_buf out,
                                              uint32_t num_heads, uint32_t suffix_length) {
    UZU_PROPAGATE(check(k, KK_ATTENTION_TWO_PASS2, cb));
    UZU_REQUIRE(partials.buffer && sums.buffer && maxs.buffer && out.buffer, "attention_two_pass2: all buffers are required");
    return k::attention_two_pass2(cb_stream(cb), (const float*)bptr(partials), (const float*)bptr(sums), (const float*)bptr(maxs), bptr(out), k->t[0],
                                  k->f[0], num_heads, suffix_length);
}

// ------------------------------------------------------------------------------------- KVCacheUpdate / SigmoidGate / GatedActMul
uzu_status uzu_hip_kv_cache_update_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out) {
    REQ_DT(t, "kv_cache_update");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_KV_CACHE_UPDATE, out, &k));
    k->t[0] = t;
    return UZU_OK;
}
uzu_status uzu_hip_kv_cache_update_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf in_place_keys, uzu_buf in_place_values,
                                          const uzu_kv_copy* copies, uint32_t copy_count, uint32_t element_dim) {
    UZU_PROPAGATE(check(k, KK_KV_CACHE_UPDATE, cb));
    UZU_REQUIRE(in_place_keys.buffer && in_place_values.buffer && (copies || !copy_count), "kv_cache_update: null argument");
    return k::kv_cache_update(cb_stream(cb), bptr(in_place_keys), bptr(in_place_values), k->t[0], copies, copy_count, element_dim);
}

uzu_status uzu_hip_sigmoid_gate_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out) {
    REQ_DT(t, "sigmoid_gate");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_SIGMOID_GATE, out, &k));
    k->t[0] = t;
    return UZU_OK;
}
uzu_status uzu_hip_sigmoid_gate_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf gate, uzu_buf output, uint32_t total_elements) {
    UZU_PROPAGATE(check(k, KK_SIGMOID_GATE, cb));
    UZU_REQUIRE(gate.buffer && output.buffer, "sigmoid_gate: null buffer");
    return k::sigmoid_gate(cb_stream(cb), bptr(gate), bptr(output), k->t[0], total_elements);
}

// Kernel-owned scratch (grow-only).  Regrown at encode time OUTSIDE stream capture only: the stream-ordered pool is not trusted
// on this ROCm build (runtime.hip) and a captured graph must not reference a block that may be regrown.
static uzu_status kernel_scratch(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, size_t need, const char* what, void** out) {
    if (need > k->scratch_bytes) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(cb_stream(cb), &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
            set_error("%s: this call needs %zu bytes of scratch, more than the kernel owns; encode once outside a graph-captured command buffer "
                      "first (the block cannot grow during capture)", what, need);
            return UZU_ERR_UNSUPPORTED;
        }
        UZU_HIP_TRY(hipStreamSynchronize(cb_stream(cb))); // earlier encodes on this stream may still use the old block
        if (k->scratch) (void)hipFree(k->scratch);
        k->scratch = nullptr, k->scratch_bytes = 0;
        UZU_HIP_TRY(hipMalloc(&k->scratch, need));
        k->scratch_bytes = need;
    }
    *out = k->scratch;
    return UZU_OK;
}

uzu_status uzu_hip_gated_act_mul_create(uzu_hip_context* ctx, uint32_t t, uint32_t ops, uint32_t interleaved, uint32_t use_hadamard,
                                        uint32_t activation_scale_group_size, uint32_t sum_group_size, uzu_hip_kernel** out) {
    REQ_DT(t, "gated_act_mul");
    // gated_act_mul.rs:36-42
    UZU_REQUIRE(ops <= 2u, "gated_act_mul: unknown op %u", ops);
    UZU_REQUIRE(ops == 0u || use_hadamard, "quantized gate activation requires RHT");
    if (ops != 0u) {
        UZU_REQUIRE(activation_scale_group_size && activation_scale_group_size % 32u == 0 && activation_scale_group_size <= 256u &&
                        (activation_scale_group_size & (activation_scale_group_size - 1)) == 0,
                    "gated_act_mul: activation_scale_group_size must be 32, 64, 128 or 256 (got %u)", activation_scale_group_size);
        UZU_REQUIRE(ops != 2u || (sum_group_size && sum_group_size % 32u == 0 && sum_group_size <= 256u && (sum_group_size & (sum_group_size - 1)) == 0),
                    "gated_act_mul: sum_group_size must be 32, 64, 128 or 256 (got %u)", sum_group_size);
    }
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_GATED_ACT_MUL, out, &k));
    k->t[0] = t;
    k->f[0] = interleaved, k->f[1] = ops, k->f[2] = use_hadamard, k->f[3] = activation_scale_group_size, k->f[4] = sum_group_size;
    return UZU_OK;
}
// RHT variants (gated_act_mul.rs:47-118): the gated product is rounded to T first (mod.rs:5-12), then sign factors, the 32-point
// butterfly and, for the Quantize ops, quantize_transformed_row -- i.e. GatedActMul followed by ActivationTransform{InputRht |
// Quantize | QuantizeWithGroupSums} on the rounded products; run as those two kernels (in place on fp_out, or through
// kernel-owned scratch when there is no fp_out), bit-exact like each of them.
uzu_status uzu_hip_gated_act_mul_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf act_operand, uzu_buf value_operand, uzu_buf fp_out, uzu_buf q_out,
                                        uzu_buf scales_out, uzu_buf group_sums_out, uzu_buf hadamard_factors, uint32_t gated_dim, uint32_t batch_dim,
                                        uint32_t value_offset, uint32_t value_row_stride, uint32_t act_type) {
    UZU_PROPAGATE(check(k, KK_GATED_ACT_MUL, cb));
    const bool interleaved = k->f[0], use_hadamard = k->f[2];
    const uint32_t ops = k->f[1];
    const bool quantize = ops != 0u, sums = ops == 2u;
    UZU_REQUIRE(act_operand.buffer, "gated_act_mul: act_operand is required");
    UZU_REQUIRE((fp_out.buffer != nullptr) == !quantize, quantize ? "gated_act_mul: fp_out given to a Quantize kernel" : "FP gate activation requires fp_out");
    UZU_REQUIRE((value_operand.buffer == nullptr) == interleaved, "gated_act_mul: value_operand presence must equal !interleaved");
    UZU_REQUIRE((q_out.buffer != nullptr) == quantize && (scales_out.buffer != nullptr) == quantize, "gated_act_mul: q_out / scales_out presence must equal (ops is a Quantize op)");
    UZU_REQUIRE((group_sums_out.buffer != nullptr) == sums, "gated_act_mul: group_sums_out presence must equal QuantizeWithGroupSums");
    UZU_REQUIRE((hadamard_factors.buffer != nullptr) == use_hadamard, "gated_act_mul: hadamard_factors presence must equal use_hadamard");
    UZU_REQUIRE(act_type <= 4, "gated_act_mul: unknown activation type %u", act_type);
    UZU_REQUIRE(!use_hadamard || gated_dim % 32u == 0, "gated_act_mul: RHT needs gated_dim %% 32 == 0 (got %u)", gated_dim);
    UZU_REQUIRE(!quantize || gated_dim % k->f[3] == 0, "gated_act_mul: gated_dim %u is not a multiple of the scale group %u", gated_dim, k->f[3]);
    UZU_REQUIRE(!sums || gated_dim % k->f[4] == 0, "gated_act_mul: gated_dim %u is not a multiple of the sum group %u", gated_dim, k->f[4]);
    void* product = bptr(fp_out);
    if (quantize) UZU_PROPAGATE(kernel_scratch(k, cb, (size_t)batch_dim * gated_dim * (k->t[0] == UZU_F32 ? 4 : 2), "gated_act_mul", &product));
    UZU_PROPAGATE(k::gated_act_mul(cb_stream(cb), bptr(act_operand), bptr(value_operand), product, k->t[0], gated_dim, batch_dim, value_offset,
                                   value_row_stride, act_type, interleaved));
    if (!use_hadamard) return UZU_OK;
    const uint32_t at_op = ops == 0u ? UZU_ACTIVATION_TRANSFORM_INPUT_RHT : ops == 1u ? UZU_ACTIVATION_TRANSFORM_QUANTIZE : UZU_ACTIVATION_TRANSFORM_QUANTIZE_WITH_GROUP_SUMS;
    return k::activation_transform(cb_stream(cb), quantize ? product : nullptr, quantize ? nullptr : product, (int8_t*)bptr(q_out), (float*)bptr(scales_out),
                                   (int32_t*)bptr(group_sums_out), (const int32_t*)bptr(hadamard_factors), k->t[0], batch_dim, gated_dim, at_op, k->f[3], k->f[4]);
}

// ------------------------------------------------------------------------------------- Embeddings
uzu_status uzu_hip_quantized_embedding_lookup_create(uzu_hip_context* ctx, uint32_t t, uint32_t group_size, uint32_t quantization_mode,
                                                     uint32_t quantization_method, uint32_t use_hadamard, uzu_hip_kernel** out) {
    REQ_DT(t, "quantized_embedding_lookup");
    // use_hadamard (quant_embedding.metal:92-98; `unimplemented!` in the reference's CPU kernel, quant_embedding.rs:32-34): OutputRht of the
    // dequantised, rounded row = the lookup followed by ActivationTransform::OutputRht in place (bit-identical to the fused Metal form)
    UZU_UNSUPPORTED(quantization_mode == UZU_QMODE_I8, "quantized_embedding_lookup: I8 mode is not produced by weight matrices");
    UZU_REQUIRE(group_size > 0 && quantization_method <= 2, "quantized_embedding_lookup: bad group size / method");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_QUANT_EMBEDDING, out, &k));
    k->t[0] = t;
    k->f[0] = group_size, k->f[1] = quantization_mode == UZU_QMODE_U4 ? 4 : 8, k->f[2] = quantization_method, k->f[3] = use_hadamard;
    return UZU_OK;
}
uzu_status uzu_hip_quantized_embedding_lookup_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf token_ids, uzu_buf weights, uzu_buf scales,
                                                     uzu_buf zero_points, uzu_buf biases, uzu_buf output, uzu_buf output_hadamard_factors,
                                                     uint32_t batch_size, uint32_t vocab_size, uint32_t model_dim, float input_scale) {
    UZU_PROPAGATE(check(k, KK_QUANT_EMBEDDING, cb));
    UZU_REQUIRE(token_ids.buffer && weights.buffer && scales.buffer && output.buffer, "quantized_embedding_lookup: null buffer");
    UZU_REQUIRE((zero_points.buffer != nullptr) == (k->f[2] == 1), "ScaleZeroPoint quantized embedding requires zero_points");
    UZU_REQUIRE((biases.buffer != nullptr) == (k->f[2] == 0), "ScaleBias quantized embedding requires biases");
    const bool use_hadamard = k->f[3];
    UZU_REQUIRE((output_hadamard_factors.buffer != nullptr) == use_hadamard, "quantized_embedding_lookup: hadamard factors presence must equal use_hadamard");
    UZU_REQUIRE(!use_hadamard || model_dim % 32u == 0, "quantized_embedding_lookup: use_hadamard needs model_dim %% 32 == 0 (got %u)", model_dim);
    UZU_PROPAGATE(k::quantized_embedding_lookup(cb_stream(cb), (const uint32_t*)bptr(token_ids), (const uint8_t*)bptr(weights), bptr(scales),
                                                (const uint8_t*)bptr(zero_points), bptr(biases), bptr(output), k->t[0], batch_size, vocab_size, model_dim,
                                                input_scale, k->f[0], k->f[1], k->f[2]));
    if (!use_hadamard) return UZU_OK;
    return k::activation_transform(cb_stream(cb), nullptr, bptr(output), nullptr, nullptr, nullptr, (const int32_t*)bptr(output_hadamard_factors), k->t[0], batch_size, model_dim,
                                   UZU_ACTIVATION_TRANSFORM_OUTPUT_RHT, 0, 0);
}
uzu_status uzu_hip_full_precision_embedding_lookup_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out) {
    REQ_DT(t, "full_precision_embedding_lookup");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_FP_EMBEDDING, out, &k));
    k->t[0] = t;
    return UZU_OK;
}
uzu_status uzu_hip_full_precision_embedding_lookup_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf token_ids, uzu_buf weights, uzu_buf output,
                                                          uint32_t batch_size, uint32_t vocab_size, uint32_t model_dim, float input_scale) {
    UZU_PROPAGATE(check(k, KK_FP_EMBEDDING, cb));
    UZU_REQUIRE(token_ids.buffer && weights.buffer && output.buffer, "full_precision_embedding_lookup: null buffer");
    return k::full_precision_embedding_lookup(cb_stream(cb), (const uint32_t*)bptr(token_ids), bptr(weights), bptr(output), k->t[0], batch_size,
                                              vocab_size, model_dim, input_scale);
}

// ------------------------------------------------------------------------------------- LogitTransform / Tensor*
uzu_status uzu_hip_logit_transform_create(uzu_hip_context* ctx, uint32_t t, uint32_t has_soft_cap, uzu_hip_kernel** out) {
    REQ_DT(t, "logit_transform");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_LOGIT_TRANSFORM, out, &k));
    k->t[0] = t, k->f[0] = has_soft_cap;
    return UZU_OK;
}
uzu_status uzu_hip_logit_transform_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf logits, uint32_t length, float scale, float soft_cap) {
    UZU_PROPAGATE(check(k, KK_LOGIT_TRANSFORM, cb));
    UZU_REQUIRE(logits.buffer, "logit_transform: null buffer");
    return k::logit_transform(cb_stream(cb), bptr(logits), k->t[0], length, scale, soft_cap, k->f[0]);
}
uzu_status uzu_hip_tensor_add_bias_create(uzu_hip_context* ctx, uint32_t t, uint32_t bias_t, uint32_t in_place, uzu_hip_kernel** out) {
    REQ_DT(t, "tensor_add_bias");
    REQ_DT(bias_t, "tensor_add_bias");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_TENSOR_ADD_BIAS, out, &k));
    k->t[0] = t, k->t[1] = bias_t, k->f[0] = in_place;
    return UZU_OK;
}
uzu_status uzu_hip_tensor_add_bias_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf input, uzu_buf bias, uzu_buf output, uint32_t num_cols, uint32_t length) {
    UZU_PROPAGATE(check(k, KK_TENSOR_ADD_BIAS, cb));
    UZU_REQUIRE((input.buffer == nullptr) == (k->f[0] != 0), "tensor_add_bias: input presence must equal !in_place");
    UZU_REQUIRE(bias.buffer && output.buffer && num_cols, "tensor_add_bias: null argument");
    return k::tensor_add_bias(cb_stream(cb), bptr(input), bptr(bias), bptr(output), k->t[0], k->t[1], num_cols, length);
}
uzu_status uzu_hip_tensor_add_scale_create(uzu_hip_context* ctx, uint32_t t, uint32_t in_place, uzu_hip_kernel** out) {
    REQ_DT(t, "tensor_add_scale");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_TENSOR_ADD_SCALE, out, &k));
    k->t[0] = t, k->f[0] = in_place;
    return UZU_OK;
}
uzu_status uzu_hip_tensor_add_scale_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf input, uzu_buf bias, uzu_buf output, uint32_t num_cols,
                                           uint32_t length, float scale) {
    UZU_PROPAGATE(check(k, KK_TENSOR_ADD_SCALE, cb));
    UZU_REQUIRE(bias.buffer && output.buffer && num_cols, "tensor_add_scale: null argument");
    return k::tensor_add_scale(cb_stream(cb), bptr(input), bptr(bias), bptr(output), k->t[0], num_cols, length, scale);
}
uzu_status uzu_hip_tensor_add_swap_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out) {
    REQ_DT(t, "tensor_add_swap");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_TENSOR_ADD_SWAP, out, &k));
    k->t[0] = t;
    return UZU_OK;
}
uzu_status uzu_hip_tensor_add_swap_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf skip, uzu_buf main_buffer, uint32_t length) {
    UZU_PROPAGATE(check(k, KK_TENSOR_ADD_SWAP, cb));
    UZU_REQUIRE(skip.buffer && main_buffer.buffer, "tensor_add_swap: null buffer");
    return k::tensor_add_swap(cb_stream(cb), bptr(skip), bptr(main_buffer), k->t[0], length);
}
uzu_status uzu_hip_tensor_copy_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out) {
    REQ_DT(t, "tensor_copy");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_TENSOR_COPY, out, &k));
    k->t[0] = t;
    return UZU_OK;
}
uzu_status uzu_hip_tensor_copy_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf src, uzu_buf dst, uint32_t length) {
    UZU_PROPAGATE(check(k, KK_TENSOR_COPY, cb));
    UZU_REQUIRE(src.buffer && dst.buffer, "tensor_copy: null buffer");
    return k::tensor_copy(cb_stream(cb), bptr(src), bptr(dst), k->t[0], length);
}

// ------------------------------------------------------------------------------------- UnifiedSampling
// unified_sampling.rs:13-32: every specialisation.  Transient arg-max partials live in a kernel-owned grow-only block that is
// allocated at creation (64 rows) and regrown at encode time OUTSIDE stream capture only: the stream-ordered pool is not
// trusted on this ROCm build (runtime.hip) and a captured graph must not reference a block that may be regrown.
static uzu_status sampling_scratch(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uint32_t batch_size, void** out) {
    const size_t need = k::unified_sampling_scratch_bytes(batch_size) > k::argmax_scratch_bytes(batch_size) ? k::unified_sampling_scratch_bytes(batch_size)
                                                                                                         : k::argmax_scratch_bytes(batch_size);
    return kernel_scratch(k, cb, need, "unified_sampling", out);
}
uzu_status uzu_hip_unified_sampling_create(uzu_hip_context* ctx, uint32_t t, uint32_t is_stochastic, uint32_t has_bitmask, uint32_t has_temperature,
                                           uint32_t has_top_k, uint32_t has_top_p, uint32_t has_min_p, uzu_hip_kernel** out) {
    REQ_DT(t, "unified_sampling");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_UNIFIED_SAMPLING, out, &k));
    k->t[0] = t;
    k->f[0] = is_stochastic, k->f[1] = has_bitmask, k->f[2] = has_temperature, k->f[3] = has_top_k, k->f[4] = has_top_p, k->f[5] = has_min_p;
    const size_t bytes = k::unified_sampling_scratch_bytes(64);
    if (hipMalloc(&k->scratch, bytes) != hipSuccess) {
        (void)hipGetLastError();
        uzu_hip_kernel_destroy(k);
        *out = nullptr;
        set_error("unified_sampling: cannot allocate %zu bytes of scratch", bytes);
        return UZU_ERR_OUT_OF_MEMORY;
    }
    k->scratch_bytes = bytes;
    return UZU_OK;
}
uzu_status uzu_hip_unified_sampling_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf logits, uzu_buf output, uzu_buf seeds, uzu_buf bitmask,
                                           float temperature, uint32_t top_k, float top_p, float min_p, uint32_t vocab_size, uint32_t batch_size) {
    UZU_PROPAGATE(check(k, KK_UNIFIED_SAMPLING, cb));
    UZU_REQUIRE(logits.buffer && output.buffer, "unified_sampling: null logits / output");
    UZU_REQUIRE((seeds.buffer != nullptr) == (k->f[0] != 0), "unified_sampling: seeds presence must equal is_stochastic");
    UZU_REQUIRE((bitmask.buffer != nullptr) == (k->f[1] != 0), "unified_sampling: bitmask presence must equal has_bitmask");
    UZU_REQUIRE(!k->f[2] || temperature > 0.0f, "unified_sampling: temperature must be positive");
    void* scratch = nullptr;
    UZU_PROPAGATE(sampling_scratch(k, cb, batch_size, &scratch));
    const bool plain_greedy = !k->f[0] && !k->f[1] && !k->f[2] && !k->f[3] && !k->f[4] && !k->f[5];
    if (plain_greedy) return k::argmax(cb_stream(cb), bptr(logits), k->t[0], (uint32_t*)bptr(output), vocab_size, batch_size, scratch);
    k::UnifiedSamplingParams p{};
    p.logits = bptr(logits), p.dt = k->t[0], p.output = (uint32_t*)bptr(output);
    p.seeds = (const uint64_t*)bptr(seeds), p.bitmask = (const uint32_t*)bptr(bitmask);
    p.has_temperature = k->f[2], p.has_top_k = k->f[3], p.has_top_p = k->f[4], p.has_min_p = k->f[5];
    p.temperature = temperature, p.top_k = top_k, p.top_p = top_p, p.min_p = min_p;
    p.vocab_size = vocab_size, p.batch_size = batch_size;
    return k::unified_sampling(cb_stream(cb), p, scratch);
}

// ------------------------------------------------------------------------------------- Gated DeltaNet
uzu_status uzu_hip_delta_net_conv_update_create(uzu_hip_context* ctx, uint32_t t, uint32_t has_bias, uzu_hip_kernel** out) {
    UZU_UNSUPPORTED(t != UZU_BF16, "delta_net_conv_update: only T = BF16 is instantiated (LanguageModel data type)");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_DN_CONV_UPDATE, out, &k));
    k->f[0] = has_bias;
    return UZU_OK;
}
uzu_status uzu_hip_delta_net_conv_update_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf conv_weight, uzu_buf bias, uzu_buf in_out, uzu_buf state,
                                                uint32_t kernel_size, uint32_t conv_dim, uint32_t state_stride) {
    UZU_PROPAGATE(check(k, KK_DN_CONV_UPDATE, cb));
    UZU_REQUIRE(conv_weight.buffer && in_out.buffer && state.buffer, "delta_net_conv_update: null buffer");
    UZU_REQUIRE((bias.buffer != nullptr) == (k->f[0] != 0), "delta_net_conv_update: bias presence must equal has_bias");
    UZU_REQUIRE(kernel_size >= 2, "delta_net_conv_update: kernel_size must be >= 2");
    return k::delta_net_conv_update(cb_stream(cb), (const float*)bptr(conv_weight), (const float*)bptr(bias), (uint16_t*)bptr(in_out),
                                    (float*)bptr(state), kernel_size, conv_dim, state_stride);
}
uzu_status uzu_hip_delta_net_update_create(uzu_hip_context* ctx, uint32_t t, uint32_t head_k_dim, uzu_hip_kernel** out) {
    UZU_UNSUPPORTED(t != UZU_BF16 || head_k_dim != 128, "delta_net_update: variants are (T = BF16, HEAD_K_DIM = 128)");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_DN_UPDATE, out, &k));
    k->f[0] = head_k_dim;
    return UZU_OK;
}
uzu_status uzu_hip_delta_net_update_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf in_proj, uzu_buf a_log, uzu_buf dt_bias, uzu_buf norm_weight,
                                           uzu_buf state, uzu_buf out, uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_v_dim,
                                           uint32_t key_dim, uint32_t value_dim, float norm_epsilon) {
    UZU_PROPAGATE(check(k, KK_DN_UPDATE, cb));
    UZU_REQUIRE(in_proj.buffer && a_log.buffer && dt_bias.buffer && norm_weight.buffer && state.buffer && out.buffer, "delta_net_update: null buffer");
    return k::delta_net_update(cb_stream(cb), (const uint16_t*)bptr(in_proj), (const float*)bptr(a_log), (const float*)bptr(dt_bias),
                               (const float*)bptr(norm_weight), (float*)bptr(state), (uint16_t*)bptr(out), num_v_heads, num_k_heads, k->f[0], head_v_dim,
                               key_dim, value_dim, norm_epsilon);
}
uzu_status uzu_hip_conv1d_pack_create(uzu_hip_context* ctx, uint32_t state_t, uint32_t input_t, uzu_hip_kernel** out) {
    UZU_UNSUPPORTED(state_t != UZU_F32 || input_t != UZU_BF16, "conv1d_pack: only (StateT = F32, InputT = BF16) is instantiated (delta_net.rs:216-217)");
    uzu_hip_kernel* k;
    return make_kernel(ctx, KK_CONV1D_PACK, out, &k);
}
uzu_status uzu_hip_conv1d_pack_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf state_in, uzu_buf x, uzu_buf padded, uint32_t state_stride,
                                      uint32_t row_stride, uint32_t suffix_len, uint32_t num_channels) {
    UZU_PROPAGATE(check(k, KK_CONV1D_PACK, cb));
    UZU_REQUIRE(state_in.buffer && x.buffer && padded.buffer, "conv1d_pack: null buffer");
    return k::conv1d_pack(cb_stream(cb), (const float*)bptr(state_in), (const uint16_t*)bptr(x), (float*)bptr(padded), state_stride, row_stride,
                          suffix_len, num_channels);
}
uzu_status uzu_hip_delta_net_conv_scan_create(uzu_hip_context* ctx, uint32_t t, uint32_t has_bias, uzu_hip_kernel** out) {
    UZU_UNSUPPORTED(t != UZU_BF16, "delta_net_conv_scan: only T = BF16 is instantiated");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_DN_CONV_SCAN, out, &k));
    k->f[0] = has_bias;
    return UZU_OK;
}
uzu_status uzu_hip_delta_net_conv_scan_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf conv_padded, uzu_buf conv_weight, uzu_buf bias,
                                              uzu_buf in_proj, uzu_buf state_out, uint32_t suffix_len, uint32_t kernel_size, uint32_t row_stride,
                                              uint32_t state_stride, uint32_t conv_dim, uint32_t out_stride) {
    UZU_PROPAGATE(check(k, KK_DN_CONV_SCAN, cb));
    UZU_REQUIRE(conv_padded.buffer && conv_weight.buffer && in_proj.buffer && state_out.buffer, "delta_net_conv_scan: null buffer");
    UZU_REQUIRE((bias.buffer != nullptr) == (k->f[0] != 0), "delta_net_conv_scan: bias presence must equal has_bias");
    return k::delta_net_conv_scan(cb_stream(cb), (const float*)bptr(conv_padded), (const float*)bptr(conv_weight), (const float*)bptr(bias),
                                  (uint16_t*)bptr(in_proj), (float*)bptr(state_out), suffix_len, kernel_size, row_stride, state_stride, conv_dim,
                                  out_stride);
}
uzu_status uzu_hip_delta_net_prefill_prep_create(uzu_hip_context* ctx, uint32_t t, uint32_t qk_t, uint32_t head_k_dim, uint32_t write_log_decay,
                                                 uint32_t write_compact_v, uzu_hip_kernel** out) {
    // the two instantiations the DeltaNet block creates (delta_net.rs:246-265): flat prefill (QKT = f32, decay, no compact V) and the
    // speculated-tree prep (QKT = T, log decay, compact V)
    const bool flat = qk_t == UZU_F32 && !write_log_decay && !write_compact_v, tree = qk_t == UZU_BF16 && write_log_decay && write_compact_v;
    UZU_UNSUPPORTED(t != UZU_BF16 || head_k_dim != 128 || !(flat || tree),
                    "delta_net_prefill_prep: variants are (T = BF16, QKT = F32, decay, no compact V) and (T = QKT = BF16, log decay, compact V)");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_DN_PREFILL_PREP, out, &k));
    k->f[0] = tree ? 1u : 0u;
    return UZU_OK;
}
uzu_status uzu_hip_delta_net_prefill_prep_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf in_proj, uzu_buf a_log, uzu_buf dt_bias, uzu_buf q_norm_out,
                                                 uzu_buf k_norm_out, uzu_buf compact_v_out, uzu_buf beta_out, uzu_buf decay_out, uint32_t num_v_heads,
                                                 uint32_t num_k_heads, uint32_t key_dim, uint32_t value_dim, uint32_t suffix_len) {
    UZU_PROPAGATE(check(k, KK_DN_PREFILL_PREP, cb));
    UZU_REQUIRE(in_proj.buffer && a_log.buffer && dt_bias.buffer && q_norm_out.buffer && k_norm_out.buffer && beta_out.buffer && decay_out.buffer,
                "delta_net_prefill_prep: null buffer");
    UZU_REQUIRE((compact_v_out.buffer != nullptr) == (k->f[0] != 0), "compact V output presence mismatch");
    if (k->f[0]) { // tree prep: q / k rounded to the activation type, log decays, the value section copied out
        UZU_REQUIRE(num_k_heads && key_dim == num_k_heads * 128 && num_v_heads && value_dim == num_v_heads * 128, "delta_net_prefill_prep: head_k_dim and head_v_dim are 128");
        return k::delta_net_tree_prep(cb_stream(cb), (const uint16_t*)bptr(in_proj), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, (const float*)bptr(a_log),
                                      (const float*)bptr(dt_bias), (uint16_t*)bptr(q_norm_out), (uint16_t*)bptr(k_norm_out), (uint16_t*)bptr(compact_v_out), (float*)bptr(beta_out),
                                      (float*)bptr(decay_out), suffix_len, 0, num_k_heads, num_v_heads, 128, 128, false, true);
    }
    return k::delta_net_prefill_prep(cb_stream(cb), (const uint16_t*)bptr(in_proj), (const float*)bptr(a_log), (const float*)bptr(dt_bias),
                                     (float*)bptr(q_norm_out), (float*)bptr(k_norm_out), (float*)bptr(beta_out), (float*)bptr(decay_out), num_v_heads,
                                     num_k_heads, 128, key_dim, value_dim, suffix_len);
}
uzu_status uzu_hip_delta_net_prefill_create(uzu_hip_context* ctx, uint32_t t, uint32_t head_k_dim, uzu_hip_kernel** out) {
    UZU_UNSUPPORTED(t != UZU_BF16 || head_k_dim != 128, "delta_net_prefill: variants are (T = BF16, HEAD_K_DIM = 128)");
    uzu_hip_kernel* k;
    return make_kernel(ctx, KK_DN_PREFILL, out, &k);
}
uzu_status uzu_hip_delta_net_prefill_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf q_norm, uzu_buf k_norm, uzu_buf beta, uzu_buf decay,
                                            uzu_buf in_proj, uzu_buf state, uzu_buf out, uint32_t num_v_heads, uint32_t num_k_heads, uint32_t head_v_dim,
                                            uint32_t key_dim, uint32_t value_dim, uint32_t suffix_len, uint32_t num_dv_groups) {
    UZU_PROPAGATE(check(k, KK_DN_PREFILL, cb));
    UZU_REQUIRE(q_norm.buffer && k_norm.buffer && beta.buffer && decay.buffer && in_proj.buffer && state.buffer && out.buffer, "delta_net_prefill: null buffer");
    (void)num_dv_groups;
    return k::delta_net_prefill(cb_stream(cb), (const float*)bptr(q_norm), (const float*)bptr(k_norm), (const float*)bptr(beta), (const float*)bptr(decay),
                                (const uint16_t*)bptr(in_proj), (float*)bptr(state), (uint16_t*)bptr(out), num_v_heads, num_k_heads, 128, head_v_dim,
                                key_dim, value_dim, suffix_len);
}
uzu_status uzu_hip_delta_net_norm_gate_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out) {
    UZU_UNSUPPORTED(t != UZU_BF16, "delta_net_norm_gate: only T = BF16 is instantiated");
    uzu_hip_kernel* k;
    return make_kernel(ctx, KK_DN_NORM_GATE, out, &k);
}
uzu_status uzu_hip_delta_net_norm_gate_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf in_out, uzu_buf in_proj, uzu_buf norm_weight,
                                              uint32_t num_v_heads, uint32_t head_v_dim, uint32_t value_dim, uint32_t conv_dim, uint32_t total_proj_dim,
                                              float norm_epsilon, uint32_t suffix_len) {
    UZU_PROPAGATE(check(k, KK_DN_NORM_GATE, cb));
    UZU_REQUIRE(in_out.buffer && in_proj.buffer && norm_weight.buffer, "delta_net_norm_gate: null buffer");
    return k::delta_net_norm_gate(cb_stream(cb), (uint16_t*)bptr(in_out), (const uint16_t*)bptr(in_proj), (const float*)bptr(norm_weight), num_v_heads,
                                  head_v_dim, value_dim, conv_dim, total_proj_dim, norm_epsilon, suffix_len);
}

// ---- Gated DeltaNet over a speculated tree (cpu/kernel/gdn/tree_verify/*.rs; csrc/k_deltanet_tree.hip) ----
uzu_status uzu_hip_conv_tree_scan_create(uzu_hip_context* ctx, uint32_t t, uint32_t kernel_size, uint32_t has_bias, uzu_hip_kernel** out) {
    UZU_UNSUPPORTED(t != UZU_BF16, "conv_tree_scan: only T = BF16 is instantiated");
    UZU_REQUIRE(kernel_size >= 2, "conv_tree_scan: kernel_size must be >= 2");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_CONV_TREE_SCAN, out, &k));
    k->f[0] = kernel_size, k->f[1] = has_bias;
    return UZU_OK;
}
uzu_status uzu_hip_conv_tree_scan_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf in_proj, uzu_buf conv_weight, uzu_buf bias, uzu_buf base_state, uzu_buf parents,
                                         uzu_buf out_proj, uzu_buf suffix_state, uint32_t suffix_len, uint32_t total_proj_dim, uint32_t conv_dim) {
    UZU_PROPAGATE(check(k, KK_CONV_TREE_SCAN, cb));
    UZU_REQUIRE(in_proj.buffer && conv_weight.buffer && base_state.buffer && parents.buffer && out_proj.buffer && suffix_state.buffer, "conv_tree_scan: null buffer");
    UZU_REQUIRE((bias.buffer != nullptr) == (k->f[1] != 0), "conv_tree_scan: bias presence must equal has_bias");
    // the DeltaNet in-projection row: [q | k | v | z | beta | a] with 128-wide heads => total - conv = 130 Hv, conv = 256 Hk + 128 Hv
    const uint32_t rest = total_proj_dim - conv_dim;
    UZU_UNSUPPORTED(total_proj_dim <= conv_dim || rest % 130 || conv_dim <= (rest / 130) * 128 || (conv_dim - (rest / 130) * 128) % 256,
                    "conv_tree_scan: rows are not a DeltaNet in-projection with 128-wide heads (conv_dim %u of %u)", conv_dim, total_proj_dim);
    const uint32_t Hv = rest / 130, Hk = (conv_dim - Hv * 128) / 256;
    return k::delta_net_tree_prep(cb_stream(cb), (const uint16_t*)bptr(in_proj), (const float*)bptr(conv_weight), (const float*)bptr(bias), (const float*)bptr(base_state),
                                  (const int32_t*)bptr(parents), (uint16_t*)bptr(out_proj), (float*)bptr(suffix_state), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                  nullptr, suffix_len, k->f[0], Hk, Hv, 128, 128, true, false);
}
uzu_status uzu_hip_delta_net_tree_verify_create(uzu_hip_context* ctx, uint32_t t, uint32_t num_k_heads, uint32_t num_v_heads, uint32_t head_k_dim, uint32_t head_v_dim,
                                                uzu_hip_kernel** out) {
    UZU_UNSUPPORTED(t != UZU_BF16 || head_k_dim != 128 || head_v_dim != 128, "delta_net_tree_verify: T = BF16 with 128-wide heads");
    UZU_REQUIRE(num_k_heads && num_v_heads && num_v_heads % num_k_heads == 0, "delta_net_tree_verify: value heads must be a multiple of key heads");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_DN_TREE_VERIFY, out, &k));
    k->f[0] = num_k_heads, k->f[1] = num_v_heads;
    return UZU_OK;
}
uzu_status uzu_hip_delta_net_tree_verify_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf q, uzu_buf k_norm, uzu_buf v, uzu_buf trie, uzu_buf log_decay, uzu_buf beta,
                                                uzu_buf h0, uzu_buf output, uint32_t tree_size) {
    UZU_PROPAGATE(check(k, KK_DN_TREE_VERIFY, cb));
    UZU_REQUIRE(q.buffer && k_norm.buffer && v.buffer && trie.buffer && log_decay.buffer && beta.buffer && h0.buffer && output.buffer, "delta_net_tree_verify: null buffer");
    return k::delta_net_tree_verify(cb_stream(cb), (const uint16_t*)bptr(q), (const uint16_t*)bptr(k_norm), (const uint16_t*)bptr(v), (const uint32_t*)bptr(trie),
                                    (const float*)bptr(log_decay), (const float*)bptr(beta), (const float*)bptr(h0), (uint16_t*)bptr(output), tree_size, k->f[0], k->f[1], 128,
                                    128);
}
uzu_status uzu_hip_state_advance_create(uzu_hip_context* ctx, uint32_t t, uint32_t head_k_dim, uint32_t num_v_heads, uint32_t num_k_heads, uzu_hip_kernel** out) {
    UZU_UNSUPPORTED(t != UZU_BF16 || head_k_dim != 128, "state_advance: variants are (T = BF16, HEAD_K_DIM = 128)");
    UZU_REQUIRE(num_k_heads && num_v_heads && num_v_heads % num_k_heads == 0, "state_advance: value heads must be a multiple of key heads");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_STATE_ADVANCE, out, &k));
    k->f[0] = num_v_heads, k->f[1] = num_k_heads;
    return UZU_OK;
}
uzu_status uzu_hip_state_advance_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf k_norm, uzu_buf v, uzu_buf log_decay, uzu_buf beta, uzu_buf accepted_indices,
                                        uzu_buf state, uint32_t accepted_len) {
    UZU_PROPAGATE(check(k, KK_STATE_ADVANCE, cb));
    UZU_REQUIRE(k_norm.buffer && v.buffer && log_decay.buffer && beta.buffer && accepted_indices.buffer && state.buffer, "state_advance: null buffer");
    return k::delta_net_state_advance(cb_stream(cb), (const uint16_t*)bptr(k_norm), (const uint16_t*)bptr(v), (const float*)bptr(log_decay), (const float*)bptr(beta),
                                      (const uint32_t*)bptr(accepted_indices), (float*)bptr(state), accepted_len, k->f[0], k->f[1], 128);
}

// ---- the tree speculators' kernels (cpu/kernel/attention/ancestor_attention.rs, cpu/kernel/weaver/*.rs; csrc/k_speculator.hip) ----
uzu_status uzu_hip_ancestor_attention_create(uzu_hip_context* ctx, uint32_t head_dim, uint32_t num_heads, uzu_hip_kernel** out) {
    UZU_UNSUPPORTED(head_dim != 128, "ancestor_attention: variants are HEAD_DIM = 128");
    UZU_REQUIRE(num_heads > 0, "ancestor_attention: num_heads must be positive");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_ANCESTOR_ATTENTION, out, &k));
    k->f[0] = head_dim, k->f[1] = num_heads;
    return UZU_OK;
}
uzu_status uzu_hip_ancestor_attention_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf prefix_kv, uzu_buf node_kv, uzu_buf current_qkv, uzu_buf cosines, uzu_buf sines,
                                             uzu_buf node_metadata, uzu_buf ancestor_indices, uzu_buf ancestor_counts, uzu_buf node_indices, uzu_buf output, uint32_t rows,
                                             uint32_t prefix_length, uint32_t ancestor_stride, uint32_t node_capacity, uint32_t max_depth, float scale) {
    UZU_PROPAGATE(check(k, KK_ANCESTOR_ATTENTION, cb));
    UZU_REQUIRE((prefix_kv.buffer || !prefix_length) && node_kv.buffer && current_qkv.buffer && cosines.buffer && sines.buffer && node_metadata.buffer && ancestor_indices.buffer &&
                    ancestor_counts.buffer && node_indices.buffer && output.buffer, "ancestor_attention: null buffer");
    return k::ancestor_attention(cb_stream(cb), (const uint16_t*)bptr(prefix_kv), (uint16_t*)bptr(node_kv), (const uint16_t*)bptr(current_qkv), (const float*)bptr(cosines),
                                 (const float*)bptr(sines), (const uint32_t*)bptr(node_metadata), (const uint32_t*)bptr(ancestor_indices), (const uint32_t*)bptr(ancestor_counts),
                                 (const uint32_t*)bptr(node_indices), (uint16_t*)bptr(output), rows, prefix_length, ancestor_stride, node_capacity, max_depth, scale, k->f[1], k->f[0]);
}
uzu_status uzu_hip_weaver_frontier_select_create(uzu_hip_context* ctx, uzu_hip_kernel** out) {
    uzu_hip_kernel* k;
    return make_kernel(ctx, KK_WEAVER_SELECT, out, &k);
}
uzu_status uzu_hip_weaver_frontier_select_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf frontier, uzu_buf packed_tree, uzu_buf slot_ancestors, uzu_buf node_token_ids,
                                                 uzu_buf node_metadata, uzu_buf node_ancestor_indices, uzu_buf node_valid, uzu_buf candidate_pool_ids, uzu_buf candidate_pool_logits,
                                                 uzu_buf node_candidate_ids, uzu_buf node_candidate_logits, uint32_t frontier_capacity, uint32_t tree_slot_count, uint32_t node_count,
                                                 uint32_t batch_start_slot, uint32_t ancestor_stride, uint32_t max_depth, uint32_t lookahead_count, uint32_t candidate_depth_count,
                                                 uint32_t candidates_per_depth) {
    UZU_PROPAGATE(check(k, KK_WEAVER_SELECT, cb));
    UZU_REQUIRE(frontier.buffer && packed_tree.buffer && slot_ancestors.buffer && node_token_ids.buffer && node_metadata.buffer && node_ancestor_indices.buffer && node_valid.buffer &&
                    candidate_pool_ids.buffer && candidate_pool_logits.buffer && node_candidate_ids.buffer && node_candidate_logits.buffer, "weaver_frontier_select: null buffer");
    return k::weaver_frontier_select(cb_stream(cb), (uint32_t*)bptr(frontier), (uint32_t*)bptr(packed_tree), (uint32_t*)bptr(slot_ancestors), (uint32_t*)bptr(node_token_ids),
                                     (uint32_t*)bptr(node_metadata), (uint32_t*)bptr(node_ancestor_indices), (uint32_t*)bptr(node_valid), (const uint32_t*)bptr(candidate_pool_ids),
                                     (const float*)bptr(candidate_pool_logits), (uint32_t*)bptr(node_candidate_ids), (float*)bptr(node_candidate_logits), frontier_capacity,
                                     tree_slot_count, node_count, batch_start_slot, ancestor_stride, max_depth, lookahead_count, candidate_depth_count, candidates_per_depth);
}
uzu_status uzu_hip_weaver_frontier_insert_children_create(uzu_hip_context* ctx, uzu_hip_kernel** out) {
    uzu_hip_kernel* k;
    return make_kernel(ctx, KK_WEAVER_INSERT, out, &k);
}
uzu_status uzu_hip_weaver_frontier_insert_children_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf packed_tree, uzu_buf node_metadata, uzu_buf node_valid, uzu_buf child_ids,
                                                          uzu_buf child_logprobs, uzu_buf frontier, uint32_t frontier_capacity, uint32_t tree_slot_count, uint32_t node_count,
                                                          uint32_t expand_width) {
    UZU_PROPAGATE(check(k, KK_WEAVER_INSERT, cb));
    UZU_REQUIRE(packed_tree.buffer && node_metadata.buffer && node_valid.buffer && child_ids.buffer && child_logprobs.buffer && frontier.buffer, "weaver_frontier_insert_children: null buffer");
    return k::weaver_frontier_insert_children(cb_stream(cb), (const uint32_t*)bptr(packed_tree), (const uint32_t*)bptr(node_metadata), (const uint32_t*)bptr(node_valid),
                                              (const uint32_t*)bptr(child_ids), (const float*)bptr(child_logprobs), (uint32_t*)bptr(frontier), frontier_capacity, tree_slot_count,
                                              node_count, expand_width);
}
uzu_status uzu_hip_weaver_top_children_create(uzu_hip_context* ctx, uzu_hip_kernel** out) {
    uzu_hip_kernel* k;
    return make_kernel(ctx, KK_WEAVER_TOP_CHILDREN, out, &k);
}
uzu_status uzu_hip_weaver_top_children_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf residual_logits, uzu_buf candidate_logits, uzu_buf candidate_ids, uzu_buf depth_seeds,
                                              uzu_buf node_metadata, uzu_buf output_token_ids, uzu_buf output_model_logprobs, uint32_t rows, uint32_t candidates,
                                              uint32_t expand_width, uint32_t vocab_size) {
    UZU_PROPAGATE(check(k, KK_WEAVER_TOP_CHILDREN, cb));
    UZU_REQUIRE(residual_logits.buffer && candidate_logits.buffer && candidate_ids.buffer && depth_seeds.buffer && node_metadata.buffer && output_token_ids.buffer &&
                    output_model_logprobs.buffer, "weaver_top_children: null buffer");
    return k::weaver_top_children(cb_stream(cb), (const uint16_t*)bptr(residual_logits), (const float*)bptr(candidate_logits), (const uint32_t*)bptr(candidate_ids),
                                  (const uint64_t*)bptr(depth_seeds), (const uint32_t*)bptr(node_metadata), (uint32_t*)bptr(output_token_ids), (float*)bptr(output_model_logprobs),
                                  rows, candidates, expand_width, vocab_size);
}

// ------------------------------------------------------------------------------------- Mixture of experts (k_moe.hip; bf16 only)
#define REQ_BF16(dt, what) UZU_UNSUPPORTED((dt) != UZU_BF16, what ": data type %u (the MoE kernels are built for BF16, LanguageModel's data type)", (unsigned)(dt))
uzu_status uzu_hip_moe_router_top_k_create(uzu_hip_context* ctx, uint32_t scalar_t, uint32_t has_biases, uint32_t has_router_scales, uint32_t has_per_expert_scales,
                                           uint32_t has_router_input_scale, uint32_t normalize_router_input, uzu_hip_kernel** out) {
    REQ_BF16(scalar_t, "moe_router_top_k");
    // MoeBlock::new instantiates (true, false, false, false, false) (mod.rs:170-172); the Gemma-4 router inputs are not built
    UZU_UNSUPPORTED(has_router_scales || has_per_expert_scales || has_router_input_scale || normalize_router_input, "moe_router_top_k: router scales / input normalisation");
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_MOE_ROUTER_TOPK, out, &k));
    k->f[0] = has_biases;
    return UZU_OK;
}
uzu_status uzu_hip_moe_router_top_k_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf input, uzu_buf weight, uzu_buf bias, uzu_buf topk_ids, uzu_buf topk_probs, uint32_t t, uint32_t d_model,
                                           uint32_t e, uint32_t top_k, uint32_t renorm) {
    UZU_PROPAGATE(check(k, KK_MOE_ROUTER_TOPK, cb));
    UZU_REQUIRE(input.buffer && weight.buffer && topk_ids.buffer && topk_probs.buffer && (!k->f[0] || bias.buffer), "moe_router_top_k: null buffer");
    return k::moe_router_topk(cb_stream(cb), (const uint16_t*)bptr(input), (const uint16_t*)bptr(weight), k->f[0] ? (const uint16_t*)bptr(bias) : nullptr, (int32_t*)bptr(topk_ids),
                              (uint16_t*)bptr(topk_probs), t, d_model, e, top_k, renorm);
}
uzu_status uzu_hip_moe_counts_offsets_fused_create(uzu_hip_context* ctx, uzu_hip_kernel** out) {
    uzu_hip_kernel* k;
    return make_kernel(ctx, KK_MOE_COUNTS_OFFSETS, out, &k);
}
uzu_status uzu_hip_moe_counts_offsets_fused_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf topk_ids, uzu_buf offsets, uzu_buf sum_k_out, uzu_buf partials, uint32_t t, uint32_t e,
                                                   uint32_t top_k) {
    UZU_PROPAGATE(check(k, KK_MOE_COUNTS_OFFSETS, cb));
    UZU_REQUIRE(topk_ids.buffer && offsets.buffer && sum_k_out.buffer, "moe_counts_offsets_fused: null buffer");
    return k::moe_counts_offsets(cb_stream(cb), (const int32_t*)bptr(topk_ids), (uint32_t*)bptr(offsets), (uint32_t*)bptr(sum_k_out), (uint32_t*)bptr(partials), t, e, top_k);
}
// MoeBlockBasesFromPartials + MoeScatterBucketsMap (+ MoePassABuildRowMap) as ONE kernel: block bases / allocations are Metal's way of ordering its threadgroups; here the
// bucket order is (token, slot) by construction.  Arguments = the map kernel's (scatter_buckets.rs:24-41) without the block tables, plus the row -> expert map.
uzu_status uzu_hip_moe_scatter_buckets_map_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out) {
    REQ_BF16(t, "moe_scatter_buckets_map");
    uzu_hip_kernel* k;
    return make_kernel(ctx, KK_MOE_SCATTER, out, &k);
}
uzu_status uzu_hip_moe_scatter_buckets_map_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf topk_ids, uzu_buf topk_probs, uzu_buf offsets, uzu_buf out_ids, uzu_buf out_probs, uint32_t t,
                                                  uint32_t e, uint32_t top_k, uzu_buf tok2row, uzu_buf row_expert_map) {
    UZU_PROPAGATE(check(k, KK_MOE_SCATTER, cb));
    UZU_REQUIRE(topk_ids.buffer && topk_probs.buffer && offsets.buffer && out_ids.buffer && out_probs.buffer && tok2row.buffer && row_expert_map.buffer, "moe_scatter_buckets_map: null buffer");
    return k::moe_scatter_buckets(cb_stream(cb), (const int32_t*)bptr(topk_ids), (const uint16_t*)bptr(topk_probs), (const uint32_t*)bptr(offsets), (int32_t*)bptr(out_ids),
                                  (uint16_t*)bptr(out_probs), (int32_t*)bptr(tok2row), (uint32_t*)bptr(row_expert_map), t, e, top_k);
}
uzu_status uzu_hip_moe_gather_x_perm_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out) {
    REQ_BF16(t, "moe_gather_x_perm");
    uzu_hip_kernel* k;
    return make_kernel(ctx, KK_MOE_GATHER, out, &k);
}
uzu_status uzu_hip_moe_gather_x_perm_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf x, uzu_buf bucketed_ids, uzu_buf x_perm, uzu_buf sumk_buf, uint32_t d_model, uint32_t t, uint32_t top_k) {
    UZU_PROPAGATE(check(k, KK_MOE_GATHER, cb));
    UZU_REQUIRE(x.buffer && bucketed_ids.buffer && x_perm.buffer && sumk_buf.buffer, "moe_gather_x_perm: null buffer");
    return k::moe_gather(cb_stream(cb), (const uint16_t*)bptr(x), (const int32_t*)bptr(bucketed_ids), (uint16_t*)bptr(x_perm), (const uint32_t*)bptr(sumk_buf), d_model, t, top_k);
}
// MoeExperts{Decode,Prefill}PassA without Metal's tile map / indirect dispatch buffer: rows are found through the row -> expert map, `capacity` rows are launched
uzu_status uzu_hip_moe_experts_pass_a_create(uzu_hip_context* ctx, uint32_t t, uint32_t gating_sel, uzu_hip_kernel** out) {
    REQ_BF16(t, "moe_experts_pass_a");
    UZU_REQUIRE(gating_sel <= 3, "moe_experts_pass_a: gating_sel %u", gating_sel);
    uzu_hip_kernel* k;
    UZU_PROPAGATE(make_kernel(ctx, KK_MOE_PASS_A, out, &k));
    k->f[0] = gating_sel;
    return UZU_OK;
}
uzu_status uzu_hip_moe_experts_pass_a_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf x_perm, uzu_buf row_expert_map, uzu_buf sumk_buf, uzu_buf w13_all, uzu_buf up_biases, uzu_buf hidden_out,
                                             uint32_t d_model, uint32_t d_ff, float gate_clip_min, float gate_clip_max, float up_clip_min, float up_clip_max, float silu_alpha, uint32_t capacity) {
    UZU_PROPAGATE(check(k, KK_MOE_PASS_A, cb));
    UZU_REQUIRE(x_perm.buffer && row_expert_map.buffer && sumk_buf.buffer && w13_all.buffer && up_biases.buffer && hidden_out.buffer, "moe_experts_pass_a: null buffer");
    const k::MoeExpertParams q{d_model, d_ff, k->f[0], gate_clip_min, gate_clip_max, up_clip_min, up_clip_max, silu_alpha};
    return k::moe_experts_pass_a(cb_stream(cb), (const uint16_t*)bptr(x_perm), (const uint32_t*)bptr(row_expert_map), (const uint32_t*)bptr(sumk_buf), (const uint16_t*)bptr(w13_all),
                                 (const uint16_t*)bptr(up_biases), (float*)bptr(hidden_out), q, capacity);
}
uzu_status uzu_hip_moe_experts_down_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out) {
    REQ_BF16(t, "moe_experts_down");
    uzu_hip_kernel* k;
    return make_kernel(ctx, KK_MOE_DOWN, out, &k);
}
uzu_status uzu_hip_moe_experts_down_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf hidden, uzu_buf row_expert_map, uzu_buf sumk_buf, uzu_buf w2_all, uzu_buf down_biases, uzu_buf y_out,
                                           uint32_t d_model, uint32_t d_ff, uint32_t capacity) {
    UZU_PROPAGATE(check(k, KK_MOE_DOWN, cb));
    UZU_REQUIRE(hidden.buffer && row_expert_map.buffer && sumk_buf.buffer && w2_all.buffer && down_biases.buffer && y_out.buffer, "moe_experts_down: null buffer");
    return k::moe_experts_down(cb_stream(cb), (const float*)bptr(hidden), (const uint32_t*)bptr(row_expert_map), (const uint32_t*)bptr(sumk_buf), (const uint16_t*)bptr(w2_all),
                               (const uint16_t*)bptr(down_biases), (uint16_t*)bptr(y_out), d_model, d_ff, capacity);
}
uzu_status uzu_hip_moe_finalize_create(uzu_hip_context* ctx, uint32_t t, uzu_hip_kernel** out) {
    REQ_BF16(t, "moe_finalize");
    uzu_hip_kernel* k;
    return make_kernel(ctx, KK_MOE_FINALIZE, out, &k);
}
uzu_status uzu_hip_moe_finalize_encode(uzu_hip_kernel* k, uzu_hip_cmdbuf* cb, uzu_buf tok2row, uzu_buf probs, uzu_buf y_partial, uzu_buf y, uint32_t t_count, uint32_t d_model, uint32_t top_k) {
    UZU_PROPAGATE(check(k, KK_MOE_FINALIZE, cb));
    UZU_REQUIRE(tok2row.buffer && probs.buffer && y_partial.buffer && y.buffer, "moe_finalize: null buffer");
    return k::moe_finalize(cb_stream(cb), (const int32_t*)bptr(tok2row), (const uint16_t*)bptr(probs), (const uint16_t*)bptr(y_partial), (uint16_t*)bptr(y), t_count, d_model, top_k);
}

} // extern "C"
