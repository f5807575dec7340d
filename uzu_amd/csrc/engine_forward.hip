// engine_forward.hip -- model driver above the kernel boundary (include/uzu_hip_engine.h): the forward encoders.
//
// Restates, for one sequence, the op order of the reference's backend-generic graph code:
//   Decoder::encode            BU/../encodable_block/decoder.rs:138-203
//   Transformer::encode        BU/../encodable_block/transformer.rs:226-329
//   TransformerLayer::encode   BU/../encodable_block/transformer_layer.rs:194-238
//   Attention::attend          BU/../encodable_block/mixer/attention/mode.rs:45-144
//   AttentionCores::encode     BU/../encodable_block/mixer/attention/core/mod.rs:81-93
//   DeltaNet::encode           BU/../encodable_block/mixer/delta_net.rs:473-645
//   DenseMlp::encode           BU/../encodable_block/mlp/dense.rs:32-48
//   Embedding::encode_readout  BU/../encodable_block/embedding.rs:374-456
//   LanguageModelStream        BU/../engine/language_model/stream/stream.rs:190-345 (prefill), 593-751 (decode)
// MI355X execution strategy: weights, KV cache and DeltaNet state resident in HBM; the context length,
// the next input token and the sampled-token history live in device memory, so ONE captured hipGraph is
// replayed for every decode step and steps are chained without a host round trip.
#include "engine_types.h"

using namespace uzu;
using namespace uzu::eng;

namespace uzu {
namespace eng {


// ---------------------------------------------------------------------------------- encoding helpers

// `row_parallel`: under tensor parallelism this linear's K is split over the ranks (out-proj, down-proj): the matmul
// writes f32 partial sums, the ranks all-reduce them, and the sum is rounded to bf16 into `output`.
//
// RHT linears (RHTLinearWrapper::encode_input, linear/rht_wrapper.rs:215-298, full-precision activation format): InputRht on a
// copy of the rows (the reference transforms its own allocation in place), the inner matmul without its bias, OutputRht in place
// on the result, then the bias (MatmulDOps::rht_factors, kernel.rs:296-303).
// QLoRALinearWrapper::encode (linear/qlora_wrapper.rs:177-251): intermediate = x down^T; base input = InputRht of a copy of the rows (when the
// spec carries signs); output = base matmul (no bias); output += intermediate up^T (MatmulDOps::accumulate); OutputRht in place.
void linear_qlora(Enc& e, const DLinear& L, const uint16_t* input, uint16_t* output, uint32_t batch) {
    uzu_hip_model* m = e.m;
    auto fp = [&](const uint16_t* a, const uint16_t* b, uint16_t* d, uint32_t n, uint32_t k, bool accumulate) {
        k::MatmulParams p{};
        p.a = a, p.b = b, p.d = d, p.w_dt = p.a_dt = p.d_dt = UZU_BF16, p.b_kind = UZU_MATMUL_B_FULL_PRECISION, p.bits = 16, p.ab_scale = 1.0f;
        p.accumulate = accumulate ? 1u : 0u, p.m = batch, p.n = n, p.k = k;
        const char* variant = "matmul";
        e.begin();
        const uzu_status r = k::matmul(e.s, p, m->ctx->num_cus, &variant);
        e.run(r, "matmul_adapter", k::matmul_algorithmic_bytes(p));
    };
    fp(input, L.adapter_down, m->lora_scratch, L.lora_rank, L.k, false);
    const uint16_t* base_input = input;
    if (L.in_signs) {
        RUN("activation_transform", 0, k::activation_transform(e.s, input, m->rht_scratch, nullptr, nullptr, nullptr, L.in_signs, UZU_BF16, batch, L.k,
                                                                UZU_ACTIVATION_TRANSFORM_INPUT_RHT, 0, 0));
        base_input = m->rht_scratch;
    }
    k::MatmulParams p{};
    p.a = base_input, p.b = L.w, p.scales = L.scales, p.biases = L.biases, p.zero_points = L.zp, p.d = output;
    p.w_dt = p.a_dt = p.d_dt = UZU_BF16;
    p.b_kind = L.method == UZU_QUANT_SCALE_BIAS ? UZU_MATMUL_B_SCALE_BIAS : L.method == UZU_QUANT_SCALE_ZERO_POINT ? UZU_MATMUL_B_SCALE_ZERO_POINT : UZU_MATMUL_B_SCALE_SYMMETRIC;
    p.bits = L.bits, p.group_size = L.group, p.ab_scale = 1.0f, p.m = batch, p.n = L.n, p.k = L.k;
    const char* variant = "matmul";
    e.begin();
    const uzu_status r = k::matmul(e.s, p, m->ctx->num_cus, &variant);
    e.run(r, variant, k::matmul_algorithmic_bytes(p));
    fp(m->lora_scratch, L.adapter_up, output, L.n, L.lora_rank, true);
    if (L.out_signs)
        RUN("activation_transform", 0, k::activation_transform(e.s, nullptr, output, nullptr, nullptr, nullptr, L.out_signs, UZU_BF16, batch, L.n,
                                                                UZU_ACTIVATION_TRANSFORM_OUTPUT_RHT, 0, 0));
}

// the large-tile GEMM's offset tables where the engine has them: coefficients from load time, row sums of `normed` from its normalisation
static void offset_tables(uzu_hip_model* m, const DLinear& L, const uint16_t* input, uint32_t batch, k::MatmulParams* p) {
    static const bool enabled = [] { // UZU_GEMM_TABLES=0: the GEMM's own pre-pass launch every time (A/B runs)
        const char* v = lab_env("UZU_GEMM_TABLES");
        return !v || atoi(v) != 0;
    }();
    if (!enabled || batch < 128) return;
    p->pre_coef = L.coef;
    if (input == m->normed && m->rs_rows == batch && m->rs_k == L.k && m->rs_group == L.group) p->pre_rowsum = m->rowsum;
    const uzu_hip_model::FiledRowSums* f = input == m->gated ? &m->rs_gated : input == m->delta_out ? &m->rs_delta : nullptr;
    if (f && f->rows == batch && f->k == L.k && f->part && L.group % f->part == 0) {
        p->pre_rowsum = f->buf;
        p->rowsum_parts_log2 = 31u - (uint32_t)__builtin_clz(L.group / f->part);
    }
}
// whether the producer of a prefill GEMM's activation rows should file their row sums in parts of `part` columns for `consumer` (the large-tile kernel reads them
// instead of launching its pre-pass): same conditions as the normalisation's (norm_params), plus a part that divides the quant group a power-of-two times
bool rowsum_wanted(const uzu_hip_model* m, const DLinear& consumer, uint32_t rows, uint32_t part, const uzu_hip_model::FiledRowSums& f) {
    static const bool enabled = [] { // UZU_GEMM_TABLES=0 (lab builds): the GEMM's own pre-pass launch every time
        const char* v = lab_env("UZU_GEMM_TABLES");
        return !v || atoi(v) != 0;
    }();
    static const int lab_mask = [] { // UZU_LAB_RS_MASK (lab builds): bit 0 = the GatedActMul epilogue files, bit 1 = the DeltaNet norm-gate files
        const char* v = lab_env("UZU_LAB_RS_MASK");
        return v ? atoi(v) : 3;
    }();
    if (!(lab_mask & (&f == &m->rs_gated ? 1 : 2))) return false;
    // Measured (tools/rs_llama.sh, profiles/r6_rowsum_producers.txt): at K = 14336 / 4096 (Llama-3-8B) the pre-pass launch is cheaper than it looks -- it leaves the
    // rows it sums in the Infinity Cache for the GEMM behind it -- and filing the sums elsewhere LOSES 0.2-0.9 % of a pass; at K <= 3584 (the 0.8B's projections) it wins
    if (consumer.k > 4096) return false;
    if (!enabled || rows < 128 || !consumer.coef || consumer.in_signs || consumer.lora_rank || k::exact_mode() || !f.buf || !part || consumer.group % part || consumer.k % part) return false;
    const uint32_t ratio = consumer.group / part;
    if ((ratio & (ratio - 1)) || ratio > 4) return false;
    return (size_t)(consumer.k / part) * ((rows + 3) & ~3u) <= f.floats;
}

void linear(Enc& e, const DLinear& L, const uint16_t* input, uint16_t* output, uint32_t batch, bool row_parallel, PostNorm* post) {
    if (L.lora_rank) return linear_qlora(e, L, input, output, batch); // (tensor-parallel shards of QLoRA linears are refused by the planner)
    const bool exchange = row_parallel && e.m->tp != nullptr;
    if (L.in_signs) {
        RUN("activation_transform", 0, k::activation_transform(e.s, input, e.m->rht_scratch, nullptr, nullptr, nullptr, L.in_signs, UZU_BF16, batch, L.k,
                                                                UZU_ACTIVATION_TRANSFORM_INPUT_RHT, 0, 0));
        input = e.m->rht_scratch;
    }
    k::MatmulParams p{};
    p.a = input, p.b = L.w, p.scales = L.scales, p.biases = L.biases, p.zero_points = L.zp, p.d = output, p.bias = L.out_biases;
    p.w_dt = p.a_dt = p.d_dt = UZU_BF16;
    if (L.out_signs) p.bias = nullptr; // bias_after_rht
    if (exchange) p.d = e.m->tp_buf, p.d_dt = UZU_F32;
    p.b_kind = L.method == UZU_QUANT_NONE ? UZU_MATMUL_B_FULL_PRECISION
             : L.method == UZU_QUANT_SCALE_BIAS ? UZU_MATMUL_B_SCALE_BIAS
             : L.method == UZU_QUANT_SCALE_ZERO_POINT ? UZU_MATMUL_B_SCALE_ZERO_POINT : UZU_MATMUL_B_SCALE_SYMMETRIC;
    p.bits = L.bits, p.group_size = L.group, p.ab_scale = 1.0f;
    p.m = batch, p.n = L.n, p.k = L.k;
    offset_tables(e.m, L, input, batch, &p);
    if (post && !exchange && !L.out_signs && batch >= 128) p.post_norm = &post->p, p.post_norm_done = &post->done;
    const char* variant = "matmul";
    e.begin();
    const uzu_status r = k::matmul(e.s, p, e.m->ctx->num_cus, &variant);
    e.run(r, post && post->done ? "gemm_q_mfma128+norm" : variant, k::matmul_algorithmic_bytes(p));
    if (exchange) {
        const size_t count = (size_t)batch * L.n;
        RUN("all_reduce", count * 4, tp::all_reduce_sum_f32(e.m->tp, e.s, e.m->tp_buf, count, output)); // sums rounded to bf16 into `output`
    }
    if (L.out_signs) {
        RUN("activation_transform", 0, k::activation_transform(e.s, nullptr, output, nullptr, nullptr, nullptr, L.out_signs, UZU_BF16, batch, L.n,
                                                                UZU_ACTIVATION_TRANSFORM_OUTPUT_RHT, 0, 0));
        if (L.out_biases) RUN("tensor_add_bias", 0, k::tensor_add_bias(e.s, output, L.out_biases, output, UZU_BF16, UZU_BF16, L.n, (size_t)batch * L.n));
    }
}

// up projection + GatedActMul in one kernel (the matrix-core GEMM's epilogue pairs the up and gate columns of an output);
// false = not available for this shape / mode: the caller runs the two kernels
bool linear_gated(Enc& e, const DLinear& L, const uint16_t* input, uint16_t* gated_out, uint32_t batch, uint32_t act_type, const DLinear* consumer) {
    e.m->rs_gated.rows = 0; // whoever writes `gated` next, the filed sums are stale
    static const bool enabled = [] {
        const char* v = lab_env("UZU_GEMM_ACT");
        return !v || atoi(v) != 0;
    }();
    // RHT / QLoRA linears run as their wrappers compose them: the fused GEMM knows nothing of the adapter term (x down^T) up^T
    if (!enabled || L.in_signs || L.out_signs || L.out_biases || L.lora_rank || L.method == UZU_QUANT_NONE || (e.m->flags & UZU_MODEL_NO_FUSION)) return false;
    k::MatmulParams p{};
    p.a = input, p.b = L.w, p.scales = L.scales, p.biases = L.biases, p.zero_points = L.zp, p.d = gated_out;
    p.w_dt = p.a_dt = p.d_dt = UZU_BF16;
    p.b_kind = L.method == UZU_QUANT_SCALE_BIAS ? UZU_MATMUL_B_SCALE_BIAS : L.method == UZU_QUANT_SCALE_ZERO_POINT ? UZU_MATMUL_B_SCALE_ZERO_POINT : UZU_MATMUL_B_SCALE_SYMMETRIC;
    p.bits = L.bits, p.group_size = L.group, p.ab_scale = 1.0f;
    p.m = batch, p.n = L.n, p.k = L.k;
    if (!k::matmul_act_mul_supported(e.s, p, e.m->ctx->num_cus)) return false;
    p.act_mul = 1, p.act_type = act_type;
    offset_tables(e.m, L, input, batch, &p);
    // the down projection's row sums from this epilogue (64 gated columns per workgroup): no pre-pass launch in front of it
    const bool file = consumer && gated_out == e.m->gated && consumer->k == L.n / 2 && k::gemm_q_mfma128_supported(p, e.m->ctx->num_cus) && rowsum_wanted(e.m, *consumer, batch, 64, e.m->rs_gated);
    if (file) p.gated_rowsum_out = e.m->rs_gated.buf;
    const char* variant = "matmul";
    e.begin();
    const uzu_status r = k::matmul(e.s, p, e.m->ctx->num_cus, &variant);
    e.run(r, "gemm_q_mfma+act", k::matmul_algorithmic_bytes(p));
    if (file) e.m->rs_gated.rows = batch, e.m->rs_gated.k = L.n / 2, e.m->rs_gated.part = 64;
    return true;
}

bool norm_fusable(const DNorm& N);
// Two or three rows (small speculative verify passes, prefill tails): the Normalization as the PROLOGUE of the linear that reads its rows
// (k_gemv_rows.hip, RowsNorm) -- one launch instead of two, the rows bit-identical to the separate kernel's.  mode 1 copy / 2 add; the residual rows go
// from `sc_in` to `sc_out` (two buffers: the other workgroups still read sc_in); `normed_out` (optional) receives the normalised rows for a second
// linear.  gated: L is the fused up | gate matrix and `output` = GatedActMul of its halves.  false = not available: the caller runs the separate kernels.
bool linear_normed(Enc& e, const DNorm& N, int mode, const DLinear& L, const uint16_t* x, const uint16_t* sc_in, uint16_t* sc_out, uint16_t* normed_out, uint16_t* output,
                   uint32_t rows, bool gated, uint32_t act_type) {
    const char* env = tune_env("rows_norm"); // =0: the separate Normalization launch (A/B runs, tests; read when a pass is encoded or captured)
    const bool enabled = !env || atoi(env) != 0;
    uzu_hip_model* m = e.m;
    // rows <= 3 only: EVERY workgroup of the linear normalises all the rows it stages (hundreds of workgroups x rows x two passes through the L2), which
    // costs more than the launch it saves from 4 rows on -- measured on Qwen3.5-0.8B (profiles/r5_verify_cost.json, `rows_norm_ab`): 2 nodes 1560 -> 1492 us,
    // 4 nodes 1603 -> 1656, 8 nodes 1711 -> 1996, 16 nodes 1959 -> 2794 with the prologue at every size
    static const uint32_t max_rows = [] {
        const char* v = lab_env("UZU_ROWS_NORM_MAX");
        return (uint32_t)(v && atoi(v) > 0 ? atoi(v) : 3);
    }();
    if (!enabled || rows < 2 || rows > max_rows || rows > 16 || k::exact_mode() || (m->flags & UZU_MODEL_NO_FUSION) || !norm_fusable(N) || mode == 0) return false;
    if (L.in_signs || L.out_signs || L.lora_rank || L.method == UZU_QUANT_NONE || L.bits != 4 || (gated && L.out_biases)) return false;
    k::MatmulParams p{};
    p.a = x, p.b = L.w, p.scales = L.scales, p.biases = L.biases, p.zero_points = L.zp, p.d = output, p.bias = L.out_biases;
    p.w_dt = p.a_dt = p.d_dt = UZU_BF16;
    p.b_kind = L.method == UZU_QUANT_SCALE_BIAS ? UZU_MATMUL_B_SCALE_BIAS : L.method == UZU_QUANT_SCALE_ZERO_POINT ? UZU_MATMUL_B_SCALE_ZERO_POINT : UZU_MATMUL_B_SCALE_SYMMETRIC;
    p.bits = L.bits, p.group_size = L.group, p.ab_scale = 1.0f, p.m = rows, p.n = L.n, p.k = L.k;
    if (gated) p.act_mul = 1, p.act_type = act_type;
    if (!k::gemv_rows_norm_supported(p)) return false;
    k::RowsNorm rn{};
    rn.scales = N.scales, rn.eps = N.eps, rn.offset = N.offset, rn.full_layer = N.full_layer, rn.residual_add = mode == 2;
    rn.shortcut_in = mode == 2 ? sc_in : nullptr, rn.shortcut_out = sc_out, rn.normed_out = normed_out;
    e.begin();
    e.run(k::gemv_rows_mfma(e.s, p, &rn), gated ? "gemv_rows[norm+up+act]" : "gemv_rows[norm+linear]", k::matmul_algorithmic_bytes(p));
    return true;
}

// mode: 0 none, 1 copy, 2 add (ShortcutMode, encodable_block/normalization.rs:22-27)
// `consumer`: the quantised linear that reads `output` next as a prefill GEMM: the kernel then files the group row sums of the rows it writes
k::NormParams norm_params(Enc& e, const DNorm& N, const uint16_t* input, uint16_t* output, uint16_t* shortcut, int mode, uint32_t rows, uint32_t dim, const DLinear* consumer) {
    k::NormParams p{};
    p.input = input, p.scales = N.scales, p.biases = N.biases, p.output = output, p.shortcut = mode ? shortcut : nullptr;
    p.io_dt = UZU_BF16, p.affine_dt = UZU_F32;
    p.batch_size = rows, p.element_count = dim;
    p.epsilon = N.eps, p.scale_offset = N.offset, p.post_layer_scalar = N.scalar_mode ? N.scalar : 1.0f;
    p.scale_residual_sum = N.scalar_mode == 1, p.scale_output = N.scalar_mode == 2;
    p.subtract_mean = N.subtract_mean, p.full_layer = N.full_layer;
    p.copy_to_shortcut = mode != 0, p.residual_add = mode == 2;
    uzu_hip_model* m = e.m;
    // (dim <= 2048: see rowsum_wanted -- at Llama-3-8B's 4096 the GEMM's own pre-pass, which warms the cache with the rows, is the faster route by 0.7-0.9 % of a pass)
    if (consumer && consumer->coef && !consumer->in_signs && !consumer->lora_rank && output == m->normed && rows >= 128 && consumer->k == dim && dim <= 2048 && !k::exact_mode() &&
        k::normalization_rowsum_supported(dim, consumer->group) && (size_t)(dim / consumer->group) * ((rows + 3) & ~3u) <= m->rowsum_floats)
        p.rowsum_out = m->rowsum, p.rowsum_group = consumer->group;
    return p;
}
// book-keeping of the filed row sums, at the point where the normalisation `p` is ISSUED (its parameters may have been drawn up earlier: PostNorm)
void norm_issued(uzu_hip_model* m, const k::NormParams& p) {
    if (p.output != m->normed) return;
    m->rs_rows = 0; // whatever was filed for the old rows is stale
    if (p.rowsum_out) m->rs_rows = p.batch_size, m->rs_k = p.element_count, m->rs_group = p.rowsum_group;
}
void norm(Enc& e, const DNorm& N, const uint16_t* input, uint16_t* output, uint16_t* shortcut, int mode, uint32_t rows, uint32_t dim, const DLinear* consumer) {
    const k::NormParams p = norm_params(e, N, input, output, shortcut, mode, rows, dim, consumer);
    norm_issued(e.m, p);
    RUN("normalization", 0, k::normalization(e.s, p));
}

// whole logit rows for stochastic sampling on a vocab shard (allocated on first use: set_sampling / a stochastic tree pass under TP)
uzu_status ensure_tp_gather(uzu_hip_model* m, uint32_t rows) {
    if (!m->tp || m->tp_gather_rows >= rows) return UZU_OK;
    void* p = nullptr;
    if (m->tp_gather_f32) dev_free(m, m->tp_gather_f32), dev_free(m, m->tp_gather_bf16);
    m->tp_gather_f32 = nullptr, m->tp_gather_bf16 = nullptr, m->tp_gather_rows = 0;
    UZU_PROPAGATE(dev_alloc(m, (size_t)rows * m->d.vocab_size * 4, &p));
    m->tp_gather_f32 = (float*)p;
    UZU_PROPAGATE(dev_alloc(m, (size_t)rows * m->d.vocab_size * 2, &p));
    m->tp_gather_bf16 = (uint16_t*)p;
    m->tp_gather_rows = rows;
    return UZU_OK;
}

uzu_status ensure_partials(uzu_hip_model* m, uint32_t rows, uint32_t head_dim) {
    if (rows <= m->partial_rows) return UZU_OK;
    // regrow: the old blocks stay allocated until the model is destroyed -- captured decode graphs (this state's and every
    // other uzu_hip_state's graph_two on the unfused path) carry their addresses in attention_two_pass1/2 nodes, and nothing
    // re-captures them on a regrow.  They are small (decode rows) next to the prefill-sized blocks that replace them.
    void* p;
    UZU_PROPAGATE(dev_alloc(m, (size_t)rows * 32 * head_dim * 4, &p));
    m->partials = (float*)p;
    UZU_PROPAGATE(dev_alloc(m, (size_t)rows * 32 * 4, &p));
    m->sums = (float*)p;
    UZU_PROPAGATE(dev_alloc(m, (size_t)rows * 32 * 4, &p));
    m->maxs = (float*)p;
    m->partial_rows = rows;
    return UZU_OK;
}

// The sequences of one forward pass.  n == 0: the bound state, `count` rows (the plain single-sequence pass).  n >= 1:
// `count` rows per sequence, sequence q in rows [q * count, (q + 1) * count) of every activation buffer: the linear
// layers, norms and element-wise kernels run once over all n * count rows (one GEMM with M = n * count: the weights are
// streamed once for all sequences), attention / KV append / DeltaNet run per sequence on its own state.  The reference
// has no cross-sequence batching: per sequence the arithmetic is that of the single-sequence pass.

void attention_core(Enc& e, DLayer& L, uint32_t batch, size_t row0, bool fuse_norms);

static bool attention_fusions_enabled() {
    const char* e = tune_env("attn_fused");
    return !e || atoi(e) != 0;
}
// `first_done`: the layer's first projection (qkv; the DeltaNet in-projection) has been run with its Normalization prologue already (linear_normed)
void attention_mixer(Enc& e, DLayer& L, const uint16_t* hidden, uint16_t* out, const Seqs& q, PostNorm* post, bool first_done) {
    uzu_hip_model* m = e.m;
    // KV sharing: the packed projection yields queries only, key / value norms are dropped (mixer/attention/mod.rs:80-95,135-137)
    const uint32_t hd = L.d.head_dim, nq = L.d.num_heads, nkv = L.d.is_kv_sharing ? 0u : L.d.num_groups, total_heads = nq + 2 * nkv;
    const uint32_t rows = q.rows();
    if (L.d.has_gate) linear(e, L.gate, hidden, m->gate, rows);
    if (!first_done) linear(e, L.qkv, hidden, m->qkv, rows);
    // passes of more than one row: the head norms ride in the AttentionPrepare launch (attention_prepare_normed_kernel: bit-identical, two launches per layer fewer)
    // (UZU_HIP_TUNE=attn_fused=0: the separate QKVNorm / SigmoidGate launches; tests/test_gpu_prefill_switches.py)
    // (layers without head norms keep the flat element-wise AttentionPrepare: a wave per (row, head) is the norm's shape, not the copy's)
    const bool fuse_norms = rows > 1 && (L.qn.present || (L.kn.present && nkv) || (L.d.normalize_values && nkv)) && attention_fusions_enabled() && k::attention_prepare_normed_supported(hd);
    if (!fuse_norms) {
    if (L.qn.present)
        RUN("qkv_norm", 0, k::qkv_norm(e.s, m->qkv, UZU_BF16, L.qn.scales, rows, total_heads, hd, L.qn.eps, L.qn.offset, 0, nq, L.qn.full_layer));
    if (L.kn.present && nkv)
        RUN("qkv_norm", 0, k::qkv_norm(e.s, m->qkv, UZU_BF16, L.kn.scales, rows, total_heads, hd, L.kn.eps, L.kn.offset, nq, nkv, L.kn.full_layer));
    if (L.d.normalize_values && nkv) // AttentionConfig::value_norm_config (config/token_mixer/attention.rs:32-42): eps 1e-6, FullLayer, no scales
        RUN("qkv_norm", 0, k::qkv_norm(e.s, m->qkv, UZU_BF16, nullptr, rows, total_heads, hd, 1e-6f, 0.0f, nq + nkv, nkv, 1));
    }
    if (q.n == 0) {
        attention_core(e, L, q.count, 0, fuse_norms);
    } else {
        for (uint32_t i = 0; i < q.n; ++i) {
            bind_state(m, q.st[i]);
            attention_core(e, L, q.count, (size_t)i * q.count, fuse_norms);
        }
    }
    // (SigmoidGate: attention_core applies it per sequence -- inside the key-split merge of a prefill pass where there is one)
    linear(e, L.out, m->attn_out, out, rows, true, post);
}

// AttentionPrepare + attention of `batch` rows of the bound sequence, which start at row `row0` of qkv / queries / attn_out
void attention_core(Enc& e, DLayer& L, uint32_t batch, size_t row0, bool fuse_norms) {
    uzu_hip_model* m = e.m;
    const bool has_kv = !L.d.is_kv_sharing; // prepare_queries (mode.rs:234-259): no KV rows are written; the source layer wrote this pass's already
    const uint32_t hd = L.d.head_dim, nq = L.d.num_heads, nkv = L.d.num_groups, total_heads = nq + (has_kv ? 2 * nkv : 0);
    const uint16_t* qkv = m->qkv + row0 * total_heads * hd;
    uint16_t* queries = m->queries + row0 * nq * hd;
    uint16_t* attn_out = m->attn_out + row0 * nq * hd;
    const uint32_t rope_dim = L.d.use_rope ? L.rope_dim : 0;
    // Ring state (causal sliding window; state.rs:16-55, mode.rs:66-78): the new rows go to the suffix region behind the ring
    // (kv_token_offset = physical_prefix_length = window), the attention sees window + batch rows with ring parameters derived on the
    // device from the accepted-token count, and the rows enter the ring afterwards (encode_accept, state.rs:200-219).
    const uint32_t W = L.d.sliding_window_size;
    const uint32_t* trie = m->tree.active ? m->tree.d_trie : nullptr; // a speculated tree: RoPE positions = context + height, trie mask (mode.rs:178-192)
    if (fuse_norms) {
        const bool kv = has_kv && nkv;
        const k::PrepNorm qn{L.qn.scales, L.qn.eps, L.qn.offset, L.qn.full_layer, L.qn.present ? 1u : 0u};
        const k::PrepNorm kn{L.kn.scales, L.kn.eps, L.kn.offset, L.kn.full_layer, L.kn.present && kv ? 1u : 0u};
        const k::PrepNorm vn{nullptr, 1e-6f, 0.0f, 1u, L.d.normalize_values && kv ? 1u : 0u}; // AttentionConfig::value_norm_config (config/token_mixer/attention.rs:32-42)
        RUN("attention_prepare_normed", 0, k::attention_prepare_normed(e.s, qkv, queries, L.keys, L.values, L.rope_cos, L.rope_sin, qn, kn, vn, nq, nkv, hd, rope_dim, W, batch,
                                                                         has_kv ? 1u : 0u, m->d_ctx_len, W ? 1u : 0u, trie));
    } else
        RUN("attention_prepare", 0, k::attention_prepare(e.s, qkv, queries, L.keys, L.values, L.rope_cos, L.rope_sin, nq, nkv, hd, rope_dim, W, batch, has_kv ? 1u : 0u,
                                   m->d_ctx_len, W ? 1u : 0u, trie));
    k::AttentionParams a{};
    a.queries = queries, a.keys = L.keys, a.values = L.values;
    a.dt = UZU_BF16, a.head_dim = hd, a.gqa_factor = nq / nkv;
    a.sequence_length = batch; // + *d_ctx_len on the device
    a.k_head_stride = hd, a.k_seq_stride = nkv * hd, a.v_head_stride = hd, a.v_seq_stride = nkv * hd;
    a.scale = L.d.attention_scale != 0.0f ? L.d.attention_scale : 1.0f / sqrtf((float)hd);
    a.num_heads = nq, a.suffix_length = batch, a.is_causal = L.d.is_non_causal ? 0 : 1; // AttentionConfig::is_causal (mod.rs:166-198; mask.rs:3-61)
    a.dyn = m->d_ctx_len;
    a.trie = trie;
    if (W) a.ring_window = W, a.is_kv_cache_ring = 1, a.is_sliding_window = 1, a.sliding_window_size = W;
    if (L.sinks) a.sinks = L.sinks;
    const uint32_t physical_prefix = W ? W : m->context_length; // AttentionStateType::physical_prefix_length (state.rs:26-37)
    const size_t kv_bytes = (size_t)2 * (physical_prefix + batch) * nkv * hd * 2; // K and V rows read once
    const bool two_pass = (m->regime_override >= 0 && !W) ? m->regime_override == 1 : physical_prefix + batch > 1024; // core/mod.rs:89-92
    bool gated = false;
    if (k::attention_prefill_mfma_supported(a)) { // prefill chunk: flash-attention tiles on the matrix cores, any context length
        uint32_t gate_done = 0; // SigmoidGate inside the key-split merge launch when the pass has one (bit-identical to the separate launch)
        RUN("attention_prefill_mfma", kv_bytes, k::attention_prefill_mfma(e.s, a, attn_out, L.d.has_gate && attention_fusions_enabled() ? m->gate + row0 * nq * hd : nullptr, &gate_done));
        gated = gate_done != 0;
    } else if (two_pass) { // core/mod.rs:89-92
        RUN("attention_two_pass1", kv_bytes, k::attention_two_pass1(e.s, a, m->partials, m->sums, m->maxs));
        RUN("attention_two_pass2", 0, k::attention_two_pass2(e.s, m->partials, m->sums, m->maxs, attn_out, UZU_BF16, hd, nq, batch));
    } else {
        RUN("attention_single_pass", kv_bytes, k::attention_single_pass(e.s, a, attn_out));
    }
    if (L.d.has_gate && !gated) RUN("sigmoid_gate", 0, k::sigmoid_gate(e.s, m->gate + row0 * nq * hd, attn_out, UZU_BF16, batch * nq * hd));
    if (W && !trie && has_kv && L.last_reader == (uint32_t)(&L - m->layers.data()))
        RUN("kv_ring_insert", 0, k::kv_ring_insert(e.s, L.keys, L.values, UZU_BF16, m->d_ctx_len, batch, W, nkv * hd));
}

void delta_net_core(Enc& e, DLayer& L, uint32_t batch, size_t row0, uint32_t file_rows = 0);

// DeltaNet::encode_tree_verify (delta_net.rs:334-437): conv tree scan + tree prep (one launch), the tree-verify composite, norm-gate;
// the layer's DeltaNetSuffixStatus::Tree stays in m->tree.layers[layer] for uzu_hip_model_accept
void delta_net_tree_core(Enc& e, DLayer& L, uint32_t layer, uint32_t n) {
    uzu_hip_model* m = e.m;
    const uint32_t Hv = L.d.dn_num_heads, Hk = L.d.dn_num_groups, Dk = L.d.dn_head_dim, Dv = L.d.dn_value_head_dim;
    const uint32_t key_dim = Hk * Dk, value_dim = Hv * Dv, conv_dim = 2 * key_dim + value_dim;
    const uint32_t total_proj_dim = conv_dim + value_dim + 2 * Hv, ks = L.d.dn_kernel_size;
    uzu_hip_model::TreeLayer& T = m->tree.layers[layer];
    RUN("dn_tree_prep", 0, k::delta_net_tree_prep(e.s, m->in_proj, L.conv_w, L.conv_b, L.conv_state, m->tree.d_parents, nullptr, T.conv_states, L.a_log, L.dt_bias, m->tree.q,
                                                   T.k, T.v, T.beta, T.log_decay, n, ks, Hk, Hv, Dk, Dv, true, true));
    RUN("dn_tree_verify", (size_t)Hv * Dv * Dk * 4, k::delta_net_tree_verify(e.s, m->tree.q, T.k, T.v, m->tree.d_trie, T.log_decay, T.beta, L.ssm_state, m->delta_out, n, Hk, Hv, Dk, Dv));
    // the norm-gate reads z from the rows' pass-through section: ConvTreeScan copies those channels unchanged, so the in-proj rows serve
    RUN("delta_net_norm_gate", 0, k::delta_net_norm_gate(e.s, m->delta_out, m->in_proj, L.dn_norm, Hv, Dv, value_dim, conv_dim, total_proj_dim, L.d.dn_norm_epsilon, n));
}

void delta_net_mixer(Enc& e, DLayer& L, const uint16_t* hidden, uint16_t* out, const Seqs& q, PostNorm* post = nullptr, bool first_done = false) {
    uzu_hip_model* m = e.m;
    const uint32_t rows = q.rows();
    if (!first_done) linear(e, L.in_proj, hidden, m->in_proj, rows);
    m->rs_delta.rows = 0;
    // the norm-gate files the out-projection's row sums (one part per value head) while it writes the rows: no pre-pass launch in front of that GEMM
    const uint32_t Dv = L.d.dn_value_head_dim;
    const bool file = !m->tree.active && L.out_proj.k == L.d.dn_num_heads * Dv && rowsum_wanted(m, L.out_proj, rows, Dv, m->rs_delta);
    if (m->tree.active) { // !batch_dim.full_accept() (delta_net.rs:496-502)
        delta_net_tree_core(e, L, (uint32_t)(&L - m->layers.data()), q.count);
    } else if (q.n == 0) {
        delta_net_core(e, L, q.count, 0, file ? rows : 0);
    } else {
        for (uint32_t i = 0; i < q.n; ++i) {
            bind_state(m, q.st[i]);
            delta_net_core(e, L, q.count, (size_t)i * q.count, file ? rows : 0);
        }
    }
    if (file) m->rs_delta.rows = rows, m->rs_delta.k = L.out_proj.k, m->rs_delta.part = Dv;
    linear(e, L.out_proj, m->delta_out, out, rows, true, post);
}

// conv + delta rule + norm-gate over `batch` rows of the bound sequence, starting at row `row0` of in_proj / delta_out; file_rows != 0: the pass has that many
// rows in all and the norm-gate files their sums per value head (rs_delta)
void delta_net_core(Enc& e, DLayer& L, uint32_t batch, size_t row0, uint32_t file_rows) {
    uzu_hip_model* m = e.m;
    const uint32_t Hv = L.d.dn_num_heads, Hk = L.d.dn_num_groups, Dk = L.d.dn_head_dim, Dv = L.d.dn_value_head_dim;
    const uint32_t key_dim = Hk * Dk, value_dim = Hv * Dv, conv_dim = 2 * key_dim + value_dim;
    const uint32_t total_proj_dim = conv_dim + value_dim + 2 * Hv, ks = L.d.dn_kernel_size;
    uint16_t* in_proj = m->in_proj + row0 * total_proj_dim;
    uint16_t* delta_out = m->delta_out + row0 * value_dim;
    if (batch == 1) {
        RUN("delta_net_conv_update", 0, k::delta_net_conv_update(e.s, L.conv_w, L.conv_b, in_proj, L.conv_state, ks, conv_dim, ks - 1));
        RUN("delta_net_update", (size_t)2 * Hv * Dv * Dk * 4, k::delta_net_update(e.s, in_proj, L.a_log, L.dt_bias, L.dn_norm, L.ssm_state, delta_out, Hv, Hk, Dk, Dv, key_dim, value_dim,
                                  L.d.dn_norm_epsilon));
    } else {
        const bool chunked = m->dn_ws && k::delta_net_prefill_chunked_supported(Hv, Hk, Dk, Dv, batch);
        const bool prep_fused = chunked && k::delta_net_prefill_prep_fused_enabled(); // DeltaNetPrefillPrep inside the chunk preparation (bit-identical, one launch less)
        // ... and then the conv can run OUT OF PLACE into `padded` (its halo launch goes: the window comes from the rows, which stay raw; the chunk preparation and the
        // scans read the conv'd channels from there and the preparation writes the carried conv state)
        uint16_t* conv_rows = (uint16_t*)m->padded;
        const bool conv_oop = prep_fused && (size_t)batch * conv_dim * 2 <= (size_t)(m->chunk + 8) * total_proj_dim * 4 &&
                              k::delta_net_conv_out_of_place_supported(in_proj, L.conv_w, L.conv_b, conv_rows, ks, conv_dim, total_proj_dim);
        if (conv_oop) {
            RUN("delta_net_conv_oop", 0, k::delta_net_conv_out_of_place(e.s, in_proj, L.conv_w, L.conv_b, L.conv_state, conv_rows, batch, conv_dim, total_proj_dim));
        } else if (ks <= 8 && k::delta_net_conv_fused_workspace_floats(batch, ks, conv_dim) <= (size_t)(m->chunk + 8) * total_proj_dim) {
            RUN("delta_net_conv_fused", 0, k::delta_net_conv_fused(e.s, in_proj, L.conv_w, L.conv_b, L.conv_state, m->padded, batch, ks, conv_dim, total_proj_dim));
        } else {
            RUN("conv1d_pack", 0, k::conv1d_pack(e.s, L.conv_state, in_proj, m->padded, ks - 1, total_proj_dim, batch, conv_dim));
            RUN("delta_net_conv_scan", 0, k::delta_net_conv_scan(e.s, m->padded, L.conv_w, L.conv_b, in_proj, L.conv_state, batch, ks, total_proj_dim, ks - 1, conv_dim,
                                         total_proj_dim));
        }
        if (!prep_fused)
            RUN("delta_net_prefill_prep", 0, k::delta_net_prefill_prep(e.s, in_proj, L.a_log, L.dt_bias, m->qn, m->kn, m->beta, m->decay, Hv, Hk, Dk, key_dim, value_dim, batch));
        if (prep_fused)
            RUN("delta_net_prefill_chunked", 0, k::delta_net_prefill_chunked_fused(e.s, in_proj, L.a_log, L.dt_bias, m->qn, m->kn, L.ssm_state, delta_out, m->dn_ws, Hv, Hk, Dv, key_dim,
                                                                                value_dim, batch, conv_oop ? conv_rows : nullptr, L.conv_state));
        else if (chunked)
            RUN("delta_net_prefill_chunked", 0, k::delta_net_prefill_chunked(e.s, m->qn, m->kn, m->beta, m->decay, in_proj, L.ssm_state, delta_out, m->dn_ws, Hv, Hk, Dv,
                                                                          key_dim, value_dim, batch));
        else
            RUN("delta_net_prefill", 0, k::delta_net_prefill(e.s, m->qn, m->kn, m->beta, m->decay, in_proj, L.ssm_state, delta_out, Hv, Hk, Dk, Dv, key_dim, value_dim, batch));
        const uint32_t Mp = (file_rows + 3) & ~3u;
        const bool last = row0 + batch == file_rows; // the pass's last rows: this launch also zeroes the pad rows
        RUN("delta_net_norm_gate", 0, k::delta_net_norm_gate(e.s, delta_out, in_proj, L.dn_norm, Hv, Dv, value_dim, conv_dim, total_proj_dim, L.d.dn_norm_epsilon, batch,
                                                             file_rows && batch > 1 ? m->rs_delta.buf : nullptr, Mp, (uint32_t)row0, last ? Mp : 0));
    }
}

// MoeBlock::encode (mlp/moe/mod.rs:204-350): router top-k -> counts / offsets -> scatter (+ row map) -> gather -> experts pass A / pass B -> finalize.
// (topk_ids / tok2row need no 0xFF fill here: the router writes every id, the scatter marks every entry -- valid ones with their row, the others with -1.)
void moe_mlp(Enc& e, DLayer& L, const uint16_t* input, uint16_t* output, uint32_t rows) {
    uzu_hip_model* m = e.m;
    const uzu_moe_desc& M = L.d.moe;
    const uint32_t d = m->d.model_dim, E = M.num_routed_experts, K = M.num_active_experts, F = M.expert_hidden_dim, capacity = rows * K;
    auto& S = m->moe;
    RUN("moe_router_topk", (size_t)E * d * 2, k::moe_router_topk(e.s, input, L.moe.router_weights, L.moe.router_biases, S.topk_ids, S.topk_probs, rows, d, E, K, M.router_renorm));
    RUN("moe_counts_offsets", 0, k::moe_counts_offsets(e.s, S.topk_ids, S.offsets, S.sumk, nullptr, rows, E, K));
    RUN("moe_scatter_buckets", 0, k::moe_scatter_buckets(e.s, S.topk_ids, S.topk_probs, S.offsets, S.bucketed_ids, S.bucketed_probs, S.tok2row, S.row_expert_map, rows, E, K));
    RUN("moe_gather", 0, k::moe_gather(e.s, input, S.bucketed_ids, S.x_perm, S.sumk, d, rows, K));
    const k::MoeExpertParams q{d, F, M.gating_sel, M.gate_clip_min, M.gate_clip_max, M.up_clip_min, M.up_clip_max, M.silu_alpha};
    const size_t active = (size_t)(capacity < E ? capacity : E); // experts touched, at most
    RUN("moe_experts_pass_a", active * 2 * F * d * 2, k::moe_experts_pass_a(e.s, S.x_perm, S.row_expert_map, S.sumk, L.moe.w13, L.moe.up_biases, S.hidden, q, capacity));
    RUN("moe_experts_down", active * F * d * 2, k::moe_experts_down(e.s, S.hidden, S.row_expert_map, S.sumk, L.moe.w2, L.moe.down_biases, S.y_partial, d, F, capacity));
    RUN("moe_finalize", 0, k::moe_finalize(e.s, S.tok2row, S.topk_probs, S.y_partial, output, rows, d, K));
}

__global__ void commit_kernel(uint32_t* ctx_len, uint32_t* tokens, const uint32_t* out_token, uint32_t* sampled, uint32_t count, uint32_t has_token) {
    const uint32_t len = *ctx_len;
    if (has_token) {
        const uint32_t t = *out_token;
        sampled[len + count - 1] = t;
        tokens[0] = t;
    }
    *ctx_len = len + count;
}

// One forward pass.  `seqs` null: `count` tokens of the bound state, already in m->d_tokens.  `seqs` non-null: `count`
// tokens for each of the `nseq` states, token ids already in m->batch_tokens (row q * count + i); see struct Seqs.
// `sample` => output norm + readout + argmax on the last row (of every sequence).
uzu_status encode_forward(uzu_hip_model* m, hipStream_t s, uint32_t count, bool sample, uzu_hip_state** seqs, uint32_t nseq) {
    Enc e{m, s};
    e.prof = (std::vector<ProfEntry>*)m->prof_sink;
    m->launches = 0;
    const uint32_t d = m->d.model_dim;
    Seqs q;
    q.st = seqs, q.n = seqs ? nseq : 0, q.count = count;
    const uint32_t rows = q.rows();
    const uint32_t* token_ids = seqs ? m->batch_tokens : m->d_tokens;
    uint16_t* hidden = m->hidden;
    if (m->headless) { // the rows are in `hidden` already (a draft model embeds through its target: engine_drafter.hip)
    } else if (m->embedding.method == UZU_QUANT_NONE)
        RUN("full_precision_embedding_lookup", 0, k::full_precision_embedding_lookup(s, token_ids, m->embedding.w, hidden, UZU_BF16, rows, m->d.vocab_size, d, m->d.input_scale));
    else
        RUN("quantized_embedding_lookup", 0, k::quantized_embedding_lookup(s, token_ids, (const uint8_t*)m->embedding.w, m->embedding.scales, m->embedding.zp, m->embedding.biases,
                                            hidden, UZU_BF16, rows, m->d.vocab_size, d, m->d.input_scale, m->embedding.group, m->embedding.bits,
                                            m->embedding.method));
    // EmbeddingTable with output Hadamard factors (embedding_table.rs:34-125; quant_embedding.metal, use_hadamard): OutputRht of the rows
    if (m->embedding.out_signs)
        RUN("activation_transform", 0, k::activation_transform(s, nullptr, hidden, nullptr, nullptr, nullptr, m->embedding.out_signs, UZU_BF16, rows, d,
                                                                UZU_ACTIVATION_TRANSFORM_OUTPUT_RHT, 0, 0));
    // Decoder::encode (decoder.rs:149-154): the embedding norm, no shortcut
    if (m->embedding_norm.present) {
        norm(e, m->embedding_norm, hidden, m->normed, nullptr, 0, rows, d);
        RUN("tensor_copy", 0, k::tensor_copy(s, m->normed, hidden, UZU_BF16, rows * d));
    }
    // PerLayerEmbedding::encode (per_layer_embedding.rs:108-147): per_layer_inputs [rows][layers][ple_dim] = token table row * (ple_embed_scale *
    // input_scale) + projection_norm(model_projection(embedded rows)) [ScaleOutput(input_scale); epsilon / model_projection_scale^2 at load]
    if (m->d.has_ple) {
        const uint32_t total = m->d.num_layers * m->d.ple_dim;
        const DLinear& T = m->ple_token_embedding;
        const float fused_token_scale = m->d.ple_embed_scale * m->d.ple_input_scale; // per_layer_embedding.rs:103
        if (T.method == UZU_QUANT_NONE)
            RUN("full_precision_embedding_lookup", 0, k::full_precision_embedding_lookup(s, token_ids, T.w, m->ple_token, UZU_BF16, rows, m->d.ple_vocab_size, total, fused_token_scale));
        else
            RUN("quantized_embedding_lookup", 0, k::quantized_embedding_lookup(s, token_ids, (const uint8_t*)T.w, T.scales, T.zp, T.biases, m->ple_token, UZU_BF16, rows,
                                                m->d.ple_vocab_size, total, fused_token_scale, T.group, T.bits, T.method));
        linear(e, m->ple_model_projection, hidden, m->ple_projected, rows);
        norm(e, m->ple_projection_norm, m->ple_projected, m->ple_inputs, nullptr, 0, rows * m->d.num_layers, m->d.ple_dim);
        RUN("tensor_add_scale", 0, k::tensor_add_scale(s, m->ple_token, m->ple_inputs, m->ple_inputs, UZU_BF16, rows * total, rows * total, 1.0f));
    }
    // Prefill-sized passes: a row-parallel projection whose rows go straight into the next Normalization hands that normalisation to its GEMM
    // (PostNorm: the split-K reduction, the epilogue and the normalisation of a row are one launch).  `hidden_normed`: the pre-mixer normalisation of
    // the layer about to run has been done that way by the previous layer's down projection.
    bool hidden_normed = false;
    // the residual rows: in place in m->shortcut, except where a Normalization rides in a few-rows linear's prologue (linear_normed: 2 .. 16 rows, one
    // sequence) -- those read sc_cur and write the other buffer of the pair
    uint16_t* sc_cur = m->shortcut;
    auto sc_other = [&]() { return sc_cur == m->shortcut ? m->shortcut_b : m->shortcut; };
    const bool few_rows = !seqs && m->shortcut_b != nullptr;
    // Transformer::prefill_cache_layer_count (transformer.rs:186-199,239-243): a pass that produces no output (a prefill chunk that is not the
    // prompt's last) only has to fill the caches -- it stops behind the last layer that owns a state; trailing KV-sharing layers write nothing
    uint32_t layer_count = m->d.num_layers;
    if (!sample && !m->tree.active && !m->taps && m->feature_layers.empty())
        while (layer_count > 1 && m->layers[layer_count - 1].d.mixer_kind == UZU_MIXER_ATTENTION && m->layers[layer_count - 1].d.is_kv_sharing) --layer_count;
    for (uint32_t l = 0; l < layer_count; ++l) {
        DLayer& L = m->layers[l];
        const uint16_t* h = hidden;
        bool first_done = false;
        if (L.pre_mixer.present) {
            if (!hidden_normed) {
                const bool att = L.d.mixer_kind == UZU_MIXER_ATTENTION;
                // (a gated attention layer's gate projection reads the same normalised rows: the prologue files them in m->normed)
                if (few_rows && linear_normed(e, L.pre_mixer, l > 0 ? 2 : 1, att ? L.qkv : L.in_proj, hidden, sc_cur, sc_other(), att && L.d.has_gate ? m->normed : nullptr,
                                              att ? m->qkv : m->in_proj, rows, false, 0)) {
                    first_done = true, sc_cur = sc_other();
                } else {
                    norm(e, L.pre_mixer, hidden, m->normed, sc_cur, l > 0 ? 2 : 1, rows, d, att ? &L.qkv : &L.in_proj);
                }
            }
            h = m->normed;
        } else {
            RUN("tensor_copy", 0, k::tensor_copy(s, hidden, sc_cur, UZU_BF16, rows * d));
        }
        hidden_normed = false;
        PostNorm mlp_norm; // the pre-MLP normalisation, offered to the mixer's out projection when nothing sits between them
        const bool offer_mlp = !L.post_mixer.present && rows >= 128 && !L.pre_mlp.scalar_mode;
        if (offer_mlp) mlp_norm.p = norm_params(e, L.pre_mlp, m->mixed, m->normed, sc_cur, 2, rows, d, L.d.mlp_kind == UZU_MLP_MOE ? nullptr : &L.up);
        if (L.d.mixer_kind == UZU_MIXER_ATTENTION)
            attention_mixer(e, L, h, m->mixed, q, offer_mlp ? &mlp_norm : nullptr, first_done);
        else
            delta_net_mixer(e, L, h, m->mixed, q, offer_mlp ? &mlp_norm : nullptr, first_done);
        const uint16_t* mixed = m->mixed;
        if (L.post_mixer.present) {
            norm(e, L.post_mixer, m->mixed, m->normed, nullptr, 0, rows, d);
            RUN("tensor_copy", 0, k::tensor_copy(s, m->normed, m->mixed, UZU_BF16, rows * d));
        }
        bool mlp_done = false; // pre-MLP Normalization + up | gate + GatedActMul as ONE few-rows launch
        if (offer_mlp) {
            norm_issued(m, mlp_norm.p);
            if (!mlp_norm.done) RUN("normalization", 0, k::normalization(e.s, mlp_norm.p));
        } else if (few_rows && L.d.mlp_kind != UZU_MLP_MOE && !L.post_mixer.present && linear_normed(e, L.pre_mlp, 2, L.up, mixed, sc_cur, sc_other(), nullptr, m->gated, rows, true, L.d.activation)) {
            mlp_done = true, sc_cur = sc_other();
        } else {
            norm(e, L.pre_mlp, mixed, m->normed, sc_cur, 2, rows, d, L.d.mlp_kind == UZU_MLP_MOE ? nullptr : &L.up);
        }
        const bool is_moe = L.d.mlp_kind == UZU_MLP_MOE;
        if (is_moe) {
            moe_mlp(e, L, m->normed, hidden, rows);
        } else if (!mlp_done && !linear_gated(e, L.up, m->normed, m->gated, rows, L.d.activation, &L.down)) { // prefill-sized rows: GatedActMul in the GEMM's epilogue
            linear(e, L.up, m->normed, m->up, rows);
            RUN("gated_act_mul", 0, k::gated_act_mul(s, m->up, nullptr, m->gated, UZU_BF16, L.d.hidden_dim, rows, 0, 0, L.d.activation, 1));
        }
        // the next layer's pre-mixer normalisation rides on this layer's down projection (not past the last layer: the output norm takes one row)
        PostNorm next_norm;
        const bool offer_next = !is_moe && rows >= 128 && !L.post_mlp.present && !L.d.has_ple && l + 1 < layer_count && m->layers[l + 1].pre_mixer.present;
        if (offer_next) {
            const DLayer& Nx = m->layers[l + 1];
            next_norm.p = norm_params(e, Nx.pre_mixer, hidden, m->normed, sc_cur, 2, rows, d, Nx.d.mixer_kind == UZU_MIXER_ATTENTION ? &Nx.qkv : &Nx.in_proj);
        }
        if (!is_moe) linear(e, L.down, m->gated, hidden, rows, true, offer_next ? &next_norm : nullptr);
        if (offer_next && next_norm.done) {
            norm_issued(m, next_norm.p);
            hidden_normed = true;
        }
        if (L.post_mlp.present) {
            norm(e, L.post_mlp, hidden, m->normed, nullptr, 0, rows, d);
            RUN("tensor_copy", 0, k::tensor_copy(s, m->normed, hidden, UZU_BF16, rows * d));
        }
        if (L.d.has_ple) {
            // PerLayerEmbeddingProjection::encode (per_layer_embedding.rs:217-270): shortcut += hidden; gate(shortcut) -> act(gate) * this layer's slice
            // of per_layer_inputs -> projection -> norm; shortcut = (shortcut + normed) * post_layer_scalar; hidden = 0 (transformer_layer.rs:231)
            const uint32_t length = rows * d, pd = L.d.ple_dim;
            RUN("tensor_add_bias", 0, k::tensor_add_bias(s, nullptr, hidden, sc_cur, UZU_BF16, UZU_BF16, length, length));
            linear(e, L.ple_gate, sc_cur, m->ple_gate_out, rows);
            RUN("gated_act_mul", 0, k::gated_act_mul(s, m->ple_gate_out, m->ple_inputs, m->ple_activated, UZU_BF16, pd, rows, l * pd, m->d.num_layers * pd, L.d.ple_activation, 0));
            linear(e, L.ple_projection, m->ple_activated, m->mixed, rows);
            norm(e, L.ple_norm, m->mixed, m->normed, nullptr, 0, rows, d);
            RUN("tensor_add_scale", 0, k::tensor_add_scale(s, nullptr, m->normed, sc_cur, UZU_BF16, length, length, L.d.has_post_layer_scalar ? L.d.post_layer_scalar : 1.0f));
            HIPCHK(hipMemsetAsync(hidden, 0, (size_t)length * 2, s));
        }
        // a ring whose rows later layers of this pass still had to read takes the pass's suffix rows now (DLayer::last_reader; at the latest
        // behind the last layer this pass runs)
        if (!m->tree.active)
            for (uint32_t o = 0; o <= l; ++o) {
                DLayer& Lo = m->layers[o];
                if (Lo.d.mixer_kind != UZU_MIXER_ATTENTION || Lo.d.is_kv_sharing || !Lo.d.sliding_window_size || Lo.last_reader == o) continue; // (== o: inserted by its own launch sequence)
                if (!(Lo.last_reader == l || (l + 1 == layer_count && Lo.last_reader > l))) continue;
                for (uint32_t i = 0; i < (q.n ? q.n : 1u); ++i) {
                    if (q.n) bind_state(m, q.st[i]);
                    RUN("kv_ring_insert", 0, k::kv_ring_insert(s, Lo.keys, Lo.values, UZU_BF16, m->d_ctx_len, count, Lo.d.sliding_window_size, Lo.d.num_groups * Lo.d.head_dim));
                }
            }
        // (a PLE layer has folded its output into the shortcut and zeroed `hidden`: its tap is the residual row, what capture_residual would file)
        if (m->taps && !seqs) RUN("tensor_copy", 0, k::tensor_copy(s, L.d.has_ple ? sc_cur : hidden, m->taps + ((size_t)l * m->chunk) * d, UZU_BF16, count * d));
        // Transformer::capture_residual (transformer.rs:160-171,285-293): TensorAddScale(shortcut, hidden, 1.0) of the rows, per tapped layer
        if (!seqs)
            for (size_t fi = 0; fi < m->feature_layers.size(); ++fi)
                if (m->feature_layers[fi] == l)
                    RUN("tensor_add_scale", 0, k::tensor_add_scale(s, sc_cur, hidden, m->features + (fi * m->chunk) * d, UZU_BF16, count * d, count * d, 1.0f));
    }
    m->tap_rows = count;
    m->feature_rows = seqs ? 0 : count;
    m->last_shortcut = sc_cur;
    if (m->headless) { // a draft pass: no output norm / read-out / commit of its own (engine_drafter.hip does what follows)
        if (e.st != UZU_OK) return e.st;
        const hipError_t herr = hipGetLastError();
        if (herr != hipSuccess) {
            set_error("engine: draft pass launch failed: %s", hipGetErrorString(herr));
            return UZU_ERR_HIP;
        }
        return UZU_OK;
    }
    if (m->tree.active) {
        // a tree pass: output norm, read-out and greedy sampling of EVERY node (output_range 0..size, stream.rs:618-628), no commit
        norm(e, m->output_norm, hidden, m->tree.normed, sc_cur, 2, count, d);
        m->final_hidden = m->tree.normed, m->final_hidden_rows = count;
        DLinear ro = m->d.tied_embeddings ? m->embedding : m->output_embedding;
        ro.in_signs = m->d.tied_embeddings ? m->embedding.out_signs : m->output_embedding.in_signs;
        ro.out_signs = nullptr;
        linear(e, ro, m->tree.normed, m->tree.logits, count);
        if (m->d.logit_scale != 1.0f || m->d.logit_soft_cap != 0.0f)
            RUN("logit_transform", 0, k::logit_transform(s, m->tree.logits, UZU_BF16, ro.n * count, m->d.logit_scale, m->d.logit_soft_cap, m->d.logit_soft_cap != 0.0f));
        if (m->sampling.on) {
            // every node draws with ITS seed: the trie's token_seeds when the caller passed them (stream.rs:694), else the seed of its position,
            // PRng::derive(context + height) (dflash_tfm.rs:267,304)
            if (!m->tree.host_seeds) RUN("derive_tree_seeds", 0, k::derive_tree_seeds(s, m->sampling.seed, m->d_ctx_len, m->tree.d_trie, count, m->tree.d_seeds));
            k::UnifiedSamplingParams sp = m->sampling.p;
            sp.logits = m->tree.logits, sp.dt = UZU_BF16, sp.output = m->tree.d_sampled, sp.seeds = m->tree.d_seeds, sp.vocab_size = ro.n, sp.batch_size = count;
            if (m->tp) { // the whole rows on every rank: same seeds, same distribution => the same token everywhere
                RUN("tp_gather_logits", (size_t)m->d.vocab_size * 4 * count, tp::gather_logits(m->tp, s, m->tree.logits, ro.n, m->vocab_offset, m->d.vocab_size, count, m->tp_gather_f32, m->tp_gather_bf16));
                sp.logits = m->tp_gather_bf16, sp.vocab_size = m->d.vocab_size;
            }
            RUN("unified_sampling", (size_t)sp.vocab_size * 2 * count, k::unified_sampling(s, sp, m->tree.sampling_scratch));
        } else {
            RUN("argmax", (size_t)ro.n * 2 * count, k::argmax(s, m->tree.logits, UZU_BF16, m->tree.d_sampled, ro.n, count, m->tree.argmax_scratch));
            if (m->tp) { // vocab-sharded read-out: one (logit, global index) key per node, reduced with max
                RUN("tp_keys", 0, tp::keys_from_tokens(s, m->tree.logits, ro.n, m->tree.d_sampled, m->vocab_offset, m->tp_key, count));
                RUN("all_reduce", 8 * count, tp::all_reduce_max_u64(m->tp, s, m->tp_key, count));
                RUN("tp_tokens", 0, tp::tokens_from_keys(s, m->tp_key, m->tree.d_sampled, count));
            }
        }
        if (e.st != UZU_OK) return e.st;
        hipError_t terr = hipGetLastError();
        if (terr != hipSuccess) {
            set_error("engine: tree pass launch failed: %s", hipGetErrorString(terr));
            return UZU_ERR_HIP;
        }
        return UZU_OK;
    }
    for (uint32_t i = 0; i < (seqs ? nseq : 1u); ++i) { // per sequence: sample from its last row, then commit
        if (seqs) bind_state(m, seqs[i]);
        if (sample) {
            const size_t last = ((size_t)i * count + count - 1) * d;
            norm(e, m->output_norm, hidden + last, m->last_normed, sc_cur + last, 2, 1, d);
            m->final_hidden = m->last_normed, m->final_hidden_rows = seqs ? 0 : 1;
            // Embedding::encode_readout (embedding.rs:374-456): the read-out's private InputRht -- a tied table's output signs
            // (embedding.rs:167-173) or the untied output embedding's input signs (embedding.rs:255-274) -- then the plain matmul
            DLinear ro = m->d.tied_embeddings ? m->embedding : m->output_embedding;
            ro.in_signs = m->d.tied_embeddings ? m->embedding.out_signs : m->output_embedding.in_signs;
            ro.out_signs = nullptr;
            linear(e, ro, m->last_normed, m->logits, 1);
            if (m->d.logit_scale != 1.0f || m->d.logit_soft_cap != 0.0f)
                RUN("logit_transform", 0, k::logit_transform(s, m->logits, UZU_BF16, ro.n, m->d.logit_scale, m->d.logit_soft_cap, m->d.logit_soft_cap != 0.0f));
            if (m->sampling.on) { // stream.rs:248-258: seed = PRng::derive(position of the sampled row), then UnifiedSampling
                RUN("derive_seed", 0, k::derive_seed(s, m->sampling.seed, m->d_ctx_len, count - 1, m->d_seed));
                k::UnifiedSamplingParams sp = m->sampling.p;
                sp.logits = m->logits, sp.dt = UZU_BF16, sp.output = m->d_out_token, sp.seeds = m->d_seed, sp.vocab_size = ro.n, sp.batch_size = 1;
                if (m->tp) { // the whole row on every rank (tp::gather_logits): same seed, same distribution => the same token everywhere
                    RUN("tp_gather_logits", (size_t)m->d.vocab_size * 4, tp::gather_logits(m->tp, s, m->logits, ro.n, m->vocab_offset, m->d.vocab_size, 1, m->tp_gather_f32, m->tp_gather_bf16));
                    sp.logits = m->tp_gather_bf16, sp.vocab_size = m->d.vocab_size;
                }
                RUN("unified_sampling", (size_t)sp.vocab_size * 2, k::unified_sampling(s, sp, m->sampling_scratch));
            } else {
                RUN("argmax", (size_t)ro.n * 2, k::argmax(s, m->logits, UZU_BF16, m->d_out_token, ro.n, 1, m->argmax_scratch));
            }
            if (m->tp && !m->sampling.on) { // vocab-sharded read-out: every rank contributes (logit, global index) of its local winner
                RUN("tp_key", 0, tp::key_from_token(s, m->logits, m->d_out_token, m->vocab_offset, m->tp_key));
                RUN("all_reduce", 8, tp::all_reduce_max_u64(m->tp, s, m->tp_key, 1));
                RUN("tp_token", 0, tp::token_from_key(s, m->tp_key, m->d_out_token));
            }
        }
        // an earlier launch failed: leave the device-side context length / next token untouched so that they keep agreeing
        // with the host mirror (m->context_length is only advanced by the callers on success)
        if (e.st != UZU_OK) return e.st;
        hipLaunchKernelGGL(commit_kernel, dim3(1), dim3(1), 0, s, m->d_ctx_len, m->d_tokens, m->d_out_token, m->d_sampled, count, sample ? 1u : 0u);
        ++m->launches;
    }
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        set_error("engine: forward launch failed: %s", hipGetErrorString(err));
        return UZU_ERR_HIP;
    }
    return UZU_OK;
}

// ---------------------------------------------------------------------------------- fused decode step
k::DecGemvParams dec_gemv_base(const DLinear& L, const uint16_t* x, uint16_t* out) {
    k::DecGemvParams p{};
    p.w[0] = (const uint8_t*)L.w, p.scales[0] = (const uint16_t*)L.scales, p.biases[0] = (const uint16_t*)L.biases, p.zp[0] = L.zp;
    p.out_bias[0] = L.out_signs ? nullptr : (const uint16_t*)L.out_biases, p.out[0] = out, p.n[0] = L.n; // bias_after_rht: with the OutputRht, later
    p.k = L.k, p.bits = L.bits, p.group_size = L.group;
    p.b_kind = L.method == UZU_QUANT_SCALE_BIAS ? UZU_MATMUL_B_SCALE_BIAS
             : L.method == UZU_QUANT_SCALE_ZERO_POINT ? UZU_MATMUL_B_SCALE_ZERO_POINT : UZU_MATMUL_B_SCALE_SYMMETRIC;
    p.x = x;
    return p;
}
void dec_add_second(k::DecGemvParams& p, const DLinear& L, uint16_t* out) {
    p.w[1] = (const uint8_t*)L.w, p.scales[1] = (const uint16_t*)L.scales, p.biases[1] = (const uint16_t*)L.biases, p.zp[1] = L.zp;
    p.out_bias[1] = L.out_signs ? nullptr : (const uint16_t*)L.out_biases, p.out[1] = out, p.n[1] = L.n;
}
// mode: 1 copy, 2 add
void dec_add_norm(k::DecGemvParams& p, const DNorm& N, int mode, const uint16_t* sc_in, uint16_t* sc_out) {
    p.norm_scales = N.scales;
    p.norm_plain = N.scales == nullptr;
    p.norm_eps = N.eps, p.norm_offset = N.offset, p.norm_full_layer = N.full_layer;
    p.residual_add = mode == 2;
    p.shortcut_in = mode == 2 ? sc_in : nullptr;
    p.shortcut_out = sc_out;
    // Lab builds only (WRONG numerics, timing only): UZU_LAB_NORM_NOADD=1 drops the shortcut row load, the add and workgroup 0's residual store from every normed
    // decode GEMV -- everything a producer-side residual + sum-of-squares split could take out of the consumer's prologue (an upper bound)
    static const bool no_add = [] {
        const char* v = lab_env("UZU_LAB_NORM_NOADD");
        return v && atoi(v) != 0;
    }();
    if (no_add) p.residual_add = 0, p.shortcut_in = nullptr, p.shortcut_out = nullptr;
}
size_t dec_gemv_bytes(const k::DecGemvParams& p) {
    size_t b = 0;
    for (int i = 0; i < 2; ++i) {
        if (!p.n[i]) continue;
        const size_t groups = (p.k + p.group_size - 1) / p.group_size;
        b += (size_t)p.n[i] * p.k * p.bits / 8 + (size_t)p.n[i] * groups * 2;
        if (p.b_kind == UZU_MATMUL_B_SCALE_BIAS) b += (size_t)p.n[i] * groups * 2;
        if (p.b_kind == UZU_MATMUL_B_SCALE_ZERO_POINT) b += (size_t)p.n[i] * (p.bits == 4 ? (groups + 1) / 2 : groups);
        b += (size_t)p.n[i] * 2;
    }
    return b + (size_t)p.k * 2;
}
void dec_gemv(Enc& e, const k::DecGemvParams& p, const char* name, uint32_t* grid_out = nullptr) {
    e.begin();
    e.run(k::gemv_dec(e.s, p, e.m->ctx->num_cus, grid_out), name, dec_gemv_bytes(p));
}

// out-proj / down-proj of the fused decode step; under tensor parallelism: f32 partials -> all-reduce -> bf16
void dec_gemv_row_parallel(Enc& e, k::DecGemvParams p, const char* name) {
    uzu_hip_model* m = e.m;
    if (!m->tp) return dec_gemv(e, p, name);
    uint16_t* out = p.out[0];
    p.out_f32 = m->tp_buf;
    dec_gemv(e, p, name);
    RUN("all_reduce", (size_t)p.n[0] * 4, tp::all_reduce_sum_f32(m->tp, e.s, m->tp_buf, p.n[0], out));
}

bool linear_rht(const DLinear& L) { return L.in_signs || L.out_signs; }
bool linear_fusable(const DLinear& L) {
    if (!L.w || L.lora_rank) return false; // QLoRA linears run as their wrapper composes them (unfused decode)
    // RHT linears (round 4): InputRht in the Normalization prologue of the GEMV or as a launch of its own in front of a plain-row GEMV, OutputRht
    // (+ bias) in the prologue of the next normalised GEMV or as a launch of its own (encode_decode_fused); sign vectors of +-1 only
    if (linear_rht(L) && !(L.in_bits && L.out_bits && L.n % 32 == 0)) return false;
    if (L.method == UZU_QUANT_NONE || (L.bits != 4 && L.bits != 8)) return false;
    return L.k % 32 == 0 && L.group % 32 == 0 && (L.group & (L.group - 1)) == 0 && L.k <= 32768;
}
bool norm_fusable(const DNorm& N) { return N.present && !N.subtract_mean && !N.biases && !N.scalar_mode; }
// the fused Normalization prologue needs model_dim % 1024 == 0 and <= 8192 (k_decode.hip)
bool dim_fusable(uint32_t d) { return d % 1024 == 0 && d <= 8192; }

bool model_fusable(const uzu_hip_model* m) {
    if (m->headless || !m->feature_layers.empty()) return false; // draft cores; a target whose passes file hidden features: the one-kernel-per-reference-kernel pass
    if (m->gemma_options) return false; // post-layer scalars, embedding norm, KV sharing, value normalisation, per-layer embeddings: the one-kernel-per-reference-kernel pass
    if (m->d.logit_scale != 1.0f || m->d.logit_soft_cap != 0.0f) return false;
    if (!dim_fusable(m->d.model_dim)) return false;
    if (!norm_fusable(m->output_norm)) return false;
    if (!linear_fusable(m->d.tied_embeddings ? m->embedding : m->output_embedding)) return false;
    if (m->embedding.in_signs || m->embedding.out_signs) return false; // RHT embedding rows: the commit kernel's lookup has no transform
    for (const DLayer& L : m->layers) {
        if (!norm_fusable(L.pre_mixer) || !norm_fusable(L.pre_mlp) || L.post_mixer.present || L.post_mlp.present) return false;
        if (L.d.mlp_kind == UZU_MLP_MOE) return false; // MoE layers: the one-kernel-per-reference-kernel pass
        if (!linear_fusable(L.up) || !linear_fusable(L.down)) return false;
        if (L.d.mixer_kind == UZU_MIXER_ATTENTION) {
            if (!linear_fusable(L.qkv) || !linear_fusable(L.out)) return false;
            if (L.d.has_gate && (!linear_fusable(L.gate) || L.gate.bits != L.qkv.bits || L.gate.group != L.qkv.group || L.gate.method != L.qkv.method)) return false;
            // (the two matrices of the fused launch share one prologue: with different input transforms the gate gets a launch of its own)
            if (!(L.d.head_dim == 64 || L.d.head_dim == 128 || L.d.head_dim == 256)) return false;
            if (L.d.sliding_window_size || L.d.has_sinks || L.d.is_non_causal) return false; // ring KV state / sinks / block attention: the one-kernel-per-reference-kernel path (attn_dec has neither)
            if ((L.qn.present && (L.qn.subtract_mean || L.qn.biases)) || (L.kn.present && (L.kn.subtract_mean || L.kn.biases))) return false;
        } else {
            if (!linear_fusable(L.in_proj) || !linear_fusable(L.out_proj)) return false;
            if (linear_rht(L.in_proj) && m->tp) return false; // (the stand-alone conv / update kernels of the RHT route are not sharded here)
            if (L.d.dn_head_dim != 128 || L.d.dn_value_head_dim > 512 || L.d.dn_kernel_size != 4) return false; // conv epilogue of the in-proj GEMV
            {   // norm-gate prologue of the out-proj GEMV (k_decode.hip): chunks of 8 outputs, <= 4 chunks per thread
                const uint32_t dv = L.d.dn_value_head_dim, kk = L.d.dn_num_heads * dv, nchunks = kk / 8, per = nchunks > 256 ? nchunks / 256 : 1;
                if (dv < 8 || (dv & (dv - 1)) || kk > 8192 || (nchunks > 256 && (nchunks % 256 || per > 4)) || (dv / 8) % per) return false;
            }
        }
    }
    return true;
}

// The commit kernel of a single-GPU fused step also writes the sampled token's embedding row (the next step's input).
bool commit_embeds(const uzu_hip_model* m) { return m->tp == nullptr; }

void encode_embed_row0(Enc& e) {
    uzu_hip_model* m = e.m;
    const uint32_t d = m->d.model_dim;
    if (m->embedding.method == UZU_QUANT_NONE)
        RUN("full_precision_embedding_lookup", 0, k::full_precision_embedding_lookup(e.s, m->d_tokens, m->embedding.w, m->hidden, UZU_BF16, 1, m->d.vocab_size, d, m->d.input_scale));
    else
        RUN("quantized_embedding_lookup", 0, k::quantized_embedding_lookup(e.s, m->d_tokens, (const uint8_t*)m->embedding.w, m->embedding.scales, m->embedding.zp, m->embedding.biases,
                                            m->hidden, UZU_BF16, 1, m->d.vocab_size, d, m->d.input_scale, m->embedding.group, m->embedding.bits, m->embedding.method));
}

// One decode step (count == 1, sampling) with the fused kernels of k_decode.hip.  `with_embed`: look the input token's
// embedding row up first (the first step after a prefill / set_next_token; later steps find it written by the commit).
uzu_status encode_decode_fused(uzu_hip_model* m, hipStream_t s, bool with_embed) {
    Enc e{m, s};
    e.prof = (std::vector<ProfEntry>*)m->prof_sink;
    m->launches = 0;
    const uint32_t d = m->d.model_dim;
    uint16_t* hidden = m->hidden;
    if (with_embed) encode_embed_row0(e);
    uint16_t* sc[2] = {m->shortcut, m->shortcut_b};
    int cur = 1; // the first norm (copy mode) writes sc[0]
    // RHT linears (RHTLinearWrapper, linear/rht_wrapper.rs:215-298) inside the fused step.  The raw output row of an out-projection / down
    // projection with Hadamard factors stays `pending`: the next GEMV with a Normalization prologue applies its OutputRht + bias to the row it
    // loads anyway (PRO == 3 instance of gemv_dec_kernel); a consumer that cannot (act-mul / conv epilogue) gets it flushed by the reference's
    // own two kernels first.  Rounding points are the unfused path's, so the two stay bit-identical.
    const DLinear* pending = nullptr;
    uint16_t* pending_row = nullptr;
    auto out_transform = [&](const DLinear& L, uint16_t* row) {
        if (!L.out_signs) return;
        RUN("activation_transform", 0, k::activation_transform(s, nullptr, row, nullptr, nullptr, nullptr, L.out_signs, UZU_BF16, 1, L.n, UZU_ACTIVATION_TRANSFORM_OUTPUT_RHT, 0, 0));
        if (L.out_biases) RUN("tensor_add_bias", 0, k::tensor_add_bias(s, row, L.out_biases, row, UZU_BF16, UZU_BF16, L.n, L.n));
    };
    auto flush_pending = [&]() {
        if (pending) out_transform(*pending, pending_row);
        pending = nullptr;
    };
    auto in_transform = [&](const DLinear& L, const uint16_t* row) -> const uint16_t* {
        if (!L.in_signs) return row;
        RUN("activation_transform", 0, k::activation_transform(s, row, m->rht_scratch, nullptr, nullptr, nullptr, L.in_signs, UZU_BF16, 1, L.k, UZU_ACTIVATION_TRANSFORM_INPUT_RHT, 0, 0));
        return m->rht_scratch;
    };
    // `own`: the linear(s) behind this prologue; `epilogue`: the launch has an act-mul / conv epilogue (no Hadamard instance exists for those)
    auto next_norm = [&](k::DecGemvParams& p, const DNorm& N, int mode, const DLinear* own = nullptr, bool epilogue = false) {
        if (pending && epilogue) flush_pending();
        dec_add_norm(p, N, mode, sc[cur], sc[cur ^ 1]);
        cur ^= 1;
        if (pending) p.x_rht_bits = pending->out_bits, p.x_rht_bias = (const uint16_t*)pending->out_biases, pending = nullptr;
        if (own && own->in_bits) p.in_rht_bits = own->in_bits;
    };
    for (uint32_t l = 0; l < m->d.num_layers; ++l) {
        DLayer& L = m->layers[l];
        if (L.d.mixer_kind == UZU_MIXER_ATTENTION) {
            const uint32_t hd = L.d.head_dim, nq = L.d.num_heads, nkv = L.d.num_groups;
            k::DecGemvParams p = dec_gemv_base(L.qkv, hidden, m->qkv);
            const bool own_gate = L.d.has_gate && (linear_rht(L.qkv) != linear_rht(L.gate) || L.qkv.in_words != L.gate.in_words);
            if (L.d.has_gate && !own_gate) dec_add_second(p, L.gate, m->gate);
            next_norm(p, L.pre_mixer, l > 0 ? 2 : 1, &L.qkv);
            dec_gemv(e, p, own_gate ? "gemv_dec[norm+qkv]" : "gemv_dec[norm+qkv+gate]");
            if (own_gate) { // its own InputRht: the same Normalization again, of the residual row the launch above has just written (copy mode, nothing stored)
                k::DecGemvParams g = dec_gemv_base(L.gate, sc[cur], m->gate);
                dec_add_norm(g, L.pre_mixer, 1, nullptr, nullptr);
                g.in_rht_bits = L.gate.in_bits;
                dec_gemv(e, g, "gemv_dec[norm+gate]");
            }
            if (L.qkv.out_bits && L.d.has_gate && L.gate.out_bits) // both rows in one launch
                RUN("rht_out_rows", 0, k::rht_out_rows(s, m->qkv, L.qkv.out_bits, (const uint16_t*)L.qkv.out_biases, L.qkv.n, m->gate, L.gate.out_bits,
                                                        (const uint16_t*)L.gate.out_biases, L.gate.n, nullptr, nullptr, nullptr, 0, 0));
            else {
                out_transform(L.qkv, m->qkv);
                if (L.d.has_gate) out_transform(L.gate, m->gate);
            }
            k::AttnDecParams a{};
            a.qkv = m->qkv, a.keys = L.keys, a.values = L.values, a.cosines = L.rope_cos, a.sines = L.rope_sin, a.ctx_len = m->d_ctx_len;
            a.q_norm = {L.qn.present, L.qn.full_layer, L.qn.eps, L.qn.offset, L.qn.scales};
            a.k_norm = {L.kn.present, L.kn.full_layer, L.kn.eps, L.kn.offset, L.kn.scales};
            a.num_heads = nq, a.gqa_factor = nq / nkv, a.head_dim = hd, a.rope_dim = L.d.use_rope ? L.rope_dim : 0;
            a.scale = L.d.attention_scale != 0.0f ? L.d.attention_scale : 1.0f / sqrtf((float)hd);
            a.partials = m->dec_partials, a.sums = m->dec_sums, a.maxs = m->dec_maxs, a.cache_rows = m->max_positions;
            const size_t kv_bytes = (size_t)2 * (m->context_length + 1) * nkv * hd * 2;
            if (k::attn_dec_fused_supported(nq, nq / nkv, hd, m->dec_splits, m->ctx->num_cus)) {
                // "a fused SDPA decode kernel": pass 2 + SigmoidGate inside the launch (every workgroup merges its slice of its KV-head group's rows)
                a.tickets = m->dec_tickets, a.gate = L.d.has_gate ? m->gate : nullptr, a.out = m->attn_out;
                RUN("attn_dec", kv_bytes, k::attn_dec(s, a, m->dec_splits));
            } else {
                RUN("attn_dec", kv_bytes, k::attn_dec(s, a, m->dec_splits));
                RUN("attn_merge", 0, k::attn_merge(s, m->dec_partials, m->dec_sums, m->dec_maxs, L.d.has_gate ? m->gate : nullptr, m->attn_out, nq, hd, m->dec_splits));
            }
            if (L.out.in_bits && k::gemv_dec_plain_in_rht_supported(L.out.k, L.out.bits)) { // the out-projection's InputRht in the GEMV's registers (round 5)
                k::DecGemvParams op = dec_gemv_base(L.out, m->attn_out, m->mixed);
                op.in_rht_bits = L.out.in_bits;
                dec_gemv_row_parallel(e, op, "gemv_dec[out_proj]");
            } else {
                dec_gemv_row_parallel(e, dec_gemv_base(L.out, in_transform(L.out, m->attn_out), m->mixed), "gemv_dec[out_proj]");
            }
            if (L.out.out_signs) pending = &L.out, pending_row = m->mixed;
        } else {
            const uint32_t Hv = L.d.dn_num_heads, Hk = L.d.dn_num_groups, Dk = L.d.dn_head_dim, Dv = L.d.dn_value_head_dim;
            k::DecGemvParams p = dec_gemv_base(L.in_proj, hidden, m->in_proj);
            const uint32_t conv_dim = 2 * Hk * Dk + Hv * Dv;
            if (linear_rht(L.in_proj)) {
                // the conv needs the OutputRht of the row it convolves
                next_norm(p, L.pre_mixer, l > 0 ? 2 : 1, &L.in_proj);
                k::DecGemvParams ps = p;
                ps.ep_out_bits = L.in_proj.out_bits, ps.ep_bias = (const uint16_t*)L.in_proj.out_biases;
                ps.conv_w = L.conv_w, ps.conv_b = L.conv_b, ps.conv_state = L.conv_state, ps.conv_dim = conv_dim, ps.conv_ks = L.d.dn_kernel_size;
                if (L.in_proj.out_bits && k::gemv_dec_stripe_supported(ps, m->ctx->num_cus)) {
                    // round 5: the projection's workgroups own whole 32-row Hadamard blocks and finish them themselves (k_decode.hip, PRO == 5)
                    dec_gemv(e, ps, "gemv_dec[norm+in_proj+rht+conv]");
                } else { // projection, the transform, then DeltaNetConvUpdate as a launch of its own
                    dec_gemv(e, p, "gemv_dec[norm+in_proj]");
                    RUN("rht_out_rows", 0, k::rht_out_rows(s, m->in_proj, L.in_proj.out_bits, (const uint16_t*)L.in_proj.out_biases, L.in_proj.n, nullptr, nullptr, nullptr, 0, L.conv_w,
                                                            L.conv_b, L.conv_state, L.d.dn_kernel_size, conv_dim));
                }
            } else {
                next_norm(p, L.pre_mixer, l > 0 ? 2 : 1, nullptr, true);
                // DeltaNetConvUpdate rides in the in-proj epilogue: the lane that finishes a conv channel's row convolves it
                p.conv_w = L.conv_w, p.conv_b = L.conv_b, p.conv_state = L.conv_state, p.conv_dim = conv_dim, p.conv_ks = L.d.dn_kernel_size;
                dec_gemv(e, p, "gemv_dec[norm+in_proj+conv]");
            }
            {
                k::DeltaDecParams q{};
                q.in_proj = m->in_proj, q.a_log = L.a_log, q.dt_bias = L.dt_bias, q.state = L.ssm_state, q.o = m->dn_o, q.sz = m->dn_sz;
                q.num_v_heads = Hv, q.num_k_heads = Hk, q.head_v_dim = Dv, q.key_dim = Hk * Dk, q.value_dim = Hv * Dv;
                RUN("delta_dec", (size_t)2 * Hv * Dv * Dk * 4, k::delta_dec(s, q));
                // ... and the RMSNorm * SiLU(z) gate in the out-proj prologue (it needs all Dv outputs of a head)
                k::DecGemvParams op = dec_gemv_base(L.out_proj, m->delta_out, m->mixed);
                op.dg_o = m->dn_o, op.dg_sz = m->dn_sz, op.dg_w = L.dn_norm, op.dg_dv = Dv, op.dg_eps = L.d.dn_norm_epsilon;
                // an RHT out-projection (round 5: until then delta_net_update on 16 workgroups + an InputRht launch + a plain-row GEMV): its InputRht is
                // applied to the gated row inside the norm-gate prologue; its OutputRht (+ bias) rides in the next normalised GEMV's prologue as before
                op.in_rht_bits = L.out_proj.in_bits;
                dec_gemv_row_parallel(e, op, "gemv_dec[gate+out_proj]");
                if (L.out_proj.out_signs) pending = &L.out_proj, pending_row = m->mixed;
            }
        }
        const uint16_t* down_in = nullptr; // the down projection's input row once its InputRht has been applied
        if (linear_rht(L.up)) { // GatedActMul needs the OutputRht of both halves: projection, then the transform (+ bias) and the product
            k::DecGemvParams up = dec_gemv_base(L.up, m->mixed, m->up);
            next_norm(up, L.pre_mlp, 2, &L.up);
            down_in = L.down.in_bits ? m->rht_scratch : m->gated;
            k::DecGemvParams us = up;
            us.act_mul = 1, us.act_type = L.d.activation, us.out[0] = (uint16_t*)down_in;
            us.ep_out_bits = L.up.out_bits, us.ep_bias = (const uint16_t*)L.up.out_biases, us.ep_next_in_bits = L.down.in_bits;
            if (L.up.out_bits && k::gemv_dec_stripe_supported(us, m->ctx->num_cus)) {
                // round 5: up | gate rows, their OutputRht (+ bias), GatedActMul and the down projection's InputRht in ONE launch (k_decode.hip, PRO == 5)
                dec_gemv(e, us, "gemv_dec[norm+up+rht+act]");
            } else {
                dec_gemv(e, up, "gemv_dec[norm+up]");
                // ... as ONE launch with the down projection's InputRht (a thread per stripe; the reference's four kernels on one row)
                RUN("rht_mlp_join", 0, k::rht_mlp_join(s, m->up, L.up.out_bits, (const uint16_t*)L.up.out_biases, L.down.in_bits, (uint16_t*)down_in, L.d.hidden_dim, L.d.activation));
            }
        } else {
            k::DecGemvParams up = dec_gemv_base(L.up, m->mixed, m->gated);
            next_norm(up, L.pre_mlp, 2, nullptr, true);
            up.act_mul = 1, up.act_type = L.d.activation;
            dec_gemv(e, up, "gemv_dec[norm+up+act]");
        }
        dec_gemv_row_parallel(e, dec_gemv_base(L.down, down_in ? down_in : in_transform(L.down, m->gated), hidden), "gemv_dec[down]");
        if (L.down.out_signs) pending = &L.down, pending_row = hidden;
        if (m->taps) flush_pending(); // (debug taps hold finished rows)
        if (m->taps) RUN("tensor_copy", 0, k::tensor_copy(s, hidden, m->taps + ((size_t)l * m->chunk) * d, UZU_BF16, d));
    }
    m->tap_rows = 1;
    const DLinear& ro = m->d.tied_embeddings ? m->embedding : m->output_embedding;
    k::DecGemvParams r = dec_gemv_base(ro, hidden, m->logits);
    next_norm(r, m->output_norm, 2); // (an RHT read-out is not fused: model_fusable)
    r.normed_out = m->last_normed;
    r.part_val = m->amax_val, r.part_idx = m->amax_idx, r.part_capacity = kArgmaxPartials;
    uint32_t grid = 0;
    dec_gemv(e, r, "gemv_dec[norm+readout+argmax]", &grid);
    if (m->tp && m->sampling.on) { // stochastic sampling over the gathered row (stream.rs:598-600 seed), then the plain commit with that token
        RUN("derive_seed", 0, k::derive_seed(s, m->sampling.seed, m->d_ctx_len, 0, m->d_seed));
        RUN("tp_gather_logits", (size_t)m->d.vocab_size * 4, tp::gather_logits(m->tp, s, m->logits, ro.n, m->vocab_offset, m->d.vocab_size, 1, m->tp_gather_f32, m->tp_gather_bf16));
        k::UnifiedSamplingParams sp = m->sampling.p;
        sp.logits = m->tp_gather_bf16, sp.dt = UZU_BF16, sp.output = m->d_out_token, sp.seeds = m->d_seed, sp.vocab_size = m->d.vocab_size, sp.batch_size = 1;
        RUN("unified_sampling", (size_t)m->d.vocab_size * 2, k::unified_sampling(s, sp, m->sampling_scratch));
        k::CommitEmbed eb{};
        eb.token_in = m->d_out_token;
        RUN("argmax_commit", 0, k::argmax_commit(s, m->amax_val, m->amax_idx, grid, m->d_ctx_len, m->d_tokens, m->d_out_token, m->d_sampled, &eb));
    } else if (m->tp) {
        RUN("tp_argmax_key", 0, tp::argmax_key(s, m->amax_val, m->amax_idx, grid, m->vocab_offset, m->tp_key));
        RUN("all_reduce", 8, tp::all_reduce_max_u64(m->tp, s, m->tp_key, 1));
        RUN("tp_commit_key", 0, tp::commit_key(s, m->tp_key, m->d_ctx_len, m->d_tokens, m->d_out_token, m->d_sampled));
    } else {
        k::CommitEmbed eb{};
        eb.weights = (const uint8_t*)m->embedding.w, eb.scales = (const uint16_t*)m->embedding.scales, eb.zero_points = m->embedding.zp;
        eb.biases = (const uint16_t*)m->embedding.biases, eb.output = hidden;
        eb.vocab_size = m->d.vocab_size, eb.model_dim = d, eb.group_size = m->embedding.group, eb.bits = m->embedding.bits, eb.method = m->embedding.method;
        eb.input_scale = m->d.input_scale;
        if (m->sampling.on) { // stream.rs:598-600: the seed of a decode step is derived from the context length before it
            RUN("derive_seed", 0, k::derive_seed(s, m->sampling.seed, m->d_ctx_len, 0, m->d_seed));
            k::UnifiedSamplingParams sp = m->sampling.p;
            sp.logits = m->logits, sp.dt = UZU_BF16, sp.output = m->d_out_token, sp.seeds = m->d_seed, sp.vocab_size = ro.n, sp.batch_size = 1;
            RUN("unified_sampling", (size_t)ro.n * 2, k::unified_sampling(s, sp, m->sampling_scratch));
            eb.token_in = m->d_out_token;
        }
        RUN("argmax_commit", 0, k::argmax_commit(s, m->amax_val, m->amax_idx, grid, m->d_ctx_len, m->d_tokens, m->d_out_token, m->d_sampled, &eb));
    }
    if (e.st != UZU_OK) return e.st;
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        set_error("engine: fused decode launch failed: %s", hipGetErrorString(err));
        return UZU_ERR_HIP;
    }
    return UZU_OK;
}

// (the reference-order mode runs the one-kernel-per-reference-kernel path, eagerly: its kernels take scratch from the stream workspace)
bool decode_is_fused(const uzu_hip_model* m) { return m->fusable && !(m->flags & UZU_MODEL_NO_FUSION) && !k::exact_mode(); }

// eager decode step; keeps `hidden_ready` in step with what the step left behind
uzu_status encode_decode(uzu_hip_model* m, hipStream_t s) {
    if (decode_is_fused(m)) {
        const uzu_status st = encode_decode_fused(m, s, !m->hidden_ready);
        m->hidden_ready = st == UZU_OK && commit_embeds(m);
        return st;
    }
    m->hidden_ready = false;
    return encode_forward(m, s, 1, true);
}

uzu_status build_decode_graph(uzu_hip_model* m, hipGraphExec_t* out, bool two_pass) {
    hipStream_t s = m->ctx->stream;
    HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    m->regime_override = two_pass ? 1 : 0;
    // the captured step starts at layer 0 when its own commit leaves the next embedding row behind (enqueue_decode looks
    // the first one up eagerly); otherwise the lookup is part of the graph
    uzu_status st = decode_is_fused(m) ? encode_decode_fused(m, s, !commit_embeds(m)) : encode_forward(m, s, 1, true);
    m->regime_override = -1;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(s, &g);
    if (st != UZU_OK) {
        if (g) (void)hipGraphDestroy(g);
        return st;
    }
    if (e != hipSuccess) {
        set_error("engine: graph capture failed: %s", hipGetErrorString(e));
        return UZU_ERR_HIP;
    }
    HIPCHK(hipGraphInstantiate(out, g, nullptr, nullptr, 0));
    HIPCHK(hipGraphDestroy(g));
    return UZU_OK;
}

uzu_status enqueue_decode(uzu_hip_model* m, uint32_t steps) {
    for (uint32_t i = 0; i < steps; ++i) {
        UZU_REQUIRE(m->context_length + 1 <= m->d.max_context_length, "decode: context length %u exceeds max_context_length %u", m->context_length + 1,
                    m->d.max_context_length);
        if ((m->flags & UZU_MODEL_NO_GRAPH) || k::exact_mode()) {
            UZU_PROPAGATE(encode_decode(m, m->ctx->stream));
        } else {
            const bool two = m->context_length + 1 > 1024;
            hipGraphExec_t* g = two ? &m->graph_two : &m->graph_single;
            if (!*g) UZU_PROPAGATE(build_decode_graph(m, g, two));
            if (decode_is_fused(m) && commit_embeds(m) && !m->hidden_ready) { // first step after a prefill / set_next_token
                Enc e{m, m->ctx->stream};
                encode_embed_row0(e);
                UZU_PROPAGATE(e.st);
            }
            HIPCHK(hipGraphLaunch(*g, m->ctx->stream));
            m->hidden_ready = decode_is_fused(m) && commit_embeds(m);
        }
        m->context_length += 1;
    }
    return UZU_OK;
}

} // namespace eng
} // namespace uzu
