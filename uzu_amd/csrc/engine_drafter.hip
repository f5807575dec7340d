// engine_drafter.hip -- the DFlash draft model of the reference's tree speculator on the HIP engine (include/uzu_hip_engine.h, "speculator").
//
// Restates
//   DFlash::{new, empty_state, encode_accept, encode_draft}   BU/../encodable_block/dflash.rs:41-346
//   Attention::append_projected_kv                            BU/../encodable_block/mixer/attention/mode.rs:146-169
//   QKVNorm::encode_key_value                                 BU/../encodable_block/mixer/attention/qkv_norm.rs:128-175
//   the Argmax construction's greedy sampling                 BU/../speculators/dflash_tfm.rs:167-217
//   the target's hidden-feature taps                          BU/../encodable_block/transformer.rs:160-171,285-293; stream.rs:213-214,632-633
// The draft layers are ordinary TransformerLayers: they run as a HEADLESS uzu_hip_model (no embedding, no read-out of its own) through the same
// encode_forward as the target's; rows are looked up in, and read out through, the TARGET's embedding (dflash.rs:285,335).  Tree shaping is host code
// (uzu_amd/speculator.py <- dflash_tfm.rs:133-343, uzu_amd/trie.py <- trie.rs).
#include "engine_types.h"

using namespace uzu;
using namespace uzu::eng;

struct uzu_hip_drafter {
    uzu_hip_context* ctx = nullptr;
    uzu_hip_model* target = nullptr; // borrowed: embedding lookup / read-out, the feature taps
    uzu_hip_model* core = nullptr;   // the draft layers + their KV state (AttentionState::Full of context_capacity + chunk rows)
    uzu_dflash_desc d{};             // scalars only
    std::vector<uint32_t> target_layer_ids;
    DLinear context_projection, state_kv_projection;
    DNorm context_norm, output_norm;
    uint32_t layer_kv_dim = 0;
    uint16_t *packed = nullptr;       // [chunk][num_target_layers * d]
    uint16_t *projected_kv = nullptr; // [chunk][num_layers * layer_kv_dim]
    uint16_t *draft_hidden = nullptr; // [block][d]: output norm of every draft row
    float* logits = nullptr;          // f32 [block - 1][vocab rows]
    uint32_t *d_accepted = nullptr, *d_tokens_out = nullptr, *d_draft_tokens = nullptr;
    void* argmax_scratch = nullptr;
    std::vector<uint32_t> last_tokens;
    uint32_t last_rows = 0;
    float last_draft_ms = 0.f, last_accept_ms = 0.f;
};

namespace {

// rows accepted[t] of the tapped layers' feature blocks, side by side per token: packed[t][f][:] = features[f][accepted[t]][:]  (dflash.rs:214-229)
__global__ void gather_features_kernel(const uint16_t* features, size_t feature_stride, const uint32_t* accepted, uint16_t* packed, uint32_t nf, uint32_t d, uint32_t n) {
    const size_t total = (size_t)n * nf * (d / 8);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t c = i % (d / 8);
        const uint32_t f = (i / (d / 8)) % nf;
        const uint32_t t = i / ((size_t)(d / 8) * nf);
        const uint4* src = (const uint4*)(features + (size_t)f * feature_stride + (size_t)accepted[t] * d);
        ((uint4*)(packed + ((size_t)t * nf + f) * d))[c] = src[c];
    }
}
// key | value rows of draft layer l out of the all-layers projection: out[t][:] = projected_kv[t][l * kvd : (l + 1) * kvd]  (dflash.rs:242-254)
__global__ void take_layer_kv_kernel(const uint16_t* projected_kv, uint16_t* out, uint32_t nl, uint32_t l, uint32_t kvd, uint32_t n) {
    const size_t total = (size_t)n * (kvd / 8);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t c = i % (kvd / 8), t = i / (kvd / 8);
        ((uint4*)(out + (size_t)t * kvd))[c] = ((const uint4*)(projected_kv + ((size_t)t * nl + l) * kvd))[c];
    }
}
__global__ void fill_draft_tokens_kernel(uint32_t* tokens, uint32_t first, uint32_t mask, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tokens[i] = i == 0 ? first : mask;
}
__global__ void advance_context_kernel(uint32_t* ctx_len, uint32_t n) { *ctx_len += n; }

} // namespace

extern "C" {

// ---- the target's hidden-feature taps ----------------------------------------------------------------------------------------------------------------
uzu_status uzu_hip_model_set_feature_layers(uzu_hip_model* m, const uint32_t* layer_ids, uint32_t count) {
    UZU_REQUIRE(m && (layer_ids || !count), "model_set_feature_layers: null argument");
    UZU_UNSUPPORTED(m->tp != nullptr && count, "model_set_feature_layers: hidden-feature taps of a tensor-parallel shard");
    for (uint32_t i = 0; i < count; ++i) UZU_REQUIRE(layer_ids[i] < m->d.num_layers, "model_set_feature_layers: layer %u of %u", layer_ids[i], m->d.num_layers);
    (void)hipSetDevice(m->ctx->device);
    HIPCHK(hipStreamSynchronize(m->ctx->stream));
    if (m->features) dev_free(m, m->features), m->features = nullptr;
    m->feature_layers.assign(layer_ids, layer_ids + count);
    m->feature_rows = 0;
    if (count) {
        void* p = nullptr;
        UZU_PROPAGATE(dev_alloc(m, (size_t)count * m->chunk * m->d.model_dim * 2, &p, poison_level() < 2));
        m->features = (uint16_t*)p;
    }
    // the decode step of such a model files the features as well: it runs the one-kernel-per-reference-kernel pass (captured graphs of the other form are stale)
    m->fusable = model_fusable(m);
    ++m->sampling_epoch;
    HIPCHK(hipStreamSynchronize(m->ctx->stream));
    return UZU_OK;
}

// bf16 [rows][model_dim] of tap `index` (the order of set_feature_layers) for the rows of the last pass
uzu_status uzu_hip_model_read_features(uzu_hip_model* m, uint32_t index, uint16_t* out, uint32_t* rows) {
    UZU_REQUIRE(m && index < m->feature_layers.size(), "model_read_features: tap %u of %zu", index, m ? m->feature_layers.size() : (size_t)0);
    HIPCHK(hipStreamSynchronize(m->ctx->stream));
    if (out) HIPCHK(hipMemcpy(out, m->features + (size_t)index * m->chunk * m->d.model_dim, (size_t)m->feature_rows * m->d.model_dim * 2, hipMemcpyDeviceToHost));
    if (rows) *rows = m->feature_rows;
    return UZU_OK;
}

// DecoderEncodeOutput::final_hidden of the last prefill (1 row: the sampled one) or tree pass (every node): the output-norm rows, bf16 [rows][model_dim]
// (ForwardPassChaining's output_norm, stream.rs:466-476: the Weaver construction's prefix row 0).  `out` may be null (rows only).
uzu_status uzu_hip_model_read_final_hidden(uzu_hip_model* m, uint16_t* out, uint32_t capacity_rows, uint32_t* rows) {
    UZU_REQUIRE(m && m->final_hidden && m->final_hidden_rows, "model_read_final_hidden: no prefill / tree pass has run");
    UZU_REQUIRE(!out || capacity_rows >= m->final_hidden_rows, "model_read_final_hidden: %u rows, room for %u", m->final_hidden_rows, capacity_rows);
    HIPCHK(hipStreamSynchronize(m->ctx->stream));
    if (out) HIPCHK(hipMemcpy(out, m->final_hidden, (size_t)m->final_hidden_rows * m->d.model_dim * 2, hipMemcpyDeviceToHost));
    if (rows) *rows = m->final_hidden_rows;
    return UZU_OK;
}

// ---- the draft model ---------------------------------------------------------------------------------------------------------------------------------
void uzu_hip_drafter_destroy(uzu_hip_drafter* f) {
    if (!f) return;
    if (f->core) uzu_hip_model_destroy(f->core); // (the drafter's own tensors and scratch were allocated through the core model: freed with it)
    delete f;
}

uzu_status uzu_hip_drafter_create(uzu_hip_context* ctx, uzu_hip_model* target, const uzu_dflash_desc* desc, uzu_hip_drafter** out) {
    UZU_REQUIRE(ctx && target && desc && out, "drafter_create: null argument");
    UZU_REQUIRE(desc->num_layers > 0 && desc->layers && desc->num_target_layers > 0 && desc->target_layer_ids, "drafter_create: no layers / no target layers");
    UZU_REQUIRE(desc->model_dim == target->d.model_dim, "drafter_create: draft model_dim %u != target model_dim %u (the draft model embeds through the target's table)", desc->model_dim,
                target->d.model_dim);
    UZU_REQUIRE(desc->block_size >= 2 && desc->block_size <= k::kDnTreeMaxNodes, "drafter_create: block_size %u outside 2..%u", desc->block_size, k::kDnTreeMaxNodes);
    UZU_REQUIRE(desc->context_capacity > 0 && desc->context_capacity <= desc->rope.max_sequence_length, "drafter_create: state capacity %u exceeds the RoPE capacity %u (dflash.rs:179)",
                desc->context_capacity, desc->rope.max_sequence_length);
    UZU_UNSUPPORTED(target->tp != nullptr, "drafter_create: a tensor-parallel target");
    const uint32_t d = desc->model_dim, nl = desc->num_layers, nf = desc->num_target_layers;
    for (uint32_t l = 0; l < nl; ++l) { // DFlashNewError::InvalidAttentionConfig (dflash.rs:118-120)
        const uzu_layer_desc& L = desc->layers[l];
        UZU_REQUIRE(L.mixer_kind == UZU_MIXER_ATTENTION && !L.is_kv_sharing && !L.has_ple, "drafter_create: DFlash layers must use attention mixers that own their state");
        UZU_UNSUPPORTED(L.sliding_window_size != 0, "drafter_create: sliding-window draft layers");
        UZU_REQUIRE(L.num_groups == desc->layers[0].num_groups && L.head_dim == desc->layers[0].head_dim, "drafter_create: the layers' key / value widths differ (one state_kv_projection feeds them all)");
    }
    for (uint32_t i = 0; i < nf; ++i) UZU_REQUIRE(desc->target_layer_ids[i] < target->d.num_layers, "drafter_create: target layer %u of %u", desc->target_layer_ids[i], target->d.num_layers);
    const uint32_t kvd = 2 * desc->layers[0].num_groups * desc->layers[0].head_dim;
    UZU_REQUIRE(desc->context_projection.n == d && desc->context_projection.k == d * nf && desc->state_kv_projection.k == d && desc->state_kv_projection.n == nl * kvd,
                "drafter_create: context / state_kv projection shapes inconsistent");
    UZU_REQUIRE(d % 8 == 0 && kvd % 8 == 0, "drafter_create: row widths must be multiples of 8 elements");
    (void)hipSetDevice(ctx->device);
    auto* f = new uzu_hip_drafter();
    f->ctx = ctx, f->target = target, f->d = *desc, f->layer_kv_dim = kvd;
    f->d.layers = nullptr, f->d.target_layer_ids = nullptr;
    f->target_layer_ids.assign(desc->target_layer_ids, desc->target_layer_ids + nf);
    // the layer stack: a headless model (no embedding, no read-out); its KV state holds context_capacity + chunk rows
    uzu_model_desc core{};
    core.vocab_size = 0, core.model_dim = d, core.num_layers = nl, core.tied_embeddings = 1, core.input_scale = 1.0f, core.logit_scale = 1.0f;
    core.max_context_length = desc->context_capacity, core.rope = desc->rope, core.output_norm = desc->output_norm, core.layers = desc->layers;
    uzu_status st = uzu_hip_model_create(ctx, &core, UZU_MODEL_NO_GRAPH, &f->core);
    if (st != UZU_OK) {
        delete f;
        return st;
    }
    uzu_hip_model* c = f->core;
    c->headless = true;
    c->fusable = false;
    auto fail = [&](uzu_status s) {
        uzu_hip_drafter_destroy(f);
        return s;
    };
#define TRY(x) do { st = (x); if (st != UZU_OK) return fail(st); } while (0)
    TRY(upload_linear(c, desc->context_projection, &f->context_projection));
    TRY(upload_linear(c, desc->state_kv_projection, &f->state_kv_projection));
    TRY(upload_norm(c, desc->context_norm, d, &f->context_norm));
    f->output_norm = c->output_norm;
    if (!f->context_norm.present || !f->output_norm.present) return fail((set_error("drafter_create: context_norm and output_norm are required"), UZU_ERR_INVALID_ARGUMENT));
    void* p = nullptr;
    const bool zero = poison_level() < 2;
    const uint32_t vocab_rows = uzu_hip_model_logit_count(target);
    TRY(dev_alloc(c, (size_t)c->chunk * nf * d * 2, &p, zero)); f->packed = (uint16_t*)p;
    TRY(dev_alloc(c, (size_t)c->chunk * nl * kvd * 2, &p, zero)); f->projected_kv = (uint16_t*)p;
    TRY(dev_alloc(c, (size_t)desc->block_size * d * 2, &p, zero)); f->draft_hidden = (uint16_t*)p;
    TRY(dev_alloc(c, (size_t)(desc->block_size - 1) * vocab_rows * 4, &p, zero)); f->logits = (float*)p;
    TRY(dev_alloc(c, (size_t)c->chunk * 4, &p, zero)); f->d_accepted = (uint32_t*)p;
    TRY(dev_alloc(c, (size_t)desc->block_size * 4, &p, zero)); f->d_tokens_out = (uint32_t*)p;
    TRY(dev_alloc(c, (size_t)desc->block_size * 4, &p, zero)); f->d_draft_tokens = (uint32_t*)p;
    TRY(dev_alloc(c, k::argmax_scratch_bytes(desc->block_size), &f->argmax_scratch));
#undef TRY
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail((set_error("drafter_create: synchronisation after load failed"), UZU_ERR_HIP));
    *out = f;
    return UZU_OK;
}

uzu_status uzu_hip_drafter_reset(uzu_hip_drafter* f) {
    UZU_REQUIRE(f, "drafter_reset: null drafter");
    return uzu_hip_model_reset(f->core);
}
uint32_t uzu_hip_drafter_context_length(const uzu_hip_drafter* f) { return f ? f->core->context_length : 0; }

// DFlash::encode_accept (dflash.rs:190-271) over rows `accepted_indices` of the target's LAST pass (its feature taps must be exactly this drafter's
// target_layer_ids, in order: uzu_hip_model_set_feature_layers)
uzu_status uzu_hip_drafter_accept(uzu_hip_drafter* f, const uint32_t* accepted_indices, uint32_t count) {
    UZU_REQUIRE(f && (accepted_indices || !count), "drafter_accept: null argument");
    if (!count) return UZU_OK; // dflash.rs:197-199
    uzu_hip_model *t = f->target, *c = f->core;
    UZU_REQUIRE(t->feature_layers == f->target_layer_ids, "drafter_accept: the target's feature taps are not this drafter's target_layer_ids (uzu_hip_model_set_feature_layers)");
    UZU_REQUIRE(count <= c->chunk, "drafter_accept: %u tokens, at most %u per call", count, c->chunk);
    for (uint32_t i = 0; i < count; ++i) UZU_REQUIRE(accepted_indices[i] < t->feature_rows, "drafter_accept: accepted index %u out of the last pass's %u rows", accepted_indices[i], t->feature_rows);
    UZU_REQUIRE(c->context_length + count <= f->d.context_capacity, "drafter_accept: DFlash state capacity %u exceeded (%u + %u)", f->d.context_capacity, c->context_length, count);
    (void)hipSetDevice(f->ctx->device);
    hipStream_t s = f->ctx->stream;
    const uint32_t d = f->d.model_dim, nf = f->d.num_target_layers, nl = f->d.num_layers, kvd = f->layer_kv_dim;
    Enc e{c, s};
    HIPCHK(hipEventRecord(c->ev0, s));
    HIPCHK(hipMemcpyAsync(f->d_accepted, accepted_indices, (size_t)count * 4, hipMemcpyHostToDevice, s));
    {
        const size_t total = (size_t)count * nf * (d / 8);
        const uint32_t blocks = (uint32_t)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
        hipLaunchKernelGGL(gather_features_kernel, dim3(blocks), dim3(256), 0, s, t->features, (size_t)t->chunk * d, f->d_accepted, f->packed, nf, d, count);
        ++c->launches;
    }
    linear(e, f->context_projection, f->packed, c->mixed, count);
    norm(e, f->context_norm, c->mixed, c->normed, nullptr, 0, count, d);
    linear(e, f->state_kv_projection, c->normed, f->projected_kv, count);
    for (uint32_t l = 0; l < nl; ++l) {
        DLayer& L = c->layers[l];
        const uint32_t nkv = L.d.num_groups, hd = L.d.head_dim;
        {
            const size_t total = (size_t)count * (kvd / 8);
            const uint32_t blocks = (uint32_t)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
            hipLaunchKernelGGL(take_layer_kv_kernel, dim3(blocks), dim3(256), 0, s, f->projected_kv, c->qkv, nl, l, kvd, count);
            ++c->launches;
        }
        // Attention::append_projected_kv (mode.rs:146-169): QKVNorm::encode_key_value (q_heads = 0), AttentionPrepare with no query heads, accept of all rows
        if (L.kn.present) RUN("qkv_norm", 0, k::qkv_norm(s, c->qkv, UZU_BF16, L.kn.scales, count, 2 * nkv, hd, L.kn.eps, L.kn.offset, 0, nkv, L.kn.full_layer));
        if (L.d.normalize_values) RUN("qkv_norm", 0, k::qkv_norm(s, c->qkv, UZU_BF16, nullptr, count, 2 * nkv, hd, 1e-6f, 0.0f, nkv, nkv, 1));
        RUN("attention_prepare", 0, k::attention_prepare(s, c->qkv, c->queries, L.keys, L.values, L.rope_cos, L.rope_sin, 0, nkv, hd, L.d.use_rope ? L.rope_dim : 0, 0, count, 1u,
                                                          c->d_ctx_len, 0u, nullptr));
    }
    hipLaunchKernelGGL(advance_context_kernel, dim3(1), dim3(1), 0, s, c->d_ctx_len, count);
    HIPCHK(hipEventRecord(c->ev1, s));
    UZU_PROPAGATE(e.st);
    HIPCHK(hipStreamSynchronize(s)); // accepted_indices is caller memory
    (void)hipEventElapsedTime(&f->last_accept_ms, c->ev0, c->ev1);
    c->context_length += count;
    return UZU_OK;
}

// DFlash::encode_draft (dflash.rs:273-345) + the greedy tokens of the Argmax construction (dflash_tfm.rs:167-217): `batch_size` rows = [target_output_token,
// mask, mask, ...] -> tokens_out [batch_size - 1] (the greedy token of every lookahead row; null: not wanted).  Nothing is accepted.
uzu_status uzu_hip_drafter_draft(uzu_hip_drafter* f, uint32_t target_output_token, uint32_t batch_size, uint32_t* tokens_out) {
    UZU_REQUIRE(f, "drafter_draft: null drafter");
    uzu_hip_model *t = f->target, *c = f->core;
    UZU_REQUIRE(batch_size >= 2 && batch_size <= f->d.block_size, "drafter_draft: batch size %u outside 2..block size %u (dflash.rs:283)", batch_size, f->d.block_size);
    UZU_REQUIRE(c->context_length + batch_size <= f->d.rope.max_sequence_length, "drafter_draft: block positions exceed the RoPE capacity (dflash.rs:284-287)");
    (void)hipSetDevice(f->ctx->device);
    hipStream_t s = f->ctx->stream;
    const uint32_t d = f->d.model_dim, rows = batch_size - 1, vocab_rows = uzu_hip_model_logit_count(t);
    Enc et{t, s}; // the target's embedding / read-out (its InputRht scratch, its launch counter)
    HIPCHK(hipEventRecord(c->ev0, s));
    hipLaunchKernelGGL(fill_draft_tokens_kernel, dim3(1), dim3(64), 0, s, f->d_draft_tokens, target_output_token, f->d.mask_token_id, batch_size);
    {
        Enc& e = et; // Embedding::encode_lookup of the target into the core's `hidden` (embedding.rs:345-372)
        if (t->embedding.method == UZU_QUANT_NONE)
            RUN("full_precision_embedding_lookup", 0, k::full_precision_embedding_lookup(s, f->d_draft_tokens, t->embedding.w, c->hidden, UZU_BF16, batch_size, t->d.vocab_size, d, t->d.input_scale));
        else
            RUN("quantized_embedding_lookup", 0, k::quantized_embedding_lookup(s, f->d_draft_tokens, (const uint8_t*)t->embedding.w, t->embedding.scales, t->embedding.zp, t->embedding.biases, c->hidden,
                                                                                UZU_BF16, batch_size, t->d.vocab_size, d, t->d.input_scale, t->embedding.group, t->embedding.bits, t->embedding.method));
        if (t->embedding.out_signs)
            RUN("activation_transform", 0, k::activation_transform(s, nullptr, c->hidden, nullptr, nullptr, nullptr, t->embedding.out_signs, UZU_BF16, batch_size, d, UZU_ACTIVATION_TRANSFORM_OUTPUT_RHT, 0, 0));
    }
    UZU_PROPAGATE(et.st);
    // the draft layers over context + block (flat topology, positions context .. context + batch: dflash.rs:296-323); two-pass attention scratch as for a prefill pass
    {
        uint32_t max_heads = 0, max_hd = 0;
        for (auto& L : c->layers) max_heads = max_heads > L.d.num_heads ? max_heads : L.d.num_heads, max_hd = max_hd > L.d.head_dim ? max_hd : L.d.head_dim;
        if (c->context_length + batch_size > 1024) UZU_PROPAGATE(ensure_partials(c, batch_size * max_heads, max_hd));
    }
    UZU_PROPAGATE(encode_forward(c, s, batch_size, false));
    {
        Enc e{c, s};
        norm(e, f->output_norm, c->hidden, f->draft_hidden, c->last_shortcut, 2, batch_size, d); // ShortcutMode::Add, every row (dflash.rs:325-328)
        UZU_PROPAGATE(e.st);
    }
    {
        // Embedding::encode_readout(batch_size - 1, rows 1.., DataType::F32) (dflash.rs:330-335; embedding.rs:374-456): the target's read-out, widened output
        Enc& e = et;
        const DLinear& src = t->d.tied_embeddings ? t->embedding : t->output_embedding;
        const int32_t* in_signs = t->d.tied_embeddings ? t->embedding.out_signs : t->output_embedding.in_signs;
        const uint16_t* a = f->draft_hidden + d;
        if (in_signs) {
            RUN("activation_transform", 0, k::activation_transform(s, a, t->rht_scratch, nullptr, nullptr, nullptr, in_signs, UZU_BF16, rows, d, UZU_ACTIVATION_TRANSFORM_INPUT_RHT, 0, 0));
            a = t->rht_scratch;
        }
        k::MatmulParams p{};
        p.a = a, p.b = src.w, p.scales = src.scales, p.biases = src.biases, p.zero_points = src.zp, p.d = f->logits;
        p.w_dt = p.a_dt = UZU_BF16, p.d_dt = UZU_F32;
        p.b_kind = src.method == UZU_QUANT_NONE ? UZU_MATMUL_B_FULL_PRECISION
                 : src.method == UZU_QUANT_SCALE_BIAS ? UZU_MATMUL_B_SCALE_BIAS
                 : src.method == UZU_QUANT_SCALE_ZERO_POINT ? UZU_MATMUL_B_SCALE_ZERO_POINT : UZU_MATMUL_B_SCALE_SYMMETRIC;
        p.bits = src.bits, p.group_size = src.group, p.ab_scale = 1.0f, p.m = rows, p.n = src.n, p.k = src.k;
        const char* variant = "matmul";
        e.begin();
        const uzu_status r = k::matmul(s, p, t->ctx->num_cus, &variant);
        e.run(r, variant, k::matmul_algorithmic_bytes(p));
        if (t->d.logit_scale != 1.0f || t->d.logit_soft_cap != 0.0f)
            RUN("logit_transform", 0, k::logit_transform(s, f->logits, UZU_F32, vocab_rows * rows, t->d.logit_scale, t->d.logit_soft_cap, t->d.logit_soft_cap != 0.0f));
        // Sampling::new(DataType::F32, vocab), SamplingMethod::Greedy over the lookahead rows (dflash_tfm.rs:112,181-203)
        RUN("argmax", (size_t)vocab_rows * 4 * rows, k::argmax(s, f->logits, UZU_F32, f->d_tokens_out, vocab_rows, rows, f->argmax_scratch));
        UZU_PROPAGATE(e.st);
    }
    HIPCHK(hipEventRecord(c->ev1, s));
    f->last_tokens.resize(rows);
    HIPCHK(hipMemcpyAsync(f->last_tokens.data(), f->d_tokens_out, (size_t)rows * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    (void)hipEventElapsedTime(&f->last_draft_ms, c->ev0, c->ev1);
    f->last_rows = batch_size;
    if (tokens_out) memcpy(tokens_out, f->last_tokens.data(), (size_t)rows * 4);
    return UZU_OK;
}

// outputs of the last draft: draft_hidden bf16 [rows][model_dim] (DFlashOutput::draft_hidden), logits f32 [rows - 1][vocab rows] (DFlashOutput::logits); either may be null
uzu_status uzu_hip_drafter_read_draft(uzu_hip_drafter* f, uint16_t* draft_hidden_out, float* logits_out, uint32_t* rows) {
    UZU_REQUIRE(f && f->last_rows >= 2, "drafter_read_draft: no draft has run");
    HIPCHK(hipStreamSynchronize(f->ctx->stream));
    if (draft_hidden_out) HIPCHK(hipMemcpy(draft_hidden_out, f->draft_hidden, (size_t)f->last_rows * f->d.model_dim * 2, hipMemcpyDeviceToHost));
    if (logits_out) HIPCHK(hipMemcpy(logits_out, f->logits, (size_t)(f->last_rows - 1) * uzu_hip_model_logit_count(f->target) * 4, hipMemcpyDeviceToHost));
    if (rows) *rows = f->last_rows;
    return UZU_OK;
}

// device time of the last accept / draft in milliseconds (HIP events on the engine's stream)
uzu_status uzu_hip_drafter_gpu_ms(uzu_hip_drafter* f, float* accept_ms, float* draft_ms) {
    UZU_REQUIRE(f, "drafter_gpu_ms: null drafter");
    if (accept_ms) *accept_ms = f->last_accept_ms;
    if (draft_ms) *draft_ms = f->last_draft_ms;
    return UZU_OK;
}

// engine_weaver.hip: the objects and device buffers a Weaver built on this drafter works with (not part of the public header)
uzu_status uzu_hip_drafter_internal(uzu_hip_drafter* f, uzu_hip_model** core, uzu_hip_model** target, uint16_t** draft_hidden, float** logits, uint32_t* last_rows, uint32_t* block_size) {
    UZU_REQUIRE(f, "drafter: null drafter");
    if (core) *core = f->core;
    if (target) *target = f->target;
    if (draft_hidden) *draft_hidden = f->draft_hidden;
    if (logits) *logits = f->logits;
    if (last_rows) *last_rows = f->last_rows;
    if (block_size) *block_size = f->d.block_size;
    return UZU_OK;
}

} // extern "C"
