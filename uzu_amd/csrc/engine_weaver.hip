// engine_weaver.hip -- the Weaver tree constructor of the reference's speculator on the HIP engine (include/uzu_hip_engine.h, "speculator").
//
// Restates
//   Weaver::{new, encode_prefix, encode_step, encode_tree}           BU/../encodable_block/weaver.rs:166-676
//   WeaverLayer::{encode_prefix_attention, encode_post_attention}    BU/../encodable_block/weaver_layer.rs:150-200
//   Embedding::encode_readout_sparse                                 BU/../encodable_block/embedding.rs:458-530
// over kernels that already exist: RadixTopKSmall, AncestorAttention, WeaverFrontierSelect / InsertChildren / TopChildren (k_speculator.hip), the linears,
// Normalization, AttentionPrepare, the single-pass attention core, GatedActMul.  EncodedWeaverTree::read_nodes and the trie are host code (uzu_amd/speculator.py).
//
// A tree is ~25 launches per round on rows of 1 .. 32 tokens: launch-bound.  Every input of a tree (root token, depth seeds, the target's output-norm row, the
// draft model's hidden rows and logits) sits in device buffers, so the whole construction of a shape is captured ONCE into a hipGraph and replayed.
#include "engine_types.h"

using namespace uzu;
using namespace uzu::eng;

struct uzu_hip_drafter; // engine_drafter.hip
extern "C" uzu_status uzu_hip_drafter_internal(uzu_hip_drafter* f, uzu_hip_model** core, uzu_hip_model** target, uint16_t** draft_hidden, float** logits, uint32_t* last_rows, uint32_t* block_size);

namespace {
struct WLayer {
    DNorm pre_attention_norm, pre_mlp_norm;
    DLinear qkv, out, up, down;
    uint16_t *prefix_kv = nullptr, *node_kv = nullptr; // [2][max rows][d] / [2][max slots][d]
};
constexpr uint32_t kMaxRows = 32;   // FRONTIER_MAX_WIDTH; the prefix has <= max_depth + 1 <= 32 rows as well (checked at creation)
constexpr uint32_t kMaxSlots = 2048; // FRONTIER_MAX_SLOTS
} // namespace

struct uzu_hip_weaver {
    uzu_hip_context* ctx = nullptr;
    uzu_hip_drafter* drafter = nullptr;
    uzu_hip_model *core = nullptr, *target = nullptr; // the drafter's headless model (owns this object's device memory), the target (embedding / read-out)
    uzu_weaver_desc d{};
    DNorm embedding_norm, hidden_state_norm, output_norm;
    DLinear embedding_projection, hidden_state_projection, query_projection;
    std::vector<WLayer> layers;
    float *rope_cos = nullptr, *rope_sin = nullptr;
    // device state of one tree
    uint16_t *target_hidden = nullptr, *prefix_hidden = nullptr, *normed = nullptr, *residual_in = nullptr, *residual_state = nullptr, *qkv = nullptr, *queries = nullptr, *attn_out = nullptr;
    uint16_t *projected = nullptr, *mlp_in = nullptr, *up_out = nullptr, *gated = nullptr, *token_embedding = nullptr, *query = nullptr, *logit_residuals = nullptr, *rht_in = nullptr;
    uint32_t *candidate_ids = nullptr, *node_candidate_ids = nullptr, *packed_tree = nullptr, *frontier = nullptr, *slot_ancestors = nullptr, *node_token_ids = nullptr, *node_metadata = nullptr;
    uint32_t *node_ancestor_indices = nullptr, *node_valid = nullptr, *child_token_ids = nullptr;
    float *candidate_logits = nullptr, *node_candidate_logits = nullptr, *child_logprobs = nullptr;
    uint64_t* depth_seeds = nullptr;
    uint32_t* init_block = nullptr; // the initial values of a tree's state buffers, laid out back to back: one device-to-device copy per buffer restores them
    struct Graph {
        uzu_weaver_tree_shape shape;
        hipGraphExec_t exec;
        uint32_t launches;
    };
    std::vector<Graph> graphs;
    std::vector<uint32_t> host_init;
    float last_ms = 0.f;
    uint32_t last_launches = 0;
};

namespace {

uint32_t slot_count(const uzu_weaver_tree_shape& s) { return 1 + (s.rounds ? s.rounds - 1 : 0) * s.expand_per_round; }

// DenseMlp with SiLU and biases + the residual protocol of WeaverLayer::encode_post_attention (weaver_layer.rs:187-199)
void post_attention(Enc& e, uzu_hip_weaver* w, WLayer& L, uint32_t rows) {
    const uint32_t d = w->d.model_dim;
    linear(e, L.out, w->attn_out, w->projected, rows);
    norm(e, L.pre_mlp_norm, w->projected, w->mlp_in, w->residual_state, 2, rows, d);
    linear(e, L.up, w->mlp_in, w->up_out, rows);
    RUN("gated_act_mul", 0, k::gated_act_mul(e.s, w->up_out, nullptr, w->gated, UZU_BF16, w->d.hidden_dim, rows, 0, 0, UZU_ACT_SILU, 1));
    linear(e, L.down, w->gated, w->residual_in, rows);
}

// the launches of one tree (Weaver::encode_tree behind its host-side initialisation, weaver.rs:536-676)
uzu_status encode_tree_launches(uzu_hip_weaver* w, hipStream_t s, const uzu_weaver_tree_shape& shape, uint16_t* draft_hidden, float* logits) {
    uzu_hip_model *c = w->core, *t = w->target;
    Enc e{c, s};
    c->launches = 0;
    const uzu_weaver_desc& D = w->d;
    const uint32_t d = D.model_dim, heads = D.num_heads, hd = d / heads, P = D.candidate_pool_size, td = D.target_model_dim;
    const uint32_t slots = slot_count(shape), ancestor_stride = D.max_depth, frontier_capacity = slots * shape.expand_width, pool_depth_count = shape.dflash_depth - 1;
    const uint32_t vocab = uzu_hip_model_logit_count(t);
    const float scale = 1.0f / sqrtf((float)hd);
    RUN("radix_top_k_small", (size_t)pool_depth_count * vocab * 4, k::radix_top_k_small(s, logits, w->candidate_ids, w->candidate_logits, pool_depth_count, vocab, P));
    // ---- encode_prefix (weaver.rs:283-352): row 0 = the target's output-norm row, rows 1.. = the draft rows
    const uint32_t depth = shape.dflash_depth;
    RUN("tensor_copy", 0, k::tensor_copy(s, w->target_hidden, w->prefix_hidden, UZU_BF16, td));
    RUN("tensor_copy", 0, k::tensor_copy(s, draft_hidden + td, w->prefix_hidden + td, UZU_BF16, (depth - 1) * td));
    norm(e, w->hidden_state_norm, w->prefix_hidden, w->normed, nullptr, 0, depth, td);
    linear(e, w->hidden_state_projection, w->normed, w->residual_in, depth);
    for (size_t l = 0; l < w->layers.size(); ++l) {
        WLayer& L = w->layers[l];
        norm(e, L.pre_attention_norm, w->residual_in, w->normed, w->residual_state, l > 0 ? 2 : 1, depth, d);
        linear(e, L.qkv, w->normed, w->qkv, depth);
        RUN("attention_prepare", 0, k::attention_prepare(s, w->qkv, w->queries, L.prefix_kv, L.prefix_kv + (size_t)depth * d, w->rope_cos, w->rope_sin, heads, heads, hd, hd, 0, depth, 1u, nullptr, 1u, nullptr));
        if (l + 1 == w->layers.size()) break; // the last layer only contributes its keys / values (weaver.rs:343-348)
        k::AttentionParams a{};
        a.queries = w->queries, a.keys = L.prefix_kv, a.values = L.prefix_kv + (size_t)depth * d, a.dt = UZU_BF16, a.head_dim = hd, a.gqa_factor = 1, a.sequence_length = depth;
        a.k_head_stride = hd, a.k_seq_stride = d, a.v_head_stride = hd, a.v_seq_stride = d, a.scale = scale, a.num_heads = heads, a.suffix_length = depth, a.is_causal = 1;
        RUN("attention_single_pass", 0, k::attention_single_pass(s, a, w->attn_out));
        post_attention(e, w, L, depth);
    }
    // ---- rounds (encode_step, weaver.rs:354-501)
    uint32_t batch_start_slot = 0;
    for (uint32_t round = 0; round < shape.rounds; ++round) {
        const uint32_t n = round == 0 ? 1u : shape.expand_per_round;
        if (batch_start_slot > 0)
            RUN("weaver_frontier_select", 0, k::weaver_frontier_select(s, w->frontier, w->packed_tree, w->slot_ancestors, w->node_token_ids, w->node_metadata, w->node_ancestor_indices, w->node_valid,
                                                                        w->candidate_ids, w->candidate_logits, w->node_candidate_ids, w->node_candidate_logits, frontier_capacity, slots, n,
                                                                        batch_start_slot, ancestor_stride, D.max_depth, shape.max_depth - 1, shape.dflash_depth - 1, P));
        const uint32_t* batch_candidate_ids = batch_start_slot == 0 ? w->candidate_ids : w->node_candidate_ids;
        const float* batch_candidate_logits = batch_start_slot == 0 ? w->candidate_logits : w->node_candidate_logits;
        // Embedding::encode_lookup of the target (embedding.rs:345-372)
        if (t->embedding.method == UZU_QUANT_NONE)
            RUN("full_precision_embedding_lookup", 0, k::full_precision_embedding_lookup(s, w->node_token_ids, t->embedding.w, w->token_embedding, UZU_BF16, n, t->d.vocab_size, td, t->d.input_scale));
        else
            RUN("quantized_embedding_lookup", 0, k::quantized_embedding_lookup(s, w->node_token_ids, (const uint8_t*)t->embedding.w, t->embedding.scales, t->embedding.zp, t->embedding.biases, w->token_embedding,
                                                                                UZU_BF16, n, t->d.vocab_size, td, t->d.input_scale, t->embedding.group, t->embedding.bits, t->embedding.method));
        if (t->embedding.out_signs)
            RUN("activation_transform", 0, k::activation_transform(s, nullptr, w->token_embedding, nullptr, nullptr, nullptr, t->embedding.out_signs, UZU_BF16, n, td, UZU_ACTIVATION_TRANSFORM_OUTPUT_RHT, 0, 0));
        norm(e, w->embedding_norm, w->token_embedding, w->normed, nullptr, 0, n, td);
        linear(e, w->embedding_projection, w->normed, w->residual_in, n);
        for (size_t l = 0; l < w->layers.size(); ++l) {
            WLayer& L = w->layers[l];
            norm(e, L.pre_attention_norm, w->residual_in, w->normed, w->residual_state, l > 0 ? 2 : 1, n, d);
            linear(e, L.qkv, w->normed, w->qkv, n);
            // the structure-of-arrays metadata is laid out for THIS batch's node count: field f of row r at [f * n + r]
            RUN("ancestor_attention", 0, k::ancestor_attention(s, L.prefix_kv, L.node_kv, w->qkv, w->rope_cos, w->rope_sin, w->node_metadata, w->node_ancestor_indices, w->node_metadata + 1 * n,
                                                                w->node_metadata + 2 * n, w->attn_out, n, shape.dflash_depth, ancestor_stride, slots, D.max_depth, scale, heads, hd));
            post_attention(e, w, L, n);
        }
        norm(e, w->output_norm, w->residual_in, w->normed, w->residual_state, 2, n, d);
        linear(e, w->query_projection, w->normed, w->query, n);
        {   // Embedding::encode_readout_sparse (embedding.rs:458-530): residual logits at the candidates' rows of the read-out table, soft-capped when the read-out has no scale
            const DLinear& src = t->d.tied_embeddings ? t->embedding : t->output_embedding;
            const int32_t* in_signs = t->d.tied_embeddings ? t->embedding.out_signs : t->output_embedding.in_signs;
            const uint16_t* a = w->query;
            if (in_signs) {
                RUN("activation_transform", 0, k::activation_transform(s, w->query, w->rht_in, nullptr, nullptr, nullptr, in_signs, UZU_BF16, n, td, UZU_ACTIVATION_TRANSFORM_INPUT_RHT, 0, 0));
                a = w->rht_in;
            }
            k::MatmulParams p{};
            p.a = a, p.b = src.w, p.scales = src.scales, p.biases = src.biases, p.zero_points = src.zp, p.d = w->logit_residuals, p.w_dt = p.a_dt = p.d_dt = UZU_BF16;
            p.b_kind = src.method == UZU_QUANT_NONE ? UZU_MATMUL_B_FULL_PRECISION
                     : src.method == UZU_QUANT_SCALE_BIAS ? UZU_MATMUL_B_SCALE_BIAS
                     : src.method == UZU_QUANT_SCALE_ZERO_POINT ? UZU_MATMUL_B_SCALE_ZERO_POINT : UZU_MATMUL_B_SCALE_SYMMETRIC;
            p.bits = src.bits, p.group_size = src.group, p.ab_scale = 1.0f, p.gather = batch_candidate_ids, p.m = n, p.n = P, p.k = td;
            if (t->d.logit_scale == 1.0f && t->d.logit_soft_cap != 0.0f) p.has_soft_cap = 1, p.soft_cap = t->d.logit_soft_cap;
            const char* variant = "matmul";
            e.begin();
            const uzu_status r = k::matmul(s, p, t->ctx->num_cus, &variant);
            e.run(r, "matmul_gather", 0);
        }
        RUN("weaver_top_children", 0, k::weaver_top_children(s, w->logit_residuals, batch_candidate_logits, batch_candidate_ids, w->depth_seeds, w->node_metadata, w->child_token_ids, w->child_logprobs, n, P,
                                                              shape.expand_width, vocab));
        RUN("weaver_frontier_insert_children", 0, k::weaver_frontier_insert_children(s, w->packed_tree, w->node_metadata, w->node_valid, w->child_token_ids, w->child_logprobs, w->frontier,
                                                                                      frontier_capacity, slots, n, shape.expand_width));
        batch_start_slot += n;
    }
    w->last_launches = c->launches;
    if (e.st != UZU_OK) return e.st;
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        set_error("weaver: tree launch failed: %s", hipGetErrorString(err));
        return UZU_ERR_HIP;
    }
    return UZU_OK;
}

} // namespace

extern "C" {

void uzu_hip_weaver_destroy(uzu_hip_weaver* w) {
    if (!w) return;
    (void)hipStreamSynchronize(w->ctx->stream);
    for (auto& g : w->graphs) (void)hipGraphExecDestroy(g.exec);
    delete w; // (device memory belongs to the drafter's core model: released with the drafter)
}

// Weaver::new (weaver.rs:166-275).  The object's tensors and scratch are allocated through `drafter` (whose last draft the trees read): destroy it before the drafter.
uzu_status uzu_hip_weaver_create(uzu_hip_context* ctx, uzu_hip_drafter* drafter, const uzu_weaver_desc* desc, uzu_hip_weaver** out) {
    UZU_REQUIRE(ctx && drafter && desc && out && desc->layers, "weaver_create: null argument");
    UZU_REQUIRE(desc->num_layers > 0, "weaver_create: Weaver requires at least one layer");
    UZU_REQUIRE(desc->num_heads > 0 && desc->model_dim % desc->num_heads == 0, "weaver_create: model_dim must be divisible by num_heads");
    UZU_REQUIRE(desc->candidate_pool_size >= 1 && desc->candidate_pool_size <= 512, "weaver_create: candidate_pool_size must be in 1..=512, got %u", desc->candidate_pool_size);
    const uint32_t hd = desc->model_dim / desc->num_heads;
    UZU_REQUIRE(desc->rope.head_dim == hd, "weaver_create: rope head_dim %u does not match model_dim / num_heads = %u", desc->rope.head_dim, hd);
    UZU_REQUIRE(desc->rope.max_sequence_length > desc->max_depth, "weaver_create: rope max_sequence_length %u is too small for max_depth %u", desc->rope.max_sequence_length, desc->max_depth);
    UZU_UNSUPPORTED(hd != 128, "weaver_create: head_dim %u (AncestorAttention is instantiated for HEAD_DIM = 128 only, as in the reference)", hd);
    UZU_UNSUPPORTED(desc->max_depth + 1 > kMaxRows, "weaver_create: max_depth %u (prefixes of at most %u rows)", desc->max_depth, kMaxRows);
    uzu_hip_model *core = nullptr, *target = nullptr;
    uint32_t block = 0;
    UZU_PROPAGATE(uzu_hip_drafter_internal(drafter, &core, &target, nullptr, nullptr, nullptr, &block));
    UZU_REQUIRE(desc->target_model_dim == target->d.model_dim && desc->target_embedding_dim == target->d.model_dim, "weaver_create: target dims %u / %u != the target's model_dim %u",
                desc->target_model_dim, desc->target_embedding_dim, target->d.model_dim);
    UZU_UNSUPPORTED(target->tp != nullptr, "weaver_create: a tensor-parallel target (the candidate pool needs whole logit rows)");
    auto plain = [](const uzu_linear_desc& h) { return !h.input_signs && !h.output_signs && !h.lora_rank; };
    bool all_plain = plain(desc->embedding_projection) && plain(desc->hidden_state_projection) && plain(desc->query_projection);
    for (uint32_t l = 0; l < desc->num_layers; ++l)
        all_plain = all_plain && plain(desc->layers[l].qkv_projection) && plain(desc->layers[l].out_projection) && plain(desc->layers[l].up_projection) && plain(desc->layers[l].down_projection);
    UZU_UNSUPPORTED(!all_plain, "weaver_create: RHT / QLoRA wrappers on the Weaver's own linears");
    (void)hipSetDevice(ctx->device);
    auto* w = new uzu_hip_weaver();
    w->ctx = ctx, w->drafter = drafter, w->core = core, w->target = target, w->d = *desc;
    w->d.layers = nullptr;
    const uint32_t d = desc->model_dim, td = desc->target_model_dim, P = desc->candidate_pool_size;
    uzu_status st = UZU_OK;
    auto fail = [&](uzu_status s_) {
        delete w;
        return s_;
    };
#define TRY(x) do { st = (x); if (st != UZU_OK) return fail(st); } while (0)
    TRY(upload_norm(core, desc->embedding_norm, td, &w->embedding_norm));
    TRY(upload_norm(core, desc->hidden_state_norm, td, &w->hidden_state_norm));
    TRY(upload_norm(core, desc->output_norm, d, &w->output_norm));
    TRY(upload_linear(core, desc->embedding_projection, &w->embedding_projection));
    TRY(upload_linear(core, desc->hidden_state_projection, &w->hidden_state_projection));
    TRY(upload_linear(core, desc->query_projection, &w->query_projection));
    if (!w->embedding_norm.present || !w->hidden_state_norm.present || !w->output_norm.present) return fail((set_error("weaver_create: the three norms are required"), UZU_ERR_INVALID_ARGUMENT));
    UZU_REQUIRE(desc->embedding_projection.n == d && desc->embedding_projection.k == td && desc->hidden_state_projection.n == d && desc->hidden_state_projection.k == td &&
                    desc->query_projection.n == td && desc->query_projection.k == d, "weaver_create: projection shapes inconsistent");
    void* p = nullptr;
    const bool zero = poison_level() < 2;
#define BUF(field, type, elems) do { TRY(dev_alloc(core, (size_t)(elems) * sizeof(type), &p, zero)); w->field = (type*)p; } while (0)
    w->layers.resize(desc->num_layers);
    for (uint32_t l = 0; l < desc->num_layers; ++l) {
        const uzu_weaver_layer_desc& h = desc->layers[l];
        WLayer& L = w->layers[l];
        TRY(upload_norm(core, h.pre_attention_norm, d, &L.pre_attention_norm));
        TRY(upload_norm(core, h.pre_mlp_norm, d, &L.pre_mlp_norm));
        TRY(upload_linear(core, h.qkv_projection, &L.qkv));
        TRY(upload_linear(core, h.out_projection, &L.out));
        TRY(upload_linear(core, h.up_projection, &L.up));
        TRY(upload_linear(core, h.down_projection, &L.down));
        if (h.qkv_projection.n != 3 * d || h.qkv_projection.k != d || h.out_projection.n != d || h.out_projection.k != d || h.up_projection.n != 2 * desc->hidden_dim || h.up_projection.k != d ||
            h.down_projection.n != d || h.down_projection.k != desc->hidden_dim || !L.pre_attention_norm.present || !L.pre_mlp_norm.present)
            return fail((set_error("weaver_create: layer %u shapes inconsistent", l), UZU_ERR_INVALID_ARGUMENT));
        TRY(dev_alloc(core, (size_t)2 * kMaxRows * d * 2, &p, zero)); L.prefix_kv = (uint16_t*)p;
        TRY(dev_alloc(core, (size_t)2 * kMaxSlots * d * 2, &p, zero)); L.node_kv = (uint16_t*)p;
    }
    {
        std::vector<float> cosines, sines;
        rope_tables(desc->rope, desc->max_depth + 1, cosines, sines);
        const size_t saved = core->weight_bytes;
        TRY(upload(core, cosines.data(), cosines.size() * 4, &w->rope_cos));
        TRY(upload(core, sines.data(), sines.size() * 4, &w->rope_sin));
        core->weight_bytes = saved;
    }
    const uint32_t wide = d > td ? d : td;
    BUF(target_hidden, uint16_t, td);
    BUF(prefix_hidden, uint16_t, kMaxRows * td);
    BUF(normed, uint16_t, kMaxRows * wide);
    BUF(residual_in, uint16_t, kMaxRows * d);
    BUF(residual_state, uint16_t, kMaxRows * d);
    BUF(qkv, uint16_t, kMaxRows * 3 * d);
    BUF(queries, uint16_t, kMaxRows * d);
    BUF(attn_out, uint16_t, kMaxRows * d);
    BUF(projected, uint16_t, kMaxRows * d);
    BUF(mlp_in, uint16_t, kMaxRows * d);
    BUF(up_out, uint16_t, kMaxRows * 2 * desc->hidden_dim);
    BUF(gated, uint16_t, kMaxRows * desc->hidden_dim);
    BUF(token_embedding, uint16_t, kMaxRows * td);
    BUF(query, uint16_t, kMaxRows * td);
    BUF(rht_in, uint16_t, kMaxRows * td);
    BUF(logit_residuals, uint16_t, kMaxRows * P);
    BUF(candidate_ids, uint32_t, kMaxRows * P);
    BUF(candidate_logits, float, kMaxRows * P);
    BUF(node_candidate_ids, uint32_t, kMaxRows * P);
    BUF(node_candidate_logits, float, kMaxRows * P);
    BUF(packed_tree, uint32_t, 6 * kMaxSlots);
    BUF(frontier, uint32_t, 7 * kMaxSlots);
    BUF(slot_ancestors, uint32_t, kMaxSlots * desc->max_depth);
    BUF(node_token_ids, uint32_t, kMaxRows);
    BUF(node_metadata, uint32_t, 3 * kMaxRows);
    BUF(node_ancestor_indices, uint32_t, kMaxRows * desc->max_depth);
    BUF(node_valid, uint32_t, kMaxRows);
    BUF(child_token_ids, uint32_t, kMaxRows * P);
    BUF(child_logprobs, float, kMaxRows * P);
    BUF(depth_seeds, uint64_t, desc->max_depth);
    BUF(init_block, uint32_t, 6 * kMaxSlots + kMaxRows * 2);
#undef BUF
#undef TRY
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail((set_error("weaver_create: synchronisation after load failed"), UZU_ERR_HIP));
    (void)block;
    *out = w;
    return UZU_OK;
}

uint32_t uzu_hip_weaver_max_depth(const uzu_hip_weaver* w) { return w ? w->d.max_depth : 0; }

// Weaver::encode_tree (weaver.rs:503-676) over the drafter's LAST draft (its rows must be shape->dflash_depth: uzu_hip_drafter_draft) and `target_hidden_row`, the target's
// output-norm row (host, bf16 [target_model_dim]).  packed_tree_out u32 [6][slots], frontier_out u32 [7][slots * expand_width], slots = 1 + (rounds - 1) * expand_per_round.
// UZU_ERR_INVALID_ARGUMENT with "invalid Weaver tree input" = WeaverEncodeError::InvalidTreeInput.
uzu_status uzu_hip_weaver_encode_tree(uzu_hip_weaver* w, const uint16_t* target_hidden_row, const uint64_t* depth_seeds, uint32_t depth_seed_count, uint32_t root_token_id,
                                      const uzu_weaver_tree_shape* shape, uint32_t* packed_tree_out, uint32_t* frontier_out) {
    UZU_REQUIRE(w && target_hidden_row && depth_seeds && shape && packed_tree_out && frontier_out, "weaver_encode_tree: null argument");
    const uzu_weaver_desc& D = w->d;
    const uint32_t slots = slot_count(*shape);
    if (shape->tree_budget == 0 || shape->rounds == 0 || shape->max_depth < 2 || shape->max_depth > D.max_depth + 1 || shape->dflash_depth < shape->max_depth || shape->dflash_depth > D.max_depth + 1 ||
        shape->expand_per_round == 0 || shape->expand_per_round > 32 || shape->expand_width == 0 || shape->expand_width > D.candidate_pool_size || slots > kMaxSlots / shape->expand_width ||
        depth_seed_count != D.max_depth) {
        set_error("weaver_encode_tree: invalid Weaver tree input");
        return UZU_ERR_INVALID_ARGUMENT;
    }
    uint16_t* draft_hidden = nullptr;
    float* logits = nullptr;
    uint32_t last_rows = 0;
    UZU_PROPAGATE(uzu_hip_drafter_internal(w->drafter, nullptr, nullptr, &draft_hidden, &logits, &last_rows, nullptr));
    UZU_REQUIRE(last_rows == shape->dflash_depth, "weaver_encode_tree: the drafter's last draft has %u rows, the shape's dflash_depth is %u", last_rows, shape->dflash_depth);
    (void)hipSetDevice(w->ctx->device);
    hipStream_t s = w->ctx->stream;
    const uint32_t frontier_capacity = slots * shape->expand_width, rn = shape->expand_per_round;
    // host-side initial state (weaver.rs:566-612) -> one staging block -> the device buffers
    std::vector<uint32_t>& init = w->host_init;
    init.assign((size_t)6 * slots + 2 * rn, 0u);
    for (uint32_t slot = 0; slot < slots; ++slot) init[(size_t)1 * slots + slot] = 0xFFFFFFFFu; // TreeIdx::ParentSlot = FRONTIER_NO_WINNER
    init[0] = root_token_id;                 // TreeIdx::TokenId of slot 0
    init[(size_t)5 * slots] = 1;             // TreeIdx::Valid of slot 0
    init[(size_t)6 * slots] = root_token_id; // node_token_ids[0]
    init[(size_t)6 * slots + rn] = 1;        // node_valid[0]
    HIPCHK(hipEventRecord(w->core->ev0, s));
    HIPCHK(hipMemcpyAsync(w->init_block, init.data(), init.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(w->target_hidden, target_hidden_row, (size_t)D.target_model_dim * 2, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(w->depth_seeds, depth_seeds, (size_t)D.max_depth * 8, hipMemcpyHostToDevice, s));
    hipGraphExec_t exec = nullptr;
    for (auto& g : w->graphs)
        if (!memcmp(&g.shape, shape, sizeof(*shape))) exec = g.exec, w->last_launches = g.launches;
    auto body = [&]() -> uzu_status {
        HIPCHK(hipMemcpyAsync(w->packed_tree, w->init_block, (size_t)6 * slots * 4, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(w->node_token_ids, w->init_block + (size_t)6 * slots, (size_t)rn * 4, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(w->node_valid, w->init_block + (size_t)6 * slots + rn, (size_t)rn * 4, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemsetAsync(w->frontier, 0, (size_t)7 * frontier_capacity * 4, s));
        HIPCHK(hipMemsetAsync(w->slot_ancestors, 0, (size_t)slots * D.max_depth * 4, s));
        HIPCHK(hipMemsetAsync(w->node_metadata, 0, (size_t)3 * rn * 4, s));
        HIPCHK(hipMemsetAsync(w->node_ancestor_indices, 0, (size_t)rn * D.max_depth * 4, s));
        HIPCHK(hipMemsetAsync(w->node_candidate_ids, 0, (size_t)rn * D.candidate_pool_size * 4, s));
        HIPCHK(hipMemsetAsync(w->node_candidate_logits, 0, (size_t)rn * D.candidate_pool_size * 4, s));
        return encode_tree_launches(w, s, *shape, draft_hidden, logits);
    };
    if (k::exact_mode()) { // reference-order kernels take scratch from the stream workspace: eager
        UZU_PROPAGATE(body());
    } else {
        if (!exec) {
            HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            const uzu_status st = body();
            hipGraph_t g = nullptr;
            const hipError_t ce = hipStreamEndCapture(s, &g);
            if (st != UZU_OK || ce != hipSuccess) {
                if (g) (void)hipGraphDestroy(g);
                if (st == UZU_OK) set_error("weaver_encode_tree: graph capture failed: %s", hipGetErrorString(ce));
                return st != UZU_OK ? st : UZU_ERR_HIP;
            }
            HIPCHK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
            HIPCHK(hipGraphDestroy(g));
            w->graphs.push_back({*shape, exec, w->last_launches});
        }
        HIPCHK(hipGraphLaunch(exec, s));
    }
    HIPCHK(hipEventRecord(w->core->ev1, s));
    // EncodedWeaverTree: the two structure-of-arrays buffers, compacted to the shape's strides
    HIPCHK(hipMemcpyAsync(packed_tree_out, w->packed_tree, (size_t)6 * slots * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(frontier_out, w->frontier, (size_t)7 * frontier_capacity * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    (void)hipEventElapsedTime(&w->last_ms, w->core->ev0, w->core->ev1);
    return UZU_OK;
}

// device time (ms) and launches of the last tree
uzu_status uzu_hip_weaver_stats(uzu_hip_weaver* w, float* gpu_ms, uint32_t* launches) {
    UZU_REQUIRE(w, "weaver_stats: null weaver");
    if (gpu_ms) *gpu_ms = w->last_ms;
    if (launches) *launches = w->last_launches;
    return UZU_OK;
}

} // extern "C"
