/*
 * uzu_oracle_weaver.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT).  See uzu_oracle.h.
 *
 * The Weaver tree constructor of the reference's speculator, restated from
 *   Weaver::{new, encode_prefix, encode_step, encode_tree}    BU/src/encodable_block/weaver.rs:166-676
 *   WeaverLayer::{encode_prefix_attention, encode_post_attention}   BU/src/encodable_block/weaver_layer.rs:150-200
 *   CpuRadixTopKSmall::encode                                 BU/src/backends/cpu/kernel/radix_top_k_small.rs:25-79   (top-k of a row: value descending by total_cmp, ties to the lower index)
 *   Embedding::encode_readout_sparse                          BU/src/encodable_block/embedding.rs:458-530           (the read-out at gathered rows)
 *   EncodedWeaverTree::read_nodes                             BU/src/encodable_block/weaver.rs:60-113               (host side: uzu_amd/speculator.py)
 * on top of the kernels of uzu_oracle_speculator.c (AncestorAttention, WeaverFrontierSelect, WeaverFrontierInsertChildren, WeaverTopChildren) and
 * uzu_oracle_kernels.c.
 *
 * PARITY STATUS: **parity unpinned** (the reference holds no vectors for the block); tests/test_oracle_weaver.py pins the structure (tree invariants, the
 * candidate pool against a NumPy sort, round 0 against a hand-composed chain of the kernels).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "uzu_oracle_model_internal.h"

enum { FR_COUNT = 7, TR_TOKEN = 0, TR_PARENT = 1, TR_VALID = 5, TR_COUNT = 6, MD_ANCESTOR_COUNT = 1, MD_TREE_SLOT = 2, MD_COUNT = 3 };
#define FRONTIER_NO_WINNER 0xFFFFFFFFu
#define FRONTIER_MAX_SLOTS 2048u
#define FRONTIER_MAX_WIDTH 32u
#define CANDIDATES_MAX 512u

/* radix_top_k_small.rs:56-77: per row the k best columns, ordered by (value descending under f32::total_cmp, column ascending) */
static int total_cmp(float a, float b) { /* f32::total_cmp: the IEEE total order on the bit patterns */
    int32_t x, y;
    memcpy(&x, &a, 4), memcpy(&y, &b, 4);
    x ^= (int32_t)(((uint32_t)(x >> 31)) >> 1);
    y ^= (int32_t)(((uint32_t)(y >> 31)) >> 1);
    return (x > y) - (x < y);
}
void orc_radix_top_k_small(const float* input, uint32_t* output_ids, float* output_scores, uint32_t rows, uint32_t columns, uint32_t k) {
    if (!rows || !k || k > 512 || k > columns) {
        fprintf(stderr, "oracle: radix_top_k_small needs 1 <= k <= min(512, columns)\n");
        abort();
    }
    uint32_t* best = (uint32_t*)orc_xcalloc(k, 4);
    for (size_t row = 0; row < rows; ++row) {
        const float* v = input + row * columns;
        uint32_t n = 0;
        for (uint32_t c = 0; c < columns; ++c) { /* insertion into the sorted winners: the same total order as select_nth + sort_unstable_by(compare) */
            uint32_t pos = n;
            while (pos > 0) {
                const uint32_t o = best[pos - 1];
                const int cmp = total_cmp(v[c], v[o]); /* c before o iff value greater (or equal and c < o: never, c is the larger index) */
                if (cmp > 0) --pos;
                else break;
            }
            if (pos >= k) continue;
            const uint32_t last = n < k ? n : k - 1;
            for (uint32_t s = last; s > pos; --s) best[s] = best[s - 1];
            best[pos] = c;
            if (n < k) ++n;
        }
        for (uint32_t r = 0; r < k; ++r) output_ids[row * k + r] = best[r], output_scores[row * k + r] = v[best[r]];
    }
    free(best);
}

struct orc_weaver {
    uzu_weaver_desc desc;
    uzu_weaver_layer_desc* layers;
};

orc_weaver* orc_weaver_create(const uzu_weaver_desc* desc) {
    /* Weaver::new's checks (weaver.rs:172-199) */
    if (!desc->num_layers || !desc->num_heads || desc->model_dim % desc->num_heads || !desc->candidate_pool_size || desc->candidate_pool_size > CANDIDATES_MAX ||
        desc->rope.head_dim != desc->model_dim / desc->num_heads || desc->rope.max_sequence_length <= desc->max_depth) {
        fprintf(stderr, "oracle: Weaver description refused (layers, heads, candidate pool 1..512, rope head_dim = model_dim / heads, rope length > max_depth)\n");
        abort();
    }
    orc_weaver* w = (orc_weaver*)orc_xcalloc(1, sizeof(orc_weaver));
    w->desc = *desc;
    w->layers = (uzu_weaver_layer_desc*)orc_xcalloc(desc->num_layers, sizeof(uzu_weaver_layer_desc));
    memcpy(w->layers, desc->layers, sizeof(uzu_weaver_layer_desc) * desc->num_layers);
    w->desc.layers = w->layers;
    return w;
}
void orc_weaver_destroy(orc_weaver* w) {
    if (!w) return;
    free(w->layers);
    free(w);
}

/* DenseMlp with SiLU and up / down biases (weaver_layer.rs:126-135; mlp/dense.rs:32-48) */
static uint16_t* weaver_mlp(const orc_weaver* w, const uzu_weaver_layer_desc* L, uint16_t* input, uint32_t rows) {
    uint16_t* fused_up = orc_linear(&L->up_projection, input, rows);
    uint16_t* gated = (uint16_t*)orc_xcalloc((size_t)rows * w->desc.hidden_dim, 2);
    orc_gated_act_mul(fused_up, NULL, gated, ORC_BF16, w->desc.hidden_dim, rows, 0, 0, UZU_ACT_SILU, 1);
    free(fused_up);
    uint16_t* down = orc_linear(&L->down_projection, gated, rows);
    free(gated);
    return down;
}
/* WeaverLayer::encode_post_attention (weaver_layer.rs:187-199): out projection -> pre-MLP norm (adds into the residual) -> MLP */
static uint16_t* post_attention(const orc_weaver* w, const uzu_weaver_layer_desc* L, uint16_t* attention_output, uint16_t* residual_state, uint32_t rows) {
    uint16_t* projected = orc_linear(&L->out_projection, attention_output, rows);
    uint16_t* mlp_input = orc_norm(&L->pre_mlp_norm, projected, residual_state, 2, rows, w->desc.model_dim);
    free(projected);
    uint16_t* out = weaver_mlp(w, L, mlp_input, rows);
    free(mlp_input);
    return out;
}

/* Weaver::encode_tree (weaver.rs:503-676).  target_hidden bf16 [>= 1 row, target_model_dim] (row 0 is used), draft_hidden bf16 [dflash_depth, target_model_dim]
 * (rows 1.. are used), draft logits f32 [dflash_depth - 1, vocab], depth_seeds [max_depth].  Outputs: packed_tree u32 [TreeIdx::COUNT, slot_count], frontier u32
 * [FrontierIdx::COUNT, slot_count * expand_width] (the caller sizes them from the shape).  Returns 0, or 1 for WeaverEncodeError::InvalidTreeInput. */
int orc_weaver_encode_tree(const orc_weaver* w, const orc_model* target, const uint16_t* target_hidden, const uint16_t* draft_hidden, const float* logits, const uint64_t* depth_seeds,
                           uint32_t depth_seed_count, uint32_t root_token_id, const uzu_weaver_tree_shape* shape, uint32_t* packed_tree, uint32_t* frontier) {
    const uzu_weaver_desc* D = &w->desc;
    const uzu_model_desc* T = orc_model_desc(target);
    const uint32_t d = D->model_dim, heads = D->num_heads, hd = d / heads, P = D->candidate_pool_size;
    const uint32_t tree_slot_count = 1 + (shape->rounds ? shape->rounds - 1 : 0) * shape->expand_per_round, ancestor_stride = D->max_depth;
    if (shape->tree_budget == 0 || shape->rounds == 0 || shape->max_depth < 2 || shape->max_depth > D->max_depth + 1 || shape->dflash_depth < shape->max_depth ||
        shape->dflash_depth > D->max_depth + 1 || shape->expand_per_round == 0 || shape->expand_per_round > FRONTIER_MAX_WIDTH || shape->expand_width == 0 ||
        shape->expand_width > P || tree_slot_count > FRONTIER_MAX_SLOTS / shape->expand_width || depth_seed_count != D->max_depth)
        return 1;
    if (T->model_dim != D->target_model_dim || T->model_dim != D->target_embedding_dim) {
        fprintf(stderr, "oracle: the Weaver's target dims do not match the target model\n");
        abort();
    }
    const uint32_t frontier_capacity = tree_slot_count * shape->expand_width, pool_depth_count = shape->dflash_depth - 1, vocab = T->vocab_size;
    /* the candidate pool: top candidate_pool_size tokens of every lookahead row */
    uint32_t* candidate_ids = (uint32_t*)orc_xcalloc((size_t)pool_depth_count * P, 4);
    float* candidate_logits = (float*)orc_xcalloc((size_t)pool_depth_count * P, 4);
    orc_radix_top_k_small(logits, candidate_ids, candidate_logits, pool_depth_count, vocab, P);
    /* RoPE tables of positions 0 ..= max_depth */
    const uint32_t n_pos = D->max_depth + 1;
    uint32_t* positions = (uint32_t*)orc_xcalloc(n_pos, 4);
    for (uint32_t i = 0; i < n_pos; ++i) positions[i] = i;
    float* cosines = (float*)orc_xcalloc((size_t)n_pos * hd, 4);
    float* sines = (float*)orc_xcalloc((size_t)n_pos * hd, 4);
    orc_rope_tables(&D->rope, positions, n_pos, cosines, sines);
    free(positions);

    /* ---- encode_prefix (weaver.rs:283-352): row 0 = the target's output norm, rows 1.. = the draft rows */
    const uint32_t depth = shape->dflash_depth, td = D->target_model_dim;
    uint16_t* prefix_hidden = (uint16_t*)orc_xcalloc((size_t)depth * td, 2);
    memcpy(prefix_hidden, target_hidden, (size_t)td * 2);
    memcpy(prefix_hidden + td, draft_hidden + td, (size_t)(depth - 1) * td * 2);
    uint16_t* normalized_prefix = orc_norm(&D->hidden_state_norm, prefix_hidden, NULL, 0, depth, td);
    free(prefix_hidden);
    uint16_t* residual_input = orc_linear(&D->hidden_state_projection, normalized_prefix, depth);
    free(normalized_prefix);
    uint16_t* residual_state = (uint16_t*)orc_xcalloc((size_t)depth * d, 2);
    uint16_t** prefix_kv = (uint16_t**)orc_xcalloc(D->num_layers, sizeof(uint16_t*));
    const float scale = 1.0f / sqrtf((float)hd);
    for (uint32_t l = 0; l < D->num_layers; ++l) {
        const uzu_weaver_layer_desc* L = &w->layers[l];
        /* encode_prefix_attention (weaver_layer.rs:150-185): norm (layer 0 copies into the residual, later layers add), qkv, AttentionPrepare at offset 0 */
        uint16_t* attention_input = orc_norm(&L->pre_attention_norm, residual_input, residual_state, l > 0 ? 2 : 1, depth, d);
        uint16_t* qkv = orc_linear(&L->qkv_projection, attention_input, depth);
        free(attention_input);
        uint16_t* queries = (uint16_t*)orc_xcalloc((size_t)heads * depth * hd, 2);
        prefix_kv[l] = (uint16_t*)orc_xcalloc((size_t)2 * depth * d, 2); /* keys [depth, d] then values */
        orc_attention_prepare(qkv, queries, prefix_kv[l], prefix_kv[l] + (size_t)depth * d, cosines, sines, heads, heads, hd, hd, 0, depth, 1);
        free(qkv);
        if (l + 1 == D->num_layers) { /* the last layer only contributes its keys / values (weaver.rs:343-348) */
            free(queries);
            break;
        }
        orc_attention_args a;
        memset(&a, 0, sizeof(a));
        a.queries = queries, a.keys = prefix_kv[l], a.values = prefix_kv[l] + (size_t)depth * d, a.dtype = ORC_BF16, a.head_dim = hd, a.gqa_factor = 1;
        a.sequence_length = depth, a.k_head_stride = hd, a.k_seq_stride = d, a.v_head_stride = hd, a.v_seq_stride = d, a.scale = scale, a.num_heads = heads, a.suffix_length = depth;
        a.is_causal = 1;
        uint16_t* attention_output = (uint16_t*)orc_xcalloc((size_t)depth * d, 2);
        orc_attention_single_pass(&a, attention_output); /* suffix <= 17, no prefix: the single-pass core (core/mod.rs:81-93 on the CPU backend) */
        free(queries);
        free(residual_input);
        residual_input = post_attention(w, L, attention_output, residual_state, depth);
        free(attention_output);
    }
    free(residual_input);
    free(residual_state);

    /* ---- tree state (weaver.rs:566-612) */
    const size_t ts = tree_slot_count, rn = shape->expand_per_round;
    uint16_t** node_kv = (uint16_t**)orc_xcalloc(D->num_layers, sizeof(uint16_t*));
    for (uint32_t l = 0; l < D->num_layers; ++l) node_kv[l] = (uint16_t*)orc_xcalloc((size_t)2 * ts * d, 2);
    memset(packed_tree, 0, (size_t)TR_COUNT * ts * 4);
    for (size_t slot = 0; slot < ts; ++slot) packed_tree[TR_PARENT * ts + slot] = FRONTIER_NO_WINNER;
    packed_tree[TR_TOKEN * ts] = root_token_id;
    packed_tree[TR_VALID * ts] = 1;
    memset(frontier, 0, (size_t)FR_COUNT * frontier_capacity * 4);
    uint32_t* slot_ancestors = (uint32_t*)orc_xcalloc(ts * ancestor_stride, 4);
    uint32_t* node_token_ids = (uint32_t*)orc_xcalloc(rn, 4);
    uint32_t* node_valid = (uint32_t*)orc_xcalloc(rn, 4);
    node_token_ids[0] = root_token_id, node_valid[0] = 1;
    uint32_t* node_metadata = (uint32_t*)orc_xcalloc((size_t)MD_COUNT * rn, 4);
    uint32_t* node_ancestor_indices = (uint32_t*)orc_xcalloc(rn * ancestor_stride, 4);
    uint32_t* node_candidate_ids = (uint32_t*)orc_xcalloc(rn * P, 4);
    float* node_candidate_logits = (float*)orc_xcalloc(rn * P, 4);

    uint32_t batch_start_slot = 0;
    for (uint32_t round = 0; round < shape->rounds; ++round) {
        const uint32_t n = round == 0 ? 1 : shape->expand_per_round;
        /* ---- encode_step (weaver.rs:354-501).  NB the structure-of-arrays metadata is laid out for THIS batch's node count (field f at [f * n + row]) */
        if (batch_start_slot > 0)
            orc_weaver_frontier_select(frontier, packed_tree, slot_ancestors, node_token_ids, node_metadata, node_ancestor_indices, node_valid, candidate_ids, candidate_logits,
                                       node_candidate_ids, node_candidate_logits, frontier_capacity, tree_slot_count, n, batch_start_slot, ancestor_stride, D->max_depth,
                                       shape->max_depth - 1, shape->dflash_depth - 1, P);
        const uint32_t* batch_candidate_ids = batch_start_slot == 0 ? candidate_ids : node_candidate_ids;
        const float* batch_candidate_logits = batch_start_slot == 0 ? candidate_logits : node_candidate_logits;
        uint16_t* token_embedding = (uint16_t*)orc_xcalloc((size_t)n * td, 2);
        if (T->embedding.method == UZU_QUANT_NONE)
            orc_full_precision_embedding_lookup(node_token_ids, T->embedding.weights, token_embedding, ORC_BF16, n, vocab, td, T->input_scale);
        else
            orc_quantized_embedding_lookup(node_token_ids, (const uint8_t*)T->embedding.weights, T->embedding.scales, T->embedding.zero_points, T->embedding.biases, token_embedding, ORC_BF16, n,
                                           vocab, td, T->input_scale, T->embedding.group_size, T->embedding.bits, T->embedding.method);
        if (T->embedding.output_signs) orc_activation_transform(NULL, token_embedding, NULL, NULL, NULL, T->embedding.output_signs, ORC_BF16, n, td, 1, 0, 0);
        uint16_t* normalized_embedding = orc_norm(&D->embedding_norm, token_embedding, NULL, 0, n, td);
        free(token_embedding);
        uint16_t* rin = orc_linear(&D->embedding_projection, normalized_embedding, n);
        free(normalized_embedding);
        uint16_t* rstate = (uint16_t*)orc_xcalloc((size_t)n * d, 2);
        for (uint32_t l = 0; l < D->num_layers; ++l) {
            const uzu_weaver_layer_desc* L = &w->layers[l];
            uint16_t* attention_input = orc_norm(&L->pre_attention_norm, rin, rstate, l > 0 ? 2 : 1, n, d);
            uint16_t* current_qkv = orc_linear(&L->qkv_projection, attention_input, n);
            free(attention_input);
            uint16_t* attention_output = (uint16_t*)orc_xcalloc((size_t)n * d, 2);
            orc_ancestor_attention(prefix_kv[l], node_kv[l], current_qkv, cosines, sines, node_metadata, node_ancestor_indices, node_metadata + (size_t)MD_ANCESTOR_COUNT * n,
                                   node_metadata + (size_t)MD_TREE_SLOT * n, attention_output, n, shape->dflash_depth, ancestor_stride, tree_slot_count, D->max_depth, scale, heads, hd);
            free(current_qkv);
            free(rin);
            rin = post_attention(w, L, attention_output, rstate, n);
            free(attention_output);
        }
        uint16_t* normalized_output = orc_norm(&D->output_norm, rin, rstate, 2, n, d);
        free(rin);
        free(rstate);
        uint16_t* query = orc_linear(&D->query_projection, normalized_output, n);
        free(normalized_output);
        /* Embedding::encode_readout_sparse (embedding.rs:458-530): logit_residuals[r][j] = query[r] . table[candidate_ids[r][j]], soft-capped when the read-out has no scale */
        uzu_linear_desc ro = T->tied_embeddings ? T->embedding : T->output_embedding;
        const int32_t* in_signs = T->tied_embeddings ? T->embedding.output_signs : T->output_embedding.input_signs;
        uint16_t* transformed = NULL;
        const uint16_t* a_rows = query;
        if (in_signs) {
            transformed = (uint16_t*)orc_xcalloc((size_t)n * td, 2);
            orc_activation_transform(query, transformed, NULL, NULL, NULL, in_signs, ORC_BF16, n, td, 0, 0, 0);
            a_rows = transformed;
        }
        uint16_t* logit_residuals = (uint16_t*)orc_xcalloc((size_t)n * P, 2);
        orc_matmul_args g;
        memset(&g, 0, sizeof(g));
        g.a = a_rows, g.a_dtype = ORC_BF16, g.b = ro.weights, g.scales = ro.scales, g.biases = ro.biases, g.zero_points = ro.zero_points, g.w_dtype = ORC_BF16, g.method = ro.method;
        g.bits = ro.bits, g.group_size = ro.group_size, g.b_transpose = 1, g.d = logit_residuals, g.d_dtype = ORC_BF16, g.ab_scale = 1.0f, g.gather_indices = batch_candidate_ids;
        g.m = n, g.n = P, g.k = td;
        if (T->logit_scale == 1.0f && T->logit_soft_cap != 0.0f) g.has_soft_cap = 1, g.soft_cap = T->logit_soft_cap;
        orc_matmul(&g);
        free(transformed);
        free(query);
        uint32_t* child_token_ids = (uint32_t*)orc_xcalloc((size_t)n * shape->expand_width, 4);
        float* child_logprobs = (float*)orc_xcalloc((size_t)n * shape->expand_width, 4);
        orc_weaver_top_children(logit_residuals, batch_candidate_logits, batch_candidate_ids, depth_seeds, node_metadata, child_token_ids, child_logprobs, n, P, shape->expand_width, vocab);
        free(logit_residuals);
        orc_weaver_frontier_insert_children(packed_tree, node_metadata, node_valid, child_token_ids, child_logprobs, frontier, frontier_capacity, tree_slot_count, n, shape->expand_width);
        free(child_token_ids);
        free(child_logprobs);
        batch_start_slot += n;
    }
    for (uint32_t l = 0; l < D->num_layers; ++l) free(prefix_kv[l]), free(node_kv[l]);
    free(prefix_kv), free(node_kv), free(candidate_ids), free(candidate_logits), free(cosines), free(sines), free(slot_ancestors), free(node_token_ids), free(node_valid);
    free(node_metadata), free(node_ancestor_indices), free(node_candidate_ids), free(node_candidate_logits);
    return 0;
}
