/*
 * uzu_oracle_speculator.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT).  See uzu_oracle.h.
 *
 * The kernels the reference's tree speculators run next to the forward path (BU = crates/backend-uzu/src):
 *   AncestorAttention             BU/backends/cpu/kernel/attention/ancestor_attention.rs:8-139
 *   WeaverFrontierSelect          BU/backends/cpu/kernel/weaver/weaver_frontier_select.rs:7-144
 *   WeaverFrontierInsertChildren  BU/backends/cpu/kernel/weaver/weaver_frontier_insert_children.rs:16-65
 *   WeaverTopChildren             BU/backends/cpu/kernel/weaver/weaver_top_children.rs:9-53
 * with the structure-of-arrays layouts of BU/backends/common/gpu_types/weaver.rs (FrontierIdx, TreeIdx, MetadataIdx: field f of slot s
 * at [f * capacity + s]).  Same loops, same order; the float work is the attention kernel of uzu_oracle_kernels.c and two libm calls.
 * The drafter MODELS that drive these kernels (DFlash / Weaver) are out of scope; the kernels are what a hip backend has to provide.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "uzu_oracle.h"

enum { FR_TOKEN = 0, FR_PARENT, FR_DEPTH, FR_PATH_LOGPROB, FR_EDGE_LOGPROB, FR_SCORE_KEY, FR_ACTIVE, FR_COUNT };
enum { TR_TOKEN = 0, TR_PARENT, TR_DEPTH, TR_PATH_LOGPROB, TR_EDGE_LOGPROB, TR_VALID, TR_COUNT };
enum { MD_DEPTH = 0, MD_ANCESTOR_COUNT, MD_TREE_SLOT, MD_COUNT };
#define FRONTIER_NO_WINNER 0xFFFFFFFFu
#define FRONTIER_MAX_SLOTS 2048u
#define FRONTIER_MAX_WIDTH 32u
#define PADDING_DEPTH 0u
#define CANDIDATES_MAX 512u

/* ancestor_attention.rs:8-139: per row (a tree node of the drafter): half-rotation RoPE of q and k at position depth + 1, keys / values =
 * the prefix rows, the node's ancestors' rows (node_kv slots), the node's own row; non-causal single-pass attention over them for every
 * head; the node's rotated key and its value are then written to its node_kv slot.  prefix_kv: keys [prefix, model_dim] then values;
 * node_kv: keys [capacity, model_dim] then values; current_qkv [rows, 3 model_dim]; cosines / sines [max_depth + 1, head_dim]. */
void orc_ancestor_attention(const uint16_t* prefix_kv, uint16_t* node_kv, const uint16_t* current_qkv, const float* cosines, const float* sines,
                            const uint32_t* node_metadata, const uint32_t* ancestor_indices, const uint32_t* ancestor_counts, const uint32_t* node_indices,
                            uint16_t* output, uint32_t rows, uint32_t prefix_length, uint32_t ancestor_stride, uint32_t node_capacity, uint32_t max_depth,
                            float scale, uint32_t num_heads, uint32_t head_dim) {
    const size_t model_dim = (size_t)num_heads * head_dim, qkv_width = 3 * model_dim, half_dim = head_dim / 2;
    uint16_t* rq = (uint16_t*)malloc(model_dim * 2);
    uint16_t* rk = (uint16_t*)malloc(model_dim * 2);
    for (size_t row = 0; row < rows; ++row) {
        const uint16_t* cur = current_qkv + row * qkv_width;
        const uint32_t depth = node_metadata[(size_t)MD_DEPTH * rows + row];
        if (depth >= max_depth) abort(); /* "node metadata depth must be rope-safe" */
        const size_t position = (size_t)depth + 1;
        for (int component = 0; component < 2; ++component) {
            uint16_t* dst = component ? rk : rq;
            for (size_t head = 0; head < num_heads; ++head) {
                const size_t base = (size_t)component * model_dim + head * head_dim;
                for (size_t pair = 0; pair < half_dim; ++pair) {
                    const float low = orc_bf16_to_f32(cur[base + pair]), high = orc_bf16_to_f32(cur[base + half_dim + pair]);
                    const size_t index = position * head_dim + pair;
                    dst[head * head_dim + pair] = orc_f32_to_bf16(low * cosines[index] - high * sines[index]);
                    dst[head * head_dim + half_dim + pair] = orc_f32_to_bf16(high * cosines[index + half_dim] + low * sines[index + half_dim]);
                }
            }
        }
        const size_t ancestor_count = ancestor_counts[row], length = prefix_length + ancestor_count + 1;
        uint16_t* keys = (uint16_t*)calloc(length * model_dim, 2);
        uint16_t* values = (uint16_t*)calloc(length * model_dim, 2);
        memcpy(keys, prefix_kv, (size_t)prefix_length * model_dim * 2);
        memcpy(values, prefix_kv + (size_t)prefix_length * model_dim, (size_t)prefix_length * model_dim * 2);
        for (size_t offset = 0; offset < ancestor_count; ++offset) {
            const size_t ancestor = ancestor_indices[row * ancestor_stride + offset];
            if (ancestor >= node_capacity) abort();
            memcpy(keys + (prefix_length + offset) * model_dim, node_kv + ancestor * model_dim, model_dim * 2);
            memcpy(values + (prefix_length + offset) * model_dim, node_kv + (size_t)node_capacity * model_dim + ancestor * model_dim, model_dim * 2);
        }
        memcpy(keys + (length - 1) * model_dim, rk, model_dim * 2);
        memcpy(values + (length - 1) * model_dim, cur + 2 * model_dim, model_dim * 2);
        orc_attention_args a;
        memset(&a, 0, sizeof(a));
        a.queries = rq, a.keys = keys, a.values = values, a.dtype = ORC_BF16;
        a.head_dim = head_dim, a.gqa_factor = 1, a.sequence_length = (uint32_t)length;
        a.k_head_stride = head_dim, a.k_seq_stride = (uint32_t)model_dim, a.v_head_stride = head_dim, a.v_seq_stride = (uint32_t)model_dim;
        a.scale = scale, a.num_heads = num_heads, a.suffix_length = 1, a.is_causal = 0;
        orc_attention_single_pass(&a, output + row * model_dim);
        if (node_capacity > 0) {
            const size_t node = node_indices[row];
            if (node >= node_capacity) abort();
            memcpy(node_kv + node * model_dim, rk, model_dim * 2);
            memcpy(node_kv + (size_t)node_capacity * model_dim + node * model_dim, cur + 2 * model_dim, model_dim * 2);
        }
        free(keys);
        free(values);
    }
    free(rq);
    free(rk);
}

/* weaver_frontier_select.rs:7-144: node by node, the best active frontier slot -- expandable ones first, then by path score key, ties by
 * lower parent slot then lower token id -- becomes tree slot batch_start_slot + node; its ancestors are its parent's ancestors + the parent. */
void orc_weaver_frontier_select(uint32_t* frontier, uint32_t* packed_tree, uint32_t* slot_ancestors, uint32_t* node_token_ids, uint32_t* node_metadata,
                                uint32_t* node_ancestor_indices, uint32_t* node_valid, const uint32_t* candidate_pool_ids, const float* candidate_pool_logits,
                                uint32_t* node_candidate_ids, float* node_candidate_logits, uint32_t frontier_capacity_, uint32_t tree_slot_count_,
                                uint32_t node_count_, uint32_t batch_start_slot, uint32_t ancestor_stride_, uint32_t max_depth, uint32_t lookahead_count,
                                uint32_t candidate_depth_count_, uint32_t candidates_per_depth_) {
    if (frontier_capacity_ == 0 || frontier_capacity_ > FRONTIER_MAX_SLOTS || node_count_ == 0 || node_count_ > FRONTIER_MAX_WIDTH || ancestor_stride_ == 0 ||
        max_depth == 0 || tree_slot_count_ == 0 || batch_start_slot + node_count_ > tree_slot_count_ || candidate_depth_count_ == 0 || candidates_per_depth_ == 0)
        return;
    const size_t fc = frontier_capacity_, ts = tree_slot_count_, nc = node_count_, as = ancestor_stride_;
    const size_t cdc = candidate_depth_count_, cpd = candidates_per_depth_;
    for (size_t node = 0; node < nc; ++node) {
        uint32_t key = 0, parent = FRONTIER_NO_WINNER, token = FRONTIER_NO_WINNER, winner = FRONTIER_NO_WINNER;
        for (size_t slot = 0; slot < fc; ++slot) {
            if (frontier[FR_ACTIVE * fc + slot] == 0) continue;
            const uint32_t score_key = frontier[FR_SCORE_KEY * fc + slot];
            const uint32_t expandable = frontier[FR_DEPTH * fc + slot] < lookahead_count ? 1u : 0u;
            const uint32_t n0 = (expandable << 31) | (score_key >> 1), n1 = frontier[FR_PARENT * fc + slot], n2 = frontier[FR_TOKEN * fc + slot];
            if (n0 > key || (n0 == key && (n1 < parent || (n1 == parent && n2 < token)))) key = n0, parent = n1, token = n2, winner = (uint32_t)slot;
        }
        const int real = winner != FRONTIER_NO_WINNER;
        const size_t w = real ? winner : 0;
        const size_t tree_slot = (size_t)batch_start_slot + node;
#define FIELD(f) ((real ? 1u : 0u) * frontier[(f) * fc + w])
        const uint32_t tok = FIELD(FR_TOKEN), depth = FIELD(FR_DEPTH), cumulative = FIELD(FR_PATH_LOGPROB), logprob = FIELD(FR_EDGE_LOGPROB);
#undef FIELD
        packed_tree[TR_TOKEN * ts + tree_slot] = tok;
        packed_tree[TR_PARENT * ts + tree_slot] = real ? parent : FRONTIER_NO_WINNER;
        packed_tree[TR_DEPTH * ts + tree_slot] = depth;
        packed_tree[TR_PATH_LOGPROB * ts + tree_slot] = cumulative;
        packed_tree[TR_EDGE_LOGPROB * ts + tree_slot] = logprob;
        packed_tree[TR_VALID * ts + tree_slot] = real ? 1u : 0u;
        if (real) frontier[FR_ACTIVE * fc + w] = 0;
        const size_t parent_slot = (real && (size_t)parent < ts) ? parent : 0;
        for (size_t index = 0; index < as; ++index) {
            uint32_t ancestor;
            if (real && index + 1 < (size_t)depth) ancestor = slot_ancestors[parent_slot * as + index];
            else if (real && index + 1 == (size_t)depth) ancestor = (uint32_t)parent_slot;
            else ancestor = 0;
            slot_ancestors[tree_slot * as + index] = ancestor;
            node_ancestor_indices[node * as + index] = ancestor;
        }
        node_token_ids[node] = tok;
        const int expandable = depth < lookahead_count;
        node_metadata[MD_DEPTH * nc + node] = expandable ? depth : PADDING_DEPTH;
        node_metadata[MD_ANCESTOR_COUNT * nc + node] = depth;
        node_metadata[MD_TREE_SLOT * nc + node] = (uint32_t)tree_slot;
        node_valid[node] = (real && expandable) ? 1u : 0u;
        if ((size_t)depth < cdc) {
            memcpy(node_candidate_ids + node * cpd, candidate_pool_ids + (size_t)depth * cpd, cpd * 4);
            memcpy(node_candidate_logits + node * cpd, candidate_pool_logits + (size_t)depth * cpd, cpd * 4);
        }
    }
}

static uint32_t top_k_score_key(float score) {
    uint32_t bits;
    memcpy(&bits, &score, 4);
    return (bits & 0x80000000u) == 0 ? (bits ^ 0x80000000u) : ~bits;
}

/* weaver_frontier_insert_children.rs:16-65: the expand_width children of every valid node enter the frontier at parent * expand_width + e */
void orc_weaver_frontier_insert_children(const uint32_t* packed_tree, const uint32_t* node_metadata, const uint32_t* node_valid, const uint32_t* child_ids,
                                         const float* child_logprobs, uint32_t* frontier, uint32_t frontier_capacity_, uint32_t tree_slot_count_,
                                         uint32_t node_count_, uint32_t expand_width_) {
    if (frontier_capacity_ == 0 || tree_slot_count_ == 0 || expand_width_ == 0) return;
    const size_t fc = frontier_capacity_, ts = tree_slot_count_, nc = node_count_, ew = expand_width_;
    const uint32_t* parent_indices = node_metadata + MD_TREE_SLOT * nc;
    for (size_t index = 0; index < nc * ew; ++index) {
        const size_t row = index / ew;
        if (node_valid[row] == 0) continue;
        const size_t parent = parent_indices[row], slot = parent * ew + index % ew;
        if (parent >= ts || slot >= fc) continue;
        const float logprob = child_logprobs[index];
        float parent_path;
        memcpy(&parent_path, &packed_tree[TR_PATH_LOGPROB * ts + parent], 4);
        const float cumulative = parent_path + logprob;
        uint32_t cum_bits, lp_bits;
        memcpy(&cum_bits, &cumulative, 4);
        memcpy(&lp_bits, &logprob, 4);
        frontier[FR_TOKEN * fc + slot] = child_ids[index];
        frontier[FR_PARENT * fc + slot] = (uint32_t)parent;
        frontier[FR_DEPTH * fc + slot] = packed_tree[TR_DEPTH * ts + parent] + 1;
        frontier[FR_PATH_LOGPROB * fc + slot] = cum_bits;
        frontier[FR_EDGE_LOGPROB * fc + slot] = lp_bits;
        frontier[FR_SCORE_KEY * fc + slot] = top_k_score_key(cumulative);
        frontier[FR_ACTIVE * fc + slot] = 1;
    }
}

typedef struct {
    float perturbed;
    uint32_t token;
    uint32_t index;
} top_child;
static int32_t total_key(float v) { /* f32::total_cmp (core::f32): compare these as signed integers */
    int32_t bits;
    memcpy(&bits, &v, 4);
    return bits ^ (int32_t)((uint32_t)(bits >> 31) >> 1);
}
static int cmp_top_child(const void* pa, const void* pb) { /* perturbed descending, then token ascending; stable via index */
    const top_child *a = (const top_child*)pa, *b = (const top_child*)pb;
    const int32_t ka = total_key(a->perturbed), kb = total_key(b->perturbed);
    if (ka != kb) return kb > ka ? 1 : -1;
    if (a->token != b->token) return a->token < b->token ? -1 : 1;
    return a->index < b->index ? -1 : (a->index > b->index ? 1 : 0);
}

/* weaver_top_children.rs:9-53: per node, logits = candidate + residual; the expand_width best by Gumbel-perturbed logit (the node's
 * depth picks the seed), reported with their log-softmax under the unperturbed logits */
void orc_weaver_top_children(const uint16_t* residual_logits, const float* candidate_logits, const uint32_t* candidate_ids, const uint64_t* depth_seeds,
                             const uint32_t* node_metadata, uint32_t* output_token_ids, float* output_model_logprobs, uint32_t rows_, uint32_t candidates_,
                             uint32_t expand_width_, uint32_t vocab_size) {
    const size_t rows = rows_, candidates = candidates_, expand_width = expand_width_;
    if (candidates == 0 || candidates > CANDIDATES_MAX || expand_width == 0 || expand_width > candidates) return;
    float* logits = (float*)malloc(candidates * sizeof(float));
    top_child* order = (top_child*)malloc(candidates * sizeof(top_child));
    for (size_t row = 0; row < rows; ++row) {
        const size_t base = row * candidates;
        const size_t depth = node_metadata[MD_DEPTH * rows + row];
        const uint64_t seed = depth_seeds[depth];
        for (size_t i = 0; i < candidates; ++i) logits[i] = candidate_logits[base + i] + orc_bf16_to_f32(residual_logits[base + i]);
        float max = -INFINITY;
        for (size_t i = 0; i < candidates; ++i) max = fmaxf(max, logits[i]); /* f32::max: NaN-ignoring like fmaxf */
        float sum = 0.0f;
        for (size_t i = 0; i < candidates; ++i) sum += expf(logits[i] - max);
        const float log_sum = logf(sum) + max;
        for (size_t i = 0; i < candidates; ++i) {
            uint32_t offset, word;
            orc_revidx(candidate_ids[base + i], vocab_size, &offset, &word);
            order[i].perturbed = logits[i] + orc_gumbel_float(seed, offset, word);
            order[i].token = candidate_ids[base + i];
            order[i].index = (uint32_t)i;
        }
        qsort(order, candidates, sizeof(top_child), cmp_top_child);
        for (size_t rank = 0; rank < expand_width; ++rank) {
            output_token_ids[row * expand_width + rank] = order[rank].token;
            output_model_logprobs[row * expand_width + rank] = logits[order[rank].index] - log_sum;
        }
    }
    free(logits);
    free(order);
}
