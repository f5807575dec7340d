/*
 * uzu_oracle_tree_verify.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT).  See uzu_oracle.h.
 *
 * Gated DeltaNet over a speculated token TREE (BU = crates/backend-uzu/src): restatement of the reference's CPU kernels
 *   ConvTreeScan        BU/backends/cpu/kernel/gdn/tree_verify/conv_scan.rs:13-72
 *   DeltaNetPrefillPrep BU/backends/cpu/kernel/gdn/prefill_prep.rs:12-112 in the tree-prep instantiation of
 *                       BU/encodable_block/mixer/delta_net.rs:257-265 (QKT = T, write_log_decay, write_compact_v)
 *   BuildTreePrefix     BU/backends/cpu/kernel/gdn/tree_verify/prefix.rs:6-39
 *   BuildTreeGram       BU/backends/cpu/kernel/gdn/tree_verify/tree_gram.rs:13-163
 *   TreeUpdateSolve     BU/backends/cpu/kernel/gdn/tree_verify/tree_update_solve.rs:9-131
 *   BuildTreeOut        BU/backends/cpu/kernel/gdn/tree_verify/out.rs:8-88
 *   StateAdvance        BU/backends/cpu/kernel/gdn/tree_verify/state_advance.rs:9-59
 * Same loops, same f32 accumulation order.  The reference's CPU backend has the kernels but not the composition
 * (cpu/kernel/mod.rs:37: DeltaNetTreeVerify = Infallible); the order they run in is the Metal composition's
 * (BU/backends/metal/kernel/gdn/tree_verify.rs:92-187), restated in uzu_oracle_model.c.
 *
 * A trie node is {trie_start, trie_end, height} (gpu_types/trie.rs): node `col` is an ancestor-or-self of node `row`
 * iff trie_start[col] <= row <= trie_end[col] (DFS order: a subtree is a contiguous index range).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "uzu_oracle.h"

#define TV_BLOCK 16u

static inline float tv_rd(const void* base, uint32_t dt, size_t i) {
    return dt == ORC_F32 ? ((const float*)base)[i] : orc_bf16_to_f32(((const uint16_t*)base)[i]);
}
static inline void tv_wr(void* base, uint32_t dt, size_t i, float v) {
    if (dt == ORC_F32) ((float*)base)[i] = v;
    else ((uint16_t*)base)[i] = orc_f32_to_bf16(v);
}

/* conv_scan.rs:13-72.  in_proj / out_proj [suffix_len, total_proj_dim] (T), base_state f32 [conv_dim, k-1], parents i32 [suffix_len]
 * (-1 = child of the accepted context), suffix_state f32 [suffix_len, conv_dim, k-1]: the conv state a sequence would hold after
 * accepting the path that ends in that node. */
void orc_conv_tree_scan(const void* in_proj, const float* conv_weight, const float* bias, const float* base_state, const int32_t* parents,
                        void* out_proj, float* suffix_state, uint32_t dt, uint32_t suffix_len, uint32_t kernel_size, uint32_t total_proj_dim,
                        uint32_t conv_dim) {
    const size_t state_stride = kernel_size - 1;
    for (size_t node_idx = 0; node_idx < suffix_len; ++node_idx)
        for (size_t channel_idx = 0; channel_idx < total_proj_dim; ++channel_idx) {
            const size_t proj_idx = node_idx * total_proj_dim + channel_idx;
            if (channel_idx >= conv_dim) {
                if (dt == ORC_F32) ((float*)out_proj)[proj_idx] = ((const float*)in_proj)[proj_idx];
                else ((uint16_t*)out_proj)[proj_idx] = ((const uint16_t*)in_proj)[proj_idx];
                continue;
            }
            float acc = bias ? bias[channel_idx] : 0.0f;
            const size_t weight_offset = channel_idx * kernel_size;
            const size_t base_state_offset = channel_idx * state_stride;
            int32_t source_row = (int32_t)node_idx;
            for (size_t history_offset = 0; history_offset < kernel_size; ++history_offset) {
                float sample;
                if (source_row >= 0) {
                    sample = tv_rd(in_proj, dt, (size_t)source_row * total_proj_dim + channel_idx);
                } else {
                    const size_t base_state_tap = state_stride - (size_t)(-(int64_t)source_row);
                    sample = base_state[base_state_offset + base_state_tap];
                }
                const size_t weight_tap = kernel_size - 1 - history_offset;
                acc += sample * conv_weight[weight_offset + weight_tap];
                if (history_offset < state_stride) {
                    const size_t state_tap = state_stride - 1 - history_offset;
                    suffix_state[(node_idx * conv_dim + channel_idx) * state_stride + state_tap] = sample;
                }
                source_row = source_row >= 0 ? parents[source_row] : source_row - 1;
            }
            tv_wr(out_proj, dt, proj_idx, orc_activate(UZU_ACT_SILU, acc, ORC_F32));
        }
}

/* prefill_prep.rs:12-112 with QKT = T (the normalised q / k are ROUNDED to the activation type), write_log_decay = true
 * (decay_out holds log decays) and write_compact_v = true (the value section of every row copied out): delta_net.rs:257-265. */
void orc_delta_net_tree_prep(const void* in_proj, const float* a_log, const float* dt_bias, void* q_norm_out, void* k_norm_out,
                             void* compact_v_out, float* beta_out, float* log_decay_out, uint32_t dt, uint32_t num_v_heads,
                             uint32_t num_k_heads, uint32_t head_k_dim, uint32_t key_dim, uint32_t value_dim, uint32_t suffix_len) {
    const size_t conv_dim = 2 * (size_t)key_dim + value_dim;
    const size_t total_proj_dim = conv_dim + value_dim + 2 * (size_t)num_v_heads;
    const size_t groups_per_head = num_v_heads / num_k_heads;
    const size_t esz = dt == ORC_F32 ? 4 : 2;
    for (size_t token = 0; token < suffix_len; ++token) {
        const size_t tok_offset = token * total_proj_dim;
        memcpy((char*)compact_v_out + token * value_dim * esz, (const char*)in_proj + (tok_offset + 2 * key_dim) * esz, (size_t)value_dim * esz);
        for (size_t hk = 0; hk < num_k_heads; ++hk) {
            const size_t q_off = tok_offset + hk * head_k_dim;
            float q_sq = 0.0f;
            for (size_t j = 0; j < head_k_dim; ++j) {
                const float v = tv_rd(in_proj, dt, q_off + j);
                q_sq += v * v;
            }
            const float q_inv = 1.0f / sqrtf(q_sq + 1e-6f);
            const float q_scale = 1.0f / sqrtf((float)head_k_dim);
            for (size_t j = 0; j < head_k_dim; ++j)
                tv_wr(q_norm_out, dt, token * key_dim + hk * head_k_dim + j, tv_rd(in_proj, dt, q_off + j) * q_inv * q_scale);
            const size_t k_off = tok_offset + key_dim + hk * head_k_dim;
            float k_sq = 0.0f;
            for (size_t j = 0; j < head_k_dim; ++j) {
                const float v = tv_rd(in_proj, dt, k_off + j);
                k_sq += v * v;
            }
            const float k_inv = 1.0f / sqrtf(k_sq + 1e-6f);
            for (size_t j = 0; j < head_k_dim; ++j)
                tv_wr(k_norm_out, dt, token * key_dim + hk * head_k_dim + j, tv_rd(in_proj, dt, k_off + j) * k_inv);
            for (size_t group = 0; group < groups_per_head; ++group) {
                const size_t hv = hk * groups_per_head + group;
                const float beta_raw = tv_rd(in_proj, dt, tok_offset + conv_dim + value_dim + hv);
                const float beta = 1.0f / (1.0f + expf(-beta_raw));
                const float a_raw = tv_rd(in_proj, dt, tok_offset + conv_dim + value_dim + num_v_heads + hv);
                const float sp_in = a_raw + dt_bias[hv];
                const float sp = sp_in > 20.0f ? sp_in : logf(1.0f + expf(sp_in));
                beta_out[token * num_v_heads + hv] = beta;
                log_decay_out[token * num_v_heads + hv] = -expf(a_log[hv]) * sp;
            }
        }
    }
}

/* prefix.rs:6-39: prefix[row, head] = sum of log_decay over the ancestors-or-self of row, in column order.  trie: 3 u32 per node. */
void orc_build_tree_prefix(const uint32_t* trie, const float* log_decay, float* prefix, uint32_t batch_size, uint32_t tree_size,
                           uint32_t value_heads) {
    for (size_t batch = 0; batch < batch_size; ++batch) {
        const size_t trie_batch = batch * tree_size, batch_offset = batch * tree_size * value_heads;
        for (size_t row = 0; row < tree_size; ++row)
            for (size_t head = 0; head < value_heads; ++head) {
                float sum = 0.0f;
                for (size_t col = 0; col < tree_size; ++col) {
                    const uint32_t* node = trie + (trie_batch + col) * 3;
                    if ((uint32_t)row < node[0] || (uint32_t)row > node[1]) continue;
                    sum += log_decay[batch_offset + col * value_heads + head];
                }
                prefix[batch_offset + row * value_heads + head] = sum;
            }
    }
}

/* tree_gram.rs:13-163.  q / k [batch, tree, k_heads, dk] (T); prefix / beta [batch, tree, value_heads]; h0 f32 [slots, value_heads, dv, dk],
 * h0_idx i32 [batch] (-1: no initial state) -- both NULL <=> use_h0 = false.  Outputs: a_packed f32 [batch*value_heads, NB,
 * ceil(NB/2), 16, 32] (only the tiles touching the block lower triangle are written), qkd f32 [batch*value_heads, tree, tree],
 * a_inv f32 [batch*value_heads, NB, 16, 16], kh0 f32 [batch, tree, value_heads, dv]. */
void orc_build_tree_gram(const void* q, const void* k, uint32_t dt, const uint32_t* trie, const float* prefix, const float* beta, const float* h0,
                         const int32_t* h0_idx, float* a_packed, float* qkd, float* a_inv, float* kh0, float scale, uint32_t batch_size,
                         uint32_t tree_size_, uint32_t k_heads_, uint32_t value_heads_, uint32_t head_k_dim_, uint32_t head_v_dim_) {
    const size_t tree_size = tree_size_, k_heads = k_heads_, value_heads = value_heads_, head_k_dim = head_k_dim_, head_v_dim = head_v_dim_;
    const size_t B = TV_BLOCK;
    const size_t value_heads_per_key_head = value_heads / k_heads;
    const size_t num_blocks = (tree_size + B - 1) / B, num_col_pairs = (num_blocks + 1) / 2;
    const int use_h0 = h0 != NULL && h0_idx != NULL && kh0 != NULL;
    for (size_t batch = 0; batch < batch_size; ++batch)
        for (size_t hv = 0; hv < value_heads; ++hv) {
            const size_t hk = hv / value_heads_per_key_head;
            /* qkd[row, col] = scale * exp(prefix[row] - prefix[col]) * dot(q[row], k[col]) for ancestor-or-self, else 0 */
            const size_t mat_base = (batch * value_heads + hv) * tree_size * tree_size;
            for (size_t row = 0; row < tree_size; ++row) {
                const float prefix_row = prefix[(batch * tree_size + row) * value_heads + hv];
                const size_t row_off = ((batch * tree_size + row) * k_heads + hk) * head_k_dim;
                for (size_t col = 0; col < tree_size; ++col) {
                    const size_t out = mat_base + row * tree_size + col;
                    const uint32_t* node = trie + (batch * tree_size + col) * 3;
                    if ((uint32_t)row < node[0] || (uint32_t)row > node[1]) {
                        qkd[out] = 0.0f;
                        continue;
                    }
                    const size_t col_off = ((batch * tree_size + col) * k_heads + hk) * head_k_dim;
                    float qk = 0.0f;
                    for (size_t d = 0; d < head_k_dim; ++d) qk += tv_rd(q, dt, row_off + d) * tv_rd(k, dt, col_off + d);
                    const float prefix_col = prefix[(batch * tree_size + col) * value_heads + hv];
                    qkd[out] = expf(prefix_row - prefix_col) * scale * qk;
                }
            }
            /* A[row, col] = beta[row] * exp(prefix[row] - prefix[col]) * dot(k[row], k[col]) for proper ancestors, else 0 */
            const size_t a_base = (batch * value_heads + hv) * num_blocks * num_col_pairs * B * 2 * B;
            for (size_t block = 0; block < num_blocks; ++block)
                for (size_t pair = 0; pair <= block / 2; ++pair) {
                    const size_t tile_base = a_base + (block * num_col_pairs + pair) * B * 2 * B;
                    for (size_t local_row = 0; local_row < B; ++local_row)
                        for (size_t local_col = 0; local_col < 2 * B; ++local_col) {
                            const size_t row = block * B + local_row, col = pair * 2 * B + local_col;
                            float value = 0.0f;
                            if (row != col && row < tree_size && col < tree_size) {
                                const uint32_t* node = trie + (batch * tree_size + col) * 3;
                                if ((uint32_t)row >= node[0] && (uint32_t)row <= node[1]) {
                                    const size_t row_off = ((batch * tree_size + row) * k_heads + hk) * head_k_dim;
                                    const size_t col_off = ((batch * tree_size + col) * k_heads + hk) * head_k_dim;
                                    float kk = 0.0f;
                                    for (size_t d = 0; d < head_k_dim; ++d) kk += tv_rd(k, dt, row_off + d) * tv_rd(k, dt, col_off + d);
                                    const float prefix_row = prefix[(batch * tree_size + row) * value_heads + hv];
                                    const float prefix_col = prefix[(batch * tree_size + col) * value_heads + hv];
                                    const float beta_row = beta[(batch * tree_size + row) * value_heads + hv];
                                    value = beta_row * expf(prefix_row - prefix_col) * kk;
                                }
                            }
                            a_packed[tile_base + local_row * 2 * B + local_col] = value;
                        }
                }
            /* a_inv = (I + A_diag)^-1 per block (forward substitution on the strictly lower block), identity-padded */
            for (size_t block = 0; block < num_blocks; ++block) {
                const size_t block_size = tree_size - block * B < B ? tree_size - block * B : B;
                const size_t block_base = ((batch * value_heads + hv) * num_blocks + block) * B * B;
                const size_t diag_tile = a_base + (block * num_col_pairs + block / 2) * B * 2 * B;
                const size_t diag_col = (block % 2) * B;
                for (size_t row = 0; row < B; ++row)
                    for (size_t col = 0; col < B; ++col) a_inv[block_base + row * B + col] = row == col ? 1.0f : 0.0f;
                for (size_t row = 0; row < block_size; ++row)
                    for (size_t col = 0; col < row; ++col) {
                        float sum = 0.0f;
                        for (size_t prev_row = col; prev_row < row; ++prev_row)
                            sum += a_packed[diag_tile + row * 2 * B + diag_col + prev_row] * a_inv[block_base + prev_row * B + col];
                        a_inv[block_base + row * B + col] = -sum;
                    }
            }
            /* kh0[b, token, hv, dv] = dot(k[b, token, hk], h0[slot, hv, dv]) */
            if (!use_h0) continue;
            const int32_t h0_slot = h0_idx[batch];
            if (h0_slot >= 0) {
                const size_t h0_head = ((size_t)h0_slot * value_heads + hv) * head_v_dim * head_k_dim;
                for (size_t token = 0; token < tree_size; ++token) {
                    const size_t k_off = ((batch * tree_size + token) * k_heads + hk) * head_k_dim;
                    const size_t kh0_off = ((batch * tree_size + token) * value_heads + hv) * head_v_dim;
                    for (size_t dv = 0; dv < head_v_dim; ++dv) {
                        float sum = 0.0f;
                        for (size_t d = 0; d < head_k_dim; ++d) sum += tv_rd(k, dt, k_off + d) * h0[h0_head + dv * head_k_dim + d];
                        kh0[kh0_off + dv] = sum;
                    }
                }
            }
        }
}

/* tree_update_solve.rs:9-131: (I + A) U = beta * (v - exp(prefix) * kh0), block forward substitution with the inverted diagonal blocks.
 * v [batch, tree, value_heads, dv] (T); u f32 [batch*value_heads, tree, dv]. */
void orc_tree_update_solve(const float* kh0, const void* v, uint32_t dt, const float* prefix, const float* beta, const float* a_packed,
                           const float* a_inv, const int32_t* h0_idx, float* u, uint32_t batch_size, uint32_t tree_size_, uint32_t value_heads_,
                           uint32_t head_v_dim_) {
    const size_t tree_size = tree_size_, value_heads = value_heads_, head_v_dim = head_v_dim_, block_size = TV_BLOCK;
    if (batch_size == 0 || tree_size == 0 || value_heads == 0 || head_v_dim == 0) return;
    const size_t num_blocks = (tree_size + block_size - 1) / block_size, num_col_pairs = (num_blocks + 1) / 2;
    const int use_h0 = kh0 != NULL && h0_idx != NULL;
    float* acc = (float*)malloc(sizeof(float) * block_size * head_v_dim);
    for (size_t batch = 0; batch < batch_size; ++batch) {
        const int32_t h0_slot = use_h0 ? h0_idx[batch] : -1;
        for (size_t hv = 0; hv < value_heads; ++hv) {
            const size_t bvh = batch * value_heads + hv;
            for (size_t block = 0; block < num_blocks; ++block) {
                const size_t token_base = block * block_size;
                for (size_t i = 0; i < block_size * head_v_dim; ++i) acc[i] = 0.0f;
                for (size_t local_token = 0; local_token < block_size; ++local_token) {
                    const size_t token = token_base + local_token;
                    if (token >= tree_size) continue;
                    const size_t prefix_idx = (batch * tree_size + token) * value_heads + hv;
                    const float beta_val = beta[prefix_idx];
                    const float decay_from_h0 = expf(prefix[prefix_idx]);
                    for (size_t dv = 0; dv < head_v_dim; ++dv) {
                        const size_t v_idx = ((batch * tree_size + token) * value_heads + hv) * head_v_dim + dv;
                        const float v_val = tv_rd(v, dt, v_idx);
                        const float kh0_val = h0_slot >= 0 ? kh0[v_idx] : 0.0f;
                        acc[local_token * head_v_dim + dv] = beta_val * (v_val - decay_from_h0 * kh0_val);
                    }
                }
                for (size_t prev_token = 0; prev_token < token_base; ++prev_token) {
                    const size_t prev_block = prev_token / block_size, prev_local = prev_token % block_size;
                    for (size_t local_token = 0; local_token < block_size; ++local_token) {
                        const size_t token = token_base + local_token;
                        if (token >= tree_size) continue;
                        const size_t a_idx = ((bvh * num_blocks + block) * num_col_pairs + prev_block / 2) * (block_size * 2 * block_size) +
                                             local_token * (2 * block_size) + (prev_block % 2) * block_size + prev_local;
                        const float a_val = a_packed[a_idx];
                        for (size_t dv = 0; dv < head_v_dim; ++dv)
                            acc[local_token * head_v_dim + dv] -= a_val * u[(bvh * tree_size + prev_token) * head_v_dim + dv];
                    }
                }
                for (size_t local_token = 0; local_token < block_size; ++local_token) {
                    const size_t token = token_base + local_token;
                    if (token >= tree_size) continue;
                    for (size_t dv = 0; dv < head_v_dim; ++dv) {
                        float sum = 0.0f;
                        for (size_t local_prev = 0; local_prev < block_size; ++local_prev) {
                            if (token_base + local_prev >= tree_size) continue;
                            const size_t inv_idx = ((bvh * num_blocks + block) * block_size + local_token) * block_size + local_prev;
                            sum += a_inv[inv_idx] * acc[local_prev * head_v_dim + dv];
                        }
                        u[(bvh * tree_size + token) * head_v_dim + dv] = sum;
                    }
                }
            }
        }
    }
    free(acc);
}

/* out.rs:8-88: o[row] = exp(prefix[row]) * scale * (q[row] . h0) + sum_col qkd[row, col] * u[col]; o [batch, tree, value_heads, dv] */
void orc_build_tree_out(const void* q, uint32_t qk_dt, const float* prefix, const float* qkd, const float* u, const float* h0, const int32_t* h0_indices,
                        void* o, uint32_t out_dt, float scale, uint32_t batch_size, uint32_t tree_size_, uint32_t qk_heads_, uint32_t value_heads_,
                        uint32_t head_k_dim_, uint32_t head_v_dim_) {
    const size_t tree_size = tree_size_, qk_heads = qk_heads_, value_heads = value_heads_, head_k_dim = head_k_dim_, head_v_dim = head_v_dim_;
    const size_t value_heads_per_qk_head = value_heads / qk_heads;
    const int use_h0 = h0 != NULL && h0_indices != NULL;
    for (size_t batch = 0; batch < batch_size; ++batch) {
        const int32_t h0_index = use_h0 ? h0_indices[batch] : -1;
        for (size_t value_head = 0; value_head < value_heads; ++value_head) {
            const size_t qk_head = value_head / value_heads_per_qk_head;
            const size_t q_head_base = (batch * tree_size * qk_heads + qk_head) * head_k_dim;
            const size_t prefix_base = batch * tree_size * value_heads + value_head;
            const size_t qkd_base = (batch * value_heads + value_head) * tree_size * tree_size;
            const size_t u_base = (batch * value_heads + value_head) * tree_size * head_v_dim;
            const size_t out_base = ((batch * tree_size) * value_heads + value_head) * head_v_dim;
            for (size_t row = 0; row < tree_size; ++row) {
                const size_t q_row = q_head_base + row * qk_heads * head_k_dim;
                for (size_t value_col = 0; value_col < head_v_dim; ++value_col) {
                    float acc = 0.0f;
                    if (use_h0 && h0_index >= 0) {
                        const size_t h0_base = (((size_t)h0_index * value_heads + value_head) * head_v_dim + value_col) * head_k_dim;
                        float dot = 0.0f;
                        for (size_t dim = 0; dim < head_k_dim; ++dim) dot += tv_rd(q, qk_dt, q_row + dim) * h0[h0_base + dim];
                        acc += expf(prefix[prefix_base + row * value_heads]) * scale * dot;
                    }
                    for (size_t col = 0; col < tree_size; ++col)
                        acc += qkd[qkd_base + row * tree_size + col] * u[u_base + col * head_v_dim + value_col];
                    tv_wr(o, out_dt, out_base + row * value_heads * head_v_dim + value_col, acc);
                }
            }
        }
    }
}

/* state_advance.rs:9-59: the delta rule over the accepted path, on the state in place (head_v_dim == head_k_dim by construction) */
void orc_state_advance(const void* k_norm, const void* v, uint32_t dt, const float* log_decay_buf, const float* beta_buf, const uint32_t* accepted_indices,
                       float* state, uint32_t accepted_len, uint32_t num_v_heads_, uint32_t num_k_heads_, uint32_t head_k_dim_) {
    const size_t num_v_heads = num_v_heads_, num_k_heads = num_k_heads_, head_k_dim = head_k_dim_, head_v_dim = head_k_dim_;
    const size_t key_dim = num_k_heads * head_k_dim, value_dim = num_v_heads * head_v_dim;
    const size_t v_heads_per_k_head = num_v_heads / num_k_heads;
#pragma omp parallel for schedule(static)
    for (size_t hv_idx = 0; hv_idx < num_v_heads; ++hv_idx) {
        const size_t hk_idx = hv_idx / v_heads_per_k_head;
        for (size_t dv_idx = 0; dv_idx < head_v_dim; ++dv_idx) {
            const size_t state_row_offset = (hv_idx * head_v_dim + dv_idx) * head_k_dim;
            for (size_t accepted_idx = 0; accepted_idx < accepted_len; ++accepted_idx) {
                const size_t tree_idx = accepted_indices[accepted_idx];
                const size_t tree_head_offset = tree_idx * num_v_heads + hv_idx;
                const float decay = expf(log_decay_buf[tree_head_offset]);
                const float beta = beta_buf[tree_head_offset];
                const size_t k_offset = tree_idx * key_dim + hk_idx * head_k_dim;
                float kv_mem = 0.0f;
                for (size_t dk_idx = 0; dk_idx < head_k_dim; ++dk_idx) {
                    float* se = state + state_row_offset + dk_idx;
                    *se *= decay;
                    kv_mem += *se * tv_rd(k_norm, dt, k_offset + dk_idx);
                }
                const float v_value = tv_rd(v, dt, tree_idx * value_dim + hv_idx * head_v_dim + dv_idx);
                const float delta = beta * (v_value - kv_mem);
                for (size_t dk_idx = 0; dk_idx < head_k_dim; ++dk_idx) state[state_row_offset + dk_idx] += tv_rd(k_norm, dt, k_offset + dk_idx) * delta;
            }
        }
    }
}
