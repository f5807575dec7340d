/*
 * uzu_oracle_moe.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT).  See uzu_oracle.h.
 *
 * The Mixture-of-Experts MLP: MoeBlock::encode (BU/src/encodable_block/mlp/moe/mod.rs:204-350) over the reference's kernels.
 * WHERE THE REFERENCE'S CPU KERNEL HAS A BODY it is restated loop for loop:
 *   moe_router_top_k                   BU/src/backends/cpu/kernel/moe/router_topk.rs:9-141
 *   moe_counts_offsets_fused           .../counts_offsets_fused.rs:4-55
 *   moe_gather_x_perm{1,2}_d           .../gather.rs:7-40
 *   moe_experts_decode_down_fused2_d   .../experts_two_pass_decode.rs:33-77   (pass B)
 *   moe_finalize                       .../finalize.rs:7-50
 * WHERE IT IS `todo!()` (scatter_buckets.rs, tiles_map.rs, tiles_pass_a.rs, experts_two_pass_prefill.rs, pass A of experts_two_pass_decode.rs:9-31) the
 * statement of record is the Metal shader and the expectation the reference's own tests compute in-test -- **oracle from shader**, flagged per function:
 *   scatter (bucketed ids / probs / tok2row)   metal/kernel/moe/scatter_buckets.metal + moe_experts_test.rs:92-133 (scatter_by_expert: rows of an expert in
 *                                              (token, slot) order -- the only property the block's output depends on is that tok2row inverts the bucketing)
 *   pass A (hidden = act(gate) * up, f32)      metal/kernel/moe/experts_two_pass_decode.metal:12-118 (dot products from zero, bias added behind them, clamp,
 *                                              activation; the lanes' partial sums are summed sequentially here), moe_experts_test.rs:170-275 (cpu_moe_reference)
 *   prefill pass B                             = the decode pass B's arithmetic per row (experts_two_pass_prefill.metal computes the same products on simdgroup tiles)
 * The tile-map / dispatch-argument kernels (tiles_map.rs, tiles_pass_a.rs) only shape Metal's indirect dispatches and have no numerical content.
 *
 * PARITY STATUS: **parity unpinned** against the reference binary; pinned against the reference tests' in-test expectations (tests/test_oracle_moe.py replays
 * moe_router_topk_test.rs, moe_counts_offsets_fused_test.rs, moe_finalize_test.rs, moe_gather_test.rs and cpu_moe_reference on their procedural inputs).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "uzu_oracle_model_internal.h"

static inline float ld(const void* base, uint32_t dt, size_t i) {
    if (dt == ORC_F32) return ((const float*)base)[i];
    return orc_bf16_to_f32(((const uint16_t*)base)[i]);
}
static inline void st(void* base, uint32_t dt, size_t i, float v) {
    if (dt == ORC_F32) ((float*)base)[i] = v;
    else ((uint16_t*)base)[i] = orc_f32_to_bf16(v);
}

/* router_topk.rs:9-141 with has_biases = true and the other specialisations off (MoeBlock::new, mod.rs:170-172): logits = W x + b accumulated in four
 * strided partial sums (the Metal vec4 shape), top-k by insertion (ties: lower expert id), softmax over the k winners when `renorm` */
void orc_moe_router_topk(const void* input, const void* weight, const void* bias, int32_t* topk_ids, void* topk_probs, uint32_t dt, uint32_t t, uint32_t d_model, uint32_t e,
                         uint32_t k, uint32_t renorm) {
    if (d_model % 4 || k < 1 || e < k) {
        fprintf(stderr, "oracle: moe_router_topk needs d_model %% 4 == 0 and 1 <= k <= e\n");
        abort();
    }
    float* logits = (float*)orc_xcalloc((size_t)t * e, 4);
    float* best_vals = (float*)orc_xcalloc(k, 4);
    int32_t* best_ids = (int32_t*)orc_xcalloc(k, 4);
    float* exps = (float*)orc_xcalloc(k, 4);
    for (size_t token = 0; token < t; ++token) {
        for (size_t expert = 0; expert < e; ++expert) {
            float accum[4] = {0.f, 0.f, 0.f, 0.f};
            for (size_t chunk = 0; chunk < d_model; chunk += 4)
                for (int i = 0; i < 4; ++i) {
                    const size_t idx = chunk + i;
                    const float x = ld(input, dt, token * d_model + idx) * 1.0f * 1.0f * 1.0f; /* inv_rms, router_input_scale, scale = 1 */
                    accum[i] += ld(weight, dt, expert * d_model + idx) * x;
                }
            const float sum = (accum[0] + accum[1]) + (accum[2] + accum[3]);
            logits[token * e + expert] = sum + (bias ? ld(bias, dt, expert) : 0.0f);
        }
        for (uint32_t j = 0; j < k; ++j) best_vals[j] = -INFINITY, best_ids[j] = -1;
        const float* row = logits + token * e;
        for (size_t expert = 0; expert < e; ++expert) {
            const float v = row[expert];
            int insert_pos = -1;
            for (int j = (int)k - 1; j >= 0; --j)
                if (v > best_vals[j] || (v == best_vals[j] && (best_ids[j] < 0 || (int32_t)expert < best_ids[j]))) insert_pos = j;
            if (insert_pos >= 0) {
                for (int s = (int)k - 1; s > insert_pos; --s) best_vals[s] = best_vals[s - 1], best_ids[s] = best_ids[s - 1];
                best_vals[insert_pos] = v;
                best_ids[insert_pos] = (int32_t)expert;
            }
        }
        const size_t base = token * k;
        for (uint32_t kk = 0; kk < k; ++kk) topk_ids[base + kk] = best_ids[kk];
        if (renorm) {
            float max_v = -INFINITY;
            for (uint32_t kk = 0; kk < k; ++kk) max_v = fmaxf(max_v, best_vals[kk]);
            float sum = 0.0f;
            for (uint32_t kk = 0; kk < k; ++kk) {
                exps[kk] = expf(best_vals[kk] - max_v);
                sum += exps[kk];
            }
            if (sum > 0.0f) {
                for (uint32_t kk = 0; kk < k; ++kk) st(topk_probs, dt, base + kk, exps[kk] / sum * 1.0f);
            } else {
                const float uniform = 1.0f / (float)k;
                for (uint32_t kk = 0; kk < k; ++kk) st(topk_probs, dt, base + kk, uniform * 1.0f);
            }
        } else {
            for (uint32_t kk = 0; kk < k; ++kk) st(topk_probs, dt, base + kk, best_vals[kk] * 1.0f);
        }
    }
    free(logits), free(best_vals), free(best_ids), free(exps);
}

/* counts_offsets_fused.rs:4-55: histogram of the expert ids (ids outside [0, e) ignored), exclusive scan; partials = the counts */
void orc_moe_counts_offsets_fused(const int32_t* topk_ids, uint32_t* offsets, uint32_t* sum_k_out, uint32_t* partials, uint32_t t, uint32_t e, uint32_t k) {
    if (e == 0) {
        offsets[0] = 0;
        *sum_k_out = 0;
        return;
    }
    uint32_t* counts = (uint32_t*)orc_xcalloc(e, 4);
    for (size_t ti = 0; ti < t; ++ti)
        for (size_t kk = 0; kk < k; ++kk) {
            const int32_t eid = topk_ids[ti * k + kk];
            if (eid >= 0 && (uint32_t)eid < e) counts[eid] += 1;
        }
    if (partials)
        for (uint32_t i = 0; i < e; ++i) partials[i] = counts[i];
    uint32_t sum = 0;
    for (uint32_t i = 0; i < e; ++i) {
        offsets[i] = sum;
        sum += counts[i];
    }
    offsets[e] = sum;
    *sum_k_out = sum;
    free(counts);
}

/* ORACLE FROM SHADER / TEST (scatter_buckets.rs is todo!()): MoeBlockBasesFromPartials + MoeScatterBucketsMap as one step.  Rows of expert x occupy
 * [offsets[x], offsets[x + 1]) in (token, slot) order (moe_experts_test.rs:92-133 scatter_by_expert); bucketed_ids[row] = the row's token,
 * bucketed_probs[row] = its routing probability, tok2row[token * k + slot] = row (-1 for an id outside [0, e): the encode_fill(0xFF) of mod.rs:262) */
void orc_moe_scatter_buckets(const int32_t* topk_ids, const void* topk_probs, const uint32_t* offsets, int32_t* bucketed_ids, void* bucketed_probs, int32_t* tok2row, uint32_t dt,
                             uint32_t t, uint32_t e, uint32_t k) {
    uint32_t* cursor = (uint32_t*)orc_xcalloc(e ? e : 1, 4);
    for (size_t i = 0; i < (size_t)t * k; ++i) {
        const int32_t eid = topk_ids[i];
        if (eid < 0 || (uint32_t)eid >= e) {
            tok2row[i] = -1;
            continue;
        }
        const uint32_t row = offsets[eid] + cursor[eid]++;
        bucketed_ids[row] = (int32_t)(i / k);
        st(bucketed_probs, dt, row, ld(topk_probs, dt, i));
        tok2row[i] = (int32_t)row;
    }
    free(cursor);
}

/* gather.rs:7-40 */
void orc_moe_gather(const void* x, const int32_t* bucketed_ids, void* x_perm, const uint32_t* sumk_buf, uint32_t dt, uint32_t d_model, uint32_t t, uint32_t k) {
    const size_t total_rows = *sumk_buf;
    if (total_rows > (size_t)t * k) {
        fprintf(stderr, "oracle: moe_gather: %zu rows > t * k\n", total_rows);
        abort();
    }
    const size_t esz = dt == ORC_F32 ? 4 : 2;
    for (size_t row = 0; row < total_rows; ++row) {
        const int32_t token = bucketed_ids[row];
        if (token < 0) continue;
        memcpy((uint8_t*)x_perm + row * d_model * esz, (const uint8_t*)x + (size_t)token * d_model * esz, (size_t)d_model * esz);
    }
}

/* activations.h of the Metal tree = activation_type.rs:31-49: silu with alpha, tanh-form GELU; f32 in, f32 out */
static float silu_alpha(float x, float alpha) { return x / (1.0f + expf(-alpha * x)); }
static float gelu_approx(float x) { return orc_activate(UZU_ACT_GELU_APPROX, x, ORC_F32); }

/* ORACLE FROM SHADER (pass A of experts_two_pass_decode.rs:9-31 and experts_two_pass_prefill.rs are todo!()): experts_two_pass_decode.metal:12-118 per
 * (row, hidden element); gating_sel 0 GELU(up), 1 SiLU(up), 2 SwiGLU = SiLU(gate) * up, 3 GEGLU (moe_experts_test.rs:163-168).  hidden f32 [total_rows, d_ff] */
void orc_moe_experts_pass_a(const void* x_perm, const uint32_t* expert_offsets, const void* w13_all, const void* up_biases, float* hidden_out, uint32_t dt, uint32_t d_model,
                            uint32_t d_ff, uint32_t e, float gate_clip_min, float gate_clip_max, float up_clip_min, float up_clip_max, float alpha, uint32_t gating_sel) {
    for (size_t expert = 0; expert < e; ++expert) {
        const size_t w13_base = expert * 2 * (size_t)d_ff * d_model, bias_base = expert * 2 * (size_t)d_ff;
#pragma omp parallel for schedule(static) collapse(2)
        for (size_t row = expert_offsets[expert]; row < expert_offsets[expert + 1]; ++row)
            for (size_t h = 0; h < d_ff; ++h) {
                float acc_up = 0.0f, acc_gate = 0.0f;
                for (size_t d = 0; d < d_model; ++d) {
                    const float xv = ld(x_perm, dt, row * d_model + d);
                    acc_up += xv * ld(w13_all, dt, w13_base + h * d_model + d);
                    if (gating_sel > 1) acc_gate += xv * ld(w13_all, dt, w13_base + (d_ff + h) * d_model + d);
                }
                float up_val = acc_up + ld(up_biases, dt, bias_base + h);
                up_val = fminf(fmaxf(up_val, up_clip_min), up_clip_max);
                float activated;
                if (gating_sel <= 1) {
                    activated = gating_sel == 0 ? gelu_approx(up_val) : silu_alpha(up_val, alpha);
                } else {
                    float gate_val = acc_gate + ld(up_biases, dt, bias_base + d_ff + h);
                    gate_val = fminf(fmaxf(gate_val, gate_clip_min), gate_clip_max);
                    activated = (gating_sel == 2 ? silu_alpha(gate_val, alpha) : gelu_approx(gate_val)) * up_val;
                }
                hidden_out[row * d_ff + h] = activated;
            }
    }
}

/* experts_two_pass_decode.rs:33-77 (moe_experts_decode_down_fused2_d): per row and output column a fused-multiply-add chain over the hidden row, the bias
 * behind it, one rounding to T.  row_expert_map[row] = the expert of the row (MoePassABuildRowMap: derived from the offsets). */
void orc_moe_experts_down(const float* hidden, const uint32_t* row_expert_map, const void* w2_all, const void* down_biases, void* y_out, uint32_t dt, uint32_t total_rows,
                          uint32_t d_model, uint32_t d_ff, uint32_t e) {
#pragma omp parallel for schedule(static)
    for (size_t row_idx = 0; row_idx < total_rows; ++row_idx) {
        const size_t expert_idx = row_expert_map[row_idx];
        if (expert_idx >= e) {
            fprintf(stderr, "oracle: moe_experts_down: row %zu maps to expert %zu of %u\n", row_idx, expert_idx, e);
            abort();
        }
        for (size_t col = 0; col < d_model; ++col) {
            const size_t w2_col_base = expert_idx * d_model * (size_t)d_ff + col * d_ff;
            float acc = 0.0f;
            for (size_t h = 0; h < d_ff; ++h) acc = fmaf(hidden[row_idx * d_ff + h], ld(w2_all, dt, w2_col_base + h), acc); /* h_val.mul_add(w_val, acc) */
            acc += ld(down_biases, dt, expert_idx * d_model + col);
            st(y_out, dt, row_idx * d_model + col, acc);
        }
    }
}

/* finalize.rs:7-50 */
void orc_moe_finalize(const int32_t* tok2row, const void* probs, const void* y_partial, void* y, uint32_t dt, uint32_t t_count, uint32_t d_model, uint32_t k) {
    for (size_t ti = 0; ti < t_count; ++ti)
        for (size_t f = 0; f < d_model; ++f) {
            float acc = 0.0f;
            for (size_t kk = 0; kk < k; ++kk) {
                const size_t idx = ti * k + kk;
                const int32_t row = tok2row[idx];
                if (row >= 0) {
                    float prob = ld(probs, dt, idx);
                    if (!isfinite(prob)) prob = 0.0f;
                    float val = ld(y_partial, dt, (size_t)row * d_model + f);
                    if (!isfinite(val)) val = 0.0f;
                    acc += prob * val;
                }
            }
            if (!isfinite(acc)) acc = 0.0f;
            st(y, dt, ti * d_model + f, acc);
        }
}

/* MoeBlock::encode (mod.rs:204-350) for `batch` rows of bf16 input: router top-k -> counts / offsets -> scatter -> gather -> experts (pass A, pass B) ->
 * finalize.  Returns a fresh bf16 [batch, model_dim] buffer. */
uint16_t* orc_moe_block(const uzu_moe_desc* M, uint32_t model_dim, const uint16_t* input, uint32_t batch) {
    const uint32_t E = M->num_routed_experts, K = M->num_active_experts, dff = M->expert_hidden_dim;
    const size_t total = (size_t)batch * K;
    int32_t* topk_ids = (int32_t*)orc_xcalloc(total, 4);
    memset(topk_ids, 0xFF, total * 4); /* encoder.encode_fill(&mut topk_ids, 0xFF) (mod.rs:220) */
    uint16_t* topk_probs = (uint16_t*)orc_xcalloc(total, 2);
    orc_moe_router_topk(input, M->router_weights, M->router_biases, topk_ids, topk_probs, ORC_BF16, batch, model_dim, E, K, M->router_renorm);
    uint32_t* offsets = (uint32_t*)orc_xcalloc(E + 1, 4);
    uint32_t sumk = 0;
    orc_moe_counts_offsets_fused(topk_ids, offsets, &sumk, NULL, batch, E, K);
    int32_t* bucketed_ids = (int32_t*)orc_xcalloc(total, 4);
    uint16_t* bucketed_probs = (uint16_t*)orc_xcalloc(total, 2);
    int32_t* tok2row = (int32_t*)orc_xcalloc(total, 4);
    orc_moe_scatter_buckets(topk_ids, topk_probs, offsets, bucketed_ids, bucketed_probs, tok2row, ORC_BF16, batch, E, K);
    uint16_t* x_perm = (uint16_t*)orc_xcalloc(total * model_dim, 2);
    orc_moe_gather(input, bucketed_ids, x_perm, &sumk, ORC_BF16, model_dim, batch, K);
    float* hidden = (float*)orc_xcalloc(total * dff, 4);
    orc_moe_experts_pass_a(x_perm, offsets, M->w13, M->up_biases, hidden, ORC_BF16, model_dim, dff, E, M->gate_clip_min, M->gate_clip_max, M->up_clip_min, M->up_clip_max,
                           M->silu_alpha, M->gating_sel);
    uint32_t* row_expert_map = (uint32_t*)orc_xcalloc(total ? total : 1, 4); /* MoePassABuildRowMap: the expert whose segment holds the row */
    for (uint32_t x = 0; x < E; ++x)
        for (uint32_t r = offsets[x]; r < offsets[x + 1]; ++r) row_expert_map[r] = x;
    uint16_t* y_partial = (uint16_t*)orc_xcalloc(total * model_dim, 2);
    orc_moe_experts_down(hidden, row_expert_map, M->w2, M->down_biases, y_partial, ORC_BF16, sumk, model_dim, dff, E);
    uint16_t* out = (uint16_t*)orc_xcalloc((size_t)batch * model_dim, 2);
    orc_moe_finalize(tok2row, topk_probs, y_partial, out, ORC_BF16, batch, model_dim, K);
    free(topk_ids), free(topk_probs), free(offsets), free(bucketed_ids), free(bucketed_probs), free(tok2row), free(x_perm), free(hidden), free(row_expert_map), free(y_partial);
    return out;
}
