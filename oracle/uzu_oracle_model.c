/*
 * uzu_oracle_model.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT).  See uzu_oracle.h.
 *
 * Restates the op ORDER of the reference's model graph for one sequence:
 *   Decoder::encode            BU/src/encodable_block/decoder.rs:138-203
 *   Transformer::encode        BU/src/encodable_block/transformer.rs:226-329
 *   TransformerLayer::encode   BU/src/encodable_block/transformer_layer.rs:194-238
 *   Attention::attend          BU/src/encodable_block/mixer/attention/mode.rs:45-144
 *   AttentionCores::encode     BU/src/encodable_block/mixer/attention/core/mod.rs:81-93
 *                              (CPU backend: no GEMM core => two-pass iff prefix+suffix > 1024)
 *   DeltaNet::encode           BU/src/encodable_block/mixer/delta_net.rs:473-645
 *   DenseMlp::encode           BU/src/encodable_block/mlp/dense.rs:32-48
 *   Embedding::encode_readout  BU/src/encodable_block/embedding.rs:374-456
 *   Sampling (greedy)          BU/src/backends/cpu/kernel/sampling/unified_sampling.rs:90-98
 *   encode_accept              BU/src/encodable_block/mixer/attention/state.rs:174-236 (Full cache, flat
 *                              full accept => no copies, length += n)
 *   PerLayerEmbedding          BU/src/encodable_block/per_layer_embedding.rs:36-148 (model side), :150-271 (layer side)
 * and the Gemma-family layer options: per-layer RoPE configurations (transformer.rs:101-118,249-257), post-layer scalar
 * (transformer_layer.rs:61-84), embedding norm (decoder.rs:68-83,149-154), KV sharing (transformer.rs:264-275,
 * mixer/attention/mode.rs:79-84), value normalisation (mixer/attention/qkv_norm.rs:70-72,149-175)
 * on top of the kernels in uzu_oracle_kernels.c.  Activations are bf16 (LanguageModel hard-codes
 * BF16, BU/src/engine/language_model/mod.rs:74).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "uzu_oracle.h"

#include "uzu_oracle_model_internal.h"

void* orc_xcalloc(size_t n, size_t sz);
#define xcalloc orc_xcalloc
void* orc_xcalloc(size_t n, size_t sz) {
    void* p = calloc(n ? n : 1, sz);
    if (!p) {
        fprintf(stderr, "oracle: out of memory (%zu x %zu)\n", n, sz);
        abort();
    }
    return p;
}

orc_model* orc_model_create(const uzu_model_desc* desc) {
    orc_model* m = (orc_model*)xcalloc(1, sizeof(orc_model));
    m->desc = *desc;
    m->layers = (uzu_layer_desc*)xcalloc(desc->num_layers, sizeof(uzu_layer_desc));
    memcpy(m->layers, desc->layers, sizeof(uzu_layer_desc) * desc->num_layers);
    m->desc.layers = m->layers;
    if (desc->num_ropes) { /* own copy of the table of RoPE configurations (their factor arrays stay the caller's, like every tensor) */
        uzu_rope_desc* ropes = (uzu_rope_desc*)xcalloc(desc->num_ropes, sizeof(uzu_rope_desc));
        memcpy(ropes, desc->ropes, sizeof(uzu_rope_desc) * desc->num_ropes);
        m->desc.ropes = ropes;
    }
    m->states = (layer_state*)xcalloc(desc->num_layers, sizeof(layer_state));
    m->layer_outputs = (uint16_t**)xcalloc(desc->num_layers, sizeof(uint16_t*));
    m->final_hidden = (uint16_t*)xcalloc(desc->model_dim, 2);
    m->hidden_features = (uint16_t**)xcalloc(desc->num_layers, sizeof(uint16_t*));
    for (uint32_t l = 0; l < desc->num_layers; ++l) {
        const uzu_layer_desc* L = &m->layers[l];
        if (L->mixer_kind == UZU_MIXER_ATTENTION && L->is_kv_sharing) {
            /* TransformerLayerStateType::Shared(kv_source_layer_index) (transformer.rs:205-216): no state of its own */
            const uint32_t src = L->kv_source_layer_index;
            if (src >= l || m->layers[src].mixer_kind != UZU_MIXER_ATTENTION || m->layers[src].is_kv_sharing) {
                fprintf(stderr, "oracle: layer %u shares the KV state of layer %u, which is not an earlier attention layer owning its state\n", l, src);
                abort();
            }
            /* the core's ring / window specialisation comes from the layer's own config, the ring parameters from the state it reads
             * (mod.rs:166-198, core/single_pass.rs:60-70): only equal geometry is a meaningful configuration */
            if (m->layers[src].sliding_window_size != L->sliding_window_size || m->layers[src].num_groups != L->num_groups || m->layers[src].head_dim != L->head_dim) {
                fprintf(stderr, "oracle: layer %u and its KV source %u differ in window / kv heads / head_dim\n", l, src);
                abort();
            }
        } else if (L->mixer_kind == UZU_MIXER_ATTENTION) {
            /* AttentionState::create_empty (state.rs:69-136): causal + sliding window => Ring with max_length = the window */
            const size_t max_prefix = L->sliding_window_size ? L->sliding_window_size : desc->max_context_length;
            const size_t max_elements = max_prefix + ATTENTION_SUFFIX_CAPACITY;
            m->states[l].ring_max = L->sliding_window_size;
            const size_t element_size = (size_t)L->num_groups * L->head_dim;
            m->states[l].keys = (uint16_t*)xcalloc(max_elements * element_size, 2);
            m->states[l].values = (uint16_t*)xcalloc(max_elements * element_size, 2);
        } else {
            const size_t key_dim = (size_t)L->dn_num_groups * L->dn_head_dim;
            const size_t value_dim = (size_t)L->dn_num_heads * L->dn_value_head_dim;
            const size_t conv_dim = 2 * key_dim + value_dim;
            m->states[l].conv_state = (float*)xcalloc(conv_dim * (L->dn_kernel_size - 1), 4);
            m->states[l].ssm_state = (float*)xcalloc((size_t)L->dn_num_heads * L->dn_value_head_dim * L->dn_head_dim, 4);
        }
    }
    return m;
}

void orc_model_reset(orc_model* m) {
    m->context_length = 0;
    free(m->tree_parents);
    m->tree_parents = NULL, m->tree_size = 0; /* a pending tree is dropped (its per-layer buffers are replaced by the next tree pass) */
    for (uint32_t l = 0; l < m->desc.num_layers; ++l) {
        const uzu_layer_desc* L = &m->layers[l];
        m->states[l].length = 0;
        m->states[l].ring_offset = 0;
        if (L->mixer_kind == UZU_MIXER_DELTA_NET) {
            const size_t key_dim = (size_t)L->dn_num_groups * L->dn_head_dim;
            const size_t value_dim = (size_t)L->dn_num_heads * L->dn_value_head_dim;
            memset(m->states[l].conv_state, 0, (2 * key_dim + value_dim) * (L->dn_kernel_size - 1) * 4);
            memset(m->states[l].ssm_state, 0, (size_t)L->dn_num_heads * L->dn_value_head_dim * L->dn_head_dim * 4);
        }
    }
}

void orc_model_destroy(orc_model* m) {
    if (!m) return;
    for (uint32_t l = 0; l < m->desc.num_layers; ++l) {
        free(m->states[l].keys);
        free(m->states[l].values);
        free(m->states[l].conv_state);
        free(m->states[l].ssm_state);
        free(m->states[l].tree_conv_states);
        free(m->states[l].tree_k);
        free(m->states[l].tree_v);
        free(m->states[l].tree_log_decay);
        free(m->states[l].tree_beta);
        free(m->layer_outputs[l]);
        free(m->hidden_features[l]);
    }
    free(m->hidden_features);
    free(m->final_hidden_rows);
    free(m->tree_parents);
    free(m->layer_outputs);
    free(m->final_hidden);
    free(m->states);
    free(m->layers);
    if (m->desc.num_ropes) free((void*)m->desc.ropes);
    free(m);
}

/* Timing aid for bench.py's cpu_baseline leg: puts the model at context length `n` WITHOUT running a prefill -- the KV
 * rows get small deterministic values, the DeltaNet states stay as they are.  The cost of a decode step does not depend
 * on the data, only on the context length (KV rows scanned); results after this call are not reference results. */
void orc_model_fill_synthetic_context(orc_model* m, uint32_t n) {
    if (n > m->desc.max_context_length) n = m->desc.max_context_length;
    for (uint32_t l = 0; l < m->desc.num_layers; ++l) {
        const uzu_layer_desc* L = &m->layers[l];
        if (L->mixer_kind != UZU_MIXER_ATTENTION || L->is_kv_sharing) continue;
        const size_t element_size = (size_t)L->num_groups * L->head_dim;
        const uint32_t rows = m->states[l].ring_max && n > m->states[l].ring_max ? m->states[l].ring_max : n; /* a ring holds its window */
        for (size_t i = 0; i < (size_t)rows * element_size; ++i) {
            const uint32_t h = (uint32_t)(i * 2654435761u + l * 40503u);
            m->states[l].keys[i] = (uint16_t)(0x3C00u + (h >> 26));           /* bf16 in [0.0078, 0.0156) */
            m->states[l].values[i] = (uint16_t)(0xBC00u + ((h >> 20) & 63u)); /* small negative values */
        }
        m->states[l].length = rows;
        if (m->states[l].ring_max && n > m->states[l].ring_max) m->states[l].ring_offset = (n - m->states[l].ring_max) % m->states[l].ring_max;
    }
    m->context_length = n;
}

uint32_t orc_model_context_length(const orc_model* m) { return m->context_length; }
const uint16_t* orc_model_layer_output(const orc_model* m, uint32_t layer, uint32_t* rows) {
    if (rows) *rows = m->last_rows;
    return m->layer_outputs[layer];
}
const uint16_t* orc_model_final_hidden(const orc_model* m) { return m->final_hidden; }
void orc_model_capture_features(orc_model* m, uint32_t on) { m->capture_features = on; }
const uint16_t* orc_model_hidden_feature(const orc_model* m, uint32_t layer, uint32_t* rows) {
    if (rows) *rows = m->last_rows;
    return m->hidden_features[layer];
}
const uint16_t* orc_model_final_hidden_rows(const orc_model* m, uint32_t* rows) {
    if (rows) *rows = m->final_hidden_row_count;
    return m->final_hidden_rows;
}
const uzu_model_desc* orc_model_desc(const orc_model* m) { return &m->desc; }

/* Linear::encode -> MatmulKernel::encode with b_transpose = true (linear/matmul.rs:122-148) */
static void fp_matmul(const uint16_t* a, const uint16_t* b, uint16_t* d, uint32_t m, uint32_t n, uint32_t k, int accumulate) {
    orc_matmul_args g;
    memset(&g, 0, sizeof(g));
    g.a = a, g.a_dtype = ORC_BF16, g.b = b, g.w_dtype = ORC_BF16, g.method = UZU_QUANT_NONE, g.bits = 16, g.b_transpose = 1;
    g.d = d, g.d_dtype = ORC_BF16, g.ab_scale = 1.0f, g.accumulate = accumulate, g.m = m, g.n = n, g.k = k;
    orc_matmul(&g);
}

/* QLoRALinearWrapper::encode (linear/qlora_wrapper.rs:177-251): intermediate = x down^T (bf16); base input = InputRht(copy of x) when the
 * spec carries incoherence signs; output = base(x') (no bias, no output D-op); output += intermediate up^T (MatmulDOps::accumulate);
 * OutputRht in place. */
static uint16_t* linear_qlora(const uzu_linear_desc* lin, const uint16_t* input, uint32_t batch) {
    uint16_t* out = (uint16_t*)xcalloc((size_t)batch * lin->n, 2);
    uint16_t* intermediate = (uint16_t*)xcalloc((size_t)batch * lin->lora_rank, 2);
    fp_matmul(input, lin->adapter_down, intermediate, batch, lin->lora_rank, lin->k, 0);
    uint16_t* transformed = NULL;
    const uint16_t* base_input = input;
    if (lin->input_signs) {
        transformed = (uint16_t*)xcalloc((size_t)batch * lin->k, 2);
        orc_activation_transform(input, transformed, NULL, NULL, NULL, lin->input_signs, ORC_BF16, batch, lin->k, 0, 0, 0);
        base_input = transformed;
    }
    orc_matmul_args g;
    memset(&g, 0, sizeof(g));
    g.a = base_input, g.a_dtype = ORC_BF16, g.b = lin->weights, g.scales = lin->scales, g.biases = lin->biases, g.zero_points = lin->zero_points;
    g.w_dtype = ORC_BF16, g.method = lin->method, g.bits = lin->bits, g.group_size = lin->group_size, g.b_transpose = 1;
    g.d = out, g.d_dtype = ORC_BF16, g.ab_scale = 1.0f, g.m = batch, g.n = lin->n, g.k = lin->k;
    orc_matmul(&g);
    fp_matmul(intermediate, lin->adapter_up, out, batch, lin->n, lin->lora_rank, 1);
    if (lin->output_signs) orc_activation_transform(NULL, out, NULL, NULL, NULL, lin->output_signs, ORC_BF16, batch, lin->n, 1, 0, 0);
    free(transformed);
    free(intermediate);
    return out;
}

void* orc_linear_typed(const uzu_linear_desc* lin, const uint16_t* input, uint32_t batch, uint32_t d_dtype) {
    if (lin->lora_rank) {
        if (d_dtype != ORC_BF16) {
            fprintf(stderr, "oracle: a QLoRA linear with a non-bf16 output\n");
            abort();
        }
        return linear_qlora(lin, input, batch);
    }
    void* out = xcalloc((size_t)batch * lin->n, d_dtype == ORC_F32 ? 4 : 2);
    /* RHTLinearWrapper::encode_input (linear/rht_wrapper.rs:215-298), full-precision activation format: InputRht of the rows
     * (encode_fp_in_place on the wrapper's own allocation), the inner LinearMatmul with MatmulDOps::rht_factors = output signs */
    uint16_t* transformed = NULL;
    if (lin->input_signs) {
        transformed = (uint16_t*)xcalloc((size_t)batch * lin->k, 2);
        orc_activation_transform(input, transformed, NULL, NULL, NULL, lin->input_signs, ORC_BF16, batch, lin->k, 0, 0, 0);
        input = transformed;
    }
    orc_matmul_args g;
    memset(&g, 0, sizeof(g));
    g.rht_factors = lin->output_signs;
    g.a = input;
    g.a_dtype = ORC_BF16;
    g.b = lin->weights;
    g.scales = lin->scales;
    g.biases = lin->biases;
    g.zero_points = lin->zero_points;
    g.w_dtype = ORC_BF16;
    g.method = lin->method;
    g.bits = lin->bits;
    g.group_size = lin->group_size;
    g.b_transpose = 1;
    g.d = out;
    g.d_dtype = d_dtype;
    g.ab_scale = 1.0f;
    g.bias = lin->out_biases;
    g.m = batch;
    g.n = lin->n;
    g.k = lin->k;
    orc_matmul(&g);
    free(transformed);
    return out;
}
uint16_t* orc_linear(const uzu_linear_desc* lin, const uint16_t* input, uint32_t batch) { return (uint16_t*)orc_linear_typed(lin, input, batch, ORC_BF16); }
#define linear orc_linear

/* Normalization::encode (encodable_block/normalization.rs:114-146); mode: 0 none, 1 copy, 2 add */
/* scalar_mode: PostLayerScalar (normalization.rs:17-21,76-80): 0 None, 1 ScaleResidualSum(scalar), 2 ScaleOutput(scalar) */
static uint16_t* norm_scaled(const uzu_norm_desc* nd, const uint16_t* input, uint16_t* shortcut, int mode, uint32_t rows, uint32_t dim,
                             int scalar_mode, float scalar, float epsilon);
uint16_t* orc_norm(const uzu_norm_desc* nd, const uint16_t* input, uint16_t* shortcut, int mode, uint32_t rows, uint32_t dim) {
    return norm_scaled(nd, input, shortcut, mode, rows, dim, 0, 1.0f, nd->epsilon);
}
#define norm orc_norm
static uint16_t* norm_scaled(const uzu_norm_desc* nd, const uint16_t* input, uint16_t* shortcut, int mode, uint32_t rows, uint32_t dim,
                             int scalar_mode, float scalar, float epsilon) {
    uint16_t* out = (uint16_t*)xcalloc((size_t)rows * dim, 2);
    orc_norm_args g;
    memset(&g, 0, sizeof(g));
    g.input = input;
    g.scales = nd->scales;
    g.biases = nd->biases;
    g.output = out;
    g.shortcut = mode ? shortcut : NULL;
    g.io_dtype = ORC_BF16;
    g.affine_dtype = ORC_F32;
    g.batch_size = rows;
    g.element_count = dim;
    g.epsilon = epsilon;
    g.scale_offset = nd->scale_offset;
    g.post_layer_scalar = scalar_mode ? scalar : 1.0f;
    g.scale_residual_sum = scalar_mode == 1;
    g.scale_output = scalar_mode == 2;
    g.subtract_mean = nd->subtract_mean;
    g.full_layer = nd->full_layer;
    g.copy_to_shortcut = mode != 0;
    g.residual_add = mode == 2;
    orc_normalization(&g);
    return out;
}

/* the RoPE configuration of layer l: uzu_model_desc.ropes[rope_index] when the model carries several (transformer.rs:101-118) */
static const uzu_rope_desc* layer_rope(const uzu_model_desc* D, const uzu_layer_desc* L) {
    return D->num_ropes ? &D->ropes[L->rope_index] : &D->rope;
}

static uint16_t* attention_mixer(orc_model* m, uint32_t l, uint16_t* hidden, uint32_t batch, const float* cosines,
                                 const float* sines, const uint32_t* trie) {
    const uzu_layer_desc* L = &m->layers[l];
    /* MaybeMut::Const(owned layer's state) for a sharing layer (transformer.rs:264-275) */
    layer_state* st = &m->states[L->is_kv_sharing ? L->kv_source_layer_index : l];
    const uint32_t hd = L->head_dim, nq = L->num_heads, nkv = L->num_groups;
    const uint32_t nkv_proj = L->is_kv_sharing ? 0 : nkv; /* num_kv_heads = (!is_kv_sharing).then_some(num_groups) (mod.rs:80) */
    /* gate projection first, from a copy of hidden (mode.rs:54-61) */
    uint16_t* gate = NULL;
    if (L->has_gate) gate = linear(&L->gate_projection, hidden, batch);
    uint16_t* qkv = linear(&L->qkv_projection, hidden, batch);
    const uint32_t total_heads = nq + 2 * nkv_proj;
    /* QKVNorm::encode_packed (qkv_norm.rs:137-175): query, key, value heads; key / value norms are dropped with KV sharing (mod.rs:135-137),
     * head_count == 0 skips a head group */
    if (L->query_norm.present)
        orc_qkv_norm(qkv, ORC_BF16, L->query_norm.scales, batch, total_heads, hd, L->query_norm.epsilon,
                     L->query_norm.scale_offset, 0, nq, L->query_norm.full_layer);
    if (L->key_norm.present && nkv_proj)
        orc_qkv_norm(qkv, ORC_BF16, L->key_norm.scales, batch, total_heads, hd, L->key_norm.epsilon,
                     L->key_norm.scale_offset, nq, nkv, L->key_norm.full_layer);
    if (L->normalize_values && nkv_proj) /* AttentionConfig::value_norm_config (config/token_mixer/attention.rs:32-42): eps 1e-6, FullLayer, no scales */
        orc_qkv_norm(qkv, ORC_BF16, NULL, batch, total_heads, hd, 1e-6f, 0.0f, nq + nkv, nkv, 1);
    /* prepare_kv_and_queries (mode.rs:200-232): kv_token_offset = physical_prefix_length; prepare_queries (mode.rs:234-259) with KV sharing */
    uint16_t* queries = (uint16_t*)xcalloc((size_t)nq * batch * hd, 2);
    const uint32_t rope_dim = L->use_rope ? layer_rope(&m->desc, L)->head_dim : 0;
    /* kv_token_offset = state_type.physical_prefix_length(): the length of a Full cache, max_length of a Ring (state.rs:26-37) */
    const uint32_t physical_prefix = st->ring_max ? st->ring_max : st->length;
    orc_attention_prepare(qkv, queries, st->keys, st->values, cosines, sines, nq, nkv, hd, rope_dim, physical_prefix, batch, nkv_proj != 0);
    free(qkv);
    /* AttentionCores::encode (core/mod.rs:81-93) */
    orc_attention_args a;
    memset(&a, 0, sizeof(a));
    a.queries = queries;
    a.keys = st->keys;
    a.values = st->values;
    a.dtype = ORC_BF16;
    a.head_dim = hd;
    a.gqa_factor = nq / nkv;
    a.sequence_length = physical_prefix + batch; /* core/single_pass.rs:60 */
    if (st->ring_max) { /* ring_params (state.rs:39-54), sliding window (core/mod.rs:17-28) */
        a.is_kv_cache_ring = 1, a.ring_offset = st->ring_offset, a.ring_length = st->length;
        a.is_sliding_window = 1, a.sliding_window_size = L->sliding_window_size; /* the layer's own core arguments (mod.rs:166-198) */
    }
    if (L->has_sinks) a.sinks = L->sinks;
    a.k_head_stride = hd;
    a.k_seq_stride = nkv * hd;
    a.v_head_stride = hd;
    a.v_seq_stride = nkv * hd;
    a.scale = L->attention_scale != 0.0f ? L->attention_scale : 1.0f / sqrtf((float)hd);
    a.num_heads = nq;
    a.suffix_length = batch;
    a.is_causal = L->is_non_causal ? 0 : 1; /* AttentionConfig::is_causal (mod.rs:166-198): a DFlash draft block attends to all of its rows */
    a.trie = trie; /* mode.rs:178-192: a non-flat batch runs the is_trie cores with the nodes as the mask's suffix topology */
    uint16_t* out = (uint16_t*)xcalloc((size_t)batch * nq * hd, 2);
    if (physical_prefix + batch > 1024) { /* core/mod.rs:89-92 */
        const size_t rows = (size_t)batch * nq;
        float* partials = (float*)xcalloc(rows * 32 * hd, 4);
        float* sums = (float*)xcalloc(rows * 32, 4);
        float* maxs = (float*)xcalloc(rows * 32, 4);
        orc_attention_two_pass1(&a, partials, sums, maxs);
        orc_attention_two_pass2(partials, sums, maxs, out, ORC_BF16, hd, nq, batch);
        free(partials);
        free(sums);
        free(maxs);
    } else {
        orc_attention_single_pass(&a, out);
    }
    free(queries);
    if (gate) {
        orc_sigmoid_gate(gate, out, ORC_BF16, batch * nq * hd);
        free(gate);
    }
    uint16_t* projected = linear(&L->out_projection, out, batch);
    free(out);
    return projected;
}

/* DeltaNet::encode_tree_verify (delta_net.rs:334-437) with the Metal composition of DeltaNetTreeVerify::encode
 * (metal/kernel/gdn/tree_verify.rs:92-187: prefix -> gram -> solve -> out, batch 1, scale 1, h0 = the ssm state, slot 0). */
static uint16_t* delta_net_tree_verify(orc_model* m, uint32_t l, uint16_t* in_projected, uint32_t tree, const uint32_t* trie, const int32_t* parents) {
    const uzu_layer_desc* L = &m->layers[l];
    layer_state* st = &m->states[l];
    const uint32_t Hv = L->dn_num_heads, Hk = L->dn_num_groups, Dk = L->dn_head_dim, Dv = L->dn_value_head_dim;
    const uint32_t key_dim = Hk * Dk, value_dim = Hv * Dv, conv_dim = 2 * key_dim + value_dim;
    const uint32_t total_proj_dim = conv_dim + value_dim + 2 * Hv;
    const uint32_t ks = L->dn_kernel_size;
    free(st->tree_conv_states), free(st->tree_k), free(st->tree_v), free(st->tree_log_decay), free(st->tree_beta);
    st->tree_conv_states = (float*)xcalloc((size_t)tree * conv_dim * (ks - 1), 4);
    st->tree_k = (uint16_t*)xcalloc((size_t)tree * key_dim, 2);
    st->tree_v = (uint16_t*)xcalloc((size_t)tree * value_dim, 2);
    st->tree_beta = (float*)xcalloc((size_t)tree * Hv, 4);
    st->tree_log_decay = (float*)xcalloc((size_t)tree * Hv, 4);
    uint16_t* tree_projected = (uint16_t*)xcalloc((size_t)tree * total_proj_dim, 2);
    uint16_t* q = (uint16_t*)xcalloc((size_t)tree * key_dim, 2);
    orc_conv_tree_scan(in_projected, L->dn_conv_weights, L->dn_conv_biases, st->conv_state, parents, tree_projected, st->tree_conv_states, ORC_BF16,
                       tree, ks, total_proj_dim, conv_dim);
    orc_delta_net_tree_prep(tree_projected, L->dn_a_log, L->dn_dt_bias, q, st->tree_k, st->tree_v, st->tree_beta, st->tree_log_decay, ORC_BF16, Hv,
                            Hk, Dk, key_dim, value_dim, tree);
    const uint32_t nb = (tree + 15) / 16, ncp = (nb + 1) / 2;
    const int32_t h0_idx = 0;
    float* prefix = (float*)xcalloc((size_t)tree * Hv, 4);
    float* a_packed = (float*)xcalloc((size_t)Hv * nb * ncp * 16 * 32, 4);
    float* qkd = (float*)xcalloc((size_t)Hv * tree * tree, 4);
    float* a_inv = (float*)xcalloc((size_t)Hv * nb * 16 * 16, 4);
    float* kh0 = (float*)xcalloc((size_t)tree * Hv * Dv, 4);
    float* u = (float*)xcalloc((size_t)Hv * tree * Dv, 4);
    uint16_t* delta_output = (uint16_t*)xcalloc((size_t)tree * value_dim, 2);
    orc_build_tree_prefix(trie, st->tree_log_decay, prefix, 1, tree, Hv);
    orc_build_tree_gram(q, st->tree_k, ORC_BF16, trie, prefix, st->tree_beta, st->ssm_state, &h0_idx, a_packed, qkd, a_inv, kh0, 1.0f, 1, tree, Hk, Hv, Dk, Dv);
    orc_tree_update_solve(kh0, st->tree_v, ORC_BF16, prefix, st->tree_beta, a_packed, a_inv, &h0_idx, u, 1, tree, Hv, Dv);
    orc_build_tree_out(q, ORC_BF16, prefix, qkd, u, st->ssm_state, &h0_idx, delta_output, ORC_BF16, 1.0f, 1, tree, Hk, Hv, Dk, Dv);
    free(prefix), free(a_packed), free(qkd), free(a_inv), free(kh0), free(u), free(q);
    orc_delta_net_norm_gate(delta_output, tree_projected, L->dn_norm_scales, Hv, Dv, value_dim, conv_dim, total_proj_dim, L->dn_norm_epsilon, tree);
    free(tree_projected);
    uint16_t* projected = linear(&L->dn_out_proj, delta_output, tree);
    free(delta_output);
    return projected;
}

static uint16_t* delta_net_mixer(orc_model* m, uint32_t l, uint16_t* hidden, uint32_t batch, const uint32_t* trie, const int32_t* parents) {
    const uzu_layer_desc* L = &m->layers[l];
    layer_state* st = &m->states[l];
    const uint32_t Hv = L->dn_num_heads, Hk = L->dn_num_groups, Dk = L->dn_head_dim, Dv = L->dn_value_head_dim;
    const uint32_t key_dim = Hk * Dk, value_dim = Hv * Dv, conv_dim = 2 * key_dim + value_dim;
    const uint32_t total_proj_dim = conv_dim + value_dim + 2 * Hv;
    const uint32_t ks = L->dn_kernel_size;
    uint16_t* in_projected = linear(&L->dn_in_proj, hidden, batch);
    if (trie) { /* !batch_dim.full_accept() (delta_net.rs:496-502) */
        uint16_t* out = delta_net_tree_verify(m, l, in_projected, batch, trie, parents);
        free(in_projected);
        return out;
    }
    uint16_t* delta_output = (uint16_t*)xcalloc((size_t)batch * value_dim, 2);
    if (batch == 1) {
        orc_delta_net_conv_update(L->dn_conv_weights, L->dn_conv_biases, in_projected, st->conv_state, ks, conv_dim, ks - 1);
        orc_delta_net_update(in_projected, L->dn_a_log, L->dn_dt_bias, L->dn_norm_scales, st->ssm_state, delta_output, Hv,
                             Hk, Dk, Dv, key_dim, value_dim, L->dn_norm_epsilon);
    } else {
        /* delta_net.rs:533-636: conv_pack -> conv_scan -> prefill_prep -> prefill -> norm_gate.
         * NB: `padded` rows are total_proj_dim wide; only the first conv_dim channels are packed. */
        float* padded = (float*)xcalloc((size_t)(batch + ks - 1) * total_proj_dim, 4);
        orc_conv1d_pack(st->conv_state, in_projected, padded, ks - 1, total_proj_dim, batch, conv_dim);
        orc_delta_net_conv_scan(padded, L->dn_conv_weights, L->dn_conv_biases, in_projected, st->conv_state, batch, ks,
                                total_proj_dim, ks - 1, conv_dim, total_proj_dim);
        free(padded);
        float* qn = (float*)xcalloc((size_t)batch * key_dim, 4);
        float* kn = (float*)xcalloc((size_t)batch * key_dim, 4);
        float* beta = (float*)xcalloc((size_t)batch * Hv, 4);
        float* decay = (float*)xcalloc((size_t)batch * Hv, 4);
        orc_delta_net_prefill_prep(in_projected, L->dn_a_log, L->dn_dt_bias, qn, kn, beta, decay, Hv, Hk, Dk, key_dim,
                                   value_dim, batch);
        orc_delta_net_prefill(qn, kn, beta, decay, in_projected, st->ssm_state, delta_output, Hv, Hk, Dk, Dv, key_dim,
                              value_dim, batch);
        free(qn);
        free(kn);
        free(beta);
        free(decay);
        orc_delta_net_norm_gate(delta_output, in_projected, L->dn_norm_scales, Hv, Dv, value_dim, conv_dim, total_proj_dim,
                                L->dn_norm_epsilon, batch);
    }
    free(in_projected);
    uint16_t* projected = linear(&L->dn_out_proj, delta_output, batch);
    free(delta_output);
    return projected;
}

uint16_t* orc_layers_forward(orc_model* m, uint16_t* hidden, uint32_t count, const uint32_t* trie, const int32_t* parents, const uint16_t* per_layer_inputs,
                             uint16_t** shortcut_out) {
    const uzu_model_desc* D = &m->desc;
    const uint32_t d = D->model_dim;
    uint16_t* shortcut = (uint16_t*)xcalloc((size_t)count * d, 2);
    /* host RoPE tables for this pass, one pair per distinct configuration (transformer.rs:247-257) */
    const uint32_t n_ropes = D->num_ropes ? D->num_ropes : (D->rope.kind != UZU_ROPE_NONE ? 1u : 0u);
    float** cos_tabs = (float**)xcalloc(n_ropes, sizeof(float*));
    float** sin_tabs = (float**)xcalloc(n_ropes, sizeof(float*));
    if (n_ropes) {
        uint32_t* pos = (uint32_t*)xcalloc(count, 4);
        for (uint32_t i = 0; i < count; ++i) pos[i] = m->context_length + (trie ? trie[3 * i + 2] : i); /* transformer.rs:247: context + height */
        for (uint32_t r = 0; r < n_ropes; ++r) {
            const uzu_rope_desc* R = D->num_ropes ? &D->ropes[r] : &D->rope;
            cos_tabs[r] = (float*)xcalloc((size_t)count * R->head_dim, 4);
            sin_tabs[r] = (float*)xcalloc((size_t)count * R->head_dim, 4);
            orc_rope_tables(R, pos, count, cos_tabs[r], sin_tabs[r]);
        }
        free(pos);
    }
    m->last_rows = count;
    for (uint32_t l = 0; l < D->num_layers; ++l) {
        const uzu_layer_desc* L = &m->layers[l];
        uint16_t* h;
        if (L->pre_mixer_norm.present) {
            h = norm(&L->pre_mixer_norm, hidden, shortcut, l > 0 ? 2 : 1, count, d);
            free(hidden);
        } else {
            memcpy(shortcut, hidden, (size_t)count * d * 2);
            h = hidden;
        }
        const uint32_t ri = D->num_ropes ? L->rope_index : 0;
        const float* cosines = L->use_rope && n_ropes ? cos_tabs[ri] : NULL;
        const float* sines = L->use_rope && n_ropes ? sin_tabs[ri] : NULL;
        /* transformer_layer.rs:61-84: the scalar belongs to the two norms unless a PLE projection owns it */
        const int norms_scale = L->has_post_layer_scalar && !L->has_ple;
        uint16_t* mixed = L->mixer_kind == UZU_MIXER_ATTENTION ? attention_mixer(m, l, h, count, cosines, sines, trie)
                                                               : delta_net_mixer(m, l, h, count, trie, parents);
        free(h);
        if (L->post_mixer_norm.present) {
            uint16_t* t = norm(&L->post_mixer_norm, mixed, NULL, 0, count, d);
            free(mixed);
            mixed = t;
        }
        uint16_t* mlp_in = norm_scaled(&L->pre_mlp_norm, mixed, shortcut, 2, count, d, norms_scale ? 1 : 0, L->post_layer_scalar, L->pre_mlp_norm.epsilon);
        free(mixed);
        uint16_t* down;
        if (L->mlp_kind == UZU_MLP_MOE) { /* MoeBlock::encode (mlp/moe/mod.rs:204-350): uzu_oracle_moe.c */
            down = orc_moe_block(&L->moe, d, mlp_in, count);
            free(mlp_in);
        } else {
            /* DenseMlp (mlp/dense.rs:32-48): up -> GatedActMul(interleaved) -> down */
            uint16_t* fused_up = linear(&L->up_projection, mlp_in, count);
            free(mlp_in);
            uint16_t* gated = (uint16_t*)xcalloc((size_t)count * L->hidden_dim, 2);
            orc_gated_act_mul(fused_up, NULL, gated, ORC_BF16, L->hidden_dim, count, 0, 0, L->activation, 1);
            free(fused_up);
            down = linear(&L->down_projection, gated, count);
            free(gated);
        }
        if (L->post_mlp_norm.present) {
            uint16_t* t = norm_scaled(&L->post_mlp_norm, down, NULL, 0, count, d, norms_scale ? 2 : 0, L->post_layer_scalar, L->post_mlp_norm.epsilon);
            free(down);
            down = t;
        }
        if (L->has_ple) {
            /* PerLayerEmbeddingProjection::encode (per_layer_embedding.rs:217-270): shortcut += hidden; gate(shortcut) -> act(gate) * the
             * layer's slice of per_layer_inputs -> projection -> norm; shortcut = (shortcut + normed) * post_layer_scalar; hidden = 0 */
            const uint32_t length = count * d;
            orc_tensor_add_bias(NULL, down, shortcut, ORC_BF16, ORC_BF16, length, length);
            uint16_t* gate_out = linear(&L->ple_gate, shortcut, count);
            uint16_t* activated = (uint16_t*)xcalloc((size_t)count * L->ple_dim, 2);
            orc_gated_act_mul(gate_out, per_layer_inputs, activated, ORC_BF16, L->ple_dim, count, l * L->ple_dim, D->num_layers * L->ple_dim, L->ple_activation, 0);
            free(gate_out);
            uint16_t* projected = linear(&L->ple_projection, activated, count);
            free(activated);
            uint16_t* normed = norm(&L->ple_norm, projected, NULL, 0, count, d);
            free(projected);
            orc_tensor_add_scale(NULL, normed, shortcut, ORC_BF16, length, length, L->has_post_layer_scalar ? L->post_layer_scalar : 1.0f);
            free(normed);
            memset(down, 0, (size_t)length * 2); /* encoder.encode_fill(&mut hidden, 0) (transformer_layer.rs:231) */
        }
        hidden = down;
        if (m->capture_features) { /* Transformer::capture_residual (transformer.rs:160-171): TensorAddScale(shortcut, hidden, 1.0) */
            free(m->hidden_features[l]);
            m->hidden_features[l] = (uint16_t*)xcalloc((size_t)count * d, 2);
            orc_tensor_add_scale(shortcut, hidden, m->hidden_features[l], ORC_BF16, count * d, count * d, 1.0f);
        }
        free(m->layer_outputs[l]);
        m->layer_outputs[l] = (uint16_t*)xcalloc((size_t)count * d, 2);
        /* (a PLE layer has folded its output into the shortcut and zeroed `hidden`: its tap is the residual row, what capture_residual would file) */
        memcpy(m->layer_outputs[l], L->has_ple ? shortcut : hidden, (size_t)count * d * 2);
    }
    for (uint32_t r = 0; r < n_ropes; ++r) free(cos_tabs[r]), free(sin_tabs[r]);
    free(cos_tabs);
    free(sin_tabs);
    *shortcut_out = shortcut;
    return hidden;
}

/* `trie` == NULL: a flat, fully accepted pass (prefill chunk / decode step): logits + greedy token of the LAST row, then encode_accept.
 * `trie` != NULL (3 u32 per node): a speculated tree, not accepted (stream.rs:556-566,618-628): token positions = context + height, logits
 * and greedy tokens of EVERY row (output_range 0..size), the states keep the unaccepted suffix for orc_model_accept. */
static uint32_t forward_core(orc_model* m, const uint32_t* token_ids, uint32_t count, const uint32_t* trie, uint16_t* logits_out, uint32_t* tokens_out) {
    const uzu_model_desc* D = &m->desc;
    const uint32_t d = D->model_dim;
    if (m->tree_size) {
        fprintf(stderr, "oracle: forward with an unaccepted tree pending\n");
        abort();
    }
    int32_t* parents = NULL;
    if (trie) { /* BatchTopology::new (batch_topology.rs:11-37) */
        parents = (int32_t*)xcalloc(count, 4);
        uint32_t* stack = (uint32_t*)xcalloc(count + 1, 4);
        uint32_t depth = 0;
        for (uint32_t i = 0; i < count; ++i) {
            const uint32_t height = trie[3 * i + 2];
            if (height > depth) {
                fprintf(stderr, "oracle: trie node %u at height %u does not follow a node of height >= %u\n", i, height, height - 1);
                abort();
            }
            depth = height;
            parents[i] = depth ? (int32_t)stack[depth - 1] : -1;
            stack[depth++] = i;
        }
        free(stack);
    }
    if (count == 0 || count > ATTENTION_SUFFIX_CAPACITY) {
        fprintf(stderr, "oracle: forward chunk must be 1..1024 tokens\n");
        abort();
    }
    /* Embedding::encode_lookup (embedding.rs:345-372) */
    uint16_t* hidden = (uint16_t*)xcalloc((size_t)count * d, 2);
    if (D->embedding.method == UZU_QUANT_NONE)
        orc_full_precision_embedding_lookup(token_ids, D->embedding.weights, hidden, ORC_BF16, count, D->vocab_size, d,
                                            D->input_scale);
    else
        orc_quantized_embedding_lookup(token_ids, (const uint8_t*)D->embedding.weights, D->embedding.scales,
                                       D->embedding.zero_points, D->embedding.biases, hidden, ORC_BF16, count,
                                       D->vocab_size, d, D->input_scale, D->embedding.group_size, D->embedding.bits,
                                       D->embedding.method);
    /* EmbeddingTable with output Hadamard factors (embedding_table.rs:34-125, quant_embedding.metal use_hadamard): OutputRht of the row */
    if (D->embedding.output_signs)
        orc_activation_transform(NULL, hidden, NULL, NULL, NULL, D->embedding.output_signs, ORC_BF16, count, d, 1, 0, 0);
    /* Decoder::encode (decoder.rs:149-154): the embedding norm, no shortcut */
    if (D->embedding_norm.present) {
        uint16_t* t = norm(&D->embedding_norm, hidden, NULL, 0, count, d);
        free(hidden);
        hidden = t;
    }
    /* PerLayerEmbedding::encode (per_layer_embedding.rs:108-147): per_layer_inputs [count, layers, ple_dim] =
     * token table row * (ple_embed_scale * input_scale) + projection_norm(model_projection(embedded)) [ScaleOutput(input_scale),
     * epsilon / model_projection_scale^2] */
    uint16_t* per_layer_inputs = NULL;
    if (D->has_ple) {
        const uint32_t total = D->num_layers * D->ple_dim;
        uint16_t* token_ple = (uint16_t*)xcalloc((size_t)count * total, 2);
        const float fused_token_scale = D->ple_embed_scale * D->ple_input_scale; /* per_layer_embedding.rs:103 */
        const uzu_linear_desc* T = &D->ple_token_embedding;
        if (T->method == UZU_QUANT_NONE)
            orc_full_precision_embedding_lookup(token_ids, T->weights, token_ple, ORC_BF16, count, D->ple_vocab_size, total, fused_token_scale);
        else
            orc_quantized_embedding_lookup(token_ids, (const uint8_t*)T->weights, T->scales, T->zero_points, T->biases, token_ple, ORC_BF16, count,
                                           D->ple_vocab_size, total, fused_token_scale, T->group_size, T->bits, T->method);
        uint16_t* projected = linear(&D->ple_model_projection, hidden, count); /* on a copy of the rows (:125-128): `hidden` stays as it is */
        const float s2 = D->ple_model_projection_scale * D->ple_model_projection_scale;
        uint16_t* normed = norm_scaled(&D->ple_projection_norm, projected, NULL, 0, count * D->num_layers, D->ple_dim, 2, D->ple_input_scale,
                                       D->ple_projection_norm.epsilon / s2);
        free(projected);
        per_layer_inputs = (uint16_t*)xcalloc((size_t)count * total, 2);
        orc_tensor_add_scale(token_ple, normed, per_layer_inputs, ORC_BF16, count * total, count * total, 1.0f);
        free(token_ple);
        free(normed);
    }
    uint16_t* shortcut = NULL;
    hidden = orc_layers_forward(m, hidden, count, trie, parents, per_layer_inputs, &shortcut);
    free(per_layer_inputs);
    /* output_norm over the output range, shortcut add (transformer.rs:317-323): the last row of a flat pass, every row of a tree */
    const uint32_t out_rows = trie ? count : 1;
    const size_t first = (size_t)(count - out_rows) * d;
    uint16_t* normed = norm(&D->output_norm, hidden + first, shortcut + first, 2, out_rows, d);
    memcpy(m->final_hidden, normed + (size_t)(out_rows - 1) * d, (size_t)d * 2);
    if (m->capture_features) {
        free(m->final_hidden_rows);
        m->final_hidden_rows = (uint16_t*)xcalloc((size_t)out_rows * d, 2);
        memcpy(m->final_hidden_rows, normed, (size_t)out_rows * d * 2);
        m->final_hidden_row_count = out_rows;
    }
    free(hidden);
    free(shortcut);
    /* Embedding::encode_readout (embedding.rs:374-456): readout_input_hadamard = the tied table's output signs (embedding.rs:167-173) or
     * the untied output embedding's input signs (embedding.rs:255-274): InputRht on a private copy of the row, then the plain matmul */
    uzu_linear_desc ro = D->tied_embeddings ? D->embedding : D->output_embedding;
    ro.input_signs = D->tied_embeddings ? D->embedding.output_signs : D->output_embedding.input_signs;
    ro.output_signs = NULL;
    uint16_t* logits = linear(&ro, normed, out_rows);
    free(normed);
    if (D->logit_scale != 1.0f || D->logit_soft_cap != 0.0f)
        orc_logit_transform(logits, ORC_BF16, D->vocab_size * out_rows, D->logit_scale, D->logit_soft_cap, D->logit_soft_cap != 0.0f);
    uint32_t* sampled = (uint32_t*)xcalloc(out_rows, 4);
    orc_argmax(logits, ORC_BF16, sampled, D->vocab_size, out_rows);
    const uint32_t token = sampled[out_rows - 1];
    if (tokens_out) memcpy(tokens_out, sampled, (size_t)out_rows * 4);
    free(sampled);
    if (logits_out) memcpy(logits_out, logits, (size_t)D->vocab_size * out_rows * 2);
    free(logits);
    if (trie) { /* not accepted: the suffix rows stay behind the caches' logical end, the DeltaNet layers hold their Tree status */
        m->tree_size = count;
        m->tree_parents = parents;
        return token;
    }
    /* encode_accept (state.rs:174-236), flat full accept: nothing to copy on a Full cache; a Ring takes the suffix rows in order */
    for (uint32_t l = 0; l < D->num_layers; ++l) {
        if (m->layers[l].mixer_kind != UZU_MIXER_ATTENTION || m->layers[l].is_kv_sharing) continue; /* Shared layer states are skipped (transformer.rs:63-69) */
        layer_state* st = &m->states[l];
        if (!st->ring_max) {
            st->length += count;
            continue;
        }
        orc_copy* copies = (orc_copy*)xcalloc(count, sizeof(orc_copy));
        for (uint32_t idx = 0; idx < count; ++idx) {
            copies[idx].source = st->ring_max + idx;
            copies[idx].destination = (st->ring_offset + st->length) % st->ring_max;
            if (st->length < st->ring_max) st->length += 1;
            else st->ring_offset = (st->ring_offset + 1) % st->ring_max;
        }
        orc_kv_cache_update(st->keys, st->values, ORC_BF16, copies, count, m->layers[l].num_groups * m->layers[l].head_dim);
        free(copies);
    }
    m->context_length += count;
    return token;
}

uint32_t orc_model_forward(orc_model* m, const uint32_t* token_ids, uint32_t count, uint16_t* logits_out) {
    return forward_core(m, token_ids, count, NULL, logits_out, NULL);
}

void orc_model_verify_tree(orc_model* m, const uint32_t* token_ids, const uint32_t* trie, uint32_t tree_size, uint32_t* sampled_out, uint16_t* logits_out) {
    forward_core(m, token_ids, tree_size, trie, logits_out, sampled_out);
}

/* TransformerState::encode_accept over the layers (stream.rs:441-444): attention caches compact the accepted rows
 * (mixer/attention/state.rs:174-236), DeltaNet layers take the accepted node's conv state and advance the SSM state along the
 * accepted path (delta_net.rs:65-120). */
void orc_model_accept(orc_model* m, const uint32_t* accepted, uint32_t n) {
    if (!m->tree_size || !n) {
        fprintf(stderr, "oracle: accept without a pending tree / of zero indices\n");
        abort();
    }
    for (uint32_t i = 0; i < n; ++i) { /* delta_net.rs:88-90, state.rs:179 */
        const int32_t want_parent = i ? (int32_t)accepted[i - 1] : -1;
        if (accepted[i] >= m->tree_size || m->tree_parents[accepted[i]] != want_parent) {
            fprintf(stderr, "oracle: accepted indices are not a root path of the tree\n");
            abort();
        }
    }
    for (uint32_t l = 0; l < m->desc.num_layers; ++l) {
        const uzu_layer_desc* L = &m->layers[l];
        layer_state* st = &m->states[l];
        if (L->mixer_kind == UZU_MIXER_ATTENTION && L->is_kv_sharing) continue; /* TransformerLayerStateType::Shared (transformer.rs:63-69) */
        if (L->mixer_kind == UZU_MIXER_ATTENTION) {
            orc_copy* copies = (orc_copy*)xcalloc(n, sizeof(orc_copy));
            uint32_t nc = 0;
            if (!st->ring_max) {
                for (uint32_t i = 0; i < n; ++i)
                    if (accepted[i] != i) copies[nc].source = st->length + accepted[i], copies[nc].destination = st->length + i, ++nc;
                st->length += n;
            } else {
                for (uint32_t i = 0; i < n; ++i) {
                    copies[nc].source = st->ring_max + accepted[i];
                    copies[nc].destination = (st->ring_offset + st->length) % st->ring_max;
                    ++nc;
                    if (st->length < st->ring_max) st->length += 1;
                    else st->ring_offset = (st->ring_offset + 1) % st->ring_max;
                }
            }
            if (nc) orc_kv_cache_update(st->keys, st->values, ORC_BF16, copies, nc, L->num_groups * L->head_dim);
            free(copies);
        } else {
            const uint32_t Hv = L->dn_num_heads, Hk = L->dn_num_groups, Dk = L->dn_head_dim, Dv = L->dn_value_head_dim;
            const size_t conv_state_elems = (size_t)(2 * Hk * Dk + Hv * Dv) * (L->dn_kernel_size - 1);
            memcpy(st->conv_state, st->tree_conv_states + (size_t)accepted[n - 1] * conv_state_elems, conv_state_elems * 4);
            orc_state_advance(st->tree_k, st->tree_v, ORC_BF16, st->tree_log_decay, st->tree_beta, accepted, st->ssm_state, n, Hv, Hk, Dk);
            free(st->tree_conv_states), free(st->tree_k), free(st->tree_v), free(st->tree_log_decay), free(st->tree_beta);
            st->tree_conv_states = NULL, st->tree_k = st->tree_v = NULL, st->tree_log_decay = st->tree_beta = NULL;
        }
    }
    free(m->tree_parents);
    m->tree_parents = NULL, m->tree_size = 0;
    m->context_length += n;
}
