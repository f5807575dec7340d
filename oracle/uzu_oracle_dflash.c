/*
 * uzu_oracle_dflash.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT).  See uzu_oracle.h.
 *
 * The DFlash draft model of the reference's tree speculator, restated from
 *   DFlash::{new, empty_state, encode_accept, encode_draft}   BU/src/encodable_block/dflash.rs:41-346
 *   Attention::append_projected_kv                            BU/src/encodable_block/mixer/attention/mode.rs:146-169
 *   QKVNorm::encode_key_value                                 BU/src/encodable_block/mixer/attention/qkv_norm.rs:128-175
 *   DFlashTfmSpeculator::propose_tree, Argmax construction    BU/src/speculators/dflash_tfm.rs:133-218 (the chain's greedy sampling; the
 *                                                             trie itself is host code: uzu_amd/speculator.py)
 * on top of the kernels of uzu_oracle_kernels.c and the layer loop of uzu_oracle_model.c (the draft layers are ordinary TransformerLayers:
 * the model core below is an orc_model without an embedding of its own -- the draft model looks its rows up in, and reads out through, the
 * TARGET's embedding, dflash.rs:285,335).
 *
 * PARITY STATUS: as for the rest of the oracle -- the reference holds no vectors for this block: **parity unpinned**; pinned indirectly by
 * tests/test_oracle_dflash.py (the accept path against a hand-composed chain of the oracle kernels, block attention against float64).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "uzu_oracle_model_internal.h"

struct orc_dflash {
    uzu_dflash_desc desc;
    uint32_t* target_layer_ids;
    orc_model* core; /* the draft layers + their attention states (AttentionState::Full, capacity + 1024 rows: empty_state, dflash.rs:174-188) */
    uint32_t layer_kv_dim; /* 2 * num_groups * head_dim of the first layer (dflash.rs:121) */
};

orc_dflash* orc_dflash_create(const uzu_dflash_desc* desc) {
    if (!desc->num_layers || !desc->layers || !desc->num_target_layers || desc->block_size > ATTENTION_SUFFIX_CAPACITY) {
        fprintf(stderr, "oracle: DFlash description without layers / target layers, or block_size beyond the attention suffix capacity\n");
        abort();
    }
    orc_dflash* f = (orc_dflash*)orc_xcalloc(1, sizeof(orc_dflash));
    f->desc = *desc;
    f->target_layer_ids = (uint32_t*)orc_xcalloc(desc->num_target_layers, 4);
    memcpy(f->target_layer_ids, desc->target_layer_ids, (size_t)desc->num_target_layers * 4);
    f->desc.target_layer_ids = f->target_layer_ids;
    for (uint32_t l = 0; l < desc->num_layers; ++l) {
        const uzu_layer_desc* L = &desc->layers[l];
        if (L->mixer_kind != UZU_MIXER_ATTENTION || L->is_kv_sharing || L->sliding_window_size || L->has_ple) { /* DFlashNewError::InvalidAttentionConfig (dflash.rs:118-120) */
            fprintf(stderr, "oracle: DFlash layers must use plain attention mixers with a full KV state\n");
            abort();
        }
    }
    if (desc->context_capacity > desc->rope.max_sequence_length) { /* dflash.rs:179 */
        fprintf(stderr, "oracle: DFlash state capacity exceeds configured RoPE capacity\n");
        abort();
    }
    f->layer_kv_dim = 2 * desc->layers[0].num_groups * desc->layers[0].head_dim;
    uzu_model_desc core;
    memset(&core, 0, sizeof(core));
    core.vocab_size = desc->vocab_size, core.model_dim = desc->model_dim, core.num_layers = desc->num_layers, core.tied_embeddings = 1;
    core.input_scale = 1.0f, core.logit_scale = 1.0f, core.max_context_length = desc->context_capacity;
    core.rope = desc->rope, core.output_norm = desc->output_norm, core.layers = desc->layers;
    f->core = orc_model_create(&core);
    return f;
}

void orc_dflash_destroy(orc_dflash* f) {
    if (!f) return;
    orc_model_destroy(f->core);
    free(f->target_layer_ids);
    free(f);
}

void orc_dflash_reset(orc_dflash* f) { orc_model_reset(f->core); }
uint32_t orc_dflash_context_length(const orc_dflash* f) { return orc_model_context_length(f->core); }

/* DFlash::encode_accept (dflash.rs:190-271): the accepted tokens' target features -> context projection -> norm -> per-layer key / value rows
 * appended to the draft layers' caches.  target_features[i] = the feature rows of target layer target_layer_ids[i], bf16 [rows, model_dim]. */
void orc_dflash_accept(orc_dflash* f, const uint16_t* const* target_features, const uint32_t* accepted_indices, uint32_t num_tokens) {
    if (!num_tokens) return; /* dflash.rs:197-199 */
    const uzu_dflash_desc* D = &f->desc;
    orc_model* core = f->core;
    const uint32_t d = D->model_dim, nf = D->num_target_layers, nl = D->num_layers;
    const uint32_t context_length = core->context_length;
    if (context_length + num_tokens > D->context_capacity) {
        fprintf(stderr, "oracle: DFlash state capacity exceeded\n");
        abort();
    }
    /* packed_target_features [num_tokens, nf * d]: token-major, the captured layers side by side (dflash.rs:214-229) */
    uint16_t* packed = (uint16_t*)orc_xcalloc((size_t)num_tokens * nf * d, 2);
    for (uint32_t layer_index = 0; layer_index < nf; ++layer_index)
        for (uint32_t token_index = 0; token_index < num_tokens; ++token_index)
            memcpy(packed + ((size_t)token_index * nf + layer_index) * d, target_features[layer_index] + (size_t)accepted_indices[token_index] * d, (size_t)d * 2);
    uint16_t* projected = orc_linear(&D->context_projection, packed, num_tokens);
    free(packed);
    uint16_t* normalized = orc_norm(&D->context_norm, projected, NULL, 0, num_tokens, d);
    free(projected);
    uint32_t* positions = (uint32_t*)orc_xcalloc(num_tokens, 4);
    for (uint32_t i = 0; i < num_tokens; ++i) positions[i] = context_length + i; /* dflash.rs:233 */
    float* cosines = (float*)orc_xcalloc((size_t)num_tokens * D->rope.head_dim, 4);
    float* sines = (float*)orc_xcalloc((size_t)num_tokens * D->rope.head_dim, 4);
    orc_rope_tables(&D->rope, positions, num_tokens, cosines, sines);
    free(positions);
    uint16_t* projected_kv = orc_linear(&D->state_kv_projection, normalized, num_tokens); /* [num_tokens, nl * layer_kv_dim] */
    free(normalized);
    const uint32_t kvd = f->layer_kv_dim;
    for (uint32_t l = 0; l < nl; ++l) {
        const uzu_layer_desc* L = &core->layers[l];
        layer_state* st = &core->states[l];
        const uint32_t nkv = L->num_groups, hd = L->head_dim;
        uint16_t* key_value = (uint16_t*)orc_xcalloc((size_t)num_tokens * kvd, 2); /* chunk (token * nl + layer) -> row token (dflash.rs:242-254) */
        for (uint32_t t = 0; t < num_tokens; ++t) memcpy(key_value + (size_t)t * kvd, projected_kv + ((size_t)t * nl + l) * kvd, (size_t)kvd * 2);
        /* Attention::append_projected_kv (mode.rs:146-169): QKVNorm::encode_key_value = encode_packed with q_heads = 0 (qkv_norm.rs:128-175) */
        if (L->key_norm.present)
            orc_qkv_norm(key_value, ORC_BF16, L->key_norm.scales, num_tokens, 2 * nkv, hd, L->key_norm.epsilon, L->key_norm.scale_offset, 0, nkv, L->key_norm.full_layer);
        if (L->normalize_values) orc_qkv_norm(key_value, ORC_BF16, NULL, num_tokens, 2 * nkv, hd, 1e-6f, 0.0f, nkv, nkv, 1);
        /* prepare_kv_and_queries with num_q_heads = 0: RoPE on the keys, rows written at the cache's physical prefix length */
        uint16_t dummy_queries[1] = {0};
        const uint32_t rope_dim = L->use_rope ? D->rope.head_dim : 0;
        orc_attention_prepare(key_value, dummy_queries, st->keys, st->values, cosines, sines, 0, nkv, hd, rope_dim, st->length, num_tokens, 1);
        free(key_value);
        st->length += num_tokens; /* state.encode_accept(0 .. batch_dim) on a Full cache: no copies (state.rs:174-198) */
    }
    free(projected_kv);
    free(cosines);
    free(sines);
    core->context_length += num_tokens;
}

/* DFlash::encode_draft (dflash.rs:273-345): `batch_size` rows = [target_output_token, mask, mask, ...] embedded with the TARGET's table, through the draft
 * layers over context + block (nothing is accepted: the block's key / value rows stay behind the caches' logical end and are overwritten by the next
 * draft), output norm of every row -> draft_hidden bf16 [batch_size, d]; rows 1.. through the target's read-out in f32 -> logits f32 [batch_size - 1, vocab].
 * Returns, like the Argmax construction of propose_tree (dflash_tfm.rs:167-217), the greedy token of every lookahead row in tokens_out [batch_size - 1]. */
void orc_dflash_draft(orc_dflash* f, const orc_model* target, uint32_t target_output_token, uint32_t batch_size, uint16_t* draft_hidden_out, float* logits_out,
                      uint32_t* tokens_out) {
    const uzu_dflash_desc* D = &f->desc;
    const uzu_model_desc* T = orc_model_desc(target);
    orc_model* core = f->core;
    const uint32_t d = D->model_dim;
    if (batch_size < 2 || batch_size > D->block_size || core->context_length + batch_size > D->rope.max_sequence_length || T->model_dim != d) { /* dflash.rs:283-287 */
        fprintf(stderr, "oracle: DFlash draft of %u rows (block size %u) at context %u\n", batch_size, D->block_size, core->context_length);
        abort();
    }
    uint32_t* tokens = (uint32_t*)orc_xcalloc(batch_size, 4);
    for (uint32_t i = 0; i < batch_size; ++i) tokens[i] = D->mask_token_id;
    tokens[0] = target_output_token;
    /* Embedding::encode_lookup of the target (embedding.rs:345-372; the embedding NORM is the decoder's, decoder.rs:149-154, and is not applied here) */
    uint16_t* hidden = (uint16_t*)orc_xcalloc((size_t)batch_size * d, 2);
    if (T->embedding.method == UZU_QUANT_NONE)
        orc_full_precision_embedding_lookup(tokens, T->embedding.weights, hidden, ORC_BF16, batch_size, T->vocab_size, d, T->input_scale);
    else
        orc_quantized_embedding_lookup(tokens, (const uint8_t*)T->embedding.weights, T->embedding.scales, T->embedding.zero_points, T->embedding.biases, hidden, ORC_BF16,
                                       batch_size, T->vocab_size, d, T->input_scale, T->embedding.group_size, T->embedding.bits, T->embedding.method);
    if (T->embedding.output_signs) orc_activation_transform(NULL, hidden, NULL, NULL, NULL, T->embedding.output_signs, ORC_BF16, batch_size, d, 1, 0, 0);
    free(tokens);
    /* flat topology {start i, end batch - 1, height i}, positions context .. context + batch (dflash.rs:296-306): the layer loop of the model core */
    uint16_t* shortcut = NULL;
    hidden = orc_layers_forward(core, hidden, batch_size, NULL, NULL, NULL, &shortcut);
    uint16_t* draft_hidden = orc_norm(&D->output_norm, hidden, shortcut, 2, batch_size, d); /* ShortcutMode::Add (dflash.rs:148-157,325-328) */
    free(hidden);
    free(shortcut);
    if (draft_hidden_out) memcpy(draft_hidden_out, draft_hidden, (size_t)batch_size * d * 2);
    /* Embedding::encode_readout(batch_size - 1, lookahead rows, DataType::F32) (dflash.rs:330-335; embedding.rs:374-456) */
    uzu_linear_desc ro = T->tied_embeddings ? T->embedding : T->output_embedding;
    ro.input_signs = T->tied_embeddings ? T->embedding.output_signs : T->output_embedding.input_signs;
    ro.output_signs = NULL;
    const uint32_t rows = batch_size - 1;
    float* logits = (float*)orc_linear_typed(&ro, draft_hidden + d, rows, ORC_F32);
    free(draft_hidden);
    if (T->logit_scale != 1.0f || T->logit_soft_cap != 0.0f) orc_logit_transform(logits, ORC_F32, T->vocab_size * rows, T->logit_scale, T->logit_soft_cap, T->logit_soft_cap != 0.0f);
    if (logits_out) memcpy(logits_out, logits, (size_t)rows * T->vocab_size * 4);
    if (tokens_out) orc_argmax(logits, ORC_F32, tokens_out, T->vocab_size, rows); /* Sampling::new(DataType::F32, vocab), Greedy (dflash_tfm.rs:112,189-203) */
    free(logits);
}
