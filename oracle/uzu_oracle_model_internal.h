/*
 * uzu_oracle_model_internal.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT): what uzu_oracle_model.c shares with uzu_oracle_dflash.c.
 */
#ifndef UZU_ORACLE_MODEL_INTERNAL_H
#define UZU_ORACLE_MODEL_INTERNAL_H

#include "uzu_oracle.h"

#define ATTENTION_SUFFIX_CAPACITY 1024u /* mixer/attention/state.rs:14 */

typedef struct {
    uint16_t* keys;   /* bf16 [max_ctx + 1024, kv_heads*hd] */
    uint16_t* values;
    uint32_t length;  /* AttentionStateType::Full { length } | Ring { length } */
    uint32_t ring_offset, ring_max; /* AttentionStateType::Ring { offset, .., max_length } (ring_max == 0: Full), state.rs:16-55 */
    float* conv_state; /* f32 [conv_dim, k-1] */
    float* ssm_state;  /* f32 [Hv, Dv, Dk] */
    /* DeltaNetSuffixStatus::Tree (delta_net.rs:39-46): what an unaccepted tree pass leaves behind for encode_accept */
    float* tree_conv_states; /* f32 [tree, conv_dim, k-1] */
    uint16_t *tree_k, *tree_v; /* bf16 [tree, key_dim] / [tree, value_dim] */
    float *tree_log_decay, *tree_beta; /* f32 [tree, Hv] */
} layer_state;

struct orc_model {
    uzu_model_desc desc;
    uzu_layer_desc* layers;
    layer_state* states;
    uint32_t context_length;
    uint16_t** layer_outputs; /* debug taps: per layer [rows, d] of the last forward */
    uint32_t last_rows;
    uint16_t* final_hidden;
    /* DFlash speculator support (stream.rs:213-214,632-633; transformer.rs:160-171,285-293): the residual-stream features of the layers a speculator
     * taps -- capture_residual(shortcut, hidden) = TensorAddScale(shortcut, hidden, 1.0) -- and the output norm of EVERY row of the pass
     * (DecoderEncodeOutput::final_hidden, decoder.rs:190-194), kept for the last forward when capture_features != 0 */
    uint32_t capture_features;
    uint16_t** hidden_features; /* per layer [rows, d] */
    uint16_t* final_hidden_rows; /* [out_rows, d] */
    uint32_t final_hidden_row_count;
    /* an unaccepted speculated tree (orc_model_verify_tree ... orc_model_accept) */
    uint32_t tree_size;
    int32_t* tree_parents;
};


void* orc_xcalloc(size_t n, size_t sz);
/* Linear::encode (linear/matmul.rs:122-148, rht_wrapper.rs, qlora_wrapper.rs): a fresh [batch, n] bf16 buffer */
uint16_t* orc_linear(const uzu_linear_desc* lin, const uint16_t* input, uint32_t batch);
/* ... with the output element type of the call (Embedding::encode_readout(.., DataType::F32, ..), dflash.rs:335): f32 [batch, n] when d_dtype == ORC_F32 */
void* orc_linear_typed(const uzu_linear_desc* lin, const uint16_t* input, uint32_t batch, uint32_t d_dtype);
/* Normalization::encode (encodable_block/normalization.rs:114-146); mode: 0 none, 1 copy, 2 add */
uint16_t* orc_norm(const uzu_norm_desc* nd, const uint16_t* input, uint16_t* shortcut, int mode, uint32_t rows, uint32_t dim);
/* Transformer::encode's layer loop (transformer.rs:247-293) over `count` rows: consumes `hidden`, returns the last layer's output rows and the
 * residual rows in *shortcut_out (both owned by the caller).  Attention layers append the rows' keys / values behind their cache's logical end
 * and do NOT accept them. */
uint16_t* orc_layers_forward(orc_model* m, uint16_t* hidden, uint32_t count, const uint32_t* trie, const int32_t* parents, const uint16_t* per_layer_inputs,
                             uint16_t** shortcut_out);

/* MoeBlock::encode (uzu_oracle_moe.c): a fresh bf16 [batch, model_dim] buffer */
uint16_t* orc_moe_block(const uzu_moe_desc* M, uint32_t model_dim, const uint16_t* input, uint32_t batch);

#endif
