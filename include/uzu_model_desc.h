/*
 * uzu_model_desc.h -- plain-C description of a decoder-only language model as uzu's
 * `encodable_block` layer sees it after `Engine::load_language_model`
 * (crates/backend-uzu/src/engine/language_model/mod.rs:58-116).
 *
 * It is the in-memory equivalent of `config.json` + `model.safetensors`: the same numbers the
 * reference's `config/**` structs carry, and host pointers to tensors in the reference's on-disk
 * layouts.  Both the HIP engine (include/uzu_hip_engine.h) and the CPU
 * oracle (oracle/) consume this struct, so a parity test hands the SAME bytes to both.
 *
 * All pointers are HOST pointers, owned by the caller, and only need to stay valid for the
 * duration of the `*_model_create` call that receives them.
 */
#ifndef UZU_MODEL_DESC_H
#define UZU_MODEL_DESC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* backends/common/gpu_types/quantization_method.rs:4-9 */
typedef enum {
    UZU_QUANT_SCALE_BIAS = 0,       /* MLXSpec: deq = scale*q + bias                  */
    UZU_QUANT_SCALE_ZERO_POINT = 1, /* IntSpec asymmetric: deq = scale*q - scale*zp   */
    UZU_QUANT_SCALE_SYMMETRIC = 2,  /* IntSpec symmetric: deq = scale*q - scale*2^(b-1) */
    UZU_QUANT_NONE = 3              /* FullPrecisionSpec: bf16 weights [n,k]          */
} uzu_quant_method;

/* backends/common/gpu_types/activation_type.rs:8-14 */
typedef enum {
    UZU_ACT_SILU = 0,
    UZU_ACT_GELU_APPROX = 1,
    UZU_ACT_GELU_EXACT = 2,
    UZU_ACT_IDENTITY = 3,
    UZU_ACT_SOFTPLUS = 4
} uzu_activation_type;

/*
 * One `Linear` (encodable_block/linear/matmul.rs:39-47) = WeightMatrix in Layout::OutputInput
 * (encodable_block/weight_matrix.rs:101-162):
 *   weights      u8  [n, k/pack]            pack = 2 (4 bit, low nibble = even k) or 1 (8 bit)
 *   scales       bf16 [n, ceil(k/group)]
 *   biases       bf16 [n, ceil(k/group)]                  (ScaleBias only)
 *   zero_points  u8  [n, ceil(groups/pack)]               (ScaleZeroPoint only; 4 bit: nibble packed)
 *   out_biases   bf16 [n]                                 (Linear `biases`, optional)
 * method == UZU_QUANT_NONE: weights is bf16 [n,k], bits == 16, everything else NULL.
 * HybridSpec { incoherence_processing_mode: InputOutput, block 32, no adapter } (RHTLinearWrapper, linear/rht_wrapper.rs:140-298):
 *   input_signs  i32 [k], output_signs i32 [n] (+-1): y = OutputRht(W InputRht(x)) + out_biases; NULL = a plain linear.
 * Embedding tables (encodable_block/embedding.rs:126-341) reuse the struct with their own reading of the sign vectors:
 *   `embedding` / input table, HybridSpec { Output, block 32 }: output_signs i32 [model_dim]: lookup = OutputRht(dequantised row); a TIED
 *   table's read-out takes InputRht(row, the same signs) in front of the matmul (embedding.rs:161-188, 385-400);
 *   `output_embedding`, HybridSpec { Input, block 32 }: input_signs i32 [model_dim]: read-out = W InputRht(x) (embedding.rs:243-283).
 */
typedef struct {
    uint32_t n;          /* output_dim */
    uint32_t k;          /* input_dim */
    uint32_t bits;       /* 4, 8 or 16 */
    uint32_t group_size; /* 0 for full precision */
    uint32_t method;     /* uzu_quant_method */
    uint32_t reserved;
    const void* weights;
    const uint16_t* scales;
    const uint16_t* biases;
    const uint8_t* zero_points;
    const uint16_t* out_biases;
    const int32_t* input_signs;
    const int32_t* output_signs;
    /* HybridSpec { adapter_spec: LowRankSpec { rank } } (QLoRALinearWrapper, linear/qlora_wrapper.rs:61-252): bf16 adapters next to the
     * quantized base; y = [OutputRht]( W [InputRht](x) + (x down^T) up^T ); lora_rank == 0 = no adapter.  The sign vectors are optional
     * here (incoherence_block_size None); out_biases must be NULL (the reference asserts it). */
    uint32_t lora_rank;
    uint32_t reserved2;
    const uint16_t* adapter_down; /* bf16 [rank, k]  (weights.adapter.down_projection) */
    const uint16_t* adapter_up;   /* bf16 [n, rank]  (weights.adapter.up_projection) */
} uzu_linear_desc;

/* config/normalization.rs:10-18 + tensor `scales` f32 [dim] (encodable_block/normalization.rs:67-74) */
typedef struct {
    uint32_t present;       /* 0 => this optional norm is absent */
    uint32_t full_layer;    /* UpcastMode::FullLayer (1) vs OnlyNormalization (0) */
    uint32_t subtract_mean; /* LayerNorm-style mean subtraction */
    uint32_t reserved;
    float epsilon;
    float scale_offset;     /* Option<f32>::unwrap_or(0.0) */
    const float* scales;    /* f32 [dim] or NULL (has_scale == false) */
    const float* biases;    /* f32 [dim] or NULL */
} uzu_norm_desc;

typedef enum { UZU_MIXER_ATTENTION = 0, UZU_MIXER_DELTA_NET = 1 } uzu_mixer_kind;

/* config/rope/*.rs -- every AnyRoPEConfig variant (encodable_block/mixer/attention/rope.rs:13-114) */
typedef enum { UZU_ROPE_NONE = 0, UZU_ROPE_UNSCALED = 1, UZU_ROPE_LLAMA = 2, UZU_ROPE_LINEAR = 3, UZU_ROPE_YARN = 4,
               UZU_ROPE_LONGROPE = 5 } uzu_rope_kind;

typedef struct {
    uint32_t kind;                    /* uzu_rope_kind */
    uint32_t head_dim;                /* rope dim (may be < attention head_dim: partial rotary) */
    uint32_t max_sequence_length;
    uint32_t original_context_length; /* Llama */
    float base;
    float scaling_factor;             /* Llama / Linear */
    float low_frequency_factor;       /* Llama */
    float high_frequency_factor;      /* Llama */
    float beta_fast, beta_slow;       /* YaRN (config/rope/yarn_rope.rs) */
    uint32_t truncate;                /* YaRN */
    uint32_t reserved;
    const float* short_factor;        /* LongRoPE (config/rope/longrope.rs): f32 [head_dim / 2] */
    const float* long_factor;         /* LongRoPE: used when max_sequence_length > original_context_length */
} uzu_rope_desc;

/* MixtureOfExpertsConfig (config/mlp/mixture_of_experts.rs:9-22) + the tensors of the `mlp` subtree of a MoE layer (encodable_block/mlp/moe/mod.rs:114-160),
 * all in the model's data type (bf16), full precision:
 *   router.weights.weights [E, model_dim] (FullPrecisionSpec, Layout::OutputInput), router.biases [E]
 *   experts.up_projection.weights.weights [E, 2*d_ff, model_dim] (rows [0, d_ff) of an expert = up, [d_ff, 2 d_ff) = gate), experts.up_projection.biases [E, 2*d_ff]
 *   experts.down_projection.weights.weights [E, model_dim, d_ff], experts.down_projection.biases [E, model_dim]
 * What MoeBlock::new refuses is refused here as well: shared experts, an expert gate, a block without router / up / down biases, activations other than
 * SiLU / GELUApprox, more than 512 routed or 128 active experts, model_dim % 4 != 0 (mod.rs:84-112). */
typedef struct {
    uint32_t num_routed_experts;
    uint32_t num_active_experts; /* num_active_routed_experts */
    uint32_t expert_hidden_dim;  /* d_ff */
    uint32_t router_renorm;      /* routing_function == SoftmaxRouting: softmax over the k winners (mod.rs:110); else the raw logits */
    uint32_t gating_sel;         /* 2 = SwiGLU (expert activation SILU), 3 = GEGLU (GELUApprox)  (mod.rs:104-108) */
    float silu_alpha;            /* expert_config.activation.alpha() */
    float gate_clip_min, gate_clip_max, up_clip_min, up_clip_max; /* expert_config.{gate,up}_clipping; -inf / +inf = none */
    const uint16_t* router_weights;
    const uint16_t* router_biases;
    const uint16_t* w13;
    const uint16_t* w2;
    const uint16_t* up_biases;
    const uint16_t* down_biases;
} uzu_moe_desc;

typedef enum { UZU_MLP_DENSE = 0, UZU_MLP_MOE = 1 } uzu_mlp_kind;

/* config/transformer_layer.rs:8-21 with AttentionConfig / DeltaNetConfig and a DenseMLPConfig | MixtureOfExpertsConfig */
typedef struct {
    uint32_t mixer_kind; /* uzu_mixer_kind */
    uint32_t hidden_dim; /* MLP hidden (up projection has 2*hidden rows: [up ; gate]) */
    uint32_t activation; /* uzu_activation_type of the gated MLP */
    uint32_t mlp_kind;   /* uzu_mlp_kind (the field was `reserved`: 0 = the dense MLP) */

    uzu_norm_desc pre_mixer_norm;
    uzu_norm_desc post_mixer_norm;
    uzu_norm_desc pre_mlp_norm;
    uzu_norm_desc post_mlp_norm;

    /* --- attention (config/token_mixer/attention.rs:9-29) --- */
    uint32_t num_heads;
    uint32_t num_groups; /* kv heads */
    uint32_t head_dim;
    uint32_t has_gate;   /* gate_projection_config.is_some() */
    float attention_scale; /* 0 => 1/sqrt(head_dim) */
    uint32_t use_rope;
    uzu_linear_desc qkv_projection;  /* n = (heads + 2*groups) * head_dim */
    uzu_linear_desc gate_projection; /* n = heads*head_dim */
    uzu_linear_desc out_projection;  /* k = heads*head_dim */
    uzu_norm_desc query_norm;        /* scales f32 [head_dim] */
    uzu_norm_desc key_norm;
    uint32_t sliding_window_size;    /* 0 = none; causal + window => the ring KV state of mixer/attention/state.rs:20-106 */
    uint32_t has_sinks;
    const void* sinks;               /* bf16 [heads] (mixer.sinks, mixer/attention/mod.rs:162-165) or NULL */

    /* --- gated delta net (config/token_mixer/delta_net.rs) --- */
    uint32_t dn_num_heads;      /* value heads Hv */
    uint32_t dn_num_groups;     /* key heads Hk */
    uint32_t dn_head_dim;       /* Dk (128) */
    uint32_t dn_value_head_dim; /* Dv (128) */
    uint32_t dn_kernel_size;
    float dn_norm_epsilon;
    uzu_linear_desc dn_in_proj;  /* n = 2*Hk*Dk + 2*Hv*Dv + 2*Hv */
    uzu_linear_desc dn_out_proj; /* k = Hv*Dv */
    const float* dn_conv_weights; /* f32 [conv_dim, kernel_size] */
    const float* dn_conv_biases;  /* f32 [conv_dim] or NULL */
    const float* dn_a_log;        /* f32 [Hv] */
    const float* dn_dt_bias;      /* f32 [Hv] */
    const float* dn_norm_scales;  /* f32 [Dv] */

    /* --- dense MLP (encodable_block/mlp/dense.rs) --- */
    uzu_linear_desc up_projection;   /* n = 2*hidden: rows [0,h) = up, [h,2h) = gate */
    uzu_linear_desc down_projection; /* k = hidden */

    /* --- layer options of the Gemma families (config/transformer_layer.rs:16-20, config/token_mixer/attention.rs:26-28);
     *     all zero = none of them.  NB the struct has grown in place over the rounds (it is the element type of uzu_model_desc.layers, so its size is
     *     part of the ABI): uzu_hip_desc_abi() reports the sizes the library was built with -- a caller checks them once (uzu_amd/desc.py does) --- */
    uint32_t rope_index;            /* use_rope != 0: which entry of uzu_model_desc.ropes rotates this layer (transformer.rs:101-118);
                                       ignored when num_ropes == 0 (the single `rope`) */
    uint32_t has_post_layer_scalar; /* transformer_layer.rs:61-84: pre_mlp_norm scales the residual sum, post_mlp_norm (required) its
                                       output -- unless the layer has a PLE projection, which then owns the scalar */
    float post_layer_scalar;        /* tensor `post_layer_scalar` [1] */
    uint32_t is_kv_sharing;         /* AttentionConfig::is_kv_sharing == TransformerLayerConfig::kv_source_layer_index.is_some(): the packed
                                       projection yields queries only (n = heads*head_dim), no key / value norm, no KV append; the
                                       attention reads the source layer's state (mixer/attention/mode.rs:79-84, transformer.rs:264-275) */
    uint32_t kv_source_layer_index; /* an earlier attention layer that owns its state (is_kv_sharing != 0) */
    uint32_t normalize_values;      /* AttentionConfig::value_norm_config(): scale-free RMS norm (eps 1e-6, FullLayer) of the value heads */
    uint32_t has_ple;               /* ple_config.is_some(): PerLayerEmbeddingProjection (per_layer_embedding.rs:150-271) ends the layer */
    uint32_t ple_dim;
    uint32_t ple_activation;        /* uzu_activation_type */
    uint32_t is_non_causal;         /* AttentionConfig::is_causal == false (mixer/attention/mod.rs:166-198, mask.rs:3-61): every suffix row sees every suffix row --
                                       the block attention of a DFlash draft model (encodable_block/dflash.rs).  0 = causal (the field was `reserved3`: same layout) */
    uzu_linear_desc ple_gate;       /* ple.gate: n = ple_dim, k = model_dim */
    uzu_linear_desc ple_projection; /* ple.projection: n = model_dim, k = ple_dim */
    uzu_norm_desc ple_norm;         /* ple.norm: [model_dim] */

    /* --- mixture of experts (mlp_kind == UZU_MLP_MOE: up_projection / down_projection / hidden_dim / activation above are unused) --- */
    uzu_moe_desc moe;
} uzu_layer_desc;

typedef struct {
    uint32_t vocab_size;
    uint32_t model_dim;
    uint32_t num_layers;
    uint32_t tied_embeddings;
    float input_scale;   /* embedding input_scale, default 1 */
    float logit_scale;   /* default 1 */
    float logit_soft_cap; /* 0 => none */
    uint32_t max_context_length; /* KV cache capacity = max_context_length + 1024 rows (attention/state.rs:108-122) */
    uzu_rope_desc rope;
    /* embedding table, Layout::InputOutput => stored [vocab, dim/pack]; described here with
     * n = vocab, k = model_dim exactly like the readout matmul sees it (embedding.rs:374-456). */
    uzu_linear_desc embedding;
    uzu_linear_desc output_embedding; /* untied readout; ignored when tied */
    uzu_norm_desc output_norm;
    const uzu_layer_desc* layers;

    /* --- decoder options of the Gemma families; all zero = none --- */
    uint32_t num_ropes;          /* 0: every rotating layer uses `rope` above.  Else the distinct AnyRoPEConfig values of the layers in
                                    order of first use (Transformer::new, transformer.rs:101-118): layer l rotates with ropes[rope_index] */
    uint32_t has_ple;            /* DecoderConfig::ple_model_config.is_some() (decoder.rs:85-99) */
    const uzu_rope_desc* ropes;
    uzu_norm_desc embedding_norm; /* DecoderConfig::embedding_norm_config: a Normalization of the looked-up rows (decoder.rs:68-83,149-154) */
    /* PLEModelConfig (config/per_layer_embedding.rs:5-15), PerLayerEmbedding (per_layer_embedding.rs:36-148); num_layers == the layer count */
    uint32_t ple_dim;
    uint32_t ple_vocab_size;
    float ple_embed_scale;
    float ple_model_projection_scale;
    float ple_input_scale;
    uint32_t reserved;
    uzu_linear_desc ple_token_embedding;  /* per_layer_embedding.token_embedding: n = ple_vocab_size, k = num_layers*ple_dim (an embedding table) */
    uzu_linear_desc ple_model_projection; /* per_layer_embedding.model_projection: n = num_layers*ple_dim, k = model_dim */
    uzu_norm_desc ple_projection_norm;    /* per_layer_embedding.projection_norm: [ple_dim]; epsilon as configured (the engine divides it by
                                             model_projection_scale^2, per_layer_embedding.rs:75-80) */
} uzu_model_desc;

/*
 * The DFlash draft model of the tree speculator: DFlashDraftConfig (config/dflash.rs:9-23) + the tensors of the subtree
 * `speculator.draft_model` (speculators/dflash_tfm.rs:86-107, encodable_block/dflash.rs:86-172):
 *   context_projection.*   Linear  [model_dim, model_dim * num_target_layers]   the accepted tokens' target features, side by side
 *   context_norm.scales    Normalization of the projected features (no shortcut)
 *   state_kv_projection.*  Linear  [num_layers * 2 * groups * head_dim, model_dim]   keys | values of every draft layer from one projected row
 *   layers.{i}.*           TransformerLayers with attention mixers (block attention: is_non_causal as configured) and a full KV state
 *   output_norm.scales     Normalization with the shortcut added
 * The draft model has no embedding of its own: rows are looked up in, and read out through, the TARGET model's table (dflash.rs:285,335).
 */
typedef struct {
    uint32_t model_dim;
    uint32_t hidden_dim;
    uint32_t block_size;           /* rows of one draft pass (<= ATTENTION_SUFFIX_CAPACITY = 1024) */
    uint32_t mask_token_id;
    uint32_t num_target_layers;    /* target_layer_ids.len(): the target's hidden-feature taps (stream.rs:213-214,632-633) */
    uint32_t num_layers;
    uint32_t vocab_size;
    uint32_t context_capacity;     /* DFlash::empty_state(context_capacity) (dflash.rs:174-188; language_model/state.rs:46-51: the target's max context) */
    const uint32_t* target_layer_ids;
    uzu_linear_desc context_projection;
    uzu_norm_desc context_norm;
    uzu_linear_desc state_kv_projection;
    uzu_rope_desc rope;            /* rope_config: positions context .. context + rows; max_sequence_length bounds the state */
    const uzu_layer_desc* layers;
    uzu_norm_desc output_norm;
} uzu_dflash_desc;

/*
 * The Weaver tree constructor of the speculator: WeaverConfig (config/weaver.rs:5-17) + the tensors of the subtree `speculator.weaver`
 * (encodable_block/weaver.rs:166-275, weaver_layer.rs:47-148).  All norms share norm_config (their own scales); the layers' MLP is a DenseMLP with SiLU and
 * up / down biases (weaver_layer.rs:126-131); model_dim = num_heads * rope.head_dim.
 *   embedding_norm.scales [target_embedding_dim], embedding_projection.{weights.*, biases} [model_dim, target_embedding_dim]
 *   hidden_state_norm.scales [target_model_dim], hidden_state_projection.{weights.*, biases} [model_dim, target_model_dim]
 *   blocks.{i}.{pre_attention_norm, qkv_projection [3 model_dim, model_dim], out_projection [model_dim, model_dim], pre_mlp_norm, mlp.up_projection (+ biases)
 *               [2 hidden, model_dim], mlp.down_projection (+ biases) [model_dim, hidden]}
 *   output_norm.scales [model_dim], query_projection.weights.* [target_model_dim, model_dim]
 */
typedef struct {
    uzu_norm_desc pre_attention_norm;
    uzu_norm_desc pre_mlp_norm;
    uzu_linear_desc qkv_projection;
    uzu_linear_desc out_projection;
    uzu_linear_desc up_projection;
    uzu_linear_desc down_projection;
} uzu_weaver_layer_desc;

typedef struct {
    uint32_t model_dim;
    uint32_t target_model_dim;
    uint32_t target_embedding_dim;
    uint32_t num_layers;
    uint32_t num_heads;
    uint32_t hidden_dim;
    uint32_t max_depth;
    uint32_t candidate_pool_size; /* 1..512 (CANDIDATES_MAX) */
    uzu_norm_desc embedding_norm;
    uzu_linear_desc embedding_projection;
    uzu_norm_desc hidden_state_norm;
    uzu_linear_desc hidden_state_projection;
    uzu_norm_desc output_norm;
    uzu_linear_desc query_projection;
    uzu_rope_desc rope; /* head_dim = model_dim / num_heads, max_sequence_length > max_depth */
    const uzu_weaver_layer_desc* layers;
} uzu_weaver_desc;

/* WeaverTreeShape (encodable_block/weaver.rs:33-46; dflash_tfm.rs:259-279: max_depth counts the root) */
typedef struct {
    uint32_t tree_budget;
    uint32_t max_depth;
    uint32_t dflash_depth;
    uint32_t rounds;
    uint32_t expand_per_round;
    uint32_t expand_width;
} uzu_weaver_tree_shape;

#ifdef __cplusplus
}
#endif
#endif /* UZU_MODEL_DESC_H */
