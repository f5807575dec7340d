"""ctypes mirror of ``include/uzu_model_desc.h`` plus NumPy-backed builders.

The structs here are byte-for-byte the C structs; `ModelBundle` keeps every NumPy array alive
for as long as the descriptor is in use (the C side only borrows the host pointers).

Reference schema these fields come from: crates/backend-uzu/src/config/** (decoder.rs,
transformer_layer.rs, token_mixer/{attention,delta_net}.rs, normalization.rs, rope/*.rs,
weight_matrix/*.rs) and the tensor layouts in encodable_block/weight_matrix.rs:101-162.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

# enums (uzu_model_desc.h)
QUANT_SCALE_BIAS, QUANT_SCALE_ZERO_POINT, QUANT_SCALE_SYMMETRIC, QUANT_NONE = 0, 1, 2, 3
ACT_SILU, ACT_GELU_APPROX, ACT_GELU_EXACT, ACT_IDENTITY, ACT_SOFTPLUS = 0, 1, 2, 3, 4
MIXER_ATTENTION, MIXER_DELTA_NET = 0, 1
ROPE_NONE, ROPE_UNSCALED, ROPE_LLAMA, ROPE_LINEAR, ROPE_YARN, ROPE_LONGROPE = 0, 1, 2, 3, 4, 5


class LinearDesc(C.Structure):
    _fields_ = [
        ("n", C.c_uint32), ("k", C.c_uint32), ("bits", C.c_uint32), ("group_size", C.c_uint32),
        ("method", C.c_uint32), ("reserved", C.c_uint32),
        ("weights", C.c_void_p), ("scales", C.c_void_p), ("biases", C.c_void_p),
        ("zero_points", C.c_void_p), ("out_biases", C.c_void_p),
        ("input_signs", C.c_void_p), ("output_signs", C.c_void_p),
        ("lora_rank", C.c_uint32), ("reserved2", C.c_uint32), ("adapter_down", C.c_void_p), ("adapter_up", C.c_void_p),
    ]


class NormDesc(C.Structure):
    _fields_ = [
        ("present", C.c_uint32), ("full_layer", C.c_uint32), ("subtract_mean", C.c_uint32), ("reserved", C.c_uint32),
        ("epsilon", C.c_float), ("scale_offset", C.c_float),
        ("scales", C.c_void_p), ("biases", C.c_void_p),
    ]


class RopeDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32), ("head_dim", C.c_uint32), ("max_sequence_length", C.c_uint32),
        ("original_context_length", C.c_uint32),
        ("base", C.c_float), ("scaling_factor", C.c_float), ("low_frequency_factor", C.c_float),
        ("high_frequency_factor", C.c_float),
        ("beta_fast", C.c_float), ("beta_slow", C.c_float), ("truncate", C.c_uint32), ("reserved", C.c_uint32),
        ("short_factor", C.c_void_p), ("long_factor", C.c_void_p),
    ]


class MoeDesc(C.Structure):
    """include/uzu_model_desc.h::uzu_moe_desc"""
    _fields_ = [
        ("num_routed_experts", C.c_uint32), ("num_active_experts", C.c_uint32), ("expert_hidden_dim", C.c_uint32), ("router_renorm", C.c_uint32),
        ("gating_sel", C.c_uint32), ("silu_alpha", C.c_float),
        ("gate_clip_min", C.c_float), ("gate_clip_max", C.c_float), ("up_clip_min", C.c_float), ("up_clip_max", C.c_float),
        ("router_weights", C.c_void_p), ("router_biases", C.c_void_p), ("w13", C.c_void_p), ("w2", C.c_void_p), ("up_biases", C.c_void_p), ("down_biases", C.c_void_p),
    ]


MLP_DENSE, MLP_MOE = 0, 1


class LayerDesc(C.Structure):
    _fields_ = [
        ("mixer_kind", C.c_uint32), ("hidden_dim", C.c_uint32), ("activation", C.c_uint32), ("mlp_kind", C.c_uint32),
        ("pre_mixer_norm", NormDesc), ("post_mixer_norm", NormDesc), ("pre_mlp_norm", NormDesc),
        ("post_mlp_norm", NormDesc),
        ("num_heads", C.c_uint32), ("num_groups", C.c_uint32), ("head_dim", C.c_uint32), ("has_gate", C.c_uint32),
        ("attention_scale", C.c_float), ("use_rope", C.c_uint32),
        ("qkv_projection", LinearDesc), ("gate_projection", LinearDesc), ("out_projection", LinearDesc),
        ("query_norm", NormDesc), ("key_norm", NormDesc),
        ("sliding_window_size", C.c_uint32), ("has_sinks", C.c_uint32), ("sinks", C.c_void_p),
        ("dn_num_heads", C.c_uint32), ("dn_num_groups", C.c_uint32), ("dn_head_dim", C.c_uint32),
        ("dn_value_head_dim", C.c_uint32), ("dn_kernel_size", C.c_uint32), ("dn_norm_epsilon", C.c_float),
        ("dn_in_proj", LinearDesc), ("dn_out_proj", LinearDesc),
        ("dn_conv_weights", C.c_void_p), ("dn_conv_biases", C.c_void_p), ("dn_a_log", C.c_void_p),
        ("dn_dt_bias", C.c_void_p), ("dn_norm_scales", C.c_void_p),
        ("up_projection", LinearDesc), ("down_projection", LinearDesc),
        # layer options of the Gemma families (all zero = none)
        ("rope_index", C.c_uint32), ("has_post_layer_scalar", C.c_uint32), ("post_layer_scalar", C.c_float),
        ("is_kv_sharing", C.c_uint32), ("kv_source_layer_index", C.c_uint32), ("normalize_values", C.c_uint32),
        ("has_ple", C.c_uint32), ("ple_dim", C.c_uint32), ("ple_activation", C.c_uint32), ("is_non_causal", C.c_uint32),
        ("ple_gate", LinearDesc), ("ple_projection", LinearDesc), ("ple_norm", NormDesc),
        ("moe", MoeDesc),
    ]


class ModelDesc(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_uint32), ("model_dim", C.c_uint32), ("num_layers", C.c_uint32),
        ("tied_embeddings", C.c_uint32),
        ("input_scale", C.c_float), ("logit_scale", C.c_float), ("logit_soft_cap", C.c_float),
        ("max_context_length", C.c_uint32),
        ("rope", RopeDesc),
        ("embedding", LinearDesc), ("output_embedding", LinearDesc),
        ("output_norm", NormDesc),
        ("layers", C.POINTER(LayerDesc)),
        # decoder options of the Gemma families (all zero = none)
        ("num_ropes", C.c_uint32), ("has_ple", C.c_uint32), ("ropes", C.POINTER(RopeDesc)),
        ("embedding_norm", NormDesc),
        ("ple_dim", C.c_uint32), ("ple_vocab_size", C.c_uint32), ("ple_embed_scale", C.c_float),
        ("ple_model_projection_scale", C.c_float), ("ple_input_scale", C.c_float), ("reserved", C.c_uint32),
        ("ple_token_embedding", LinearDesc), ("ple_model_projection", LinearDesc), ("ple_projection_norm", NormDesc),
    ]


def _ptr(a: Optional[np.ndarray]) -> Optional[int]:
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


@dataclass
class LinearWeights:
    """One WeightMatrix (+ optional Linear biases) as NumPy arrays in the reference's layout."""
    n: int
    k: int
    bits: int                 # 4 / 8 / 16
    group_size: int
    method: int               # QUANT_*
    weights: np.ndarray       # u8 [n, k*bits/8]  or uint16(bf16) [n,k]
    scales: Optional[np.ndarray] = None       # uint16 (bf16 bits) [n, groups]
    biases: Optional[np.ndarray] = None       # uint16 [n, groups]
    zero_points: Optional[np.ndarray] = None  # u8
    out_biases: Optional[np.ndarray] = None   # uint16 [n]
    input_signs: Optional[np.ndarray] = None  # int32 [k]  (HybridSpec InputOutput: RHTLinearWrapper, rht_wrapper.rs:140-298)
    output_signs: Optional[np.ndarray] = None  # int32 [n]
    adapter_down: Optional[np.ndarray] = None  # bf16 bits [rank, k]  (HybridSpec adapter_spec LowRankSpec: QLoRALinearWrapper, qlora_wrapper.rs:61-252)
    adapter_up: Optional[np.ndarray] = None    # bf16 bits [n, rank]

    @property
    def lora_rank(self) -> int:
        return 0 if self.adapter_down is None else int(self.adapter_down.shape[0])

    def desc(self) -> LinearDesc:
        return LinearDesc(self.n, self.k, self.bits, self.group_size, self.method, 0, _ptr(self.weights),
                          _ptr(self.scales), _ptr(self.biases), _ptr(self.zero_points), _ptr(self.out_biases),
                          _ptr(self.input_signs), _ptr(self.output_signs), self.lora_rank, 0, _ptr(self.adapter_down), _ptr(self.adapter_up))

    def nbytes(self) -> int:
        return sum(a.nbytes for a in (self.weights, self.scales, self.biases, self.zero_points, self.out_biases, self.input_signs,
                                      self.output_signs, self.adapter_down, self.adapter_up) if a is not None)

    def rows(self, lo: int, hi: int) -> "LinearWeights":
        """Column-parallel shard: output rows [lo, hi) (any split is layout-safe, groups run along k)."""
        if self.input_signs is not None or self.output_signs is not None or self.adapter_down is not None:
            raise NotImplementedError("tensor-parallel shards of RHT / QLoRA (HybridSpec) linears")
        sl = slice(lo, hi)
        cp = lambda a: None if a is None else np.ascontiguousarray(a[sl])
        return LinearWeights(hi - lo, self.k, self.bits, self.group_size, self.method, cp(self.weights),
                             cp(self.scales), cp(self.biases), cp(self.zero_points), cp(self.out_biases))


@dataclass
class NormWeights:
    present: bool = True
    full_layer: bool = False
    subtract_mean: bool = False
    epsilon: float = 1e-6
    scale_offset: float = 0.0
    scales: Optional[np.ndarray] = None  # f32 [dim]
    biases: Optional[np.ndarray] = None

    def desc(self) -> NormDesc:
        return NormDesc(int(self.present), int(self.full_layer), int(self.subtract_mean), 0, self.epsilon,
                        self.scale_offset, _ptr(self.scales), _ptr(self.biases))


ABSENT_NORM = NormWeights(present=False)


@dataclass
class PleLayerWeights:
    """PLELayerConfig + the `ple` subtree of a layer (PerLayerEmbeddingProjection, per_layer_embedding.rs:150-271)."""
    ple_dim: int
    activation: int
    gate: "LinearWeights"        # [ple_dim, model_dim]
    projection: "LinearWeights"  # [model_dim, ple_dim]
    norm: "NormWeights"          # [model_dim]


@dataclass
class PleModelWeights:
    """PLEModelConfig + the `per_layer_embedding` subtree (PerLayerEmbedding, per_layer_embedding.rs:36-148)."""
    ple_dim: int
    ple_vocab_size: int
    ple_embed_scale: float
    model_projection_scale: float
    input_scale: float
    token_embedding: "LinearWeights"   # table [ple_vocab, num_layers * ple_dim]
    model_projection: "LinearWeights"  # [num_layers * ple_dim, model_dim]
    projection_norm: "NormWeights"     # [ple_dim], epsilon as configured


@dataclass
class MoeWeights:
    """MixtureOfExpertsConfig + the `mlp` subtree of a MoE layer (encodable_block/mlp/moe/mod.rs:84-200); every tensor bf16 bits (uint16)."""
    num_routed_experts: int
    num_active_experts: int
    expert_hidden_dim: int
    router_weights: np.ndarray   # [E, model_dim]
    router_biases: np.ndarray    # [E]
    w13: np.ndarray              # [E, 2 * d_ff, model_dim]
    w2: np.ndarray               # [E, model_dim, d_ff]
    up_biases: np.ndarray        # [E, 2 * d_ff]
    down_biases: np.ndarray      # [E, model_dim]
    router_renorm: bool = True   # SoftmaxRouting
    gating_sel: int = 2          # 2 SwiGLU (SiLU), 3 GEGLU (GELUApprox)
    silu_alpha: float = 1.0
    gate_clip: tuple = (float("-inf"), float("inf"))
    up_clip: tuple = (float("-inf"), float("inf"))

    def desc(self) -> MoeDesc:
        return MoeDesc(self.num_routed_experts, self.num_active_experts, self.expert_hidden_dim, int(self.router_renorm), self.gating_sel, self.silu_alpha,
                       self.gate_clip[0], self.gate_clip[1], self.up_clip[0], self.up_clip[1], _ptr(self.router_weights), _ptr(self.router_biases), _ptr(self.w13),
                       _ptr(self.w2), _ptr(self.up_biases), _ptr(self.down_biases))

    def nbytes_active(self) -> int:
        """bytes one decoded token streams: the router + the k active experts' matrices and biases"""
        per_expert = (self.w13.nbytes + self.w2.nbytes + self.up_biases.nbytes + self.down_biases.nbytes) // self.num_routed_experts
        return self.router_weights.nbytes + self.router_biases.nbytes + self.num_active_experts * per_expert


@dataclass
class LayerWeights:
    mixer_kind: int
    hidden_dim: int
    activation: int
    pre_mixer_norm: NormWeights
    pre_mlp_norm: NormWeights
    up_projection: Optional[LinearWeights]
    down_projection: Optional[LinearWeights]
    post_mixer_norm: NormWeights = field(default_factory=lambda: ABSENT_NORM)
    post_mlp_norm: NormWeights = field(default_factory=lambda: ABSENT_NORM)
    # attention
    num_heads: int = 0
    num_groups: int = 0
    head_dim: int = 0
    has_gate: bool = False
    attention_scale: float = 0.0
    use_rope: bool = False
    qkv_projection: Optional[LinearWeights] = None
    gate_projection: Optional[LinearWeights] = None
    out_projection: Optional[LinearWeights] = None
    query_norm: NormWeights = field(default_factory=lambda: ABSENT_NORM)
    key_norm: NormWeights = field(default_factory=lambda: ABSENT_NORM)
    sliding_window_size: int = 0          # 0 = full attention; causal sliding window => ring KV state (state.rs:20-106)
    sinks: Optional[np.ndarray] = None    # bf16 bits (uint16) [num_heads] or None
    # delta net
    dn_num_heads: int = 0
    dn_num_groups: int = 0
    dn_head_dim: int = 0
    dn_value_head_dim: int = 0
    dn_kernel_size: int = 0
    dn_norm_epsilon: float = 1e-6
    dn_in_proj: Optional[LinearWeights] = None
    dn_out_proj: Optional[LinearWeights] = None
    dn_conv_weights: Optional[np.ndarray] = None
    dn_conv_biases: Optional[np.ndarray] = None
    dn_a_log: Optional[np.ndarray] = None
    dn_dt_bias: Optional[np.ndarray] = None
    dn_norm_scales: Optional[np.ndarray] = None
    # layer options of the Gemma families (config/transformer_layer.rs:16-20, config/token_mixer/attention.rs:26-28)
    rope_index: int = 0                          # which of ModelBundle.ropes rotates this layer (use_rope)
    post_layer_scalar: Optional[float] = None    # has_post_layer_scalar + tensor `post_layer_scalar` [1]
    kv_source_layer_index: Optional[int] = None  # is_kv_sharing: queries only, the KV state of that earlier layer is read
    normalize_values: bool = False
    ple: Optional["PleLayerWeights"] = None      # ple_config: PerLayerEmbeddingProjection at the end of the layer
    is_non_causal: bool = False                  # AttentionConfig::is_causal == false (the block attention of a DFlash draft layer)
    moe: Optional["MoeWeights"] = None           # mlp_config = MixtureOfExpertsConfig: replaces the dense MLP (up / down projection are then None)

    def desc(self) -> LayerDesc:
        empty = LinearDesc()
        ld = lambda w: w.desc() if w is not None else empty
        return LayerDesc(
            self.mixer_kind, self.hidden_dim, self.activation, MLP_MOE if self.moe is not None else MLP_DENSE,
            self.pre_mixer_norm.desc(), self.post_mixer_norm.desc(), self.pre_mlp_norm.desc(),
            self.post_mlp_norm.desc(),
            self.num_heads, self.num_groups, self.head_dim, int(self.has_gate), self.attention_scale,
            int(self.use_rope),
            ld(self.qkv_projection), ld(self.gate_projection), ld(self.out_projection),
            self.query_norm.desc(), self.key_norm.desc(),
            int(self.sliding_window_size), int(self.sinks is not None), _ptr(self.sinks),
            self.dn_num_heads, self.dn_num_groups, self.dn_head_dim, self.dn_value_head_dim, self.dn_kernel_size,
            self.dn_norm_epsilon,
            ld(self.dn_in_proj), ld(self.dn_out_proj),
            _ptr(self.dn_conv_weights), _ptr(self.dn_conv_biases), _ptr(self.dn_a_log), _ptr(self.dn_dt_bias),
            _ptr(self.dn_norm_scales),
            ld(self.up_projection), ld(self.down_projection),
            int(self.rope_index), int(self.post_layer_scalar is not None), float(self.post_layer_scalar or 0.0),
            int(self.kv_source_layer_index is not None), int(self.kv_source_layer_index or 0), int(self.normalize_values),
            int(self.ple is not None), self.ple.ple_dim if self.ple else 0, self.ple.activation if self.ple else 0, int(self.is_non_causal),
            ld(self.ple.gate if self.ple else None), ld(self.ple.projection if self.ple else None),
            (self.ple.norm if self.ple else ABSENT_NORM).desc(),
            self.moe.desc() if self.moe is not None else MoeDesc(),
        )

    def linears(self):
        names = ("qkv_projection", "gate_projection", "out_projection", "dn_in_proj", "dn_out_proj", "up_projection",
                 "down_projection")
        out = [(n, getattr(self, n)) for n in names if getattr(self, n) is not None]
        if self.ple is not None:
            out += [("ple.gate", self.ple.gate), ("ple.projection", self.ple.projection)]
        return out


@dataclass
class RopeConfig:
    kind: int = ROPE_NONE
    head_dim: int = 0
    max_sequence_length: int = 0
    base: float = 10000.0
    scaling_factor: float = 1.0
    original_context_length: int = 0
    low_frequency_factor: float = 1.0
    high_frequency_factor: float = 1.0
    beta_fast: float = 32.0           # YaRN
    beta_slow: float = 1.0
    truncate: bool = True
    short_factor: Optional[np.ndarray] = None  # LongRoPE: f32 [head_dim / 2]
    long_factor: Optional[np.ndarray] = None

    def desc(self) -> RopeDesc:
        def ptr(a):
            if a is None:
                return None
            assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"] and a.size == self.head_dim // 2
            return a.ctypes.data
        return RopeDesc(self.kind, self.head_dim, self.max_sequence_length, self.original_context_length, self.base,
                        self.scaling_factor, self.low_frequency_factor, self.high_frequency_factor, self.beta_fast, self.beta_slow,
                        int(self.truncate), 0, ptr(self.short_factor), ptr(self.long_factor))


class DFlashDesc(C.Structure):
    """include/uzu_model_desc.h::uzu_dflash_desc"""
    _fields_ = [
        ("model_dim", C.c_uint32), ("hidden_dim", C.c_uint32), ("block_size", C.c_uint32), ("mask_token_id", C.c_uint32),
        ("num_target_layers", C.c_uint32), ("num_layers", C.c_uint32), ("vocab_size", C.c_uint32), ("context_capacity", C.c_uint32),
        ("target_layer_ids", C.POINTER(C.c_uint32)),
        ("context_projection", LinearDesc), ("context_norm", NormDesc), ("state_kv_projection", LinearDesc), ("rope", RopeDesc),
        ("layers", C.POINTER(LayerDesc)), ("output_norm", NormDesc),
    ]


@dataclass
class DFlashBundle:
    """A DFlash draft model (DFlashDraftConfig, config/dflash.rs:9-23, + the tensors of `speculator.draft_model`)."""
    name: str
    model_dim: int
    hidden_dim: int
    block_size: int
    mask_token_id: int
    target_layer_ids: List[int]
    vocab_size: int
    context_capacity: int
    context_projection: "LinearWeights"
    context_norm: "NormWeights"
    state_kv_projection: "LinearWeights"
    rope: "RopeConfig"
    layers: List["LayerWeights"]
    output_norm: "NormWeights"
    _keep: list = field(default_factory=list, repr=False)

    def desc(self) -> DFlashDesc:
        arr = (LayerDesc * len(self.layers))(*[l.desc() for l in self.layers])
        ids = (C.c_uint32 * len(self.target_layer_ids))(*self.target_layer_ids)
        self._keep += [arr, ids]
        return DFlashDesc(self.model_dim, self.hidden_dim, self.block_size, self.mask_token_id, len(self.target_layer_ids), len(self.layers), self.vocab_size,
                          self.context_capacity, C.cast(ids, C.POINTER(C.c_uint32)), self.context_projection.desc(), self.context_norm.desc(),
                          self.state_kv_projection.desc(), self.rope.desc(), C.cast(arr, C.POINTER(LayerDesc)), self.output_norm.desc())


class WeaverLayerDesc(C.Structure):
    """include/uzu_model_desc.h::uzu_weaver_layer_desc"""
    _fields_ = [("pre_attention_norm", NormDesc), ("pre_mlp_norm", NormDesc), ("qkv_projection", LinearDesc), ("out_projection", LinearDesc), ("up_projection", LinearDesc),
                ("down_projection", LinearDesc)]


class WeaverDesc(C.Structure):
    """include/uzu_model_desc.h::uzu_weaver_desc"""
    _fields_ = [
        ("model_dim", C.c_uint32), ("target_model_dim", C.c_uint32), ("target_embedding_dim", C.c_uint32), ("num_layers", C.c_uint32), ("num_heads", C.c_uint32),
        ("hidden_dim", C.c_uint32), ("max_depth", C.c_uint32), ("candidate_pool_size", C.c_uint32),
        ("embedding_norm", NormDesc), ("embedding_projection", LinearDesc), ("hidden_state_norm", NormDesc), ("hidden_state_projection", LinearDesc),
        ("output_norm", NormDesc), ("query_projection", LinearDesc), ("rope", RopeDesc), ("layers", C.POINTER(WeaverLayerDesc)),
    ]


class WeaverTreeShape(C.Structure):
    """include/uzu_model_desc.h::uzu_weaver_tree_shape (WeaverTreeShape, encodable_block/weaver.rs:33-46)"""
    _fields_ = [(n, C.c_uint32) for n in ("tree_budget", "max_depth", "dflash_depth", "rounds", "expand_per_round", "expand_width")]

    def slot_count(self) -> int:
        return 1 + max(self.rounds - 1, 0) * self.expand_per_round


@dataclass
class WeaverLayerWeights:
    pre_attention_norm: "NormWeights"
    pre_mlp_norm: "NormWeights"
    qkv_projection: "LinearWeights"
    out_projection: "LinearWeights"
    up_projection: "LinearWeights"    # + biases
    down_projection: "LinearWeights"  # + biases

    def desc(self) -> WeaverLayerDesc:
        return WeaverLayerDesc(self.pre_attention_norm.desc(), self.pre_mlp_norm.desc(), self.qkv_projection.desc(), self.out_projection.desc(), self.up_projection.desc(),
                               self.down_projection.desc())


@dataclass
class WeaverBundle:
    """The Weaver tree constructor (WeaverConfig, config/weaver.rs:5-17, + the tensors of `speculator.weaver`)."""
    name: str
    model_dim: int
    target_model_dim: int
    target_embedding_dim: int
    num_heads: int
    hidden_dim: int
    max_depth: int
    candidate_pool_size: int
    embedding_norm: "NormWeights"
    embedding_projection: "LinearWeights"
    hidden_state_norm: "NormWeights"
    hidden_state_projection: "LinearWeights"
    output_norm: "NormWeights"
    query_projection: "LinearWeights"
    rope: "RopeConfig"
    layers: List[WeaverLayerWeights]
    _keep: list = field(default_factory=list, repr=False)

    def desc(self) -> WeaverDesc:
        arr = (WeaverLayerDesc * len(self.layers))(*[l.desc() for l in self.layers])
        self._keep.append(arr)
        return WeaverDesc(self.model_dim, self.target_model_dim, self.target_embedding_dim, len(self.layers), self.num_heads, self.hidden_dim, self.max_depth,
                          self.candidate_pool_size, self.embedding_norm.desc(), self.embedding_projection.desc(), self.hidden_state_norm.desc(),
                          self.hidden_state_projection.desc(), self.output_norm.desc(), self.query_projection.desc(), self.rope.desc(), C.cast(arr, C.POINTER(WeaverLayerDesc)))


@dataclass
class ModelBundle:
    """A whole model: config + weights.  `desc()` returns a ctypes ModelDesc that borrows from self."""
    name: str
    vocab_size: int
    model_dim: int
    max_context_length: int
    rope: RopeConfig
    embedding: LinearWeights
    output_norm: NormWeights
    layers: List[LayerWeights]
    tied_embeddings: bool = True
    output_embedding: Optional[LinearWeights] = None
    input_scale: float = 1.0
    logit_scale: float = 1.0
    logit_soft_cap: float = 0.0
    ropes: Optional[List[RopeConfig]] = None     # distinct per-layer RoPE configurations (LayerWeights.rope_index); None: `rope` for all
    embedding_norm: NormWeights = field(default_factory=lambda: ABSENT_NORM)
    ple: Optional[PleModelWeights] = None
    _keep: list = field(default_factory=list, repr=False)

    def desc(self) -> ModelDesc:
        arr = (LayerDesc * len(self.layers))(*[l.desc() for l in self.layers])
        self._keep.append(arr)
        out_emb = self.output_embedding.desc() if self.output_embedding is not None else LinearDesc()
        ropes = None
        if self.ropes:
            ropes = (RopeDesc * len(self.ropes))(*[r.desc() for r in self.ropes])
            self._keep.append(ropes)
        e = LinearDesc()
        p = self.ple
        return ModelDesc(self.vocab_size, self.model_dim, len(self.layers), int(self.tied_embeddings),
                         self.input_scale, self.logit_scale, self.logit_soft_cap, self.max_context_length,
                         self.rope.desc(), self.embedding.desc(), out_emb, self.output_norm.desc(),
                         C.cast(arr, C.POINTER(LayerDesc)),
                         len(self.ropes) if self.ropes else 0, int(p is not None),
                         C.cast(ropes, C.POINTER(RopeDesc)) if ropes is not None else None,
                         self.embedding_norm.desc(),
                         p.ple_dim if p else 0, p.ple_vocab_size if p else 0, p.ple_embed_scale if p else 0.0,
                         p.model_projection_scale if p else 0.0, p.input_scale if p else 0.0, 0,
                         p.token_embedding.desc() if p else e, p.model_projection.desc() if p else e,
                         (p.projection_norm if p else ABSENT_NORM).desc())

    # ---- algorithmic bytes per decoded token (SURVEY.md §8d formula) ----
    def weight_stream_bytes(self) -> int:
        total = 0
        for l in self.layers:
            for _, w in l.linears():
                total += w.nbytes()
            if l.moe is not None:
                total += l.moe.nbytes_active()
        readout = self.embedding if self.tied_embeddings else self.output_embedding
        total += readout.nbytes()
        if self.ple is not None:
            total += self.ple.model_projection.nbytes()
        return total

    def state_bytes_per_token(self, context: int) -> int:
        total = 0
        for l in self.layers:
            if l.mixer_kind == MIXER_ATTENTION:
                rows = min(context, l.sliding_window_size) if l.sliding_window_size else context
                total += 2 * rows * l.num_groups * l.head_dim * 2             # K and V rows, bf16 (a sharing layer reads its source's)
            else:
                total += 2 * l.dn_num_heads * l.dn_value_head_dim * l.dn_head_dim * 4  # f32 state read + write
                conv_dim = 2 * l.dn_num_groups * l.dn_head_dim + l.dn_num_heads * l.dn_value_head_dim
                total += 2 * conv_dim * (l.dn_kernel_size - 1) * 4            # conv state read + write
                total += conv_dim * l.dn_kernel_size * 4                      # conv taps
        return total

    def prefill_flops(self, tokens: int, context_before: int = 0) -> float:
        """Algorithmic FLOPs of prefilling `tokens` tokens: 2*M*sum(N*K) over the linears (the
        read-out runs for the last row only) + causal attention 4*heads*hd per (query, visible key) pair."""
        macs = 0
        for l in self.layers:
            for _, w in l.linears():
                macs += w.n * w.k
            if l.moe is not None:  # router for every token + the k active experts' up | gate and down matrices
                mo = l.moe
                macs += mo.num_routed_experts * self.model_dim + mo.num_active_experts * 3 * mo.expert_hidden_dim * self.model_dim
        flops = 2.0 * tokens * macs
        readout = self.embedding if self.tied_embeddings else self.output_embedding
        flops += 2.0 * readout.n * readout.k
        pairs = tokens * context_before + tokens * (tokens + 1) / 2.0
        for l in self.layers:
            if l.mixer_kind == MIXER_ATTENTION:
                flops += 4.0 * pairs * l.num_heads * l.head_dim
            else:  # delta rule: ~6 flops per state element per token (decay, S k, rank-1 update, S q)
                flops += 6.0 * tokens * l.dn_num_heads * l.dn_value_head_dim * l.dn_head_dim
        return flops

    def decode_bytes_per_token(self, context: int) -> int:
        emb_row = self.embedding.nbytes() // self.vocab_size
        return self.weight_stream_bytes() + self.state_bytes_per_token(context) + emb_row
