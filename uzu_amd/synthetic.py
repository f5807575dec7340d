"""Synthetic, format-exact model generator (there is no network for checkpoints).

Weights are random but laid out exactly as a lalamo-exported uzu checkpoint stores them
(encodable_block/weight_matrix.rs:101-162): packed int4/int8 codes
``[N, K/pack]`` low-nibble-first, bf16 scales/biases ``[N, K/g]``, nibble-packed zero points,
f32 norm scales.  Value ranges follow the reference's own test-input generators
(crates/backend-uzu/src/tests/matmul/quant.rs:58-110), rescaled by 1/sqrt(K) so that a 24-48 layer
stack stays inside bf16 range.

Shapes: Qwen3.5-0.8B layer shapes are the reference's benchmark table
(crates/backend-uzu/src/tests/matmul/shape.rs:82-100); layer count / vocab / conv kernel are the
public model-card values.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field, replace
from typing import List, Optional

import numpy as np

from . import desc as D


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """half::bf16::from_f32 (round to nearest even) on an array; returns uint16 bit patterns."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    rounding = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    return ((u + rounding) >> np.uint32(16)).astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


@dataclass
class ModelConfig:
    name: str
    vocab_size: int
    model_dim: int
    hidden_dim: int
    layer_kinds: List[int]                 # MIXER_* per layer
    # attention
    num_heads: int
    num_groups: int
    head_dim: int
    has_gate: bool = False
    qk_norm: bool = False
    rope: D.RopeConfig = field(default_factory=D.RopeConfig)
    # delta net
    dn_num_heads: int = 0
    dn_num_groups: int = 0
    dn_head_dim: int = 128
    dn_value_head_dim: int = 128
    dn_kernel_size: int = 4
    # norms
    norm_epsilon: float = 1e-6
    norm_scale_offset: float = 0.0
    norm_full_layer: bool = False
    # quantisation of every linear + embedding
    bits: int = 4
    group_size: int = 128
    method: int = D.QUANT_SCALE_BIAS
    tied_embeddings: bool = True
    max_context_length: int = 4096
    seed: int = 42
    logit_row_sigma: float = 0.6   # log-normal spread of per-token readout row norms (peaked logits)
    rht: bool = False              # every layer linear is a HybridSpec InputOutput linear (random +-1 sign vectors; not the embeddings)
    rht_embeddings: bool = False   # HybridSpec Output / Input embeddings (embedding.rs:126-341): sign vectors over model_dim on the tables
    qlora_rank: int = 0            # every layer linear carries a bf16 low-rank adapter of this rank (QLoRALinearWrapper); with `rht` also the signs
    sliding_windows: Optional[List[int]] = None  # per attention layer (in layer order): window size, 0 = full attention (Gemma / gpt-oss pattern)
    sinks: bool = False            # every attention layer carries per-head sink logits (mixer.sinks)
    linear_biases: bool = False    # every layer linear carries an output bias (with `rht`: added behind the OutputRht, kernel.rs:296-303)
    # layer / decoder options of the Gemma families (transformer_layer.rs:61-184, decoder.rs:68-99)
    post_norms: bool = False       # post_mixer_norm + post_mlp_norm on every layer (the Gemma sandwich)
    post_layer_scalars: bool = False  # has_post_layer_scalar on every layer (needs post_norms)
    embedding_norm: bool = False
    normalize_values: bool = False
    layer_ropes: Optional[List[D.RopeConfig]] = None  # distinct RoPE configurations; rope_pattern picks one per attention layer (cyclic)
    rope_pattern: Optional[List[int]] = None
    kv_sharing: Optional[dict] = None  # {layer index: source layer index}: the layer's projection yields queries only, it reads the source's KV state
    ple_dim: int = 0               # > 0: per-layer embeddings (PLEModelConfig + a PLELayerConfig on every layer)
    ple_vocab_size: int = 0        # 0 = vocab_size
    first_layer_without_pre_mixer_norm: bool = False
    moe_experts: int = 0           # > 0: every layer's MLP is a MixtureOfExpertsConfig with this many routed experts (full-precision bf16 experts)
    moe_active: int = 2
    moe_hidden: int = 0            # expert_hidden_dim (0 = hidden_dim)
    moe_gelu: bool = False         # GEGLU experts (GELUApprox) instead of SwiGLU
    moe_clip: Optional[float] = None  # symmetric gate / up clipping at +-moe_clip
    non_causal_attention: bool = False  # AttentionConfig::is_causal == false on every attention layer (the block attention of a DFlash draft model)

    @property
    def num_layers(self) -> int:
        return len(self.layer_kinds)


def qwen35_0p8b(max_context_length: int = 4096, **kw) -> ModelConfig:
    """Qwen3.5-0.8B: 24 layers = 6 x [DeltaNet, DeltaNet, DeltaNet, gated attention]."""
    kinds = ([D.MIXER_DELTA_NET] * 3 + [D.MIXER_ATTENTION]) * 6
    cfg = ModelConfig(
        name="qwen3.5-0.8b", vocab_size=248320, model_dim=1024, hidden_dim=3584, layer_kinds=kinds,
        num_heads=8, num_groups=2, head_dim=256, has_gate=True, qk_norm=True,
        rope=D.RopeConfig(kind=D.ROPE_UNSCALED, head_dim=64, max_sequence_length=262144, base=10000000.0),
        dn_num_heads=16, dn_num_groups=16, dn_head_dim=128, dn_value_head_dim=128, dn_kernel_size=4,
        norm_epsilon=1e-6, norm_scale_offset=1.0, norm_full_layer=True,
        bits=4, group_size=128, method=D.QUANT_SCALE_BIAS, tied_embeddings=True,
        max_context_length=max_context_length,
        # seed 45: the greedy stream of the 2040-token synthetic prompt keeps moving (12 distinct tokens in 25 in the CPU
        # restatement) instead of falling into a fixed point after a few steps as most seeds do (tools/seed_search.py)
        seed=45)
    return replace(cfg, **kw)


def llama3_8b(max_context_length: int = 4096, **kw) -> ModelConfig:
    cfg = ModelConfig(
        name="llama-3-8b", vocab_size=128256, model_dim=4096, hidden_dim=14336,
        layer_kinds=[D.MIXER_ATTENTION] * 32, num_heads=32, num_groups=8, head_dim=128,
        rope=D.RopeConfig(kind=D.ROPE_LLAMA, head_dim=128, max_sequence_length=131072, base=500000.0,
                          scaling_factor=8.0, original_context_length=8192, low_frequency_factor=1.0,
                          high_frequency_factor=4.0),
        norm_epsilon=1e-5, norm_scale_offset=0.0, norm_full_layer=False,
        bits=4, group_size=128, method=D.QUANT_SCALE_BIAS, tied_embeddings=False,
        max_context_length=max_context_length)
    return replace(cfg, **kw)


def qwen3_14b_class(max_context_length: int = 8192 + 1024, **kw) -> ModelConfig:
    """BASELINE configs[4] "Qwen-14B-class int4": the public Qwen3-14B card: d 5120, 40 layers, 40 q / 8 kv
    heads of 128, q/k RMSNorm, ffn 17408, vocab 151 936, untied read-out, RoPE base 1e6."""
    cfg = ModelConfig(
        name="qwen3-14b-class", vocab_size=151936, model_dim=5120, hidden_dim=17408,
        layer_kinds=[D.MIXER_ATTENTION] * 40, num_heads=40, num_groups=8, head_dim=128, qk_norm=True,
        rope=D.RopeConfig(kind=D.ROPE_UNSCALED, head_dim=128, max_sequence_length=40960, base=1000000.0),
        norm_epsilon=1e-6, norm_scale_offset=0.0, norm_full_layer=False,
        bits=4, group_size=128, method=D.QUANT_SCALE_BIAS, tied_embeddings=False,
        max_context_length=max_context_length, seed=14)
    return replace(cfg, **kw)


def tiny_qwen(**kw) -> ModelConfig:
    """Same topology as Qwen3.5 (DeltaNet x3 + gated attention with q/k norm, partial rotary) at toy size."""
    cfg = ModelConfig(
        name="tiny-qwen", vocab_size=2048, model_dim=256, hidden_dim=512,
        layer_kinds=[D.MIXER_DELTA_NET, D.MIXER_DELTA_NET, D.MIXER_DELTA_NET, D.MIXER_ATTENTION],
        num_heads=4, num_groups=2, head_dim=64, has_gate=True, qk_norm=True,
        rope=D.RopeConfig(kind=D.ROPE_UNSCALED, head_dim=16, max_sequence_length=8192, base=10000000.0),
        dn_num_heads=4, dn_num_groups=2, dn_head_dim=128, dn_value_head_dim=128, dn_kernel_size=4,
        norm_epsilon=1e-6, norm_scale_offset=1.0, norm_full_layer=True,
        bits=4, group_size=64, method=D.QUANT_SCALE_BIAS, tied_embeddings=True, max_context_length=2048, seed=66)
    return replace(cfg, **kw)


def tiny_llama(**kw) -> ModelConfig:
    cfg = ModelConfig(
        name="tiny-llama", vocab_size=1024, model_dim=256, hidden_dim=768,
        layer_kinds=[D.MIXER_ATTENTION] * 3, num_heads=4, num_groups=2, head_dim=64,
        rope=D.RopeConfig(kind=D.ROPE_LLAMA, head_dim=64, max_sequence_length=8192, base=500000.0,
                          scaling_factor=8.0, original_context_length=1024, low_frequency_factor=1.0,
                          high_frequency_factor=4.0),
        norm_epsilon=1e-5, bits=4, group_size=32, method=D.QUANT_SCALE_ZERO_POINT, tied_embeddings=False,
        max_context_length=2048, seed=44)
    return replace(cfg, **kw)


def tiny_gemma(**kw) -> ModelConfig:
    """The layer / decoder options of the Gemma 3 / 3n families at toy size: local (sliding window, its own RoPE base) and global attention
    layers, sandwich norms, post-layer scalars, value normalisation, two trailing layers that share the KV state of an earlier layer of
    their kind, per-layer embeddings."""
    local = D.RopeConfig(kind=D.ROPE_UNSCALED, head_dim=64, max_sequence_length=8192, base=10000.0)
    glob = D.RopeConfig(kind=D.ROPE_LINEAR, head_dim=64, max_sequence_length=8192, base=1000000.0, scaling_factor=8.0)
    cfg = ModelConfig(
        name="tiny-gemma", vocab_size=1024, model_dim=256, hidden_dim=512,
        layer_kinds=[D.MIXER_ATTENTION] * 5, num_heads=4, num_groups=2, head_dim=64, qk_norm=True,
        rope=local, layer_ropes=[local, glob], rope_pattern=[0, 1, 0, 0, 1], sliding_windows=[48, 0, 48, 48, 0],
        kv_sharing={3: 0, 4: 1}, post_norms=True, post_layer_scalars=True, normalize_values=True, ple_dim=256,
        norm_epsilon=1e-6, norm_scale_offset=1.0, norm_full_layer=True,
        bits=4, group_size=32, method=D.QUANT_SCALE_BIAS, tied_embeddings=True, max_context_length=2048, seed=77)
    return replace(cfg, **kw)


PRESETS = {"tiny-gemma": tiny_gemma, "qwen3.5-0.8b": qwen35_0p8b, "llama-3-8b": llama3_8b, "qwen3-14b-class": qwen3_14b_class, "tiny-qwen": tiny_qwen, "tiny-llama": tiny_llama}


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def make_linear(cfg: ModelConfig, name: str, n: int, k: int, gain: float = 1.0,
                row_mult: Optional[np.ndarray] = None, out_bias: bool = False) -> D.LinearWeights:
    """Random quantised [n,k] matrix whose dequantised entries have std ~= gain/sqrt(k)."""
    rng = _rng(cfg.seed, name)
    bits, g, method = cfg.bits, cfg.group_size, cfg.method
    assert k % g == 0 and (k * bits) % 8 == 0
    groups = k // g
    if bits == 4:
        codes = rng.integers(0, 256, size=(n, k // 2), dtype=np.uint8)
        levels = 16
    else:
        codes = rng.integers(0, 256, size=(n, k), dtype=np.uint8)
        levels = 256
    code_std = np.sqrt((levels * levels - 1) / 12.0)
    base = gain / (np.sqrt(k) * code_std)
    scales = (base * rng.uniform(0.6, 1.4, size=(n, groups))).astype(np.float32)
    if row_mult is not None:
        scales *= row_mult.astype(np.float32)[:, None]
    scales_b = f32_to_bf16_bits(scales)
    biases_b = zero_points = None
    mid = (levels - 1) / 2.0
    if method == D.QUANT_SCALE_BIAS:
        sf = bf16_bits_to_f32(scales_b)
        biases = -mid * sf + rng.uniform(-0.03, 0.03, size=(n, groups)).astype(np.float32) * sf * 4.0
        biases_b = f32_to_bf16_bits(biases)
    elif method == D.QUANT_SCALE_ZERO_POINT:
        if bits == 4:
            zp = rng.integers(6, 10, size=(n, (groups + 1) // 2 * 2), dtype=np.uint8)
            zero_points = np.ascontiguousarray((zp[:, 0::2] | (zp[:, 1::2] << 4)).astype(np.uint8))
        else:
            zero_points = rng.integers(120, 136, size=(n, groups), dtype=np.uint8)
    ob = None
    if out_bias or (cfg.linear_biases and name.startswith("layers.")):
        ob = f32_to_bf16_bits(rng.uniform(-0.1, 0.1, size=(n,)).astype(np.float32))
    lw = D.LinearWeights(n, k, bits, g, method, codes, scales_b, biases_b, zero_points, ob)
    if cfg.rht and name.startswith("layers.") and n % 32 == 0 and k % 32 == 0:  # whole 32-wide Hadamard blocks on both sides
        signs = np.array([-1, 1], np.int32)
        lw.input_signs, lw.output_signs = np.ascontiguousarray(rng.choice(signs, k)), np.ascontiguousarray(rng.choice(signs, n))
    if cfg.qlora_rank and name.startswith("layers."):  # bf16 low-rank adapter next to the quantized base (a small correction: ~10 % of the base's output)
        r2 = _rng(cfg.seed, name + ".adapter")
        rank = cfg.qlora_rank
        lw.adapter_down = f32_to_bf16_bits((r2.normal(0.0, 1.0, size=(rank, k)) / np.sqrt(k)).astype(np.float32))
        lw.adapter_up = f32_to_bf16_bits((r2.normal(0.0, 0.1 * gain, size=(n, rank)) / np.sqrt(rank)).astype(np.float32))
    return lw


def readout_row_multipliers(cfg: ModelConfig) -> np.ndarray:
    """The log-normal per-token multipliers applied to the read-out rows (logit i scales with multiplier i)."""
    rng = _rng(cfg.seed, "row_mult")
    return np.exp(rng.normal(0.0, cfg.logit_row_sigma, size=(cfg.vocab_size,)))


def make_norm(cfg: ModelConfig, name: str, dim: int) -> D.NormWeights:
    rng = _rng(cfg.seed, name)
    centre = 0.0 if cfg.norm_scale_offset != 0.0 else 1.0
    scales = (centre + rng.uniform(-0.1, 0.1, size=(dim,))).astype(np.float32)
    return D.NormWeights(True, cfg.norm_full_layer, False, cfg.norm_epsilon, cfg.norm_scale_offset, scales, None)


def build_layers(cfg: ModelConfig) -> List[D.LayerWeights]:
    """The TransformerLayers of `cfg` (also the layers of a DFlash draft model: build_drafter)."""
    d = cfg.model_dim
    layers: List[D.LayerWeights] = []
    attn_index = 0
    for li, kind in enumerate(cfg.layer_kinds):
        p = f"layers.{li}."
        lw = D.LayerWeights(
            mixer_kind=kind, hidden_dim=cfg.hidden_dim, activation=D.ACT_SILU,
            pre_mixer_norm=D.ABSENT_NORM if (li == 0 and cfg.first_layer_without_pre_mixer_norm) else make_norm(cfg, p + "pre_mixer_norm", d),
            pre_mlp_norm=make_norm(cfg, p + "pre_mlp_norm", d),
            up_projection=None if cfg.moe_experts else make_linear(cfg, p + "mlp.up_projection", 2 * cfg.hidden_dim, d, gain=1.0),
            down_projection=None if cfg.moe_experts else make_linear(cfg, p + "mlp.down_projection", d, cfg.hidden_dim, gain=1.5),
        )
        if cfg.moe_experts:  # MoeBlock's tensors (mlp/moe/mod.rs:114-160): bf16, full precision; magnitudes as for the dense MLP
            r = _rng(cfg.seed, p + "mlp.moe")
            E, F = cfg.moe_experts, cfg.moe_hidden or cfg.hidden_dim
            bf = lambda a: f32_to_bf16_bits(np.asarray(a, np.float32))
            clip = (float("-inf"), float("inf")) if cfg.moe_clip is None else (-float(cfg.moe_clip), float(cfg.moe_clip))
            lw.moe = D.MoeWeights(
                num_routed_experts=E, num_active_experts=cfg.moe_active, expert_hidden_dim=F,
                router_weights=bf(r.normal(0.0, 1.0 / np.sqrt(d), (E, d))), router_biases=bf(r.uniform(-0.5, 0.5, (E,))),
                w13=bf(r.normal(0.0, 1.0 / np.sqrt(d), (E, 2 * F, d))), w2=bf(r.normal(0.0, 1.5 / np.sqrt(F), (E, d, F))),
                up_biases=bf(r.uniform(-0.05, 0.05, (E, 2 * F))), down_biases=bf(r.uniform(-0.05, 0.05, (E, d))),
                gating_sel=3 if cfg.moe_gelu else 2, gate_clip=clip, up_clip=clip)
        if cfg.post_norms:
            lw.post_mixer_norm, lw.post_mlp_norm = make_norm(cfg, p + "post_mixer_norm", d), make_norm(cfg, p + "post_mlp_norm", d)
        if cfg.post_layer_scalars:
            assert cfg.post_norms, "a post-layer scalar needs the post-MLP norm (transformer_layer.rs:61-66)"
            lw.post_layer_scalar = float(bf16_bits_to_f32(f32_to_bf16_bits(_rng(cfg.seed, p + "post_layer_scalar").uniform(0.5, 1.5, 1)))[0])
        if cfg.ple_dim:
            lw.ple = D.PleLayerWeights(
                ple_dim=cfg.ple_dim, activation=D.ACT_GELU_APPROX,
                gate=make_linear(cfg, p + "ple.gate", cfg.ple_dim, d, gain=1.0),
                projection=make_linear(cfg, p + "ple.projection", d, cfg.ple_dim, gain=1.0),
                norm=make_norm(cfg, p + "ple.norm", d))
        if kind == D.MIXER_ATTENTION:
            q_dim = cfg.num_heads * cfg.head_dim
            kv_dim = cfg.num_groups * cfg.head_dim
            lw.num_heads, lw.num_groups, lw.head_dim = cfg.num_heads, cfg.num_groups, cfg.head_dim
            lw.has_gate = cfg.has_gate
            lw.use_rope = cfg.rope.kind != D.ROPE_NONE
            if cfg.layer_ropes:
                lw.rope_index = int(cfg.rope_pattern[attn_index % len(cfg.rope_pattern)]) if cfg.rope_pattern else 0
            lw.normalize_values = cfg.normalize_values
            if cfg.kv_sharing and li in cfg.kv_sharing:
                lw.kv_source_layer_index = int(cfg.kv_sharing[li])
                kv_dim = 0  # the packed projection yields queries only (mixer/attention/mod.rs:89-95)
            lw.qkv_projection = make_linear(cfg, p + "mixer.qkv_projection", q_dim + 2 * kv_dim, d, gain=1.0)
            if cfg.has_gate:
                lw.gate_projection = make_linear(cfg, p + "mixer.gate_projection", q_dim, d, gain=1.0)
            lw.out_projection = make_linear(cfg, p + "mixer.out_projection", d, q_dim, gain=1.0)
            if cfg.qk_norm:
                lw.query_norm = make_norm(cfg, p + "mixer.query_norm", cfg.head_dim)
                if lw.kv_source_layer_index is None:
                    lw.key_norm = make_norm(cfg, p + "mixer.key_norm", cfg.head_dim)
            if cfg.sliding_windows is not None:
                lw.sliding_window_size = int(cfg.sliding_windows[attn_index % len(cfg.sliding_windows)])
            if cfg.sinks:
                lw.sinks = f32_to_bf16_bits(_rng(cfg.seed, p + "mixer.sinks").normal(0.0, 1.0, cfg.num_heads).astype(np.float32))
            lw.is_non_causal = cfg.non_causal_attention
            attn_index += 1
        else:
            Hv, Hk, Dk, Dv = cfg.dn_num_heads, cfg.dn_num_groups, cfg.dn_head_dim, cfg.dn_value_head_dim
            key_dim, value_dim = Hk * Dk, Hv * Dv
            conv_dim = 2 * key_dim + value_dim
            r = _rng(cfg.seed, p + "mixer.dn")
            lw.dn_num_heads, lw.dn_num_groups, lw.dn_head_dim, lw.dn_value_head_dim = Hv, Hk, Dk, Dv
            lw.dn_kernel_size = cfg.dn_kernel_size
            lw.dn_norm_epsilon = cfg.norm_epsilon
            lw.dn_in_proj = make_linear(cfg, p + "mixer.in_proj", conv_dim + value_dim + 2 * Hv, d, gain=1.0)
            lw.dn_out_proj = make_linear(cfg, p + "mixer.out_proj", d, value_dim, gain=1.0)
            lw.dn_conv_weights = r.uniform(-0.6, 0.6, size=(conv_dim, cfg.dn_kernel_size)).astype(np.float32)
            lw.dn_conv_biases = None
            lw.dn_a_log = r.uniform(-1.0, 1.0, size=(Hv,)).astype(np.float32)
            lw.dn_dt_bias = r.uniform(-1.0, 1.0, size=(Hv,)).astype(np.float32)
            lw.dn_norm_scales = (1.0 + r.uniform(-0.1, 0.1, size=(Dv,))).astype(np.float32)
        layers.append(lw)
    return layers


def build_model(cfg: ModelConfig) -> D.ModelBundle:
    d = cfg.model_dim
    row_mult = readout_row_multipliers(cfg)
    embedding = make_linear(cfg, "embedding", cfg.vocab_size, d, gain=1.0, row_mult=row_mult)
    output_embedding = None
    if not cfg.tied_embeddings:
        output_embedding = make_linear(cfg, "output_embedding", cfg.vocab_size, d, gain=1.0, row_mult=row_mult)
    if cfg.rht_embeddings:  # HybridSpec Output mode on the (input) table, Input mode on an untied output embedding (embedding.rs:126-341)
        signs = np.array([-1, 1], np.int32)
        embedding.output_signs = np.ascontiguousarray(_rng(cfg.seed, "embedding.signs").choice(signs, d))
        if output_embedding is not None:
            output_embedding.input_signs = np.ascontiguousarray(_rng(cfg.seed, "output_embedding.signs").choice(signs, d))
    layers = build_layers(cfg)
    ple = None
    if cfg.ple_dim:
        total = cfg.num_layers * cfg.ple_dim
        pv = cfg.ple_vocab_size or cfg.vocab_size
        ple = D.PleModelWeights(
            ple_dim=cfg.ple_dim, ple_vocab_size=pv, ple_embed_scale=float(np.sqrt(cfg.ple_dim)), model_projection_scale=float(1.0 / np.sqrt(d)),
            input_scale=float(1.0 / np.sqrt(2.0)),
            token_embedding=make_linear(cfg, "per_layer_embedding.token_embedding", pv, total, gain=1.0),
            model_projection=make_linear(cfg, "per_layer_embedding.model_projection", total, d, gain=1.0),
            projection_norm=make_norm(cfg, "per_layer_embedding.projection_norm", cfg.ple_dim))
    return D.ModelBundle(
        name=cfg.name, vocab_size=cfg.vocab_size, model_dim=d, max_context_length=cfg.max_context_length,
        rope=cfg.rope, embedding=embedding, output_norm=make_norm(cfg, "output_norm", d), layers=layers,
        tied_embeddings=cfg.tied_embeddings, output_embedding=output_embedding,
        ropes=list(cfg.layer_ropes) if cfg.layer_ropes else None,
        embedding_norm=make_norm(cfg, "embedding_norm", d) if cfg.embedding_norm else D.ABSENT_NORM, ple=ple)


def build_drafter(target: ModelConfig, num_layers: int = 2, block_size: int = 8, target_layer_ids: Optional[List[int]] = None, hidden_dim: int = 0, num_heads: int = 0,
                  num_groups: int = 0, head_dim: int = 0, non_causal: bool = True, qk_norm: Optional[bool] = None, mask_token_id: Optional[int] = None,
                  context_capacity: int = 0) -> "D.DFlashBundle":
    """A DFlash draft model for `target` (DFlashDraftConfig, config/dflash.rs:9-23; tensors of `speculator.draft_model`, encodable_block/dflash.rs:86-172):
    attention-only layers of the target's width, the context projection over the tapped target layers, the per-layer key / value projection."""
    ids = list(target_layer_ids) if target_layer_ids is not None else sorted({max(target.num_layers // 2 - 1, 0), target.num_layers - 1})
    assert all(0 <= i < target.num_layers for i in ids)
    rope = target.rope if target.rope.kind != D.ROPE_NONE else D.RopeConfig(kind=D.ROPE_UNSCALED, head_dim=head_dim or target.head_dim, max_sequence_length=target.max_context_length + 1024, base=10000.0)
    cfg = replace(target, name=target.name + "-dflash", layer_kinds=[D.MIXER_ATTENTION] * num_layers, hidden_dim=hidden_dim or target.hidden_dim,
                  num_heads=num_heads or target.num_heads, num_groups=num_groups or target.num_groups, head_dim=head_dim or target.head_dim, has_gate=False,
                  qk_norm=target.qk_norm if qk_norm is None else qk_norm, seed=target.seed + 7001, rope=replace(rope, head_dim=min(rope.head_dim, head_dim or target.head_dim)),
                  non_causal_attention=non_causal, sliding_windows=None, sinks=False, post_norms=False, post_layer_scalars=False, embedding_norm=False, normalize_values=False,
                  layer_ropes=None, rope_pattern=None, kv_sharing=None, ple_dim=0, first_layer_without_pre_mixer_norm=False, rht=False, rht_embeddings=False, qlora_rank=0)
    d = cfg.model_dim
    layers = build_layers(cfg)
    kv_dim = 2 * cfg.num_groups * cfg.head_dim
    return D.DFlashBundle(
        name=cfg.name, model_dim=d, hidden_dim=cfg.hidden_dim, block_size=block_size, mask_token_id=(target.vocab_size - 1) if mask_token_id is None else mask_token_id,
        target_layer_ids=ids, vocab_size=target.vocab_size, context_capacity=context_capacity or target.max_context_length,
        context_projection=make_linear(cfg, "context_projection", d, d * len(ids), gain=1.0), context_norm=make_norm(cfg, "context_norm", d),
        state_kv_projection=make_linear(cfg, "state_kv_projection", num_layers * kv_dim, d, gain=1.0), rope=cfg.rope, layers=layers, output_norm=make_norm(cfg, "output_norm", d))


def build_weaver(target: ModelConfig, model_dim: int = 0, num_layers: int = 2, num_heads: int = 4, hidden_dim: int = 0, max_depth: int = 16, candidate_pool_size: int = 16) -> "D.WeaverBundle":
    """A Weaver tree constructor for `target` (WeaverConfig, config/weaver.rs:5-17; tensors of `speculator.weaver`, encodable_block/weaver.rs:166-275)."""
    d = model_dim or target.model_dim
    assert d % num_heads == 0
    hd, hidden = d // num_heads, hidden_dim or target.hidden_dim
    cfg = replace(target, name=target.name + "-weaver", model_dim=d, hidden_dim=hidden, seed=target.seed + 9001, rht=False, rht_embeddings=False, qlora_rank=0, linear_biases=False,
                  moe_experts=0)
    rope = D.RopeConfig(kind=D.ROPE_UNSCALED, head_dim=hd, max_sequence_length=max_depth + 8, base=10000.0)
    layers = []
    for i in range(num_layers):
        p = f"blocks.{i}."
        layers.append(D.WeaverLayerWeights(
            pre_attention_norm=make_norm(cfg, p + "pre_attention_norm", d), pre_mlp_norm=make_norm(cfg, p + "pre_mlp_norm", d),
            qkv_projection=make_linear(cfg, p + "qkv_projection", 3 * d, d), out_projection=make_linear(cfg, p + "out_projection", d, d),
            up_projection=make_linear(cfg, p + "mlp.up_projection", 2 * hidden, d, out_bias=True), down_projection=make_linear(cfg, p + "mlp.down_projection", d, hidden, gain=1.5, out_bias=True)))
    td = target.model_dim
    return D.WeaverBundle(
        name=cfg.name, model_dim=d, target_model_dim=td, target_embedding_dim=td, num_heads=num_heads, hidden_dim=hidden, max_depth=max_depth, candidate_pool_size=candidate_pool_size,
        embedding_norm=make_norm(replace(cfg, model_dim=td), "embedding_norm", td), embedding_projection=make_linear(cfg, "embedding_projection", d, td, out_bias=True),
        hidden_state_norm=make_norm(replace(cfg, model_dim=td), "hidden_state_norm", td), hidden_state_projection=make_linear(cfg, "hidden_state_projection", d, td, out_bias=True),
        output_norm=make_norm(cfg, "output_norm", d), query_projection=make_linear(cfg, "query_projection", td, d), rope=rope, layers=layers)


def synthetic_prompt(length: int, vocab_size: int, variant: int = 0, suffix: int = 16) -> np.ndarray:
    """token ids = (i*7919 + 13) mod vocab.  `variant` != 0 shifts the last `suffix` ids by variant * 15485863 (mod vocab): a
    family of prompts that share all but their tail, used to pick a prompt whose greedy continuation has no near-tie (tools/stream_search.py)."""
    i = np.arange(length, dtype=np.int64)
    ids = (i * 7919 + 13) % vocab_size
    if variant:
        tail = i >= length - suffix
        ids[tail] = (ids[tail] + int(variant) * 15485863) % vocab_size
    return ids.astype(np.uint32)
