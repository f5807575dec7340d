"""CPU checks of the oracle's restatement of the Gemma-family layer / decoder options (transformer_layer.rs:61-184, transformer.rs:101-118,
205-216,249-275, decoder.rs:68-99,149-165, per_layer_embedding.rs, mixer/attention/mode.rs:79-84, qkv_norm.rs:70-72).  The reference holds
no known answers for these compositions, so each option is pinned by an identity the reference's op order implies -- a model WITH the option,
configured so that the option must not change the arithmetic, is BIT-identical to the model without it -- plus a check that the option does
change the result when it is not neutral, and that the fed-in-chunks / fed-at-once agreement of the sequence state survives it."""
import ctypes as C
from dataclasses import replace

import numpy as np
import pytest

from oracle import oracle as O
from uzu_amd import desc as D
from uzu_amd import synthetic as S


def f(b):
    return (np.asarray(b, np.uint16).astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def run(bundle, prompt, steps=4, chunks=None):
    om = O.OracleModel(bundle)
    if chunks:
        for lo, hi in zip(chunks[:-2], chunks[1:-1]):
            om.forward(prompt[lo:hi])
        tok, lg = om.forward(prompt[chunks[-2]:chunks[-1]], True)
    else:
        tok, lg = om.prefill(prompt, True)
    toks, logits = [tok], [lg]
    for _ in range(steps):
        tok, lg = om.forward([tok], True)
        toks.append(tok)
        logits.append(lg)
    om.close()
    return toks, logits


def same(a, b):
    return a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1], b[1]))


def spread(a, b):
    return max(np.abs(f(x) - f(y)).max() / f(x).std() for x, y in zip(a[1], b[1]))


BASE = dict(kv_sharing=None, post_layer_scalars=False, normalize_values=False, ple_dim=0, layer_ropes=None, rope_pattern=None)
PROMPT = S.synthetic_prompt(60, 1024)


def test_post_layer_scalar_of_one_is_the_plain_layer_and_any_other_value_is_not():
    """PostLayerScalar::ScaleResidualSum(1) / ScaleOutput(1) multiply by 1.0 and re-round a bf16 value: the same bits (normalization.rs:71-76,
    119-121 of the CPU kernel); a scalar != 1 scales the residual sum and the layer output (transformer_layer.rs:80-84)."""
    plain = S.build_model(S.tiny_gemma(**BASE))
    ones = S.build_model(S.tiny_gemma(**{**BASE, "post_layer_scalars": True}))
    for lw in ones.layers:
        lw.post_layer_scalar = 1.0
    assert same(run(plain, PROMPT), run(ones, PROMPT))
    scaled = S.build_model(S.tiny_gemma(**{**BASE, "post_layer_scalars": True}))
    assert all(lw.post_layer_scalar not in (None, 1.0) for lw in scaled.layers)
    assert spread(run(plain, PROMPT), run(scaled, PROMPT)) > 0.3


def test_post_layer_scalar_scales_residual_sum_and_layer_output_exactly_there():
    """One layer, scalar 0.5 (a power of two: exact in bf16).  With the output norm's input = hidden + shortcut, where the shortcut after the
    pre-MLP norm is (mixed + embedding) * s and the layer output is post_mlp_norm(...) * s, the tapped layer output must be exactly half of
    the tapped output of the scalar-free layer WHEN the MLP sees the same rows -- it does not (its input is normalised from the scaled sum, and
    RMS normalisation undoes a power-of-two scale exactly up to epsilon), so the comparison is: outputs equal to the plain model's times
    0.5 within 2 bf16 ulps, far from equal to the plain output itself."""
    cfg = S.tiny_gemma(**{**BASE, "layer_kinds": [D.MIXER_ATTENTION], "sliding_windows": [0], "post_layer_scalars": True})
    a = S.build_model(cfg)
    a.layers[0].post_layer_scalar = 0.5
    b = S.build_model(replace(cfg, post_layer_scalars=False))
    oa, ob = O.OracleModel(a), O.OracleModel(b)
    oa.forward(PROMPT[:9]), ob.forward(PROMPT[:9])
    ya, yb = f(oa.layer_output(0)), f(ob.layer_output(0))
    assert np.abs(ya - 0.5 * yb).max() <= 2.0 ** -6 * np.abs(yb).max()
    assert np.abs(ya - yb).max() > 0.2 * np.abs(yb).max()
    oa.close(), ob.close()


def test_identical_per_layer_rope_configurations_equal_the_single_rope_model():
    """Transformer::new dedups equal AnyRoPEConfig values (transformer.rs:109-114); two table entries with the same numbers must rotate like one."""
    cfg = S.tiny_gemma(**BASE)
    single = S.build_model(cfg)
    twin = S.build_model(replace(cfg, layer_ropes=[cfg.rope, replace(cfg.rope)], rope_pattern=[0, 1, 1, 0, 1]))
    assert twin.ropes is not None and [lw.rope_index for lw in twin.layers] == [0, 1, 1, 0, 1]
    assert same(run(single, PROMPT), run(twin, PROMPT))
    g = S.tiny_gemma()
    mixed = S.build_model(replace(cfg, layer_ropes=g.layer_ropes, rope_pattern=g.rope_pattern))
    assert spread(run(single, PROMPT), run(mixed, PROMPT)) > 0.3, "a layer rotated with another base / scaling must change the logits"


def test_embedding_norm_equals_a_table_of_pre_normalised_rows():
    """decoder.rs:149-154: Normalization (no shortcut) of the looked-up rows.  A full-precision input table that already holds
    norm(row) -- built with the oracle's own lookup + normalization KERNELS -- must give the same logits, bit for bit."""
    cfg = S.tiny_gemma(**{**BASE, "embedding_norm": True, "tied_embeddings": False})
    normed = S.build_model(cfg)
    ids = np.arange(cfg.vocab_size, dtype=np.uint32)
    E = normed.embedding
    rows = np.zeros((cfg.vocab_size, cfg.model_dim), np.uint16)
    O.call("orc_quantized_embedding_lookup", ids, E.weights, E.scales, E.zero_points, E.biases, rows, O.BF16, cfg.vocab_size, cfg.vocab_size, cfg.model_dim,
           1.0, E.group_size, E.bits, E.method)
    out = np.zeros_like(rows)
    N = normed.embedding_norm
    args = O.NormArgs(rows.ctypes.data, N.scales.ctypes.data, None, out.ctypes.data, None, O.BF16, O.F32, cfg.vocab_size, cfg.model_dim, N.epsilon, N.scale_offset, 1.0,
                      int(N.subtract_mean), int(N.full_layer), 0, 0, 0, 0)
    O.lib().orc_normalization(C.byref(args))
    table = S.build_model(replace(cfg, embedding_norm=False))
    table.embedding = D.LinearWeights(cfg.vocab_size, cfg.model_dim, 16, 0, D.QUANT_NONE, out)
    assert same(run(normed, PROMPT), run(table, PROMPT))
    assert spread(run(normed, PROMPT), run(S.build_model(replace(cfg, embedding_norm=False)), PROMPT)) > 0.3


def test_per_layer_embedding_with_a_zero_projection_norm_is_the_plain_model():
    """PerLayerEmbeddingProjection (per_layer_embedding.rs:217-270): shortcut += hidden; shortcut = (shortcut + norm(proj(act(gate(shortcut)) * ple
    slice))) * scalar; hidden = 0.  With ple.norm scales = 0 (scale_offset 0) the added term is +0 and the next Normalization adds hidden = 0 to the
    shortcut: the same residual stream, bit for bit, as the plain layer whose next Normalization adds `hidden` itself."""
    cfg = S.tiny_gemma(**{**BASE, "ple_dim": 32, "norm_scale_offset": 0.0})
    ple = S.build_model(cfg)
    for lw in ple.layers:
        lw.ple.norm = D.NormWeights(True, True, False, 1e-6, 0.0, np.zeros(cfg.model_dim, np.float32), None)
    plain = S.build_model(replace(cfg, ple_dim=0))
    assert same(run(plain, PROMPT), run(ple, PROMPT))
    live = S.build_model(cfg)
    assert spread(run(plain, PROMPT), run(live, PROMPT)) > 0.3


def test_per_layer_embedding_owns_the_post_layer_scalar():
    """transformer_layer.rs:78-84: with a PLE projection the two norms get PostLayerScalar::None and the projection's residual_combine scales
    the combined residual.  Zero ple.norm scales again: the layer reduces to `shortcut = (shortcut + hidden) * s` -- and with s = 1 to the plain layer."""
    cfg = S.tiny_gemma(**{**BASE, "ple_dim": 32, "post_layer_scalars": True, "norm_scale_offset": 0.0})
    ple = S.build_model(cfg)
    for lw in ple.layers:
        lw.ple.norm = D.NormWeights(True, True, False, 1e-6, 0.0, np.zeros(cfg.model_dim, np.float32), None)
        lw.post_layer_scalar = 1.0
    plain = S.build_model(replace(cfg, ple_dim=0, post_layer_scalars=False))
    assert same(run(plain, PROMPT), run(ple, PROMPT))
    # ONE layer, s = 0.5: the stream that reaches the output norm is exactly half of the plain layer's (a power of two: exact in bf16), which the
    # RMS norm undoes up to its epsilon -- while the norms' own scalar (which this configuration must NOT apply as well) would make it a quarter
    one = replace(cfg, layer_kinds=[D.MIXER_ATTENTION], sliding_windows=[0])
    half = S.build_model(one)
    half.layers[0].ple.norm = D.NormWeights(True, True, False, 1e-6, 0.0, np.zeros(cfg.model_dim, np.float32), None)
    half.layers[0].post_layer_scalar = 0.5
    oa, ob = O.OracleModel(half), O.OracleModel(S.build_model(replace(one, ple_dim=0, post_layer_scalars=False)))
    oa.capture_features(True), ob.capture_features(True)
    oa.forward(PROMPT[:5]), ob.forward(PROMPT[:5])
    # the PLE projection folds the layer's output into the shortcut and leaves hidden = 0 (transformer_layer.rs:231): the layer's tap is the residual row --
    # what Transformer::capture_residual(shortcut, hidden) files for it -- here exactly half of the plain layer's residual row
    assert np.array_equal(oa.layer_output(0), oa.hidden_feature(0))
    assert np.array_equal(f(oa.layer_output(0)), 0.5 * f(ob.hidden_feature(0)))
    la, lb = f(oa.final_hidden()), f(ob.final_hidden())
    assert np.abs(la - lb).max() <= 2.0 ** -6 * np.abs(lb).max()
    oa.close(), ob.close()


def test_value_normalisation_removes_the_scale_of_the_value_projection():
    """AttentionConfig::value_norm_config: scale-free RMS norm (eps 1e-6) of every value head.  Doubling the value rows of the packed
    projection (bf16 scales and biases x 2: exact) leaves the logits where they were up to the epsilon term; without the option the
    attention output doubles."""
    def doubled(bundle):
        for lw in bundle.layers:
            q = lw.qkv_projection
            v0 = (lw.num_heads + lw.num_groups) * lw.head_dim
            for arr in (q.scales, q.biases):
                arr[v0:] = S.f32_to_bf16_bits(S.bf16_bits_to_f32(arr[v0:]) * 2.0)
        return bundle
    on = S.tiny_gemma(**{**BASE, "normalize_values": True, "post_norms": False})  # (a post-mixer norm would undo the doubling by itself)
    a, b = run(S.build_model(on), PROMPT), run(doubled(S.build_model(on)), PROMPT)
    assert spread(a, b) < 0.1 and a[0] == b[0]  # (epsilon + bf16 re-rounding of the normalised values: measured 0.04 sigma)
    off = replace(on, normalize_values=False)
    assert spread(run(S.build_model(off), PROMPT), run(doubled(S.build_model(off)), PROMPT)) > 0.2
    assert spread(a, run(S.build_model(off), PROMPT)) > 0.2


def test_kv_sharing_layer_reads_the_state_its_source_layer_wrote():
    """transformer.rs:264-275 + mode.rs:79-84: a sharing layer projects queries only and attends over the source layer's keys / values --
    prefix AND this pass's suffix rows.  (1) The sequence state must not depend on how the prompt is cut into passes: one 60-token pass
    against passes of 7 + 21 + 32 tokens, full caches and wrapped rings (window 48 < 60).  (2) The source matters: pointing the last layer
    at the other owned layer changes the logits.  (3) A sharing layer owns nothing: its packed projection has heads * head_dim rows."""
    cfg = S.tiny_gemma(**{**BASE, "kv_sharing": {3: 0, 4: 1}})
    b = S.build_model(cfg)
    assert b.layers[3].qkv_projection.n == cfg.num_heads * cfg.head_dim and b.layers[3].kv_source_layer_index == 0 and b.layers[3].key_norm is D.ABSENT_NORM
    one, cut = run(b, PROMPT, 5), run(b, PROMPT, 5, chunks=[0, 7, 28, 60])
    assert one[0] == cut[0] and spread(one, cut) <= 0.25
    full = replace(cfg, sliding_windows=[0])
    one_f, cut_f = run(S.build_model(full), PROMPT, 5), run(S.build_model(full), PROMPT, 5, chunks=[0, 7, 28, 60])
    assert one_f[0] == cut_f[0] and all(np.array_equal(x, y) for x, y in zip(one_f[1], cut_f[1])), "full caches: the same keys in the same order, bit for bit"
    other = S.build_model(replace(full, kv_sharing={3: 0, 4: 2}))
    assert spread(one_f, run(other, PROMPT, 5)) > 0.2


def test_speculated_tree_on_the_gemma_options_equals_linear_decoding():
    """verify_tree + accept over a model with every option: a chain-shaped tree of the tokens the model would decode anyway samples the same
    tokens as the linear steps, and the state after accepting it continues like the linear run (sharing layers skip encode_accept,
    transformer.rs:63-69; ring layers take the accepted suffix rows)."""
    b = S.build_model(S.tiny_gemma())
    lin = run(b, PROMPT, 6)
    om = O.OracleModel(b)
    first = om.prefill(PROMPT)
    assert first == lin[0][0]
    chain = [lin[0][0], lin[0][1], lin[0][2]]
    trie = np.array([[i, len(chain) - 1, i] for i in range(len(chain))], np.uint32)  # {trie_start, trie_end, height} (subtree range, inclusive): a path
    sampled = om.verify_tree(chain, trie)
    assert list(sampled) == lin[0][1:4]
    om.accept(np.arange(len(chain), dtype=np.uint32))
    tok = int(sampled[-1])
    rest = []
    for _ in range(3):
        tok = om.forward([tok])
        rest.append(tok)
    assert rest == lin[0][4:7]
    om.close()


def test_all_options_together_feed_in_chunks_equals_feed_at_once():
    b = S.build_model(S.tiny_gemma(embedding_norm=True, first_layer_without_pre_mixer_norm=True))
    one, cut = run(b, PROMPT, 5), run(b, PROMPT, 5, chunks=[0, 13, 41, 60])
    assert one[0] == cut[0] and spread(one, cut) <= 0.25
    assert len(set(one[0])) >= 1


def test_tensor_parallel_planner_refuses_the_options_by_name():
    """uzu_amd/tp.py plans column / row splits of plain decoder layers; the options above run on one GPU only (the engine refuses such a shard too)."""
    from uzu_amd import tp
    for kw in (dict(), dict(BASE, embedding_norm=True), dict(BASE, normalize_values=True), dict(BASE, post_layer_scalars=True), dict(BASE, kv_sharing={4: 1}),
               dict(BASE, ple_dim=32)):
        with pytest.raises(NotImplementedError, match="post-layer scalars, an embedding norm, KV sharing"):
            tp.shard_bundle(S.build_model(S.tiny_gemma(**kw)), 0, 2)
    shard, _ = tp.shard_bundle(S.build_model(S.tiny_gemma(**dict(BASE, layer_ropes=S.tiny_gemma().layer_ropes, rope_pattern=[0, 1, 0, 0, 1]))), 0, 2)
    assert shard.ropes is not None and [l.rope_index for l in shard.layers] == [0, 1, 0, 0, 1], "per-layer RoPE configurations are replicated with the layers"
