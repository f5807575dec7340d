"""GPU parity tests of the Gemma-family layer / decoder options in the HIP engine (csrc/engine_forward.hip::encode_forward) against the CPU oracle:
per-layer RoPE configurations (transformer.rs:101-118,249-257), post-layer scalars (transformer_layer.rs:61-84), embedding norm
(decoder.rs:68-83,149-154), KV sharing between layers (transformer.rs:264-275, mixer/attention/mode.rs:79-84), value normalisation
(qkv_norm.rs:70-72), per-layer embeddings (per_layer_embedding.rs), a first layer without pre-mixer norm (transformer_layer.rs:217-220).

Bars: in reference-order mode (uzu_hip_set_exact) the logits of EVERY step are BIT-identical to the oracle's, all vocabulary entries -- the
composition (which kernel, which operands, which order) is then proven, only the order of f32 additions separates the production kernels;
in production mode logits within 0.25 sigma of the row, arg-max equal wherever the oracle's own top-2 gap is outside that band."""
import ctypes as C
from dataclasses import replace

import numpy as np
import pytest

from helpers import bf16, f32, quant_matrix, ulp_diff_bf16
from test_gpu_kernels import activations, hip_matmul, oracle_matmul
from oracle import oracle as O
from uzu_amd import _ffi
from uzu_amd import backend as B
from uzu_amd import desc as D
from uzu_amd import synthetic as S
from uzu_amd.engine import MODEL_BATCH, MODEL_DEBUG_TAPS, MODEL_NO_GRAPH, HipModel
from uzu_amd.trie import TrieNode

pytestmark = pytest.mark.gpu

NONE = dict(kv_sharing=None, post_layer_scalars=False, normalize_values=False, ple_dim=0, layer_ropes=None, rope_pattern=None)
G = S.tiny_gemma()
OPTIONS = {
    "post_layer_scalar": dict(NONE, post_layer_scalars=True),
    "per_layer_rope": dict(NONE, layer_ropes=G.layer_ropes, rope_pattern=G.rope_pattern),
    "embedding_norm": dict(NONE, embedding_norm=True),
    "kv_sharing_rings_and_full": dict(NONE, kv_sharing={3: 0, 4: 1}),
    "kv_sharing_full_only": dict(NONE, kv_sharing={2: 1, 3: 0, 4: 1}, sliding_windows=[0]),
    "normalize_values": dict(NONE, normalize_values=True),
    "per_layer_embedding": dict(NONE, ple_dim=256),
    "per_layer_embedding_owns_the_scalar": dict(NONE, ple_dim=256, post_layer_scalars=True),
    "no_first_pre_mixer_norm": dict(NONE, first_layer_without_pre_mixer_norm=True),
    "everything": dict(embedding_norm=True, first_layer_without_pre_mixer_norm=True),
}


def _set_exact(on):
    fn = _ffi.lib().uzu_hip_set_exact
    fn.restype, fn.argtypes = None, [C.c_int32]
    fn(1 if on else 0)


def sigma_error(want_bits, got_bits):
    w, g = f32(want_bits).astype(np.float64), f32(got_bits).astype(np.float64)
    return float(np.abs(w - g).max() / w.std())


def top2_gap(bits):
    w = np.sort(f32(bits).astype(np.float64))
    return float((w[-1] - w[-2]) / w.std())


@pytest.mark.parametrize("option", sorted(OPTIONS))
def test_option_logits_bit_identical_to_the_oracle_in_reference_order_mode(hip_ctx, option):
    """A 70-token prompt (windows of 48 wrap inside the chunk: the ring of a SOURCE layer must keep its rows until the last layer that
    shares them has run) + 8 chained decode steps (context 70..78: a shared ring wraps further, one row per step)."""
    cfg = S.tiny_gemma(**OPTIONS[option])
    bundle = S.build_model(cfg)
    prompt = S.synthetic_prompt(70, cfg.vocab_size)
    om = O.OracleModel(bundle)
    _set_exact(True)
    try:
        hm = HipModel(hip_ctx, bundle)
        o_tok, o_logits = om.prefill(prompt, True)
        h_tok = hm.prefill(prompt)
        got = hm.read_logits()
        assert np.array_equal(o_logits, got), f"prefill: {(o_logits != got).sum()} of {got.size} logits differ (max {ulp_diff_bf16(o_logits, got).max()} bf16 ulps)"
        assert h_tok == o_tok
        for step in range(8):
            o_tok, o_logits = om.forward([o_tok], True)
            toks, _ = hm.decode(1)
            got = hm.read_logits()
            assert np.array_equal(o_logits, got), f"decode step {step}: {(o_logits != got).sum()} of {got.size} logits differ"
            assert int(toks[0]) == o_tok
        hm.close()
    finally:
        _set_exact(False)
        om.close()


@pytest.mark.parametrize("option", ["everything", "kv_sharing_rings_and_full", "per_layer_embedding_owns_the_scalar"])
@pytest.mark.parametrize("flags", [0, MODEL_NO_GRAPH])
def test_option_production_kernels_within_tolerance(hip_ctx, option, flags):
    """The production kernels (few-rows / GEMV linears, wave-parallel norms and attention), teacher-forced: logits within 0.25 sigma at every
    step, arg-max equal outside near-ties; decode replayed as a hipGraph and as plain launches.  These models are not fusable
    (model_fusable): the step is the one-kernel-per-reference-kernel pass with one row."""
    cfg = S.tiny_gemma(**OPTIONS[option])
    bundle = S.build_model(cfg)
    prompt = S.synthetic_prompt(70, cfg.vocab_size)
    om = O.OracleModel(bundle)
    hm = HipModel(hip_ctx, bundle, flags)
    o_tok, o_logits = om.prefill(prompt, True)
    h_tok = hm.prefill(prompt)
    worst = sigma_error(o_logits, hm.read_logits())
    assert h_tok == o_tok or top2_gap(o_logits) < 0.05
    for step in range(10):
        hm.set_next_token(o_tok)
        o_tok, o_logits = om.forward([o_tok], True)
        toks, _ = hm.decode(1)
        worst = max(worst, sigma_error(o_logits, hm.read_logits()))
        assert int(toks[0]) == o_tok or top2_gap(o_logits) < 0.05, f"step {step}: oracle {o_tok}, hip {int(toks[0])}, gap {top2_gap(o_logits):.3f} sigma"
    assert worst <= 0.25, f"logits {worst:.3f} sigma off the oracle's"
    print(f"{option} flags={flags}: logits within {worst:.3f} sigma")
    hm.close()
    om.close()


def test_long_prompt_takes_the_prefill_gemm_paths_with_every_option(hip_ctx):
    """1500 tokens = one pass of 1500 rows here (matrix-core GEMMs with their fused epilogues, flash-attention tiles, the PostNorm hand-over --
    which a post-layer scalar or a PLE projection must switch off), two passes of <= 1024 in the oracle; rings of 48 rows wrap 31 times.
    Production mode: logits of the last prompt row and of 4 teacher-forced steps within 0.25 sigma."""
    cfg = S.tiny_gemma(embedding_norm=True, max_context_length=1600)
    bundle = S.build_model(cfg)
    prompt = S.synthetic_prompt(1500, cfg.vocab_size)
    om = O.OracleModel(bundle)
    hm = HipModel(hip_ctx, bundle)
    o_tok, o_logits = om.prefill(prompt, True)
    hm.prefill(prompt)
    worst = sigma_error(o_logits, hm.read_logits())
    for _ in range(4):
        hm.set_next_token(o_tok)
        o_tok, o_logits = om.forward([o_tok], True)
        hm.decode(1)
        worst = max(worst, sigma_error(o_logits, hm.read_logits()))
    assert worst <= 0.25, f"logits {worst:.3f} sigma off the oracle's"
    hm.close()
    om.close()


@pytest.mark.parametrize("exact", [1, 0])
def test_prompt_fed_in_two_passes_stops_the_first_behind_the_last_owning_layer(hip_ctx, exact, monkeypatch):
    """Transformer::prefill_cache_layer_count (transformer.rs:186-199,239-243): a pass without output runs only up to the last layer that owns a
    state -- here layers 3 and 4 (KV sharing) are skipped in the first pass of a 1100-token prompt (1024 + 76 rows: the reference's own passes in
    reference-order mode, UZU_PREFILL_CHUNK=1024 in production mode), and the ring of layer 0, whose last reader was skipped, takes its rows behind
    layer 2.  The oracle runs every layer in both passes: same logits (bit-identical / within tolerance), same tokens."""
    monkeypatch.setenv("UZU_PREFILL_CHUNK", "1024")
    cfg = S.tiny_gemma(embedding_norm=True, max_context_length=1200)
    bundle = S.build_model(cfg)
    prompt = S.synthetic_prompt(1100, cfg.vocab_size)
    om = O.OracleModel(bundle)
    _set_exact(bool(exact))
    try:
        hm = HipModel(hip_ctx, bundle)
        o_tok, o_logits = om.prefill(prompt, True)
        h_tok = hm.prefill(prompt)
        worst = 0.0
        for step in range(4):
            got = hm.read_logits()
            if exact:
                assert np.array_equal(o_logits, got), f"step {step}: {(o_logits != got).sum()} of {got.size} logits differ"
                assert h_tok == o_tok
            else:
                worst = max(worst, sigma_error(o_logits, got))
                assert h_tok == o_tok or top2_gap(o_logits) < 0.05
                hm.set_next_token(o_tok)
            o_tok, o_logits = om.forward([o_tok], True)
            h_tok = int(hm.decode(1)[0][0])
        assert worst <= 0.25
        hm.close()
    finally:
        _set_exact(False)
        om.close()


def test_layer_taps_with_every_option_in_reference_order_mode(hip_ctx):
    """Per-layer outputs of a prefill pass, bit-identical: a PLE layer leaves hidden = 0 (transformer_layer.rs:231) and the residual stream
    carries everything -- its tap is the residual row (what Transformer::capture_residual would file: round 6, advisor); without PLE
    the taps are the post-MLP-norm outputs, scaled by the post-layer scalar."""
    for kw in (dict(ple_dim=0), dict()):
        cfg = S.tiny_gemma(**kw)
        bundle = S.build_model(cfg)
        prompt = S.synthetic_prompt(33, cfg.vocab_size)
        om = O.OracleModel(bundle)
        om.prefill(prompt)
        _set_exact(True)
        try:
            hm = HipModel(hip_ctx, bundle, MODEL_DEBUG_TAPS)
            hm.prefill(prompt)
            for layer in range(len(bundle.layers)):
                want, got = om.layer_output(layer), hm.read_layer_output(layer)
                assert np.array_equal(want, got), f"layer {layer}: {(want != got).sum()} of {got.size} elements differ"
                assert not (want == 0).all()
            hm.close()
        finally:
            _set_exact(False)
            om.close()


def test_speculated_tree_and_accept_with_every_option(hip_ctx):
    """verify_tree / accept (stream.rs:556-628, 380-470) over the options: sharing layers attend over their source's prefix + tree rows under
    the trie mask and own nothing to accept (transformer.rs:63-69); ring sources take the accepted rows.  Reference-order mode: the logits of
    every node -- right and wrong branches -- bit-identical to the oracle's; three rounds, then plain decoding continues on the oracle's stream."""
    cfg = S.tiny_gemma(embedding_norm=True)
    bundle = S.build_model(cfg)
    prompt = S.synthetic_prompt(41, cfg.vocab_size)
    om = O.OracleModel(bundle)
    tok = om.prefill(prompt)
    want = [tok]
    for _ in range(20):
        tok = om.forward([tok])
        want.append(tok)
    om.reset()
    om.prefill(prompt)
    _set_exact(True)
    try:
        hm = HipModel(hip_ctx, bundle)
        got = [hm.prefill(prompt)]
        assert got[0] == want[0]
        depth = 3
        for rnd in range(3):
            i = len(got) - 1
            root = TrieNode(got[-1])
            node = root
            for dpt in range(1, depth + 1):
                wrong = TrieNode((want[i + dpt] + 17 * dpt) % cfg.vocab_size)
                if dpt == 2:
                    wrong.add(TrieNode((want[i + dpt] + 5) % cfg.vocab_size))
                node.add(wrong)
                child = TrieNode(want[i + dpt])
                node.add(child)
                node = child
            flat = root.linearize()
            o_sampled, o_logits = om.verify_tree(flat.token_ids(), flat.nodes(), True)
            h_sampled = hm.verify_tree(flat.token_ids(), flat.nodes())
            h_logits = hm.read_tree_logits()
            assert np.array_equal(o_logits, h_logits), f"round {rnd}: {(o_logits != h_logits).sum()} tree logits differ"
            assert list(h_sampled) == list(o_sampled)
            acc = flat.accept(h_sampled)
            assert len(acc) == depth + 1
            om.accept([idx for idx, _, _ in acc])
            hm.accept([idx for idx, _, _ in acc])
            got.extend(int(s) for _, _, s in acc)
        assert got == want[:len(got)]
        toks, _ = hm.decode(len(want) - len(got))
        assert [int(t) for t in toks] == want[len(got):]
        hm.close()
    finally:
        _set_exact(False)
        om.close()


def test_batched_prefill_and_sequence_states_with_every_option(hip_ctx):
    """Two sequences through ONE batched prefill pass (uzu_hip_model_prefill_batch: the linears see 2 x 70 rows, attention and the deferred
    ring inserts of shared states run per sequence on its own state), then chained decode per state: the tokens of each sequence alone."""
    cfg = S.tiny_gemma()
    bundle = S.build_model(cfg)
    base = S.synthetic_prompt(70, cfg.vocab_size).astype(np.int64)
    prompts = np.stack([((base * m + 3 * m) % cfg.vocab_size).astype(np.uint32) for m in (1, 7)])
    hm = HipModel(hip_ctx, bundle, MODEL_BATCH(2))
    alone, alone_logits = [], []
    for i in range(2):
        hm.reset()
        seq = [hm.prefill(prompts[i])] + [int(t) for t in hm.decode(6)[0]]
        alone.append(seq)
        alone_logits.append(hm.read_logits())
    states = [hm.new_state() for _ in range(2)]
    first = hm.prefill_batch(states, prompts)
    for i in range(2):
        hm.bind(states[i])
        seq = [int(first[i])] + [int(t) for t in hm.decode(6)[0]]
        assert sigma_error(alone_logits[i], hm.read_logits()) <= 0.25
        assert seq == alone[i], f"sequence {i}: alone {alone[i]}, batched {seq}"
    hm.bind(None)
    for st in states:
        st.close()
    hm.close()


def test_engine_refuses_inconsistent_layer_options(hip_ctx):
    """The reference's construction errors as status codes: a post-layer scalar without post-MLP norm (TransformerLayerError::
    PostLayerScalarWithoutPostMlpNorm), a KV source that is not an earlier owning attention layer, a later layer without pre-mixer norm
    (MissingPreMixerNormConfig)."""
    def broken(mutate, **kw):
        bundle = S.build_model(S.tiny_gemma(**kw))
        mutate(bundle)
        with pytest.raises(B.UzuHipError) as e:
            HipModel(hip_ctx, bundle)
        return str(e.value)
    def no_post_mlp(b):
        b.layers[1].post_mlp_norm = D.ABSENT_NORM
    assert "post_mlp_norm" in broken(no_post_mlp)
    def forward_source(b):
        b.layers[3].kv_source_layer_index = 4
    assert "shares the KV state" in broken(forward_source)
    def chained_source(b):
        b.layers[4].kv_source_layer_index = 3
    assert "shares the KV state" in broken(chained_source)
    def window_mismatch(b):
        b.layers[4].kv_source_layer_index = 0
    assert "differ in window" in broken(window_mismatch)
    def second_without_norm(b):
        b.layers[1].pre_mixer_norm = D.ABSENT_NORM
    assert "pre_mixer_norm" in broken(second_without_norm)


# ------------------------------------------------------------------------------------------ the two kernel specialisations the options reach
@pytest.mark.parametrize("which", ["scale_residual_sum", "scale_output"])
@pytest.mark.parametrize("full_layer", [0, 1])
@pytest.mark.parametrize("dim,rows", [(256, 7), (1024, 40)])
def test_normalization_post_layer_scalar_specialisations(hip_ctx, which, full_layer, dim, rows):
    """Normalization { scale_residual_sum | scale_output } (normalization.rs:71-76,119-121 of the CPU kernel) through the C ABI, both kernels
    (general and the prefill rows kernel, >= 16 rows of a multiple of 1024 elements): shortcut write-back bit-exact, output <= 1-2 bf16 ulps."""
    rng = np.random.default_rng(dim + rows + full_layer)
    x, sc = bf16(rng.normal(0, 1.5, size=(rows, dim))), bf16(rng.normal(0, 1.5, size=(rows, dim)))
    scales = rng.uniform(-0.2, 0.2, size=(dim,)).astype(np.float32)
    srs, so = int(which == "scale_residual_sum"), int(which == "scale_output")
    scalar = 0.7109375
    want, want_sc = np.zeros_like(x), sc.copy()
    args = O.NormArgs(x.ctypes.data, scales.ctypes.data, None, want.ctypes.data, want_sc.ctypes.data, O.BF16, O.F32, rows, dim, 1e-6, 1.0, scalar, 0, full_layer, 1, 1, srs, so)
    O.lib().orc_normalization(C.byref(args))
    kern = B.NormalizationKernel.new(hip_ctx, B.BF16, B.F32, B.BF16, B.F32, 0, 0, full_layer, 1, 1, 0, srs, so, 0, 1)
    bx, bs, bo, bsc = hip_ctx.buffer_from(x), hip_ctx.buffer_from(scales), hip_ctx.create_buffer(x.nbytes), hip_ctx.buffer_from(sc)
    cb = hip_ctx.create_command_buffer("norm").start_encoding()
    kern.encode(bx, bs, None, bo, bsc, None, rows, dim, 1e-6, 1.0, scalar, cb)
    cb.end_encoding().submit().wait_until_completed()
    got = bo.download(np.uint16, rows * dim).reshape(rows, dim)
    assert np.array_equal(bsc.download(np.uint16, rows * dim).reshape(rows, dim), want_sc)
    assert ulp_diff_bf16(want, got).max() <= (1.0 if full_layer else 2.0)
    assert (want == got).mean() >= 0.98


@pytest.mark.parametrize("bits,method", [(4, 0), (4, 1), (8, 2)])
@pytest.mark.parametrize("m", [1, 3, 16, 130])
@pytest.mark.parametrize("n,k", [(256, 32), (32, 256), (40, 64), (256, 96), (1280, 160)])
def test_matmul_with_the_short_reductions_of_a_per_layer_embedding(hip_ctx, bits, method, m, n, k):
    """A PLE projection reduces over ple_dim and a PLE gate has ple_dim outputs: K and N far below the layer shapes of the other matmul tests
    (K = 32: one 32-weight lane step; N = 32 / 40: a fraction of a tile) through every kernel the row count selects (decode GEMV, few-rows,
    64-tile and 128-tile GEMMs or their fall-backs).  Same bar as test_gemv_quant_variants / test_gemm_mfma_variants."""
    rng = np.random.default_rng(bits * 1000 + method * 100 + m + n + k)
    q = quant_matrix(rng, n, k, bits, 32, method)
    a = activations(rng, m, k)
    want, got = oracle_matmul(a, q, m), hip_matmul(hip_ctx, a, q, m)
    ulps = ulp_diff_bf16(want, got)
    assert ulps.max() <= 1.0, f"max {ulps.max()} bf16 ulps"
    assert (want == got).mean() >= 0.97


@pytest.mark.parametrize("exact", [1, 0])
def test_per_layer_embedding_of_32_dimensions(hip_ctx, exact):
    cfg = S.tiny_gemma(ple_dim=32)
    bundle = S.build_model(cfg)
    prompt = S.synthetic_prompt(150, cfg.vocab_size)
    om = O.OracleModel(bundle)
    _set_exact(bool(exact))
    try:
        hm = HipModel(hip_ctx, bundle)
        o_tok, o_logits = om.prefill(prompt, True)
        hm.prefill(prompt)
        for step in range(4):
            got = hm.read_logits()
            if exact:
                assert np.array_equal(o_logits, got), f"step {step}: {(o_logits != got).sum()} of {got.size} logits differ"
            else:
                assert sigma_error(o_logits, got) <= 0.25
            hm.set_next_token(o_tok)
            o_tok, o_logits = om.forward([o_tok], True)
            hm.decode(1)
        hm.close()
    finally:
        _set_exact(False)
        om.close()


def test_gemma_shaped_layers_production_against_reference_order(hip_ctx):
    """The options at layer widths of the Gemma 3n class (model_dim 2048, 8 x 256 query heads over 2 KV heads, hidden 8192, PLE of 256 dimensions, group 128,
    sliding windows of 512; 6 layers: local, local, global, local -> shares layer 0, global -> shares layer 2, local): the CPU oracle does not
    reach these widths in test time, so -- as in the configuration-scale tests of test_gpu_model.py -- reference-order mode (bit-identical to the
    oracle wherever the oracle runs, test_option_logits_bit_identical_...) is the proxy: a 700-token prompt (the rings wrap) + 3 teacher-forced
    steps, production logits within 0.25 sigma, arg-max equal outside near-ties."""
    local = D.RopeConfig(kind=D.ROPE_UNSCALED, head_dim=256, max_sequence_length=8192, base=10000.0)
    glob = D.RopeConfig(kind=D.ROPE_LINEAR, head_dim=256, max_sequence_length=8192, base=1000000.0, scaling_factor=8.0)
    cfg = S.tiny_gemma(name="gemma-shaped", vocab_size=16384, model_dim=2048, hidden_dim=8192, layer_kinds=[D.MIXER_ATTENTION] * 6, num_heads=8, num_groups=2,
                       head_dim=256, rope=local, layer_ropes=[local, glob], rope_pattern=[0, 0, 1, 0, 1, 0], sliding_windows=[512, 512, 0, 512, 0, 512],
                       kv_sharing={3: 0, 4: 2}, ple_dim=256, group_size=128, max_context_length=1024, embedding_norm=True, seed=91)
    bundle = S.build_model(cfg)
    prompt = S.synthetic_prompt(700, cfg.vocab_size)
    runs = {}
    for exact in (1, 0):
        _set_exact(bool(exact))
        try:
            hm = HipModel(hip_ctx, bundle)
            tok = hm.prefill(prompt)
            rows = [(tok, hm.read_logits())]
            for _ in range(3):
                if not exact:
                    hm.set_next_token(runs[1][len(rows) - 1][0])
                tok = int(hm.decode(1)[0][0])
                rows.append((tok, hm.read_logits()))
            runs[exact] = rows
            hm.close()
        finally:
            _set_exact(False)
    worst = 0.0
    for (want_tok, want), (got_tok, got) in zip(runs[1], runs[0]):
        worst = max(worst, sigma_error(want, got))
        assert got_tok == want_tok or top2_gap(want) < 0.05
    assert worst <= 0.25, f"production logits {worst:.3f} sigma off reference-order mode"
    print(f"gemma-shaped layers: production within {worst:.3f} sigma of reference-order mode")


def test_first_model_of_a_fresh_process_with_a_short_prompt():
    """Regression test of the fill race: `hipMemset` on the null stream returns before the fill has happened and the engine's
    stream is non-blocking, so the zero fills of a new model's scratch used to land on buffers a SHORT first pass had already written (all-zero logits,
    token 0) -- only for the first model of a process, which no in-process test can be.  A subprocess creates a 2048-wide model (~200 MB of fills), runs
    a 70-token prompt at once, production mode first, then reference-order mode: same token, logits within tolerance, no logit exactly 0."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, 'tests')
from helpers import f32
from uzu_amd import _ffi, synthetic as S, desc as D
from uzu_amd.backend import Context
from uzu_amd.engine import HipModel
rope = D.RopeConfig(kind=D.ROPE_UNSCALED, head_dim=256, max_sequence_length=8192, base=10000.0)
cfg = S.tiny_llama(name='wide', vocab_size=4096, model_dim=2048, hidden_dim=8192, layer_kinds=[D.MIXER_ATTENTION] * 3, num_heads=8, num_groups=2, head_dim=256,
                   rope=rope, group_size=128, method=D.QUANT_SCALE_BIAS, max_context_length=1024, seed=91)
bundle = S.build_model(cfg)
prompt = S.synthetic_prompt(70, cfg.vocab_size)
ctx = Context.new(0)
fn = _ffi.lib().uzu_hip_set_exact
fn.restype, fn.argtypes = None, [C.c_int32]
out = []
for exact in (0, 1):
    fn(exact)
    hm = HipModel(ctx, bundle)
    out.append((hm.prefill(prompt), f32(hm.read_logits()).astype(np.float64)))
    hm.close()
fn(0)
(pt, pl), (et, el) = out
print('RESULT', pt, et, float(np.abs(pl - el).max() / el.std()), int((pl == 0).sum()))
"""
    res = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=120)
    line = [l for l in res.stdout.splitlines() if l.startswith("RESULT")]
    assert res.returncode == 0 and line, res.stdout + res.stderr
    prod_tok, exact_tok, err, zeros = line[0].split()[1:]
    assert int(zeros) == 0 and float(err) <= 0.25 and prod_tok == exact_tok, line[0]
