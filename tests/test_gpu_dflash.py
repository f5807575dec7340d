"""GPU parity tests of the DFlash draft model and the speculative loop it closes (csrc/engine_drafter.hip <- encodable_block/dflash.rs:41-346; host logic
uzu_amd/speculator.py <- speculators/dflash_tfm.rs) against the CPU oracle (oracle/uzu_oracle_dflash.c):

  * the target's hidden-feature taps (production outputs, not UZU_MODEL_DEBUG_TAPS) == the oracle's capture_residual rows;
  * reference-order mode: draft hidden rows and the f32 draft logits BIT-IDENTICAL to the oracle's, after prefill accepts and after tree accepts; the
    proposed trie identical to the oracle's (tokens, ranges, heights, seeds);
  * production kernels: logits within the parity tolerance of the model tests (0.25 sigma), tokens equal where the oracle's decision is not a near-tie;
  * the whole speculative stream (propose -> verify -> accept on target and drafter) emits the oracle's tokens, and plain decoding's.
"""
import ctypes as C

import numpy as np
import pytest

from helpers import OracleTarget, f32
from oracle import oracle as O
from uzu_amd import _ffi
from uzu_amd import synthetic as S
from uzu_amd.engine import HipDrafter, HipModel
from uzu_amd.speculator import DFlashSpeculator, SpeculativeStream, TreeShape
from uzu_amd.trie import PRng

pytestmark = pytest.mark.gpu


def _set_exact(on):
    fn = _ffi.lib().uzu_hip_set_exact
    fn.restype, fn.argtypes = None, [C.c_int32]
    fn(1 if on else 0)


def _prompt(cfg, n=24):
    return ((S.synthetic_prompt(n, cfg.vocab_size).astype(np.int64) * 7 + 35) % cfg.vocab_size).astype(np.uint32)


def _pair(hip_ctx, cfg, **drafter_kw):
    bundle = S.build_model(cfg)
    db = S.build_drafter(cfg, **drafter_kw)
    om = O.OracleModel(bundle)
    om.capture_features(True)
    od = O.OracleDFlash(db)
    hm = HipModel(hip_ctx, bundle)
    hd = HipDrafter(hip_ctx, hm, db)
    return bundle, db, om, od, hm, hd


PRESETS = {"tiny-qwen": lambda: S.tiny_qwen(seed=34), "tiny-llama": lambda: S.tiny_llama()}


@pytest.mark.parametrize("preset", list(PRESETS))
def test_exact_mode_features_draft_and_trie_are_bit_identical(hip_ctx, preset):
    cfg = PRESETS[preset]()
    _set_exact(True)
    try:
        bundle, db, om, od, hm, hd = _pair(hip_ctx, cfg, block_size=8)
        prompt = _prompt(cfg, 40)
        o_tok, h_tok = om.prefill(prompt), hm.prefill(prompt)
        assert h_tok == o_tok
        o_feats, h_feats = [om.hidden_feature(l) for l in db.target_layer_ids], hm.hidden_features()
        assert len(h_feats) == len(o_feats) and all(np.array_equal(a, b) for a, b in zip(h_feats, o_feats))
        od.accept(o_feats, np.arange(40))
        hd.accept(h_feats, np.arange(40))
        assert hd.context_length == od.context_length == 40
        for batch in (8, 3):
            oh, ol, ot = od.draft(om, o_tok, batch)
            hh, hl, ht = hd.draft(hm, h_tok, batch, want_outputs=True)
            assert np.array_equal(hh, oh), f"draft hidden rows differ at block {batch}"
            assert np.array_equal(hl.view(np.uint32), ol.view(np.uint32)), f"f32 draft logits differ at block {batch}"
            assert np.array_equal(ht, ot)
        # the trie of the Argmax construction
        prng = PRng(99)
        ft_o = DFlashSpeculator(od).propose_tree(om, o_tok, TreeShape(tree_budget=6), prng).linearize()
        ft_h = DFlashSpeculator(hd).propose_tree(hm, h_tok, TreeShape(tree_budget=6), prng).linearize()
        assert np.array_equal(ft_h.token_ids(), ft_o.token_ids()) and np.array_equal(ft_h.nodes(), ft_o.nodes()) and np.array_equal(ft_h.token_seeds(), ft_o.token_seeds())
        # a verify pass + accept of a root path on both sides, then the drafters take the accepted rows' features: drafts stay bit-identical
        o_s, h_s = om.verify_tree(ft_o.token_ids(), ft_o.nodes()), hm.verify_tree(ft_h.token_ids(), ft_h.nodes())
        assert np.array_equal(h_s, o_s)
        full = ft_o.accept(o_s)
        idx = np.array([i for i, _, _ in full], dtype=np.uint32)
        o_feats, h_feats = [om.hidden_feature(l) for l in db.target_layer_ids], hm.hidden_features()
        assert all(np.array_equal(a, b) for a, b in zip(h_feats, o_feats))
        om.accept(idx), hm.accept(idx)
        od.accept(o_feats, idx), hd.accept(h_feats, idx)
        nxt = int(full[-1][2])
        oh, ol, ot = od.draft(om, nxt, 8)
        hh, hl, ht = hd.draft(hm, nxt, 8, want_outputs=True)
        assert np.array_equal(hl.view(np.uint32), ol.view(np.uint32)) and np.array_equal(hh, oh) and np.array_equal(ht, ot)
        hd.close(), hm.close()
    finally:
        _set_exact(False)


@pytest.mark.parametrize("preset", list(PRESETS))
def test_production_kernels_within_tolerance(hip_ctx, preset):
    cfg = PRESETS[preset]()
    bundle, db, om, od, hm, hd = _pair(hip_ctx, cfg, block_size=8)
    prompt = _prompt(cfg, 40)
    o_tok, h_tok = om.prefill(prompt), hm.prefill(prompt)
    o_feats, h_feats = [om.hidden_feature(l) for l in db.target_layer_ids], hm.hidden_features()
    for a, b in zip(h_feats, o_feats):  # residual-stream rows: a few bf16 ulps at the top binade (one ulp there = 2^-7 of the largest value)
        assert np.abs(f32(a).astype(np.float64) - f32(b)).max() <= 2.0 ** -5 * np.abs(f32(b)).max()
    od.accept(o_feats, np.arange(40))
    hd.accept(h_feats, np.arange(40))
    _, ol, ot = od.draft(om, o_tok, 8)
    _, hl, ht = hd.draft(hm, o_tok, 8, want_outputs=True)
    m = S.readout_row_multipliers(cfg).astype(np.float64)
    for r in range(7):
        w_o, w_h = ol[r].astype(np.float64), hl[r].astype(np.float64)
        sigma = (w_o / m).std()
        assert (np.abs(w_h - w_o) / m).max() <= 0.25 * sigma, f"row {r}: draft logits {(np.abs(w_h - w_o) / m).max() / sigma:.3f} sigma off the oracle's"
        if int(ht[r]) != int(ot[r]):  # only inside a near-tie of the oracle's own decision
            best, other = int(ot[r]), int(ht[r])
            assert (w_o[best] - w_o[other]) / (sigma * (m[best] + m[other])) < 0.1
    hd.close(), hm.close()


@pytest.mark.parametrize("exact", [True, False])
def test_speculative_stream_emits_the_oracles_tokens(hip_ctx, exact):
    """propose -> verify -> accept on target and drafter, ten rounds: the HIP stream == the oracle's stream == plain greedy decoding (smoke()'s model and
    prompt: every top-2 gap of that stream is >= 0.44 sigma), with the drafters' contexts in step.  Reference-order mode: every round's trie identical to the
    oracle's, node by node; production kernels: the same roots (the DRAFTED tokens of this random draft model are near-ties of its own logits -- it repeats one
    token -- so they may differ; whatever is drafted, the emitted tokens are the target's own arg-max)."""
    cfg = S.tiny_qwen(seed=34)
    _set_exact(exact)
    try:
        bundle, db, om, od, hm, hd = _pair(hip_ctx, cfg, block_size=8)
        prompt = _prompt(cfg, 24)
        plain = O.OracleModel(bundle)
        tok = plain.prefill(prompt)
        want = [tok]
        for _ in range(12):
            tok = plain.forward([tok])
            want.append(tok)
        so = SpeculativeStream(OracleTarget(om, db.target_layer_ids), DFlashSpeculator(od), seed=7, speculation_batch=6, prefill_chunk=16)
        sh = SpeculativeStream(hm, DFlashSpeculator(hd), seed=7, speculation_batch=6, prefill_chunk=16)
        assert so.prefill(prompt) == sh.prefill(prompt) == want[0]
        assert [want[0]] + so.generate(12) == want and [want[0]] + sh.generate(12) == want
        assert hd.context_length == hm.context_length == 24 + len(sh.tokens) - 1
        assert all(int(a.token_ids()[0]) == int(b.token_ids()[0]) for a, b in zip(sh.tries, so.tries))
        if exact:
            assert len(sh.tries) == len(so.tries)
            for a, b in zip(sh.tries, so.tries):
                assert np.array_equal(a.token_ids(), b.token_ids()) and np.array_equal(a.nodes(), b.nodes()) and np.array_equal(a.token_seeds(), b.token_seeds())
        hd.close(), hm.close()
    finally:
        _set_exact(False)


def test_drafter_refusals(hip_ctx):
    cfg = S.tiny_qwen(seed=34)
    bundle = S.build_model(cfg)
    hm = HipModel(hip_ctx, bundle)
    db = S.build_drafter(cfg, block_size=8)
    hd = HipDrafter(hip_ctx, hm, db)
    from uzu_amd._ffi import UzuHipError
    with pytest.raises(UzuHipError, match="out of the last pass"):  # nothing has run: no rows to accept
        hd.accept(None, [0])
    hm.prefill(_prompt(cfg, 8))
    with pytest.raises(UzuHipError, match="batch size"):
        hd.draft(hm, 1, 9)
    hm.set_feature_layers([0])
    with pytest.raises(UzuHipError, match="feature taps are not this drafter's"):
        hd.accept(None, [0])
    bad = S.build_drafter(cfg, block_size=8)
    bad.layers[1].sliding_window_size = 16
    with pytest.raises(UzuHipError, match="sliding-window"):
        HipDrafter(hip_ctx, hm, bad)
    hd.close(), hm.close()
