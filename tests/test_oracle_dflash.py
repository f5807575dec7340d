"""CPU: the oracle's DFlash draft model (oracle/uzu_oracle_dflash.c <- encodable_block/dflash.rs:41-346) and the speculator's host logic
(uzu_amd/speculator.py <- speculators/dflash_tfm.rs:133-343).  The reference holds no vectors for this block; what can be pinned without it:

  * the target's hidden features are what Transformer::capture_residual files: Normalization(output_norm) of the LAST layer's feature rows is the pass's
    final hidden, bit for bit (the output norm adds the same shortcut: transformer.rs:160-171 vs :317-323);
  * encode_accept appends: accepting rows in two calls == accepting them in one (positions continue at the drafter's context), and it really reads the
    rows it is told to (other indices / other features => other drafts);
  * encode_draft accepts nothing: repeating it reproduces it, the context stays; the block attention follows AttentionConfig::is_causal (mask.rs:3-61):
    a non-causal block's row 1 sees the rows behind it (its logits change with the block length), a causal block's does not;
  * propose_tree (Argmax): chain of the greedy lookahead tokens under the root, seeds PRng::derive(root position + depth), shape errors as the reference's.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from uzu_amd import desc as D
from uzu_amd import synthetic as S
from uzu_amd.speculator import DFlashSpeculator, InvalidTreeShape, SpeculativeStream, TreeShape
from uzu_amd.trie import PRng

from helpers import OracleTarget


@pytest.fixture(scope="module")
def setup():
    cfg = S.tiny_qwen(seed=34)
    bundle = S.build_model(cfg)
    om = O.OracleModel(bundle)
    om.capture_features(True)
    prompt = S.synthetic_prompt(24, cfg.vocab_size)
    tok = om.prefill(prompt)
    return cfg, bundle, om, prompt, tok


def drafter_for(cfg, **kw):
    db = S.build_drafter(cfg, block_size=8, **kw)
    return db, O.OracleDFlash(db)


def test_last_layer_feature_normalises_to_the_final_hidden(setup):
    cfg, bundle, om, prompt, tok = setup
    last = len(bundle.layers) - 1
    feat = om.hidden_feature(last)
    assert feat.shape == (24, cfg.model_dim)
    n = bundle.output_norm
    row = np.ascontiguousarray(feat[-1])
    got = np.zeros(cfg.model_dim, dtype=np.uint16)
    # Normalization without a shortcut on the feature row (the fields of orc_norm_args in order: input, scales, biases, output, shortcut, io / affine dtype, rows,
    # dim, eps, scale_offset, post_layer_scalar, subtract_mean, full_layer, copy_to_shortcut, residual_add, scale_residual_sum, scale_output)
    args = O.NormArgs(row.ctypes.data, n.scales.ctypes.data, None, got.ctypes.data, None, O.BF16, O.F32, 1, cfg.model_dim, n.epsilon, n.scale_offset, 1.0,
                      int(n.subtract_mean), int(n.full_layer), 0, 0, 0, 0)
    O.lib().orc_normalization(C.byref(args))
    assert np.array_equal(got.reshape(-1), om.final_hidden())
    assert np.array_equal(om.final_hidden_rows()[-1], om.final_hidden())


def test_accept_in_two_calls_equals_one_call_and_reads_the_rows_it_is_given(setup):
    cfg, bundle, om, prompt, tok = setup
    db, a = drafter_for(cfg)
    _, b = drafter_for(cfg)
    feats = [om.hidden_feature(l) for l in db.target_layer_ids]
    a.accept(feats, np.arange(24))
    b.accept(feats, np.arange(10))
    # the second call's rows are re-based: row i of a later pass == row 10 + i of the first
    b.accept([f[10:] for f in feats], np.arange(14))
    assert a.context_length == b.context_length == 24
    ha, la, ta = a.draft(om, tok, 8)
    hb, lb, tb = b.draft(om, tok, 8)
    assert np.array_equal(la, lb) and np.array_equal(ha, hb) and np.array_equal(ta, tb)
    # other rows => another state
    _, c = drafter_for(cfg)
    c.accept(feats, np.arange(24)[::-1].copy())
    assert not np.array_equal(c.draft(om, tok, 8)[1], la)
    # other features => another state
    _, e = drafter_for(cfg)
    e.accept([feats[1], feats[0]], np.arange(24))
    assert not np.array_equal(e.draft(om, tok, 8)[1], la)
    assert np.isfinite(la).all() and la.std() > 0.1


def test_draft_accepts_nothing_and_block_attention_follows_is_causal(setup):
    cfg, bundle, om, prompt, tok = setup
    db, f = drafter_for(cfg)
    assert all(l.is_non_causal for l in db.layers)
    feats = [om.hidden_feature(l) for l in db.target_layer_ids]
    f.accept(feats, np.arange(24))
    h1, l1, t1 = f.draft(om, tok, 8)
    h2, l2, t2 = f.draft(om, tok, 8)
    assert np.array_equal(l1, l2) and np.array_equal(h1, h2) and f.context_length == 24
    # non-causal: row 1 of a 4-row block sees rows 2..3, of an 8-row block rows 2..7
    _, l4, _ = f.draft(om, tok, 4)
    assert not np.array_equal(l4[0], l1[0])
    # causal: row 1 sees rows 0..1 whatever follows
    dbc, fc = drafter_for(cfg, non_causal=False)
    assert not any(l.is_non_causal for l in dbc.layers)
    fc.accept(feats, np.arange(24))
    _, c8, _ = fc.draft(om, tok, 8)
    _, c4, _ = fc.draft(om, tok, 4)
    assert np.array_equal(c8[:3], c4)
    # the greedy tokens are the arg-max of the f32 logits, ties to the lowest id (unified_sampling.rs:90-98)
    assert [int(np.argmax(r)) for r in l1] == [int(t) for t in t1]


def test_propose_tree_argmax_chain_and_shape_errors(setup):
    cfg, bundle, om, prompt, tok = setup
    db, f = drafter_for(cfg)
    f.accept([om.hidden_feature(l) for l in db.target_layer_ids], np.arange(24))
    spec = DFlashSpeculator(f)
    assert spec.hidden_feature_layer_indices() == db.target_layer_ids and not spec.has_weaver()
    prng = PRng(1234)
    trie = spec.propose_tree(om, tok, TreeShape(tree_budget=6), prng)
    flat = trie.linearize()
    _, _, toks = f.draft(om, tok, 8)
    assert flat.is_flat() and len(flat) == 6
    assert [int(t) for t in flat.token_ids()] == [tok] + [int(t) for t in toks[:5]]
    assert [int(s) for s in flat.token_seeds()] == [prng.derive(24 + depth) for depth in range(6)]
    with pytest.raises(InvalidTreeShape, match="argmax chain of 9 nodes"):
        spec.propose_tree(om, tok, TreeShape(tree_budget=9), prng)
    with pytest.raises(InvalidTreeShape, match="outside 2..=8"):
        spec.propose_tree(om, tok, TreeShape(tree_budget=4, dflash_depth_override=9), prng)
    with pytest.raises(AssertionError):
        spec.propose_tree(om, tok, TreeShape(tree_budget=1), prng)
    # a shorter override drafts a shorter block: other lookahead rows under block attention
    t3 = spec.propose_tree(om, tok, TreeShape(tree_budget=3, dflash_depth_override=3), prng).linearize()
    _, _, toks3 = f.draft(om, tok, 3)
    assert [int(t) for t in t3.token_ids()] == [tok] + [int(t) for t in toks3[:2]]


def test_speculative_stream_equals_plain_greedy_decoding():
    """Speculation never changes the stream under greedy sampling: whatever the drafter proposes, the tokens are those of plain decoding (every emitted
    token is the target's own arg-max at an accepted position); and the drafter's context follows the target's."""
    cfg = S.tiny_qwen(seed=34)
    bundle = S.build_model(cfg)
    prompt = ((S.synthetic_prompt(24, cfg.vocab_size).astype(np.int64) * 7 + 35) % cfg.vocab_size).astype(np.uint32)
    plain = O.OracleModel(bundle)
    tok = plain.prefill(prompt)
    want = [tok]
    for _ in range(12):
        tok = plain.forward([tok])
        want.append(tok)
    om = O.OracleModel(bundle)
    db, f = drafter_for(cfg)
    stream = SpeculativeStream(OracleTarget(om, db.target_layer_ids), DFlashSpeculator(f), seed=7, speculation_batch=6, prefill_chunk=16)
    first = stream.prefill(prompt)
    assert first == want[0] and f.context_length == 24
    got = stream.generate(12)
    assert [first] + got == want[:13]
    assert f.context_length == om.context_length == 24 + len(stream.tokens) - 1  # every emitted token but the pending last one has been accepted
    assert stream.rounds >= 1 and stream.proposed == 5 * stream.rounds
