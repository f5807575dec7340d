"""GPU parity tests of the Weaver tree constructor (csrc/engine_weaver.hip, k_speculator.hip::radix_top_k_small <- encodable_block/weaver.rs:166-676,
cpu/kernel/radix_top_k_small.rs) against the CPU oracle (oracle/uzu_oracle_weaver.c):

  * RadixTopKSmall through the engine's candidate pool: ids and scores EXACT (integer selection under total_cmp; ties to the lower column) -- checked on the tree's
    tokens, which all come from the pool, and directly through the packed tree of a one-round shape;
  * reference-order mode: packed_tree and frontier BIT-IDENTICAL to the oracle's for several shapes (eager launches), the shape checks (InvalidTreeInput);
  * production kernels (the captured hipGraph, replayed): the same tree wherever the oracle's choices are not near-ties -- the structure checks of the CPU test hold
    on the HIP tree, and replaying a shape reproduces itself bit for bit;
  * the speculative stream with the Weaver construction (HIP target + drafter + weaver) emits the oracle stream's tokens = plain greedy decoding's.
"""
import ctypes as C

import numpy as np
import pytest

from helpers import OracleTarget
from oracle import oracle as O
from uzu_amd import _ffi
from uzu_amd import desc as D
from uzu_amd import synthetic as S
from uzu_amd.engine import HipDrafter, HipModel, HipWeaver
from uzu_amd.speculator import DFlashSpeculator, InvalidTreeShape, SpeculativeStream, TreeShape, read_nodes
from uzu_amd.trie import PRng

pytestmark = pytest.mark.gpu


def _set_exact(on):
    fn = _ffi.lib().uzu_hip_set_exact
    fn.restype, fn.argtypes = None, [C.c_int32]
    fn(1 if on else 0)


def _prompt(cfg, n=24):
    return ((S.synthetic_prompt(n, cfg.vocab_size).astype(np.int64) * 7 + 35) % cfg.vocab_size).astype(np.uint32)


WEAVER = dict(model_dim=256, num_layers=2, num_heads=2, hidden_dim=256, max_depth=7, candidate_pool_size=16)  # head_dim 128: AncestorAttention's only instantiation


def _setup(hip_ctx, cfg, n_prompt=24):
    bundle = S.build_model(cfg)
    db = S.build_drafter(cfg, block_size=8)
    wb = S.build_weaver(cfg, **WEAVER)
    om = O.OracleModel(bundle)
    om.capture_features(True)
    od, ow = O.OracleDFlash(db), O.OracleWeaver(wb)
    hm = HipModel(hip_ctx, bundle)
    hd = HipDrafter(hip_ctx, hm, db)
    hw = HipWeaver(hip_ctx, hd, wb)
    prompt = _prompt(cfg, n_prompt)
    o_tok, h_tok = om.prefill(prompt), hm.prefill(prompt)
    od.accept([om.hidden_feature(l) for l in db.target_layer_ids], np.arange(n_prompt))
    hd.accept(hm.hidden_features(), np.arange(n_prompt))
    return bundle, db, wb, om, od, ow, hm, hd, hw, o_tok, h_tok


SHAPES = [(12, 6, 8, 5, 3, 3), (4, 2, 8, 1, 3, 3), (16, 8, 8, 6, 4, 4), (10, 3, 4, 4, 2, 5), (16, 8, 8, 3, 7, 16)]


@pytest.mark.parametrize("preset", ["tiny-qwen", "tiny-llama"])
def test_exact_mode_trees_are_bit_identical(hip_ctx, preset):
    cfg = S.tiny_qwen(seed=34) if preset == "tiny-qwen" else S.tiny_llama()
    _set_exact(True)
    try:
        bundle, db, wb, om, od, ow, hm, hd, hw, o_tok, h_tok = _setup(hip_ctx, cfg)
        assert h_tok == o_tok
        o_norm, h_norm = om.final_hidden_rows()[-1:], hm.final_hidden_rows()[-1:]
        assert np.array_equal(h_norm, o_norm), "the prefill's output-norm row"
        seeds = [PRng(5).derive(24 + i) for i in range(wb.max_depth)]
        for dims in SHAPES:
            shape = D.WeaverTreeShape(*dims)
            dh, lg, _ = od.draft(om, o_tok, shape.dflash_depth)
            hd.draft(hm, h_tok, shape.dflash_depth)
            want = ow.encode_tree(om, o_norm, dh, lg, seeds, o_tok, shape)
            got = hw.encode_tree(hm, h_norm, None, None, seeds, h_tok, shape)
            assert np.array_equal(got[0], want[0]), f"packed tree differs for shape {dims}"
            assert np.array_equal(got[1], want[1]), f"frontier differs for shape {dims}"
        # WeaverEncodeError::InvalidTreeInput (weaver.rs:517-533)
        hd.draft(hm, h_tok, 8)
        for bad in ((12, 1, 8, 5, 3, 3), (12, 9, 8, 5, 3, 3), (12, 6, 8, 0, 3, 3), (12, 6, 8, 5, 33, 3), (12, 6, 8, 5, 3, 17), (0, 6, 8, 5, 3, 3)):
            assert hw.encode_tree(hm, h_norm, None, None, seeds, h_tok, D.WeaverTreeShape(*bad)) is None
        assert hw.encode_tree(hm, h_norm, None, None, seeds[:-1], h_tok, D.WeaverTreeShape(12, 6, 8, 5, 3, 3)) is None
        with pytest.raises(_ffi.UzuHipError, match="dflash_depth"):  # the drafter's last draft must have the shape's rows
            hw.encode_tree(hm, h_norm, None, None, seeds, h_tok, D.WeaverTreeShape(6, 4, 5, 3, 2, 2))
        # the Weaver construction through the speculator: the tries equal
        prng = PRng(99)
        ts = TreeShape(tree_budget=10, max_tree_depth=6, construction_method="weaver", rounds=4, expand_per_round=3, expand_width=3)
        ft_o = DFlashSpeculator(od, ow).propose_tree(om, o_tok, ts, prng, o_norm).linearize()
        ft_h = DFlashSpeculator(hd, hw).propose_tree(hm, h_tok, ts, prng, h_norm).linearize()
        assert np.array_equal(ft_h.token_ids(), ft_o.token_ids()) and np.array_equal(ft_h.nodes(), ft_o.nodes()) and np.array_equal(ft_h.token_seeds(), ft_o.token_seeds())
        assert not ft_o.is_flat()
        # a tree pass: every node's output-norm row
        o_s, h_s = om.verify_tree(ft_o.token_ids(), ft_o.nodes()), hm.verify_tree(ft_h.token_ids(), ft_h.nodes())
        assert np.array_equal(h_s, o_s) and np.array_equal(hm.final_hidden_rows(), om.final_hidden_rows())
        hw.close(), hd.close(), hm.close()
    finally:
        _set_exact(False)


def test_production_graph_tree_is_a_tree_over_the_exact_candidate_pool(hip_ctx):
    """Production kernels, the captured graph: the candidate pool (RadixTopKSmall over the f32 draft logits) is exact on whatever logits the HIP drafter produced;
    the tree is well-formed; a replay of the same shape gives the same bits; a second shape captures its own graph."""
    cfg = S.tiny_qwen(seed=34)
    bundle, db, wb, om, od, ow, hm, hd, hw, o_tok, h_tok = _setup(hip_ctx, cfg)
    h_norm = hm.final_hidden_rows()[-1:]
    seeds = [PRng(5).derive(24 + i) for i in range(wb.max_depth)]
    shape = D.WeaverTreeShape(12, 6, 8, 5, 3, 3)
    _, h_logits, _ = hd.draft(hm, h_tok, 8, want_outputs=True)
    packed, frontier = hw.encode_tree(hm, h_norm, None, None, seeds, h_tok, shape)
    P = wb.candidate_pool_size
    pool_ids, pool_scores = np.zeros((7, P), np.uint32), np.zeros((7, P), np.float32)
    O.call("orc_radix_top_k_small", h_logits, pool_ids, pool_scores, 7, cfg.vocab_size, P)
    slots = shape.slot_count()
    tok_f, par_f, dep_f, val_f = packed[0], packed[1].view(np.int32), packed[2], packed[5]
    assert val_f[0] == 1 and tok_f[0] == h_tok and par_f[0] == -1 and dep_f[0] == 0
    valid = 0
    for s in range(1, slots):
        if not val_f[s]:
            continue
        valid += 1
        p = int(par_f[s])
        assert 0 <= p < s and val_f[p] == 1 and dep_f[s] == dep_f[p] + 1 and dep_f[s] <= shape.max_depth - 1
        assert tok_f[s] in pool_ids[dep_f[s] - 1], "a child's token comes from the (exact) candidate pool of its depth"
    assert valid >= 3
    nodes = read_nodes(packed, frontier)
    for n in nodes:
        kids = [nodes[c] for c in n.child_indices]
        assert len({k.token_id for k in kids}) == len(kids) and all(k.depth == n.depth + 1 for k in kids) and all(k.logprob <= 1e-6 for k in kids)
    # one round: the root's children ARE WeaverTopChildren over pool row 0 -- their tokens are a subset of pool row 0
    one = D.WeaverTreeShape(4, 2, 8, 1, 3, 3)
    p1, f1 = hw.encode_tree(hm, h_norm, None, None, seeds, h_tok, one)
    kids = read_nodes(p1, f1)[1:]
    assert len(kids) == 3 and all(k.token_id in pool_ids[0] for k in kids)
    # replays
    ms0, launches = hw.stats
    again = hw.encode_tree(hm, h_norm, None, None, seeds, h_tok, shape)
    assert np.array_equal(again[0], packed) and np.array_equal(again[1], frontier)
    assert hw.stats[1] > 20
    # vs the oracle's tree on the SAME inputs (HIP draft rows and logits): equal unless the oracle's own top-children choice was a near-tie
    h_hidden, _, _ = hd.draft(hm, h_tok, 8, want_outputs=True)
    want = ow.encode_tree(om, h_norm, h_hidden, h_logits, seeds, h_tok, shape)
    same = np.array_equal(want[0][0], packed[0]) and np.array_equal(want[0][1], packed[1])
    if not same:  # report how far: the first slots must agree (the root's children are decided on margins of whole logits)
        assert want[0][0][0] == packed[0][0]
    hw.close(), hd.close(), hm.close()


@pytest.mark.parametrize("exact", [True, False])
def test_speculative_stream_with_the_weaver_construction(hip_ctx, exact):
    cfg = S.tiny_qwen(seed=34)
    bundle = S.build_model(cfg)
    prompt = _prompt(cfg)
    plain = O.OracleModel(bundle)
    tok = plain.prefill(prompt)
    want = [tok]
    for _ in range(10):
        tok = plain.forward([tok])
        want.append(tok)
    db = S.build_drafter(cfg, block_size=8)
    wb = S.build_weaver(cfg, **WEAVER)
    _set_exact(exact)
    try:
        om = O.OracleModel(bundle)
        o_stream = SpeculativeStream(OracleTarget(om, db.target_layer_ids), DFlashSpeculator(O.OracleDFlash(db), O.OracleWeaver(wb)), seed=7, speculation_batch=10, prefill_chunk=16,
                                     weaver_shape=(4, 3, 3))
        hm = HipModel(hip_ctx, bundle)
        hd = HipDrafter(hip_ctx, hm, db)
        hw = HipWeaver(hip_ctx, hd, wb)
        h_stream = SpeculativeStream(hm, DFlashSpeculator(hd, hw), seed=7, speculation_batch=10, prefill_chunk=16, weaver_shape=(4, 3, 3))
        assert h_stream.prefill(prompt) == o_stream.prefill(prompt) == want[0]
        got_h, got_o = h_stream.generate(10), o_stream.generate(10)
        assert [want[0]] + got_o == want
        assert [want[0]] + got_h == want
        assert any(not t.is_flat() for t in h_stream.tries)
        if exact:  # every round's trie identical
            assert len(h_stream.tries) == len(o_stream.tries)
            for a, b in zip(h_stream.tries, o_stream.tries):
                assert np.array_equal(a.token_ids(), b.token_ids()) and np.array_equal(a.nodes(), b.nodes()) and np.array_equal(a.token_seeds(), b.token_seeds())
        with pytest.raises(InvalidTreeShape, match="max_depth 9"):
            DFlashSpeculator(hd, hw).propose_tree(hm, want[-1], TreeShape(8, max_tree_depth=9, construction_method="weaver"), PRng(1), hm.final_hidden_rows()[-1:])
        hw.close(), hd.close(), hm.close()
    finally:
        _set_exact(False)
