"""CPU: the oracle's Weaver tree constructor (oracle/uzu_oracle_weaver.c <- encodable_block/weaver.rs:166-676) and the host side of the Weaver construction
(uzu_amd/speculator.py <- speculators/dflash_tfm.rs:224-292, weaver.rs:60-113).  Pinned without the reference binary:

  * RadixTopKSmall against a NumPy sort under the reference's order (value descending by total_cmp, ties to the lower column), NaNs / signed zeros included;
  * the encoded tree is a tree: slot 0 = the root, every valid slot's parent is an earlier valid slot one level up, depths <= max_depth - 1, siblings carry
    distinct tokens drawn from their depth's candidate pool, frontier entries hang off valid slots; the shape checks of encode_tree (InvalidTreeInput);
  * round 0 by hand: the root's children = WeaverTopChildren of (pool row 0 logits + the sparse read-out of the root's query) -- composed here from the oracle
    KERNELS on the same inputs;
  * the speculative stream with the Weaver construction still emits plain greedy decoding's tokens."""
import ctypes as C

import numpy as np
import pytest

from helpers import OracleTarget, bf16, f32
from oracle import oracle as O
from uzu_amd import desc as D
from uzu_amd import synthetic as S
from uzu_amd.speculator import DFlashSpeculator, InvalidTreeShape, SpeculativeStream, TreeShape, read_nodes
from uzu_amd.trie import PRng


def total_order_key(v):
    b = np.asarray(v, np.float32).view(np.int32).astype(np.int64)
    return np.where(b < 0, b ^ 0x7FFFFFFF, b)  # f32::total_cmp


@pytest.mark.parametrize("rows,cols,k", [(1, 10, 10), (3, 1000, 7), (2, 5000, 512)])
def test_radix_top_k_small_against_a_sort(rows, cols, k):
    rng = np.random.default_rng(rows * 1000 + k)
    x = rng.normal(size=(rows, cols)).astype(np.float32)
    x[:, ::17] = x[:, 1::17][:, : x[:, ::17].shape[1]]  # ties
    if cols >= 1000:
        x[0, 5], x[0, 6], x[0, 7], x[0, 8] = np.nan, 0.0, -0.0, np.inf
    ids, scores = np.zeros((rows, k), np.uint32), np.zeros((rows, k), np.float32)
    O.call("orc_radix_top_k_small", x, ids, scores, rows, cols, k)
    for r in range(rows):
        key = total_order_key(x[r])
        want = sorted(range(cols), key=lambda c: (-int(key[c]), c))[:k]
        assert list(ids[r]) == want
        assert np.array_equal(scores[r].view(np.uint32), x[r][want].view(np.uint32))


@pytest.fixture(scope="module")
def setup():
    cfg = S.tiny_qwen(seed=34)
    bundle = S.build_model(cfg)
    om = O.OracleModel(bundle)
    om.capture_features(True)
    prompt = ((S.synthetic_prompt(24, cfg.vocab_size).astype(np.int64) * 7 + 35) % cfg.vocab_size).astype(np.uint32)
    tok = om.prefill(prompt)
    db = S.build_drafter(cfg, block_size=8)
    dr = O.OracleDFlash(db)
    dr.accept([om.hidden_feature(l) for l in db.target_layer_ids], np.arange(24))
    wb = S.build_weaver(cfg, model_dim=128, num_layers=2, num_heads=4, hidden_dim=256, max_depth=7, candidate_pool_size=16)
    wv = O.OracleWeaver(wb)
    return cfg, bundle, om, tok, db, dr, wb, wv


def encode(setup, shape):
    cfg, bundle, om, tok, db, dr, wb, wv = setup
    draft_hidden, logits, _ = dr.draft(om, tok, shape.dflash_depth)
    seeds = [PRng(5).derive(24 + i) for i in range(wb.max_depth)]
    return wv.encode_tree(om, om.final_hidden_rows()[-1:], draft_hidden, logits, seeds, tok, shape), draft_hidden, logits, seeds


def test_encoded_tree_is_a_tree_over_the_candidate_pool(setup):
    cfg, bundle, om, tok, db, dr, wb, wv = setup
    shape = D.WeaverTreeShape(12, 6, 8, 5, 3, 3)
    (packed, frontier), draft_hidden, logits, seeds = encode(setup, shape)
    slots = shape.slot_count()
    assert packed.shape == (6, slots) and frontier.shape == (7, slots * 3)
    tok_f, par_f, dep_f, val_f = packed[0], packed[1].view(np.int32), packed[2], packed[5]
    assert val_f[0] == 1 and tok_f[0] == tok and par_f[0] == -1 and dep_f[0] == 0
    pool_ids = np.zeros((7, 16), np.uint32)
    O.call("orc_radix_top_k_small", logits, pool_ids, np.zeros((7, 16), np.float32), 7, cfg.vocab_size, 16)
    for s in range(1, slots):
        if not val_f[s]:
            continue
        p = int(par_f[s])
        assert 0 <= p < s and val_f[p] == 1 and dep_f[s] == dep_f[p] + 1 and dep_f[s] <= shape.max_depth - 1
        assert tok_f[s] in pool_ids[dep_f[s] - 1], "a child's token comes from the candidate pool of its depth"
    nodes = read_nodes(packed, frontier)
    assert nodes[0].token_id == tok and nodes[0].depth == 0 and len(nodes) > 1
    for n in nodes:
        kids = [nodes[c] for c in n.child_indices]
        assert len({k.token_id for k in kids}) == len(kids) and all(k.depth == n.depth + 1 for k in kids)
        assert all(k.logprob <= 1e-6 for k in kids)  # log-softmax values
    # shape checks (weaver.rs:517-533)
    for bad in (D.WeaverTreeShape(12, 1, 8, 5, 3, 3), D.WeaverTreeShape(12, 9, 8, 5, 3, 3), D.WeaverTreeShape(12, 6, 5, 5, 3, 3), D.WeaverTreeShape(12, 6, 8, 0, 3, 3),
                D.WeaverTreeShape(12, 6, 8, 5, 33, 3), D.WeaverTreeShape(12, 6, 8, 5, 3, 17), D.WeaverTreeShape(0, 6, 8, 5, 3, 3)):
        dh, lg, _ = dr.draft(om, tok, 8)
        assert wv.encode_tree(om, om.final_hidden_rows()[-1:], dh, lg, seeds, tok, bad) is None


def test_round_zero_by_hand(setup):
    """One round: the root's children = WeaverTopChildren over pool row 0 with the residual logits of the root's query -- every step composed from the oracle's
    kernels on the same inputs (prefix of ONE layer keeps the hand composition short)."""
    cfg, bundle, om, tok, db, dr, _, _ = setup
    wb = S.build_weaver(cfg, model_dim=128, num_layers=1, num_heads=4, hidden_dim=256, max_depth=7, candidate_pool_size=16)
    wv = O.OracleWeaver(wb)
    shape = D.WeaverTreeShape(4, 2, 8, 1, 3, 3)
    draft_hidden, logits, _ = dr.draft(om, tok, 8)
    seeds = np.array([PRng(5).derive(24 + i) for i in range(7)], np.uint64)
    packed, frontier = wv.encode_tree(om, om.final_hidden_rows()[-1:], draft_hidden, logits, seeds, tok, shape)
    nodes = read_nodes(packed, frontier)
    assert len(nodes) == 4 and nodes[0].child_indices == [1, 2, 3]
    # by hand
    d, td, hd, P = 128, cfg.model_dim, 32, 16
    pool_ids, pool_logits = np.zeros((7, P), np.uint32), np.zeros((7, P), np.float32)
    O.call("orc_radix_top_k_small", logits, pool_ids, pool_logits, 7, cfg.vocab_size, P)

    def norm(nw, x, shortcut=None, mode=0):
        out = np.zeros_like(x)
        args = O.NormArgs(x.ctypes.data, nw.scales.ctypes.data, None, out.ctypes.data, shortcut.ctypes.data if shortcut is not None else None, O.BF16, O.F32, x.shape[0], x.shape[1],
                          nw.epsilon, nw.scale_offset, 1.0, 0, int(nw.full_layer), int(mode != 0), int(mode == 2), 0, 0)
        O.lib().orc_normalization(C.byref(args))
        return out

    def linear(lw, x, gather=None, n=None):
        n = n or lw.n
        out = np.zeros((x.shape[0], n), np.uint16)
        g = O.MatmulArgs()
        g.a, g.a_dtype, g.b, g.scales = x.ctypes.data, O.BF16, lw.weights.ctypes.data, lw.scales.ctypes.data
        g.biases = lw.biases.ctypes.data if lw.biases is not None else None
        g.w_dtype, g.method, g.bits, g.group_size, g.b_transpose = O.BF16, lw.method, lw.bits, lw.group_size, 1
        g.d, g.d_dtype, g.ab_scale = out.ctypes.data, O.BF16, 1.0
        g.bias = lw.out_biases.ctypes.data if lw.out_biases is not None and gather is None else None
        g.gather_indices = gather.ctypes.data if gather is not None else None
        g.m, g.n, g.k = x.shape[0], n, lw.k
        O.lib().orc_matmul(C.byref(g))
        return out
    # prefix: [target row; draft rows 1..] -> hidden_state_norm -> projection -> (one layer: it only contributes keys / values)
    prefix = np.concatenate([om.final_hidden_rows()[-1:], draft_hidden[1:]])
    rin = linear(wb.hidden_state_projection, norm(wb.hidden_state_norm, prefix))
    L = wb.layers[0]
    rstate = np.zeros_like(rin)
    qkv = linear(L.qkv_projection, norm(L.pre_attention_norm, rin, rstate, 1))
    pos = np.arange(8, dtype=np.uint32)
    cos, sin = np.zeros((8, hd), np.float32), np.zeros((8, hd), np.float32)
    rope = wb.rope.desc()
    O.lib().orc_rope_tables(C.byref(rope), pos.ctypes.data_as(C.c_void_p), C.c_uint32(8), cos.ctypes.data_as(C.c_void_p), sin.ctypes.data_as(C.c_void_p))
    queries, kv = np.zeros((4, 8, hd), np.uint16), np.zeros((2, 8, d), np.uint16)
    O.call("orc_attention_prepare", qkv, queries, kv[0], kv[1], cos, sin, 4, 4, hd, hd, 0, 8, 1)
    # the root node
    emb = np.zeros((1, td), np.uint16)
    E = bundle.embedding
    O.call("orc_quantized_embedding_lookup", np.array([tok], np.uint32), E.weights, E.scales, E.zero_points, E.biases, emb, O.BF16, 1, cfg.vocab_size, td, 1.0, E.group_size, E.bits, E.method)
    nrin = linear(wb.embedding_projection, norm(wb.embedding_norm, emb))
    nstate = np.zeros_like(nrin)
    cqkv = linear(L.qkv_projection, norm(L.pre_attention_norm, nrin, nstate, 1))
    node_kv, meta, anc = np.zeros((2, 1, d), np.uint16), np.zeros(3, np.uint32), np.zeros(7, np.uint32)
    att = np.zeros((1, d), np.uint16)
    O.call("orc_ancestor_attention", kv, node_kv, cqkv, cos, sin, meta, anc, meta[1:2], meta[2:3], att, 1, 8, 7, 1, 7, float(1.0 / np.sqrt(hd)), 4, hd)
    mlp_in = norm(L.pre_mlp_norm, linear(L.out_projection, att), nstate, 2)
    up = linear(L.up_projection, mlp_in)
    gated = np.zeros((1, 256), np.uint16)
    O.call("orc_gated_act_mul", up, None, gated, O.BF16, 256, 1, 0, 0, D.ACT_SILU, 1)
    down = linear(L.down_projection, gated)
    query = linear(wb.query_projection, norm(wb.output_norm, down, nstate, 2))
    residual = linear(E, query, gather=pool_ids[:1].copy(), n=P)
    kids, lps = np.zeros((1, 3), np.uint32), np.zeros((1, 3), np.float32)
    O.call("orc_weaver_top_children", residual, pool_logits[:1].copy(), pool_ids[:1].copy(), seeds, meta, kids, lps, 1, P, 3, cfg.vocab_size)
    assert [n.token_id for n in nodes[1:]] == [int(t) for t in kids[0]]
    assert np.array_equal(np.array([n.logprob for n in nodes[1:]], np.float32).view(np.uint32), lps[0].view(np.uint32))


def test_speculative_stream_with_the_weaver_construction_equals_plain_greedy_decoding():
    cfg = S.tiny_qwen(seed=34)
    bundle = S.build_model(cfg)
    prompt = ((S.synthetic_prompt(24, cfg.vocab_size).astype(np.int64) * 7 + 35) % cfg.vocab_size).astype(np.uint32)
    plain = O.OracleModel(bundle)
    tok = plain.prefill(prompt)
    want = [tok]
    for _ in range(10):
        tok = plain.forward([tok])
        want.append(tok)
    om = O.OracleModel(bundle)
    db = S.build_drafter(cfg, block_size=8)
    wb = S.build_weaver(cfg, model_dim=128, num_layers=2, num_heads=4, hidden_dim=256, max_depth=7, candidate_pool_size=16)
    spec = DFlashSpeculator(O.OracleDFlash(db), O.OracleWeaver(wb))
    assert spec.has_weaver()
    stream = SpeculativeStream(OracleTarget(om, db.target_layer_ids), spec, seed=7, speculation_batch=10, prefill_chunk=16, weaver_shape=(4, 3, 3))
    first = stream.prefill(prompt)
    got = stream.generate(10)
    assert [first] + got == want
    assert all(len(t) <= 10 for t in stream.tries) and any(not t.is_flat() for t in stream.tries), "the Weaver construction proposes branching trees within the budget"
    with pytest.raises(InvalidTreeShape, match="max_depth 9"):
        spec.propose_tree(OracleTarget(om, db.target_layer_ids), want[-1], TreeShape(8, max_tree_depth=9, construction_method="weaver"), PRng(1), om.final_hidden_rows()[-1:])
