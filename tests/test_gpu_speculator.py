"""GPU parity tests of the tree speculators' kernels (csrc/k_speculator.hip <- cpu/kernel/attention/ancestor_attention.rs, cpu/kernel/weaver/*.rs)
through the C ABI against the CPU oracle, on the procedural inputs of the reference's own tests (ancestor_attention_test.rs, weaver_frontier_test.rs,
weaver_top_children_test.rs) plus random ones.  Index / token / bit-pattern outputs are compared bit-exact; the attention output within two bf16
ulps or 0.02 (the reference test's own bound is 0.02); log-probabilities within 1e-5 (the reference test's bound)."""
import numpy as np
import pytest

from helpers import f32, ulp_diff_bf16
from test_gpu_kernels import run
from test_oracle_speculator import (ancestor_attention_inputs, frontier_select_inputs, insert_children_inputs, run_ancestor_attention, run_frontier_select, run_insert_children,
                                    run_top_children, top_children_inputs)
from uzu_amd import backend as B
from uzu_amd._ffi import UzuHipError as UzuError

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------ AncestorAttention
def hip_ancestor_attention(ctx, x):
    kernel = B.AncestorAttentionKernel.new(ctx, x["head_dim"], x["num_heads"])
    bufs = {k: ctx.buffer_from(x[k]) for k in ("prefix_kv", "node_kv", "current_qkv", "cosines", "sines", "node_metadata", "ancestor_indices", "ancestor_counts", "node_indices")}
    n_out = x["rows"] * x["num_heads"] * x["head_dim"]
    out = ctx.buffer_from(np.zeros(n_out, np.uint16))
    run(ctx, lambda cb: kernel.encode(bufs["prefix_kv"], bufs["node_kv"], bufs["current_qkv"], bufs["cosines"], bufs["sines"], bufs["node_metadata"], bufs["ancestor_indices"],
                                      bufs["ancestor_counts"], bufs["node_indices"], out, x["rows"], x["prefix_length"], x["ancestor_stride"], x["node_capacity"], x["max_depth"],
                                      x["scale"], cb))
    return out.download(np.uint16, n_out), bufs["node_kv"].download(np.uint16, x["node_kv"].size)


def check_ancestor_attention(ctx, x):
    want_out, want_kv = run_ancestor_attention(x)
    got_out, got_kv = hip_ancestor_attention(ctx, x)
    assert np.array_equal(got_kv, want_kv)  # the rotated keys (same f32 expression, one rounding) and the values: bit-exact
    diff = np.abs(f32(got_out) - f32(want_out))
    assert ((ulp_diff_bf16(got_out, want_out) <= 2) | (diff <= 2.0 ** -9)).all() and diff.max() <= 0.02


def test_ancestor_attention_on_the_reference_tests_inputs(hip_ctx):
    check_ancestor_attention(hip_ctx, ancestor_attention_inputs())


def test_ancestor_attention_reference_order_form_is_bit_identical(hip_ctx):
    """uzu_hip_set_exact(1): one sequential dot product per key, one online-softmax chain in key order, glibc-exact exp => the CPU kernel's bits"""
    import ctypes as C
    from uzu_amd import _ffi
    fn = _ffi.lib().uzu_hip_set_exact
    fn.restype, fn.argtypes = None, [C.c_int32]
    fn(1)
    try:
        for x in (ancestor_attention_inputs(), ancestor_attention_inputs(rows=9, prefix_length=300, ancestor_stride=6, nodes=40, num_heads=4, max_depth=7)):
            want_out, want_kv = run_ancestor_attention(x)
            got_out, got_kv = hip_ancestor_attention(hip_ctx, x)
            assert np.array_equal(got_kv, want_kv) and np.array_equal(got_out, want_out)
    finally:
        fn(0)


@pytest.mark.parametrize("rows,prefix,stride,heads", [(1, 0, 3, 16), (7, 1, 4, 2), (16, 37, 6, 8), (32, 300, 8, 16)])
def test_ancestor_attention_shapes(hip_ctx, rows, prefix, stride, heads):
    """no prefix at all (the first drafter step), one prefix row, more keys than one wave's share, a long prefix; random values; the ancestor lists
    name slots OUTSIDE this call's rows and, for the later rows, the slot of an EARLIER row (a child drafted with its parent)"""
    rng = np.random.default_rng(rows * 31 + prefix)
    nodes = rows + 24
    x = ancestor_attention_inputs(rows=rows, prefix_length=prefix, ancestor_stride=stride, nodes=nodes, num_heads=heads, max_depth=stride + 1)
    from helpers import bf16
    for name in ("prefix_kv", "node_kv", "current_qkv"):
        x[name] = bf16(rng.normal(size=x[name].size))
    first = nodes - rows
    counts = rng.integers(0, stride + 1, rows).astype(np.uint32)
    anc = np.zeros((rows, stride), np.uint32)
    for row in range(rows):
        anc[row, :counts[row]] = rng.integers(0, first, counts[row])
        if row and counts[row]:
            anc[row, counts[row] - 1] = first + rng.integers(0, row)  # an earlier row's slot
    x["ancestor_counts"], x["ancestor_indices"] = counts, anc.reshape(-1)
    x["node_metadata"][:rows] = counts  # depth = number of ancestors, as the speculator fills it
    check_ancestor_attention(hip_ctx, x)


def test_ancestor_attention_refuses_other_head_dims(hip_ctx):
    with pytest.raises(UzuError):
        B.AncestorAttentionKernel.new(hip_ctx, 64, 16)


# ------------------------------------------------------------------------------------------ Weaver frontier
def hip_frontier_select(ctx, x):
    kernel = B.WeaverFrontierSelectKernel.new(ctx)
    names = ("frontier", "tree", "slot_ancestors", "token", "metadata", "ancestors", "valid", "pool_ids", "pool_logits", "cand_ids", "cand_logits")
    bufs = {k: ctx.buffer_from(x[k]) for k in names}
    run(ctx, lambda cb: kernel.encode(*[bufs[k] for k in names], *x["scalars"], cb))
    return {k: bufs[k].download(x[k].dtype, x[k].size) for k in names}


def check_frontier_select(ctx, x):
    want = run_frontier_select(x)
    got = hip_frontier_select(ctx, x)
    for name, value in got.items():
        assert np.array_equal(value.view(np.uint32), want[name].reshape(-1).view(np.uint32)), name


def test_weaver_frontier_select_on_the_reference_tests_frontier(hip_ctx):
    check_frontier_select(hip_ctx, frontier_select_inputs())


def random_frontier(rng, fc, ts, nc, start, stride, max_depth, lookahead, cdc, cpd, active_fraction=0.5, ties=True):
    frontier = np.zeros((7, fc), np.uint32)
    frontier[0] = rng.integers(0, 50, fc)
    frontier[1] = rng.integers(0, max(start, 1), fc)
    frontier[2] = rng.integers(1, max_depth + 1, fc)
    frontier[3] = rng.integers(0, 2 ** 32, fc, dtype=np.uint64).astype(np.uint32)
    frontier[4] = rng.integers(0, 2 ** 32, fc, dtype=np.uint64).astype(np.uint32)
    frontier[5] = rng.integers(0, 8, fc) * 1000 if ties else rng.integers(0, 2 ** 32, fc, dtype=np.uint64).astype(np.uint32)
    frontier[6] = rng.random(fc) < active_fraction
    return dict(frontier=frontier.reshape(-1).copy(), tree=np.full(6 * ts, 55, np.uint32), slot_ancestors=rng.integers(0, max(start, 1), ts * stride).astype(np.uint32),
                token=np.full(nc, 66, np.uint32), metadata=np.full(3 * nc, 77, np.uint32), ancestors=np.full(nc * stride, 88, np.uint32), valid=np.full(nc, 99, np.uint32),
                pool_ids=rng.integers(0, 1000, cdc * cpd).astype(np.uint32), pool_logits=rng.normal(size=cdc * cpd).astype(np.float32), cand_ids=np.full(nc * cpd, 5, np.uint32),
                cand_logits=np.full(nc * cpd, 0.5, np.float32), scalars=(fc, ts, nc, start, stride, max_depth, lookahead, cdc, cpd))


@pytest.mark.parametrize("fc,ts,nc,start,stride,max_depth,lookahead,cdc,cpd,active,ties", [
    (64, 40, 9, 20, 5, 6, 4, 5, 7, 0.5, True),        # many ties: parent, token, then slot order decide
    (64, 40, 9, 20, 5, 6, 4, 5, 7, 0.05, True),       # fewer active slots than nodes: the tail of the batch is written invalid
    (1024, 256, 32, 96, 8, 8, 6, 8, 64, 0.7, False),  # the speculators' working sizes (FRONTIER_MAX_WIDTH nodes a call)
    (2048, 512, 32, 0, 8, 8, 8, 8, 32, 0.9, False),   # FRONTIER_MAX_SLOTS; batch at slot 0; nothing expandable is excluded (lookahead = max depth)
    (4096, 512, 32, 0, 8, 8, 8, 8, 32, 0.9, False),   # beyond the reference kernel's guards: a no-op there and here
    (1024, 256, 33, 0, 8, 8, 8, 8, 32, 0.9, False),
    (16, 8, 4, 2, 3, 4, 3, 0, 3, 1.0, True),
])
def test_weaver_frontier_select_random_bit_exact(hip_ctx, fc, ts, nc, start, stride, max_depth, lookahead, cdc, cpd, active, ties):
    rng = np.random.default_rng(fc + nc + int(active * 100))
    check_frontier_select(hip_ctx, random_frontier(rng, fc, ts, nc, start, stride, max_depth, lookahead, cdc, cpd, active, ties))


def test_weaver_frontier_select_key_zero_and_sentinel_entries(hip_ctx):
    """keys of zero still win against nothing; an entry that equals the scan's initial triple (key 0, parent and token 0xffffffff) never does"""
    rng = np.random.default_rng(5)
    x = random_frontier(rng, 16, 12, 6, 2, 3, 4, 4, 2, 3, 1.0, True)
    frontier = x["frontier"].reshape(7, 16)
    frontier[5] = 0
    frontier[2] = 4  # depth == lookahead: not expandable, so the whole key is 0
    frontier[1, :8], frontier[0, :8] = 0xFFFFFFFF, 0xFFFFFFFF
    frontier[6, 10:] = 0
    check_frontier_select(hip_ctx, x)


def hip_insert_children(ctx, x):
    kernel = B.WeaverFrontierInsertChildrenKernel.new(ctx)
    bufs = [ctx.buffer_from(x[k]) for k in ("tree", "metadata", "valid", "ids", "scores", "frontier")]
    run(ctx, lambda cb: kernel.encode(*bufs, *x["scalars"], cb))
    return bufs[5].download(np.uint32, x["frontier"].size)


def test_weaver_frontier_insert_children_on_the_reference_tests_inputs(hip_ctx):
    x = insert_children_inputs()
    assert np.array_equal(hip_insert_children(hip_ctx, x), run_insert_children(x))


@pytest.mark.parametrize("fc,ts,nc,ew", [(1024, 256, 64, 8), (64, 16, 5, 12), (8, 4, 3, 3)])
def test_weaver_frontier_insert_children_random_bit_exact(hip_ctx, fc, ts, nc, ew):
    """including a frontier too small for nodes x children (the rows past its end are dropped, as in the reference) and special values in the scores"""
    rng = np.random.default_rng(fc)
    tree = np.zeros((6, ts), np.uint32)
    tree[3] = rng.normal(size=ts).astype(np.float32).view(np.uint32)
    tree[2] = rng.integers(0, 7, ts)
    metadata = np.zeros((3, nc), np.uint32)
    metadata[2] = rng.permutation(ts)[:nc]  # a tree slot is the parent of one row only (two rows writing one frontier slot would race)
    scores = rng.normal(size=nc * ew).astype(np.float32)
    scores[::7] = -np.inf
    scores[3::11] = 0.0
    x = dict(tree=tree.reshape(-1), metadata=metadata.reshape(-1), valid=(rng.random(nc) < 0.7).astype(np.uint32), ids=rng.integers(0, 2 ** 17, nc * ew).astype(np.uint32),
             scores=scores, frontier=np.full(7 * fc, 42, np.uint32), scalars=(fc, ts, nc, ew))
    assert np.array_equal(hip_insert_children(hip_ctx, x), run_insert_children(x))


# ------------------------------------------------------------------------------------------ Weaver top children
def hip_top_children(ctx, x):
    kernel = B.WeaverTopChildrenKernel.new(ctx)
    bufs = [ctx.buffer_from(x[k]) for k in ("residual", "cand", "ids", "seeds", "metadata")]
    n = x["rows"] * x["children"]
    tokens, logprobs = ctx.buffer_from(np.zeros(n, np.uint32)), ctx.buffer_from(np.zeros(n, np.float32))
    run(ctx, lambda cb: kernel.encode(*bufs, tokens, logprobs, x["rows"], x["candidates"], x["children"], x["vocab"], cb))
    return tokens.download(np.uint32, n), logprobs.download(np.float32, n)


def check_top_children(ctx, x):
    want_tokens, want_logprobs = run_top_children(x)
    got_tokens, got_logprobs = hip_top_children(ctx, x)
    assert np.array_equal(got_tokens, want_tokens)  # same f32 sum, same Philox noise bits, same total order
    np.testing.assert_allclose(got_logprobs, want_logprobs, atol=1e-5)


def test_weaver_top_children_on_the_reference_tests_inputs(hip_ctx):
    check_top_children(hip_ctx, top_children_inputs())


@pytest.mark.parametrize("rows,candidates,children", [(1, 8, 8), (5, 64, 8), (64, 512, 8), (3, 100, 16)])
def test_weaver_top_children_shapes(hip_ctx, rows, candidates, children):
    """as many children as candidates, the speculators' working size, a candidate count that is no power of two; many equal perturbed logits cannot
    happen with Gumbel noise, equal token ids across rows can"""
    rng = np.random.default_rng(rows + candidates)
    x = top_children_inputs(rows=min(rows, 3), candidates=candidates)
    from helpers import bf16
    x.update(rows=rows, children=children, residual=bf16(rng.normal(size=rows * candidates) * 2), cand=(rng.normal(size=rows * candidates) * 2).astype(np.float32),
             ids=np.concatenate([rng.permutation(131072)[:candidates] for _ in range(rows)]).astype(np.uint32))
    metadata = np.zeros(3 * rows, np.uint32)
    metadata[:rows] = rng.integers(0, 3, rows)
    x["metadata"] = metadata
    check_top_children(hip_ctx, x)
