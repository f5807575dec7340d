"""CPU: the oracle's restatement of the tree speculators' kernels (oracle/uzu_oracle_speculator.c <- cpu/kernel/attention/ancestor_attention.rs,
cpu/kernel/weaver/*.rs) against independent NumPy statements of the same definitions, on the procedural inputs of the reference's own tests
(tests/unit/backends/common/kernel/attention/ancestor_attention_test.rs, .../weaver/weaver_frontier_test.rs, weaver_top_children_test.rs --
which compare a GPU backend with the CPU kernels and hold no literal expectations)."""
import numpy as np

from helpers import bf16, f32
from oracle import oracle as O

FR = dict(token=0, parent=1, depth=2, path=3, edge=4, key=5, active=6)
TR = dict(token=0, parent=1, depth=2, path=3, edge=4, valid=5)
MD = dict(depth=0, ancestors=1, slot=2)
NO_WINNER = 0xFFFFFFFF


# ------------------------------------------------------------------------------------------ inputs of the reference's tests
def ancestor_attention_inputs(rows=4, prefix_length=5, ancestor_stride=3, nodes=16, num_heads=16, head_dim=128, max_depth=8):
    model_dim = num_heads * head_dim
    qkv_width, kv_width = 3 * model_dim, 2 * model_dim

    def values(length, offset):
        idx = np.arange(length, dtype=np.int64) + offset
        return bf16(((idx * 17 % 251).astype(np.float32) - np.float32(125.0)) / np.float32(128.0))
    first_output_node = nodes - rows
    node_indices = np.arange(first_output_node, nodes, dtype=np.uint32)
    ancestor_counts = np.array([row % (ancestor_stride + 1) for row in range(rows)], np.uint32)
    ancestor_indices = np.zeros(rows * ancestor_stride, np.uint32)
    for row in range(rows):
        for offset in range(int(ancestor_counts[row])):
            ancestor_indices[row * ancestor_stride + offset] = (row * ancestor_stride + offset) % first_output_node
    half = head_dim // 2
    cosines = np.zeros(((max_depth + 1), head_dim), np.float32)
    sines = np.zeros_like(cosines)
    for position in range(max_depth + 1):
        pair = np.arange(half, dtype=np.float32)
        angle = (np.float32(position) / np.power(np.float32(10000.0), np.float32(2.0) * pair / np.float32(head_dim))).astype(np.float32)
        cosines[position, :half] = cosines[position, half:] = np.cos(angle)
        sines[position, :half] = sines[position, half:] = np.sin(angle)
    metadata = np.zeros(rows * 3, np.uint32)
    metadata[MD["depth"] * rows:(MD["depth"] + 1) * rows] = [(row * 3) % max_depth for row in range(rows)]
    return dict(prefix_kv=values(prefix_length * kv_width, 0), node_kv=values(nodes * kv_width, 11), current_qkv=values(rows * qkv_width, 29), cosines=cosines,
                sines=sines, node_metadata=metadata, ancestor_indices=ancestor_indices, ancestor_counts=ancestor_counts, node_indices=node_indices, rows=rows,
                prefix_length=prefix_length, ancestor_stride=ancestor_stride, node_capacity=nodes, max_depth=max_depth, scale=np.float32(1.0 / np.sqrt(head_dim)),
                num_heads=num_heads, head_dim=head_dim)


def run_ancestor_attention(x):
    node_kv = x["node_kv"].copy()
    out = np.zeros(x["rows"] * x["num_heads"] * x["head_dim"], np.uint16)
    O.call("orc_ancestor_attention", x["prefix_kv"], node_kv, x["current_qkv"], x["cosines"], x["sines"], x["node_metadata"], x["ancestor_indices"], x["ancestor_counts"],
           x["node_indices"], out, x["rows"], x["prefix_length"], x["ancestor_stride"], x["node_capacity"], x["max_depth"], x["scale"], x["num_heads"], x["head_dim"])
    return out, node_kv


def test_ancestor_attention_against_float64():
    x = ancestor_attention_inputs()
    out, node_kv = run_ancestor_attention(x)
    H, D, rows, P, cap = x["num_heads"], x["head_dim"], x["rows"], x["prefix_length"], x["node_capacity"]
    md = H * D
    half = D // 2
    prefix_k, prefix_v = f32(x["prefix_kv"][:P * md]).reshape(P, H, D), f32(x["prefix_kv"][P * md:]).reshape(P, H, D)
    node_k, node_v = f32(x["node_kv"][:cap * md]).reshape(cap, H, D), f32(x["node_kv"][cap * md:]).reshape(cap, H, D)
    cur = f32(x["current_qkv"]).reshape(rows, 3, H, D)
    got = f32(out).reshape(rows, H, D)
    new_k = f32(node_kv[:cap * md]).reshape(cap, H, D)
    new_v = f32(node_kv[cap * md:]).reshape(cap, H, D)
    for row in range(rows):
        pos = int(x["node_metadata"][row]) + 1
        c, s = x["cosines"][pos], x["sines"][pos]

        def rotate(v):  # half-rotation RoPE, rounded to bf16 as the kernel stores it
            low, high = v[:, :half], v[:, half:]
            return f32(bf16(np.concatenate([low * c[:half] - high * s[:half], high * c[half:] + low * s[half:]], axis=1)))
        q, k_new = rotate(cur[row, 0]), rotate(cur[row, 1])
        anc = [int(a) for a in x["ancestor_indices"][row * x["ancestor_stride"]:row * x["ancestor_stride"] + int(x["ancestor_counts"][row])]]
        keys = np.concatenate([prefix_k, node_k[anc].reshape(len(anc), H, D), k_new[None]]).astype(np.float64)
        vals = np.concatenate([prefix_v, node_v[anc].reshape(len(anc), H, D), cur[row, 2][None]]).astype(np.float64)
        scores = np.einsum("hd,lhd->hl", q.astype(np.float64), keys) * float(x["scale"])
        w = np.exp(scores - scores.max(axis=1, keepdims=True))
        want = np.einsum("hl,lhd->hd", w / w.sum(axis=1, keepdims=True), vals)
        assert np.abs(got[row] - want).max() <= 2.0 ** -8 * max(1.0, np.abs(want).max())
        node = int(x["node_indices"][row])
        assert np.array_equal(new_k[node], k_new) and np.array_equal(new_v[node], cur[row, 2])  # the node's slot: rotated key, its value
    untouched = [n for n in range(cap) if n not in set(int(v) for v in x["node_indices"])]
    assert np.array_equal(new_k[untouched], node_k[untouched]) and np.array_equal(new_v[untouched], node_v[untouched])


def frontier_select_inputs():
    """weaver_frontier_test.rs:16-62"""
    frontier = np.zeros(7 * 8, np.uint32)
    rows = [(9, 1, 1, 0x3f000000, 100, 1), (8, 0, 2, 0x3f000001, 100, 1), (7, 0, 2, 0x3f000002, 100, 1), (7, 0, 2, 0x3f000003, 100, 1), (2, 1, 3, 0x3f000004, 300, 1),
            (0, 0, 0, 0x3f000005, 200, 0), (4, 1, 3, 0x3f000006, 80, 1), (5, 1, 1, 0x3f000007, 70, 1)]
    for slot, (token, parent, depth, cum, key, active) in enumerate(rows):
        for lane, value in enumerate([token, parent, depth, cum, 0xbf800000, key, active]):
            frontier[lane * 8 + slot] = value
    return dict(frontier=frontier, tree=np.full(6 * 7, 55, np.uint32), slot_ancestors=np.arange(7 * 3, dtype=np.uint32), token=np.full(4, 66, np.uint32),
                metadata=np.full(3 * 4, 77, np.uint32), ancestors=np.full(4 * 3, 88, np.uint32), valid=np.full(4, 99, np.uint32), pool_ids=np.arange(12, dtype=np.uint32),
                pool_logits=np.arange(12, dtype=np.float32), cand_ids=np.zeros(4 * 3, np.uint32), cand_logits=np.zeros(4 * 3, np.float32),
                scalars=(8, 7, 4, 2, 3, 4, 3, 4, 3))


def run_frontier_select(x):
    y = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in x.items()}
    O.call("orc_weaver_frontier_select", y["frontier"], y["tree"], y["slot_ancestors"], y["token"], y["metadata"], y["ancestors"], y["valid"], y["pool_ids"], y["pool_logits"],
           y["cand_ids"], y["cand_logits"], *y["scalars"])
    return y


def test_weaver_frontier_select_on_the_reference_tests_frontier():
    """The selection rule read off weaver_frontier_select.rs:62-80 and applied by hand to the reference test's frontier: lookahead 3, so the depth-3
    slots are not expandable and rank below every expandable one whatever their score; among the expandable slots key 100 (>> 1 = 50) four
    times -> lower parent slot, then lower token id, then the first slot; every pick leaves the frontier."""
    y = run_frontier_select(frontier_select_inputs())
    fc, ts, nc = 8, 7, 4
    tree = y["tree"].reshape(6, ts)
    # expandable slots by (key >> 1 desc, parent asc, token asc): slot 2 / 3 (token 7, parent 0; slot 2 first), slot 1 (token 8, parent 0), slot 0 (token 9, parent 1)
    assert tree[TR["token"], 2:6].tolist() == [7, 7, 8, 9]
    assert tree[TR["parent"], 2:6].tolist() == [0, 0, 0, 1] and tree[TR["depth"], 2:6].tolist() == [2, 2, 2, 1] and tree[TR["valid"], 2:6].tolist() == [1, 1, 1, 1]
    assert tree[TR["path"], 2:6].tolist() == [0x3f000002, 0x3f000003, 0x3f000001, 0x3f000000]
    assert y["frontier"].reshape(7, fc)[FR["active"]].tolist() == [0, 0, 0, 0, 1, 0, 1, 1]
    assert tree[:, :2].tolist() == [[55, 55]] * 6 and tree[:, 6].tolist() == [55] * 6  # slots outside the batch are not touched
    md = y["metadata"].reshape(3, nc)
    assert md[MD["depth"]].tolist() == [2, 2, 2, 1] and md[MD["ancestors"]].tolist() == [2, 2, 2, 1] and md[MD["slot"]].tolist() == [2, 3, 4, 5]
    assert y["valid"].tolist() == [1, 1, 1, 1] and y["token"].tolist() == [7, 7, 8, 9]
    # ancestors of a depth-2 node with parent slot 0: [ancestors of slot 0][0], slot 0; of the depth-1 node with parent 1: [1]
    anc = y["ancestors"].reshape(nc, 3)
    assert anc.tolist() == [[0, 0, 0], [0, 0, 0], [0, 0, 0], [1, 0, 0]]
    assert np.array_equal(y["slot_ancestors"].reshape(ts, 3)[2:6], anc)
    assert y["cand_ids"].reshape(nc, 3).tolist() == [[6, 7, 8], [6, 7, 8], [6, 7, 8], [3, 4, 5]]  # the candidate pool row of the node's depth


def test_weaver_frontier_select_random_against_a_numpy_restatement():
    rng = np.random.default_rng(3)
    fc, ts, nc, start, stride, max_depth, lookahead, cdc, cpd = 64, 40, 9, 20, 5, 6, 4, 5, 7
    frontier = np.zeros((7, fc), np.uint32)
    frontier[FR["token"]] = rng.integers(0, 50, fc)
    frontier[FR["parent"]] = rng.integers(0, start, fc)
    frontier[FR["depth"]] = rng.integers(1, 6, fc)
    frontier[FR["path"]] = rng.integers(0, 2 ** 32, fc, dtype=np.uint64).astype(np.uint32)
    frontier[FR["edge"]] = rng.integers(0, 2 ** 32, fc, dtype=np.uint64).astype(np.uint32)
    frontier[FR["key"]] = rng.integers(0, 8, fc) * 1000  # many ties
    frontier[FR["active"]] = rng.integers(0, 2, fc)
    x = dict(frontier=frontier.reshape(-1).copy(), tree=np.full(6 * ts, 55, np.uint32), slot_ancestors=rng.integers(0, start, ts * stride).astype(np.uint32),
             token=np.zeros(nc, np.uint32), metadata=np.zeros(3 * nc, np.uint32), ancestors=np.zeros(nc * stride, np.uint32), valid=np.zeros(nc, np.uint32),
             pool_ids=rng.integers(0, 1000, cdc * cpd).astype(np.uint32), pool_logits=rng.normal(size=cdc * cpd).astype(np.float32), cand_ids=np.zeros(nc * cpd, np.uint32),
             cand_logits=np.zeros(nc * cpd, np.float32), scalars=(fc, ts, nc, start, stride, max_depth, lookahead, cdc, cpd))
    y = run_frontier_select(x)
    # restatement: sort the active slots once by the rule; pick k = the k-th of that order (a pick only removes itself)
    active = [s for s in range(fc) if frontier[FR["active"], s]]
    order = sorted(active, key=lambda s: (-(((1 if frontier[FR["depth"], s] < lookahead else 0) << 31) | (int(frontier[FR["key"], s]) >> 1)), int(frontier[FR["parent"], s]),
                                          int(frontier[FR["token"], s]), s))
    tree = y["tree"].reshape(6, ts)
    for node in range(nc):
        s = order[node]
        slot = start + node
        assert tree[TR["token"], slot] == frontier[FR["token"], s] and tree[TR["parent"], slot] == frontier[FR["parent"], s] and tree[TR["valid"], slot] == 1
        depth = int(frontier[FR["depth"], s])
        assert y["metadata"].reshape(3, nc)[MD["depth"], node] == (depth if depth < lookahead else 0) and y["valid"][node] == int(depth < lookahead)
        parent = int(frontier[FR["parent"], s])
        want_anc = [int(x["slot_ancestors"][parent * stride + i]) if i + 1 < depth else (parent if i + 1 == depth else 0) for i in range(stride)]
        assert y["ancestors"].reshape(nc, stride)[node].tolist() == want_anc
    assert sorted(s for s in range(fc) if y["frontier"].reshape(7, fc)[FR["active"], s]) == sorted(set(active) - set(order[:nc]))


def insert_children_inputs():
    """weaver_frontier_test.rs:64-85"""
    tree = np.zeros(6 * 4, np.uint32)
    tree[TR["path"] * 4:(TR["path"] + 1) * 4] = np.array([0.5, -1.0, 2.0, 4.0], np.float32).view(np.uint32)
    tree[TR["depth"] * 4:(TR["depth"] + 1) * 4] = [0, 2, 4, 6]
    metadata = np.zeros(3 * 3, np.uint32)
    metadata[MD["slot"] * 3:(MD["slot"] + 1) * 3] = [1, 3, 0]
    return dict(tree=tree, metadata=metadata, valid=np.array([1, 0, 1], np.uint32), ids=np.arange(10, 19, dtype=np.uint32),
                scores=np.array([-0.1, -0.2, -0.3, 8.0, 8.0, 8.0, 0.1, 0.2, 0.3], np.float32), frontier=np.full(7 * 16, 42, np.uint32), scalars=(16, 4, 3, 3))


def run_insert_children(x):
    frontier = x["frontier"].copy()
    O.call("orc_weaver_frontier_insert_children", x["tree"], x["metadata"], x["valid"], x["ids"], x["scores"], frontier, *x["scalars"])
    return frontier


def test_weaver_frontier_insert_children_on_the_reference_tests_inputs():
    x = insert_children_inputs()
    fr = run_insert_children(x).reshape(7, 16)
    # row 0 (valid, parent slot 1: path -1.0, depth 2) -> frontier slots 3, 4, 5; row 1 invalid; row 2 (parent 0: path 0.5, depth 0) -> slots 0, 1, 2
    for slot, (tok, parent, depth, path, edge) in {3: (10, 1, 3, -1.0 + np.float32(-0.1), -0.1), 4: (11, 1, 3, -1.0 + np.float32(-0.2), -0.2), 5: (12, 1, 3, -1.0 + np.float32(-0.3), -0.3),
                                                   0: (16, 0, 1, np.float32(0.5) + np.float32(0.1), 0.1), 1: (17, 0, 1, np.float32(0.5) + np.float32(0.2), 0.2),
                                                   2: (18, 0, 1, np.float32(0.5) + np.float32(0.3), 0.3)}.items():
        path_bits = int(np.float32(path).view(np.uint32))
        assert fr[:, slot].tolist() == [tok, parent, depth, path_bits, int(np.float32(edge).view(np.uint32)),
                                        (path_bits ^ 0x80000000) if path_bits < 0x80000000 else (~path_bits & 0xFFFFFFFF), 1]
    assert (fr[:, 6:] == 42).all()


def top_children_inputs(rows=3, candidates=512):
    """weaver_top_children_test.rs:13-17, 68-77"""
    idx = np.arange(rows * candidates, dtype=np.float32)
    residual = bf16(np.round(np.cos(idx * np.float32(0.017)) * np.float32(4.0)) * np.float32(0.125))
    cand = (np.round(np.sin(idx * np.float32(0.011)) * np.float32(3.0)) * np.float32(0.125)).astype(np.float32)
    ids = np.concatenate([70000 + (row * candidates + np.arange(candidates)[::-1]) for row in range(rows)]).astype(np.uint32)
    metadata = np.zeros(rows * 3, np.uint32)
    metadata[:rows] = [0, 1, 2][:rows]
    return dict(residual=residual, cand=cand, ids=ids, seeds=np.array([0x9E3779B97F4A7C15, 0xD1B54A32D192ED03, 0x2545F4914F6CDD1D], np.uint64), metadata=metadata, rows=rows,
                candidates=candidates, children=8, vocab=131072)


def run_top_children(x):
    tokens, logprobs = np.zeros(x["rows"] * x["children"], np.uint32), np.zeros(x["rows"] * x["children"], np.float32)
    O.call("orc_weaver_top_children", x["residual"], x["cand"], x["ids"], x["seeds"], x["metadata"], tokens, logprobs, x["rows"], x["candidates"], x["children"], x["vocab"])
    return tokens, logprobs


def test_weaver_top_children_against_float64():
    x = top_children_inputs()
    tokens, logprobs = run_top_children(x)
    C_, W = x["candidates"], x["children"]
    lib = O.lib()
    lib.orc_gumbel_float.restype = __import__("ctypes").c_float
    import ctypes as C
    lib.orc_gumbel_float.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
    for row in range(x["rows"]):
        logits = x["cand"][row * C_:(row + 1) * C_] + f32(x["residual"][row * C_:(row + 1) * C_])  # f32 + f32: exact statement of the kernel's sum
        ids = x["ids"][row * C_:(row + 1) * C_]
        noise = np.empty(C_, np.float32)
        for i, tok in enumerate(ids):
            off, word = C.c_uint32(), C.c_uint32()
            lib.orc_revidx(C.c_uint32(int(tok)), C.c_uint32(x["vocab"]), C.byref(off), C.byref(word))
            noise[i] = lib.orc_gumbel_float(C.c_uint64(int(x["seeds"][int(x["metadata"][row])])), off, word)
        perturbed = (logits + noise).astype(np.float32)
        order = sorted(range(C_), key=lambda i: (-float(perturbed[i]), int(ids[i])))[:W]
        assert tokens[row * W:(row + 1) * W].tolist() == [int(ids[i]) for i in order]
        l64 = logits.astype(np.float64)
        log_softmax = l64 - (np.log(np.exp(l64 - l64.max()).sum()) + l64.max())
        np.testing.assert_allclose(logprobs[row * W:(row + 1) * W], log_softmax[order], atol=1e-5)  # 512 f32 exps summed in order (the reference test allows 1e-5 too)
    assert len(set(tokens.tolist())) == x["rows"] * W
