"""The known answers the reference's OWN tests hold for this path, replayed against the oracle (and the host-side trie port).

The reference's kernel tests are differential (CPU kernel vs the other backends) and store no output files, but a number of them compute their
expectation INSIDE the test, independently of any kernel, from procedural inputs -- and run the CPU backend against it (`for_each_backend!`
includes Cpu).  Those closed forms are the golden vectors the reference's CPU path is pinned by; this file evaluates the same formulae on the
same inputs (NumPy, bf16 through half::bf16::from_f32's rounding) and holds `oracle/` to them, bit for bit where the reference asserts
equality, at the reference's tolerance otherwise.  Together with tests/test_oracle_kernels.py::test_gated_act_mul_reference_known_answer
([10, 40, 150, 240], gated_act_mul_test.rs:139-160), ::test_attention_*_against_float64 (attention_test.rs:26-124, the reference's independent
f32 attention), tests/test_oracle_sampling.py (gumbel_test.rs extremes, Random123 Philox vectors) and
tests/test_oracle_tree_verify.py::test_tree_gram_on_the_reference_tests_procedural_inputs (tree_gram_test.rs:36-55) this is every literal or
in-test expectation under BU/tests/unit that touches SURVEY.md section 8's rows; what remains unpinned is everything the reference only ever
compares backend-against-backend (matmul, normalization, attention cores at bf16, DeltaNet update): see DESIGN.md section 5.

    BU = /root/reference/crates/backend-uzu
"""
import ctypes as C

import numpy as np
import pytest

from helpers import bf16, f32
from oracle import oracle as O
from uzu_amd.trie import DuplicateTokenId, PRng, TrieNode


def typed(values, dt):
    """T::from(f32): f32 stays, bf16 rounds to nearest even; returned as the storage array the oracle reads."""
    v = np.asarray(values, dtype=np.float32)
    return v.copy() if dt == O.F32 else bf16(v)


def as_f32(a, dt):
    return a.astype(np.float32) if dt == O.F32 else f32(a)


def t_add(a, b, dt):
    """`a + b` on two T values: f32 add, then T::from (half's Add impl for bf16)."""
    return typed(as_f32(a, dt) + as_f32(b, dt), dt)


def ramp(n, fn, scale=30.0):
    """T::from((i as f32).sin() * 30f32): sine / cosine evaluated in f32."""
    return (fn(np.arange(n, dtype=np.float32)).astype(np.float32) * np.float32(scale)).astype(np.float32)


DTYPES = [pytest.param(O.F32, id="f32"), pytest.param(O.BF16, id="bf16")]


# ------------------------------------------------------------------------------------------ tensor kernels (a10)
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("in_place", [False, True])
def test_tensor_add_bias_known_answer(dt, in_place):
    """BU/tests/unit/backends/common/kernel/tensor_add_bias_test.rs:22-51: length 1025, 129 columns, input sin(i) * 30, bias cos(i) * 30,
    expected[i] = input[i] + bias[i % 129] in T."""
    length, num_cols = 1025, 129
    x, b = typed(ramp(length, np.sin), dt), typed(ramp(num_cols, np.cos), dt)
    want = t_add(x, b[np.arange(length) % num_cols], dt)
    out = x.copy() if in_place else np.zeros_like(x)
    O.call("orc_tensor_add_bias", None if in_place else x, b, out, dt, dt, num_cols, length)
    assert np.array_equal(out, want)


@pytest.mark.parametrize("dt", DTYPES)
def test_tensor_add_swap_and_copy_known_answers(dt):
    """tensor_add_swap_test.rs:19-38 (skip sin(i) * 30, main cos(i) * 30, both buffers end as skip + main) and tensor_copy_test.rs:18-34."""
    length = 1025
    skip, main = typed(ramp(length, np.sin), dt), typed(ramp(length, np.cos), dt)
    want = t_add(skip, main, dt)
    O.call("orc_tensor_add_swap", skip, main, dt, length)
    assert np.array_equal(skip, want) and np.array_equal(main, want)
    src = typed(ramp(length, np.sin), dt)
    dst = np.zeros_like(src)
    O.call("orc_tensor_copy", src, dst, dt, length)
    assert np.array_equal(dst, src)


# ------------------------------------------------------------------------------------------ embedding lookup (a9)
@pytest.mark.parametrize("dt", DTYPES)
def test_full_precision_embedding_lookup_known_answer(dt):
    """embedding/full_precision_embedding_test.rs:22-48: vocab 11, model_dim 513, ids [3, 7, 1, 10, 0, 5, 8], weights sin(i) * 30, input_scale 2:
    expected = weights[row] * T::from(2.0) in T -- the reference asserts EQUALITY."""
    vocab, dim, scale = 11, 513, 2.0
    ids = np.array([3, 7, 1, 10, 0, 5, 8], np.uint32)
    w = typed(ramp(vocab * dim, np.sin), dt).reshape(vocab, dim)
    want = typed(as_f32(w[ids], dt) * np.float32(scale), dt)
    out = np.zeros((ids.size, dim), w.dtype)
    O.call("orc_full_precision_embedding_lookup", ids, w, out, dt, ids.size, vocab, dim, scale)
    assert np.array_equal(out, want)


# ------------------------------------------------------------------------------------------ KV cache compaction (a7)
def kv_update(keys, values, dt, copies, element_dim):
    flat = np.array(copies, np.uint32).reshape(-1, 2) if copies else np.zeros((1, 2), np.uint32)  # orc_copy {source, destination} pairs
    O.call("orc_kv_cache_update", keys, values, dt, flat, len(copies), element_dim)


@pytest.mark.parametrize("dt", DTYPES)
def test_kv_cache_update_without_copies_leaves_the_caches_alone(dt):
    """kv_cache_update_test.rs:120-138 (`no_copy`): keys 1 + 0.1 i, values 100 + 0.1 i, 4 rows of 2 x 3 -- expected = the inputs."""
    total = 4 * 6
    keys = typed(1.0 + np.arange(total, dtype=np.float32) * np.float32(0.1), dt)
    values = typed(100.0 + np.arange(total, dtype=np.float32) * np.float32(0.1), dt)
    k0, v0 = keys.copy(), values.copy()
    kv_update(keys, values, dt, [], 6)
    assert np.array_equal(keys, k0) and np.array_equal(values, v0)


def test_kv_cache_update_dependent_copy_pattern_known_answer():
    """kv_cache_update_test.rs:308-385: 15 tokens x 3 heads x 7 channels, element (token, head, channel) = token * 1e6 + head * 100 + channel * 10
    (+ 1000 for values), ten copies whose later sources are earlier destinations; expected = the copies applied IN ORDER per element
    (apply_copies_3d, :277-291) -- asserted with equality."""
    seq, heads, hd = 15, 3, 7
    t, h, c = np.meshgrid(np.arange(seq), np.arange(heads), np.arange(hd), indexing="ij")
    keys = (t * 1_000_000 + h * 100 + c * 10).astype(np.float32)
    values = keys + np.float32(1000)
    copies = [(0, 14), (3, 11), (6, 8), (9, 5), (12, 2), (2, 12), (5, 9), (8, 6), (11, 3), (14, 0)]
    want_k, want_v = keys.copy(), values.copy()
    for s, d in copies:
        want_k[d], want_v[d] = want_k[s].copy(), want_v[s].copy()
    k, v = np.ascontiguousarray(keys.reshape(seq, heads * hd)), np.ascontiguousarray(values.reshape(seq, heads * hd))
    kv_update(k, v, O.F32, copies, heads * hd)
    assert np.array_equal(k.reshape(want_k.shape), want_k) and np.array_equal(v.reshape(want_v.shape), want_v)
    assert np.array_equal(want_k[0], keys[0]) and np.array_equal(want_k[14], keys[0]), "the pattern's point: row 0 -> 14 -> back to 0 leaves row 0's data in both"


# ------------------------------------------------------------------------------------------ matmul gather D-op (a1)
def matmul(a, b, m, n, k, dt, method=3, q=None, gather=None, soft_cap=None):
    d = np.zeros((m, n), a.dtype)
    g = O.MatmulArgs()
    g.a, g.a_dtype = a.ctypes.data, dt
    if q is None:
        g.b, g.w_dtype, g.method, g.bits = b.ctypes.data, dt, 3, 16
    else:
        g.b, g.scales = q["weights"].ctypes.data, q["scales"].ctypes.data
        g.biases = q["biases"].ctypes.data if q["biases"] is not None else None
        g.zero_points = q["zero_points"].ctypes.data if q["zero_points"] is not None else None
        g.w_dtype, g.method, g.bits, g.group_size = dt, q["method"], q["bits"], q["group_size"]
    g.b_transpose, g.d, g.d_dtype, g.ab_scale = 1, d.ctypes.data, dt, 1.0
    g.has_soft_cap, g.soft_cap = int(soft_cap is not None), soft_cap or 0.0
    g.gather_indices = gather.ctypes.data if gather is not None else None
    g.m, g.n, g.k = m, n, k
    O.lib().orc_matmul(C.byref(g))
    return d


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("soft_cap", [None, 15.0])
def test_gemv_gather_equals_the_dense_result_at_the_ids_full_precision(dt, soft_cap):
    """matmul/gemv_test.rs:175-241 (`fp_gather_case`, run on Cpu first): m 4, k 128, vocab 256, 8 ids per row; a[i] = (i % 13) * 0.1 - 0.6,
    weights[i] = (i % 17) * 0.1 - 0.8, ids[i] = (i * 37 + 11) % vocab; the gathered product [m, 8] must be the dense [m, vocab] product at those
    columns (the same accumulation: equal, the reference allows 0.1 / 0.01)."""
    m, k, vocab, per_row = 4, 128, 256, 8
    a = typed((np.arange(m * k) % 13).astype(np.float32) * np.float32(0.1) - np.float32(0.6), dt).reshape(m, k)
    w = typed((np.arange(vocab * k) % 17).astype(np.float32) * np.float32(0.1) - np.float32(0.8), dt).reshape(vocab, k)
    ids = ((np.arange(m * per_row) * 37 + 11) % vocab).astype(np.uint32)
    dense = matmul(a, w, m, vocab, k, dt, soft_cap=soft_cap)
    gathered = matmul(a, w, m, per_row, k, dt, gather=ids, soft_cap=soft_cap)
    want = np.take_along_axis(dense, ids.reshape(m, per_row).astype(np.int64), axis=1)
    assert np.array_equal(gathered, want)


@pytest.mark.parametrize("bits,method", [(4, 0), (4, 1), (4, 2), (8, 1)])
def test_gemv_gather_equals_the_dense_result_at_the_ids_quantized(bits, method):
    """gemv_test.rs:243-273: m 8, k 128, vocab 64, 8 ids per row, group 32, the four (bits, method) pairs of the reference (its weights come
    from rand's SmallRng, which is not reproduced: own seed)."""
    from helpers import quant_matrix
    m, k, vocab, per_row = 8, 128, 64, 8
    rng = np.random.default_rng(0x5EED)
    q = quant_matrix(rng, vocab, k, bits, 32, method)
    a = bf16(rng.uniform(-1, 1, size=(m, k)))
    ids = ((np.arange(m * per_row) * 37 + 11) % vocab).astype(np.uint32)
    dense = matmul(a, None, m, vocab, k, O.BF16, q=q)
    gathered = matmul(a, None, m, per_row, k, O.BF16, q=q, gather=ids)
    assert np.array_equal(gathered, np.take_along_axis(dense, ids.reshape(m, per_row).astype(np.int64), axis=1))


# ------------------------------------------------------------------------------------------ Gated DeltaNet (a12, f4)
def test_delta_net_prefill_prep_compact_v_known_answer():
    """gdn/delta_net_test.rs:485-548: Qwen3.5 head counts (48 value / 16 key heads of 128), 8 tokens, in_proj[i] = bf16((i % 37) * 0.02 - 0.3),
    a_log[i] = -1.5 + 0.05 i, dt_bias[i] = 0.3 + 0.02 i; the compact V the prep kernel writes must EQUAL the value section of every row
    (`assert_eq!(ref_v, expected_v)` on the Cpu result)."""
    Hv, Hk, Dk, Dv, n = 48, 16, 128, 128, 8
    key_dim, value_dim = Hk * Dk, Hv * Dv
    total = 2 * key_dim + value_dim + value_dim + 2 * Hv
    in_proj = bf16((np.arange(n * total) % 37).astype(np.float32) * np.float32(0.02) - np.float32(0.3)).reshape(n, total)
    a_log = (-1.5 + np.arange(Hv, dtype=np.float32) * np.float32(0.05)).astype(np.float32)
    dt_bias = (0.3 + np.arange(Hv, dtype=np.float32) * np.float32(0.02)).astype(np.float32)
    q, kk = np.zeros((n, key_dim), np.uint16), np.zeros((n, key_dim), np.uint16)
    v = np.zeros((n, value_dim), np.uint16)
    beta, decay = np.zeros((n, Hv), np.float32), np.zeros((n, Hv), np.float32)
    O.call("orc_delta_net_tree_prep", in_proj, a_log, dt_bias, q, kk, v, beta, decay, O.BF16, Hv, Hk, Dk, key_dim, value_dim, n)
    assert np.array_equal(v, in_proj[:, 2 * key_dim:2 * key_dim + value_dim])
    # what the same test then holds the other backends to (1e-4 abs + 1e-3 rel against the Cpu result) is, for the oracle, the float64 form:
    qf = f32(in_proj[:, :key_dim]).astype(np.float64).reshape(n, Hk, Dk)
    want_q = qf / np.sqrt((qf * qf).sum(-1, keepdims=True) + 1e-6) / np.sqrt(Dk)
    assert np.abs(f32(q).reshape(n, Hk, Dk) - want_q).max() <= 1e-4 + 1e-3 * np.abs(want_q).max() + 2.0 ** -9 * np.abs(want_q).max()  # (+ the bf16 rounding of the output)
    assert np.all((beta > 0) & (beta < 1)) and np.all(decay < 0), "beta = sigmoid(.), log decay = -exp(a_log) * softplus(.) < 0"


@pytest.mark.parametrize("tree_size", [49, 64, 128])
def test_build_tree_prefix_known_answer(tree_size):
    """gdn/tree_verify/prefix_test.rs:16-45: a chain of `tree_size` nodes ({node, size - 1, node}), 5 heads, log_decay[i] = -0.001 - 0.0001 i;
    expected[row, head] = the f32 sum over the row's ancestors-or-self, within 1e-6 (asserted on Cpu and every other backend)."""
    heads = 5
    trie = np.array([[node, tree_size - 1, node] for node in range(tree_size)], np.uint32)
    log_decay = (np.float32(-0.001) - np.arange(tree_size * heads, dtype=np.float32) * np.float32(0.0001)).astype(np.float32)
    prefix = np.zeros((tree_size, heads), np.float32)
    O.call("orc_build_tree_prefix", trie, log_decay, prefix, 1, tree_size, heads)
    want = np.zeros((tree_size, heads), np.float32)
    for row in range(tree_size):
        for head in range(heads):
            acc = np.float32(0.0)
            for token in range(row + 1):  # `.sum::<f32>()`: sequential f32 adds
                acc = np.float32(acc + (np.float32(-0.001) - np.float32(token * heads + head) * np.float32(0.0001)))
            want[row, head] = acc
    assert np.abs(prefix - want).max() <= 1e-6


# ------------------------------------------------------------------------------------------ host trie (f4): BU/tests/unit/trie_test.rs, literally
def test_prng_derive_is_the_references_finaliser():
    """sampling/prng.rs: MurmurHash3's 64-bit finaliser of seed + index; fmix64(0) = 0 is what trie_test.rs's `verify_sprout(&root, 0)` relies on,
    the other values are the published fmix64 test values of 1 and 2^64 - 1."""
    assert PRng(0).derive(0) == 0
    assert PRng(0).derive(1) == 0xB456BCFC34C2CB2C
    assert PRng(1).derive(0) == 0xB456BCFC34C2CB2C
    assert PRng((1 << 64) - 1).derive(0) == 0x64B5720B4B825F21
    assert PRng((1 << 64) - 1).derive(1) == 0  # wrapping_add


def test_trie_manual_sprout_stick_bush_tree():
    """trie_test.rs:5-196."""
    root = TrieNode(0, 0, 0.0)
    flat = root.linearize()
    assert len(flat) == 1 and flat.index(root) == 0
    assert flat.index(TrieNode(1, 0, 0.0)) is None and flat.index(TrieNode(0, 1, 0.0)) is None and flat.index(TrieNode(0, 0, 0.0)) is None
    assert list(flat.token_ids()) == [0] and list(flat.nodes()[:, 2]) == [0] and list(flat.token_seeds()) == [0]

    rng = PRng(0)
    stick = TrieNode(9, rng.derive(9), 0.0)
    for i in range(8, 0, -1):
        parent = TrieNode(i, rng.derive(i), 0.0)
        parent.add(stick)
        stick = parent
    root = TrieNode(0, rng.derive(0), 0.0)
    root.add(stick)
    flat = root.linearize()
    ids, heights, seeds = list(flat.token_ids()), list(flat.nodes()[:, 2]), list(flat.token_seeds())
    assert len(flat) == 10 and len(ids) == len(heights) == len(seeds) == 10
    node = root
    assert ids[flat.index(node)] == 0 and heights[flat.index(node)] == 0 and seeds[flat.index(node)] == rng.derive(0)
    for i in range(1, 10):
        node = node.get(i)
        assert node.token == i and node.seed == rng.derive(i)
        pos = flat.index(node)
        assert ids[pos] == i and heights[pos] == i and seeds[pos] == rng.derive(i)
    assert flat.is_flat()

    def grown(with_leaves):
        root = TrieNode(0, rng.derive(0), 0.0)
        root.add(TrieNode(1, rng.derive(1), 0.0))
        with pytest.raises(DuplicateTokenId):
            root.add(TrieNode(1, rng.derive(1), 0.0))
        with pytest.raises(DuplicateTokenId):
            root.add(TrieNode(1, 10, 0.0))
        mid_b, mid_c = TrieNode(2, rng.derive(1), 0.0), TrieNode(3, rng.derive(1), 0.0)
        if with_leaves:
            mid_b.add(TrieNode(10, rng.derive(2), 0.0))
            mid_c.add(TrieNode(20, rng.derive(2), 0.0))
            mid_c.add(TrieNode(21, rng.derive(2), 0.0))
        root.add(mid_b)
        root.add(mid_c)
        return root

    bush = grown(False)
    flat = bush.linearize()
    ids, heights, seeds = list(flat.token_ids()), list(flat.nodes()[:, 2]), list(flat.token_seeds())
    assert len(flat) == 4
    assert ids[flat.index(bush)] == 0 and heights[flat.index(bush)] == 0 and seeds[flat.index(bush)] == rng.derive(0)
    for leaf_token in (1, 2, 3):
        leaf = bush.get(leaf_token)
        pos = flat.index(leaf)
        assert leaf.token == leaf_token and leaf.seed == rng.derive(1) and ids[pos] == leaf_token and heights[pos] == 1 and seeds[pos] == rng.derive(1)

    tree = grown(True)
    flat = tree.linearize()
    ids, heights, seeds = list(flat.token_ids()), list(flat.nodes()[:, 2]), list(flat.token_seeds())
    assert len(flat) == 7
    for mid in (1, 2, 3):
        pos = flat.index(tree.get(mid))
        assert ids[pos] == mid and heights[pos] == 1 and seeds[pos] == rng.derive(1)
    for mid, leaf_token in ((2, 10), (3, 20), (3, 21)):
        pos = flat.index(tree.get(mid).get(leaf_token))
        assert ids[pos] == leaf_token and heights[pos] == 2 and seeds[pos] == rng.derive(2)
    # (token_subtrie_ranges: a node's subtree is a contiguous DFS range -- what the trie attention mask and BuildTreePrefix read)
    assert flat.nodes().tolist() == [[0, 6, 0], [1, 1, 1], [2, 3, 1], [3, 3, 2], [4, 6, 1], [5, 5, 2], [6, 6, 2]]


def sample_tree():
    """trie_test.rs:187-198"""
    root = TrieNode(0, 0, 0.0)
    mid_a = TrieNode(1, 1, -0.1)
    mid_a.add(TrieNode(4, 2, -0.4))
    mid_b = TrieNode(2, 1, -0.2)
    mid_b.add(TrieNode(5, 2, -2.8))
    root.add(mid_a)
    root.add(mid_b)
    root.add(TrieNode(3, 1, -0.3))
    return root


@pytest.mark.parametrize("budget,tokens", [(4, [0, 1, 2, 3]), (2, [0, 1]), (6, [0, 1, 4, 2, 5, 3]), (100, [0, 1, 4, 2, 5, 3])])
def test_trie_prune_to_budget(budget, tokens):
    """trie_test.rs:200-221"""
    trie = sample_tree()
    trie.prune_to_budget(budget)
    assert trie.node_count() == len(tokens) and list(trie.linearize().token_ids()) == tokens
    if budget == 4:
        assert [np.float32(trie.get(t).logprob) for t in (1, 2, 3)] == [np.float32(-0.1), np.float32(-0.2), np.float32(-0.3)]


def test_trie_prune_to_budget_tie_keeps_parent():
    """trie_test.rs:223-234"""
    root = TrieNode(0, 0, 0.0)
    child = TrieNode(1, 1, 0.0)
    child.add(TrieNode(2, 2, 0.0))
    root.add(child)
    root.add(TrieNode(3, 1, 0.0))
    root.prune_to_budget(2)
    assert root.node_count() == 2 and list(root.linearize().token_ids()) == [0, 1]


def test_trie_flat_chain_carries_the_position_seeds():
    """trie.rs:140-156: node i of a flat trie behind `prefix_length` accepted tokens carries PRng::derive(prefix_length + i) -- the seed the
    linear decode step at that position would draw with (stream.rs:248-258), so stochastic verification reproduces linear sampling."""
    rng = PRng(1234)
    flat = TrieNode.flat([7, 8, 9], prefix_length=40, prng=rng).linearize()
    assert list(flat.token_seeds()) == [rng.derive(40), rng.derive(41), rng.derive(42)] and flat.is_flat()
