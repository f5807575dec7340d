"""CPU: the oracle's restatement of the DeltaNet tree-verify kernels (oracle/uzu_oracle_tree_verify.c <- cpu/kernel/gdn/tree_verify/*.rs)
pinned by INDEPENDENT float64 statements of the same math: the chunked form (prefix sums, Gram tiles, block forward substitution) must equal
the plain delta-rule recurrence walked along every root path of the tree; plus the host trie (uzu_amd/trie.py <- trie.rs) and the
model-level verify -> accept protocol (stream.rs:380-470, 556-628) against linear decoding.

The reference holds no known-answer vectors for these kernels (its tests compare Metal with the CPU kernels on procedural inputs,
tests/unit/backends/common/kernel/gdn/tree_verify/*_test.rs); the procedural inputs of tree_gram_test.rs are reused below."""
import numpy as np
import pytest

from oracle import oracle as O
from uzu_amd import synthetic as S
from uzu_amd.trie import DuplicateTokenId, FlatTrie, TrieNode, parents as trie_parents

F32 = O.F32


def random_tree(n, rng, branch=0.45):
    """DFS-ordered random tree: returns (nodes uint32 [n,3], parents int32 [n])."""
    root = TrieNode(0)
    path = [root]
    count = 1
    while count < n:
        # either go deeper from the current leaf or back up and branch
        while len(path) > 1 and rng.random() < branch:
            path.pop()
        child = TrieNode(count)
        path[-1].add(child)
        path.append(child)
        count += 1
    flat = root.linearize()
    return flat.nodes(), flat.parents()


def path_of(parents, node):
    out = []
    while node >= 0:
        out.append(node)
        node = int(parents[node])
    return out[::-1]


def delta_rule_f64(h0, q, k, v, log_decay, beta, path, hk_of):
    """S <- exp(g) S; delta = beta (v - S k); S <- S + delta k^T along `path`; returns (state after the path, output of its last node).
    h0 [Hv, Dv, Dk]; q / k [n, Hk, Dk]; v [n, Hv, Dv]; log_decay / beta [n, Hv]."""
    S = h0.astype(np.float64).copy()
    Hv = S.shape[0]
    for node in path:
        for hv in range(Hv):
            kk = k[node, hk_of(hv)].astype(np.float64)
            S[hv] *= np.exp(np.float64(log_decay[node, hv]))
            delta = np.float64(beta[node, hv]) * (v[node, hv].astype(np.float64) - S[hv] @ kk)
            S[hv] += np.outer(delta, kk)
    last = path[-1]
    out = np.stack([S[hv] @ q[last, hk_of(hv)].astype(np.float64) for hv in range(Hv)])
    return S, out


@pytest.mark.parametrize("n", [1, 7, 16, 23, 40])
def test_tree_verify_chunked_form_equals_the_delta_rule_along_every_root_path(n):
    rng = np.random.default_rng(100 + n)
    Hk, Hv, Dk, Dv = 2, 4, 128, 48
    nodes, par = random_tree(n, rng)
    q = (rng.normal(size=(n, Hk, Dk)) / np.sqrt(Dk)).astype(np.float32)
    k = rng.normal(size=(n, Hk, Dk)).astype(np.float32)
    k /= np.linalg.norm(k, axis=-1, keepdims=True)
    v = rng.normal(size=(n, Hv, Dv)).astype(np.float32)
    log_decay = -rng.uniform(0.01, 0.6, size=(n, Hv)).astype(np.float32)
    beta = rng.uniform(0.1, 0.9, size=(n, Hv)).astype(np.float32)
    h0 = (rng.normal(size=(1, Hv, Dv, Dk)) * 0.3).astype(np.float32)
    h0_idx = np.zeros(1, np.int32)
    nb = (n + 15) // 16
    ncp = (nb + 1) // 2
    prefix = np.zeros((n, Hv), np.float32)
    a_packed = np.zeros((Hv, nb, ncp, 16, 32), np.float32)
    qkd = np.zeros((Hv, n, n), np.float32)
    a_inv = np.zeros((Hv, nb, 16, 16), np.float32)
    kh0 = np.zeros((n, Hv, Dv), np.float32)
    u = np.zeros((Hv, n, Dv), np.float32)
    out = np.zeros((n, Hv, Dv), np.float32)
    O.call("orc_build_tree_prefix", nodes, log_decay, prefix, 1, n, Hv)
    O.call("orc_build_tree_gram", q, k, F32, nodes, prefix, beta, h0, h0_idx, a_packed, qkd, a_inv, kh0, 1.0, 1, n, Hk, Hv, Dk, Dv)
    O.call("orc_tree_update_solve", kh0, v, F32, prefix, beta, a_packed, a_inv, h0_idx, u, 1, n, Hv, Dv)
    O.call("orc_build_tree_out", q, F32, prefix, qkd, u, h0, h0_idx, out, F32, 1.0, 1, n, Hk, Hv, Dk, Dv)
    # prefix = sum of the log decays over the ancestors-or-self
    for node in range(n):
        want = sum(log_decay[a].astype(np.float64) for a in path_of(par, node))
        np.testing.assert_allclose(prefix[node], want, rtol=1e-5, atol=1e-6)
    hk_of = lambda hv: hv // (Hv // Hk)
    worst = 0.0
    for node in range(n):
        _, want = delta_rule_f64(h0[0], q, k, v, log_decay, beta, path_of(par, node), hk_of)
        err = np.abs(out[node] - want).max() / max(np.abs(want).max(), 1e-3)
        worst = max(worst, err)
    assert worst < 2e-4, f"tree of {n}: chunked form is {worst:.2e} (relative) off the path recurrence"
    # without an initial state (use_h0 = false): the same tree from the zero state
    u2, out2 = np.zeros_like(u), np.zeros_like(out)
    O.call("orc_build_tree_gram", q, k, F32, nodes, prefix, beta, None, None, a_packed, qkd, a_inv, None, 1.0, 1, n, Hk, Hv, Dk, Dv)
    O.call("orc_tree_update_solve", None, v, F32, prefix, beta, a_packed, a_inv, None, u2, 1, n, Hv, Dv)
    O.call("orc_build_tree_out", q, F32, prefix, qkd, u2, None, None, out2, F32, 1.0, 1, n, Hk, Hv, Dk, Dv)
    for node in range(n):
        _, want = delta_rule_f64(np.zeros_like(h0[0]), q, k, v, log_decay, beta, path_of(par, node), hk_of)
        assert np.abs(out2[node] - want).max() / max(np.abs(want).max(), 1e-3) < 2e-4


def test_tree_gram_on_the_reference_tests_procedural_inputs():
    """tree_gram_test.rs:36-55,205-216 (make_inputs / build_trie): batch 0 is a chain (every node's subtree runs to the end), batch 1 a
    star (root + leaves); batch 1 has no initial state (h0_idx -1: the kh0-skip path).  Against float64 statements of the definitions in
    tree_gram.rs:44-60, 63-96, 98-128, 130-160."""
    B, Hk, Hv, Dk, Dv, T = 2, 2, 6, 128, 80, 37
    q_len = B * T * Hk * Dk
    scalar_len = B * T * Hv
    i = np.arange(q_len, dtype=np.float32)
    q = (np.sin(i * np.float32(0.017)) * np.float32(0.2) + np.float32(0.01)).astype(np.float32).reshape(B, T, Hk, Dk)
    k = (np.cos(i * np.float32(0.019)) * np.float32(0.18) - np.float32(0.02)).astype(np.float32).reshape(B, T, Hk, Dk)
    j = np.arange(scalar_len)
    prefix = (-(j % T).astype(np.float32) * np.float32(0.01) - (j % Hv).astype(np.float32) * np.float32(0.003)).astype(np.float32).reshape(B, T, Hv)
    beta = (np.float32(0.25) + (np.sin(j.astype(np.float32) * np.float32(0.013)) + 1) * np.float32(0.2)).astype(np.float32).reshape(B, T, Hv)
    h = np.arange(B * Hv * Dv * Dk, dtype=np.float32)
    h0 = (np.sin(h * np.float32(0.007)) * np.float32(0.05) - np.float32(0.01)).astype(np.float32).reshape(B, Hv, Dv, Dk)
    h0_idx = np.array([0, -1], np.int32)
    last = T - 1
    trie = np.array([[t, last, t] for t in range(T)] + [[0, last, 0]] + [[t, t, 1] for t in range(1, T)], np.uint32).reshape(B, T, 3)
    nb, ncp = (T + 15) // 16, ((T + 15) // 16 + 1) // 2
    a_packed = np.zeros((B * Hv, nb, ncp, 16, 32), np.float32)
    qkd = np.zeros((B * Hv, T, T), np.float32)
    a_inv = np.zeros((B * Hv, nb, 16, 16), np.float32)
    kh0 = np.zeros((B, T, Hv, Dv), np.float32)
    scale = 0.5
    O.call("orc_build_tree_gram", q, k, F32, trie, prefix, beta, h0, h0_idx, a_packed, qkd, a_inv, kh0, scale, B, T, Hk, Hv, Dk, Dv)
    anc = np.zeros((B, T, T), bool)  # anc[b, row, col]: col is an ancestor-or-self of row
    for b in range(B):
        for col in range(T):
            anc[b, trie[b, col, 0]:trie[b, col, 1] + 1, col] = True
    q64, k64, p64 = q.astype(np.float64), k.astype(np.float64), prefix.astype(np.float64)
    for b in range(B):
        for hv in range(Hv):
            hk = hv // (Hv // Hk)
            decay = np.exp(p64[b, :, hv][:, None] - p64[b, :, hv][None, :])
            want_qkd = np.where(anc[b], scale * decay * (q64[b, :, hk] @ k64[b, :, hk].T), 0.0)
            np.testing.assert_allclose(qkd[b * Hv + hv], want_qkd, rtol=2e-5, atol=2e-6)
            A = np.where(anc[b] & ~np.eye(T, dtype=bool), beta[b, :, hv].astype(np.float64)[:, None] * decay * (k64[b, :, hk] @ k64[b, :, hk].T), 0.0)
            for blk in range(nb):
                rows = slice(blk * 16, min(T, blk * 16 + 16))
                nr = rows.stop - rows.start
                for pair in range(blk // 2 + 1):
                    cols = slice(pair * 32, min(T, pair * 32 + 32))
                    nc = cols.stop - cols.start
                    tile = a_packed[b * Hv + hv, blk, pair]
                    np.testing.assert_allclose(tile[:nr, :nc], A[rows, cols], rtol=2e-5, atol=2e-6)
                    assert not tile[nr:].any() and not tile[:, nc:].any()
                want_inv = np.eye(16)
                want_inv[:nr, :nr] = np.linalg.inv(np.eye(nr) + A[rows, rows])
                np.testing.assert_allclose(a_inv[b * Hv + hv, blk], want_inv, rtol=1e-4, atol=1e-5)
            if h0_idx[b] >= 0:
                np.testing.assert_allclose(kh0[b, :, hv], k64[b, :, hk] @ h0[h0_idx[b], hv].astype(np.float64).T, rtol=2e-5, atol=2e-6)
            else:
                assert not kh0[b, :, hv].any()


def test_state_advance_is_the_delta_rule_over_the_accepted_path():
    rng = np.random.default_rng(7)
    n, Hk, Hv, D = 12, 2, 4, 128
    nodes, par = random_tree(n, rng)
    k = rng.normal(size=(n, Hk, D)).astype(np.float32)
    k /= np.linalg.norm(k, axis=-1, keepdims=True)
    v = rng.normal(size=(n, Hv, D)).astype(np.float32)
    log_decay = -rng.uniform(0.01, 0.6, size=(n, Hv)).astype(np.float32)
    beta = rng.uniform(0.1, 0.9, size=(n, Hv)).astype(np.float32)
    h0 = (rng.normal(size=(Hv, D, D)) * 0.3).astype(np.float32)
    leaf = n - 1
    path = path_of(par, leaf)
    state = h0.copy()
    O.call("orc_state_advance", k, v, F32, log_decay, beta, np.array(path, np.uint32), state, len(path), Hv, Hk, D)
    want, _ = delta_rule_f64(h0, k, k, v, log_decay, beta, path, lambda hv: hv // (Hv // Hk))
    np.testing.assert_allclose(state, want, rtol=1e-4, atol=1e-5)


def test_conv_tree_scan_equals_conv_update_along_every_root_path():
    """ConvTreeScan (tree_verify/conv_scan.rs) against DeltaNetConvUpdate (conv_update.rs) applied token by token along each root path:
    suffix_state is the conv state after the path (exact copies), the activated outputs agree to one bf16 ulp (the tap sum runs
    newest-first in the tree scan and oldest-first in the update)."""
    rng = np.random.default_rng(9)
    n, conv_dim, extra, ks = 11, 96, 24, 4
    total = conv_dim + extra
    nodes, par = random_tree(n, rng)
    f32_to_bf16 = S.f32_to_bf16_bits
    x = f32_to_bf16(rng.normal(size=(n, total)).astype(np.float32))
    w = rng.uniform(-0.6, 0.6, size=(conv_dim, ks)).astype(np.float32)
    bias = rng.uniform(-0.1, 0.1, size=(conv_dim,)).astype(np.float32)
    base = rng.normal(size=(conv_dim, ks - 1)).astype(np.float32)
    out = np.zeros_like(x)
    suffix_state = np.zeros((n, conv_dim, ks - 1), np.float32)
    O.call("orc_conv_tree_scan", x, w, bias, base, par, out, suffix_state, O.BF16, n, ks, total, conv_dim)
    assert np.array_equal(out[:, conv_dim:], x[:, conv_dim:])  # the non-conv channels pass through
    for node in range(n):
        state = base.copy()
        row = None
        for a in path_of(par, node):
            row = x[a, :conv_dim].copy()
            O.call("orc_delta_net_conv_update", w, bias, row, state, ks, conv_dim, ks - 1)
        assert np.array_equal(state, suffix_state[node])
        got, want = S.bf16_bits_to_f32(out[node, :conv_dim]), S.bf16_bits_to_f32(row)
        assert np.all(np.abs(got - want) <= np.maximum(np.abs(want), 1e-3) * 2.0 ** -7)


def test_host_trie_linearize_parents_and_accept():
    root = TrieNode(5)
    a, b = TrieNode(7), TrieNode(9)
    root.add(a), root.add(b)
    a.add(TrieNode(1)), a.add(TrieNode(2)), b.add(TrieNode(3))
    with pytest.raises(DuplicateTokenId):
        root.add(TrieNode(7))
    flat = root.linearize()
    assert flat.token_ids().tolist() == [5, 7, 1, 2, 9, 3]
    assert flat.nodes().tolist() == [[0, 5, 0], [1, 3, 1], [2, 2, 2], [3, 3, 2], [4, 5, 1], [5, 5, 2]]
    assert flat.parents().tolist() == [-1, 0, 1, 1, 0, 4] and not flat.is_flat()
    assert flat.accept([7, 2, 0, 0, 0, 0]) == [(0, 5, 7), (1, 7, 2), (3, 2, 0)]      # root -> 7 -> 2, then the tree has no 0
    assert flat.accept([4, 2, 0, 0, 0, 0]) == [(0, 5, 4)]                            # the model disagrees at the root
    assert flat.accept([9, 0, 0, 0, 3, 8]) == [(0, 5, 9), (4, 9, 3), (5, 3, 8)]
    chain = TrieNode.flat([4, 5, 6]).linearize()
    assert chain.is_flat() and chain.nodes().tolist() == [[0, 2, 0], [1, 2, 1], [2, 2, 2]] and trie_parents(chain.nodes()).tolist() == [-1, 0, 1]


def linear_stream(om, prompt, steps):
    tok, lg = om.prefill(prompt, True)
    out, logits = [tok], [lg]
    for _ in range(steps):
        tok, lg = om.forward([tok], True)
        out.append(tok)
        logits.append(lg)
    return out, logits


def logit_error_sigma(want_bits, got_bits, row_mult):
    """worst |logit difference| in row-normalised units (tests/test_gpu_model.py::logits_close: logit i and its error scale with the
    synthetic read-out row's multiplier m_i)"""
    w, g = S.bf16_bits_to_f32(want_bits).astype(np.float64) / row_mult, S.bf16_bits_to_f32(got_bits).astype(np.float64) / row_mult
    return float(np.abs(w - g).max() / w.std())


# prompt multipliers whose linear greedy streams have every top-2 gap >= 0.3 sigma (a random tiny transformer's stream is either
# varied with near-ties or decided and soon a fixed point -- profiles/r4_stream_search.txt; these are the most varied decided ones)
@pytest.mark.parametrize("preset,mult", [("tiny-qwen", 81), ("tiny-qwen", 121), ("tiny-qwen", 151), ("tiny-llama", 191)])
def test_model_verify_then_accept_equals_linear_decoding(preset, mult):
    """The speculative step of LanguageModelStream (stream.rs:380-470, 556-628) on the oracle model: a tree that contains the true
    continuation among wrong branches is verified in ONE pass, FlatTrie::accept finds the true path, encode_accept leaves the sequence
    in the state linear decoding would have reached -- the tokens that follow are the linear stream's, and the logits of every node on
    the accepted path are the linear step's logits within 0.2 sigma (row-normalised, the unit of every model-level tolerance here).  (DeltaNet tree-verify rounds the normalised q / k to bf16 and
    sums in another order than the one-token update: tolerance class, for the reference as much as for this restatement.)"""
    cfg = S.PRESETS[preset]()
    bundle = S.build_model(cfg)
    row_mult = S.readout_row_multipliers(cfg)
    base = S.synthetic_prompt(37, cfg.vocab_size).astype(np.int64)
    prompt = ((base * mult + 11 * mult) % cfg.vocab_size).astype(np.uint32)
    om = O.OracleModel(bundle)
    depth, rounds = 4, 3
    want, want_logits = linear_stream(om, prompt, rounds * (depth + 1) + 4)
    om.reset()
    got = [om.prefill(prompt)]
    assert got[0] == want[0]
    worst = 0.0
    for rnd in range(rounds):
        # propose: the true continuation want[i+1 .. i+depth] as a chain, a wrong sibling at every level (one of them with a child)
        i = len(got) - 1
        root = TrieNode(got[-1])
        node = root
        for d in range(1, depth + 1):
            wrong = TrieNode((want[i + d] + 17 * d) % cfg.vocab_size)
            if d == 2:
                wrong.add(TrieNode((want[i + d] + 5) % cfg.vocab_size))
            node.add(wrong)
            true_child = TrieNode(want[i + d])
            node.add(true_child)
            node = true_child
        flat = root.linearize()
        assert not flat.is_flat()
        sampled, logits = om.verify_tree(flat.token_ids(), flat.nodes(), True)
        accepted = flat.accept(sampled)
        assert [t for _, t, _ in accepted] == want[i:i + len(accepted)], "the accepted path is not the linear stream"
        assert len(accepted) == depth + 1, f"round {rnd}: only {len(accepted)} of {depth + 1} nodes accepted"
        for j, (idx, _, _) in enumerate(accepted):  # node j of the path consumed want[i + j]: its logits are linear step i + j + 1's
            worst = max(worst, logit_error_sigma(want_logits[i + j + 1], logits[idx], row_mult))
        om.accept([idx for idx, _, _ in accepted])
        got.extend(int(s) for _, _, s in accepted)
    assert got == want[:len(got)], f"speculative {got}\nlinear      {want}"
    assert om.context_length == len(prompt) + len(got) - 1
    assert worst <= 0.2, f"verify-pass logits are {worst:.3f} sigma off the linear steps'"
    print(f"{preset} x{mult}: verify-pass logits within {worst:.3f} sigma of the linear steps'")
    # and plain decoding carries on from the accepted state
    tok = got[-1]
    for step in range(len(got), len(want)):
        tok = om.forward([tok])
        assert tok == want[step]
    om.close()


def test_model_verify_with_nothing_right_accepts_the_root_only():
    cfg = S.tiny_qwen()
    bundle = S.build_model(cfg)
    prompt = ((S.synthetic_prompt(37, cfg.vocab_size).astype(np.int64) * 81 + 891) % cfg.vocab_size).astype(np.uint32)
    om = O.OracleModel(bundle)
    want, _ = linear_stream(om, prompt, 6)
    om.reset()
    tok = om.prefill(prompt)
    root = TrieNode(tok)
    bad = TrieNode((want[1] + 1) % cfg.vocab_size)
    bad.add(TrieNode(3))
    root.add(bad)
    flat = root.linearize()
    sampled = om.verify_tree(flat.token_ids(), flat.nodes())
    accepted = flat.accept(sampled)
    assert accepted == [(0, tok, want[1])]
    om.accept([0])
    assert om.forward([want[1]]) == want[2]
    om.close()


def test_oracle_tree_verify_matches_the_committed_fixture():
    """tests/golden/tree_verify.json (make_tree_verify_golden.py): the oracle's verify / accept path frozen -- sampled tokens, a digest of
    every node's logits, the accepted paths, the tokens decoded afterwards."""
    import hashlib
    import json
    import os
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tree_verify.json")))
    for preset, case in fx.items():
        cfg = S.PRESETS[preset]()
        bundle = S.build_model(cfg)
        mult = case["prompt_multiplier"]
        prompt = ((S.synthetic_prompt(case["prompt_len"], cfg.vocab_size).astype(np.int64) * mult + 11 * mult) % cfg.vocab_size).astype(np.uint32)
        om = O.OracleModel(bundle)
        assert linear_stream(om, prompt, len(case["linear_stream"]) - 1)[0] == case["linear_stream"]
        om.reset()
        om.prefill(prompt)
        for rnd in case["rounds"]:
            sampled, logits = om.verify_tree(np.array(rnd["token_ids"], np.uint32), np.array(rnd["nodes"], np.uint32), True)
            assert [int(t) for t in sampled] == rnd["sampled"]
            assert [hashlib.sha256(logits[n].tobytes()).hexdigest()[:16] for n in range(len(rnd["sampled"]))] == rnd["logit_sha256_16"]
            om.accept(rnd["accepted"])
        tok = rnd["sampled"][rnd["accepted"][-1]]
        after = []
        for _ in range(len(case["decoded_after"])):
            tok = om.forward([tok])
            after.append(tok)
        assert after == case["decoded_after"]
        om.close()
