"""CPU: the oracle's Mixture-of-Experts MLP (oracle/uzu_oracle_moe.c <- encodable_block/mlp/moe/mod.rs:204-350) against the expectations the reference's own tests
hold the kernels to -- computed here, as there, independently of any kernel:

  * router top-k: the logits in the CPU kernel's four-accumulator order (numpy float32 emulation), ids by (value desc, expert id asc), softmax over the winners
    (moe_router_topk_test.rs: ids equal, probabilities within 2e-2 / 5e-2);
  * counts / offsets: histogram + exclusive scan, ids outside [0, e) ignored (moe_counts_offsets_fused_test.rs);
  * scatter: tok2row inverts the bucketing, an expert's rows in (token, slot) order (moe_experts_test.rs:92-133 scatter_by_expert); gather copies rows (moe_gather_test.rs);
  * experts + finalize: cpu_moe_reference (moe_experts_test.rs:170-275) restated in float64: SwiGLU / GEGLU with clipping, y[t] = sum_k prob * (hidden W2^T + b)
    (the reference's tolerance for the bf16 block output: 1e-2 relative to the output scale);
  * finalize: non-finite terms are dropped (moe_finalize_test.rs)."""
import ctypes as C
from dataclasses import replace

import numpy as np
import pytest

from helpers import bf16, f32
from oracle import oracle as O
from uzu_amd import synthetic as S


def router_f32_emulation(x, w, b):
    """logits[t, e] in router_topk.rs:57-84's order: four strided f32 accumulators, (a0 + a1) + (a2 + a3), + bias"""
    xf, wf = f32(x).astype(np.float32), f32(w).astype(np.float32)
    t, d = xf.shape
    acc = np.zeros((t, wf.shape[0], 4), np.float32)
    for c in range(0, d, 4):
        acc += (wf[None, :, c:c + 4] * xf[:, None, c:c + 4]).astype(np.float32)
    s = (acc[..., 0] + acc[..., 1]).astype(np.float32) + (acc[..., 2] + acc[..., 3]).astype(np.float32)
    return (s + f32(b).astype(np.float32)[None, :]).astype(np.float32)


def run_router(x, w, b, k, renorm):
    t, d = x.shape
    e = w.shape[0]
    ids = np.full((t, k), -1, np.int32)
    probs = np.zeros((t, k), np.uint16)
    O.call("orc_moe_router_topk", x, w, b, ids, probs, O.BF16, t, d, e, k, int(renorm))
    return ids, probs


@pytest.mark.parametrize("t,d,e,k,renorm", [(1, 64, 4, 1, True), (5, 128, 16, 4, True), (3, 256, 128, 8, False), (2, 512, 512, 128, True)])
def test_router_topk_against_the_reference_tests_expectation(t, d, e, k, renorm):
    rng = np.random.default_rng(1234)
    x, w, b = bf16(rng.uniform(-1, 1, (t, d))), bf16(rng.uniform(-1, 1, (e, d))), bf16(rng.uniform(-0.5, 0.5, e))
    ids, probs = run_router(x, w, b, k, renorm)
    logits = router_f32_emulation(x, w, b)
    for ti in range(t):
        order = sorted(range(e), key=lambda j: (-float(logits[ti, j]), j))[:k]
        assert list(ids[ti]) == order
        best = logits[ti, order].astype(np.float32)
        if renorm:
            ex = np.exp((best - best.max()).astype(np.float32)).astype(np.float32)
            want = ex / ex.sum(dtype=np.float32)
            assert np.abs(f32(probs[ti]) - want).max() <= 2e-2
            assert abs(float(f32(probs[ti]).sum()) - 1.0) <= 2e-2
        else:
            assert np.array_equal(probs[ti], bf16(best))


def test_router_ties_keep_the_lower_expert_id():
    x = bf16(np.ones((1, 8)))
    w = bf16(np.stack([np.full(8, v) for v in (0.5, 1.0, 1.0, 0.25, 1.0)]))
    ids, probs = run_router(x, w, bf16(np.zeros(5)), 3, True)
    assert list(ids[0]) == [1, 2, 4]
    assert np.array_equal(probs[0], bf16(np.full(3, 1.0 / 3.0, np.float32)))


def counts_offsets(ids, t, e, k):
    offsets, sumk, partials = np.zeros(e + 1, np.uint32), np.zeros(1, np.uint32), np.zeros(max(e, 1), np.uint32)
    O.call("orc_moe_counts_offsets_fused", ids, offsets, sumk, partials, t, e, k)
    return offsets, int(sumk[0]), partials


def test_counts_offsets_histogram_scan_and_ignored_ids():
    rng = np.random.default_rng(7)
    t, e, k = 37, 24, 4
    ids = rng.integers(-2, e + 3, (t, k)).astype(np.int32)  # some negative, some >= e
    offsets, sumk, partials = counts_offsets(ids, t, e, k)
    valid = ids[(ids >= 0) & (ids < e)]
    counts = np.bincount(valid, minlength=e)
    assert np.array_equal(partials[:e], counts) and sumk == counts.sum() == int(offsets[e])
    assert np.array_equal(offsets[:e], np.concatenate([[0], np.cumsum(counts)[:-1]]))


def test_scatter_inverts_the_bucketing_in_token_slot_order_and_gather_copies_rows():
    rng = np.random.default_rng(11)
    t, e, k, d = 29, 8, 3, 64
    ids = rng.integers(0, e, (t, k)).astype(np.int32)
    ids[4, 1] = -1  # one slot routed nowhere
    probs = bf16(rng.uniform(0.1, 1.0, (t, k)))
    offsets, sumk, _ = counts_offsets(ids, t, e, k)
    b_ids, b_probs, tok2row = np.full(t * k, -7, np.int32), np.zeros(t * k, np.uint16), np.full(t * k, -7, np.int32)
    O.call("orc_moe_scatter_buckets", ids, probs, offsets, b_ids, b_probs, tok2row, O.BF16, t, e, k)
    assert tok2row[4 * k + 1] == -1 and sumk == t * k - 1
    cursor = offsets[:e].astype(np.int64).copy()
    for i, eid in enumerate(ids.reshape(-1)):  # scatter_by_expert (moe_experts_test.rs:113-123)
        if eid < 0:
            continue
        assert tok2row[i] == cursor[eid] and b_ids[cursor[eid]] == i // k and b_probs[cursor[eid]] == probs.reshape(-1)[i]
        cursor[eid] += 1
    assert np.array_equal(cursor, offsets[1:])
    x = bf16(rng.normal(size=(t, d)))
    x_perm = np.zeros((t * k, d), np.uint16)
    O.call("orc_moe_gather", x, b_ids, x_perm, np.array([sumk], np.uint32), O.BF16, d, t, k)
    assert all(np.array_equal(x_perm[r], x[b_ids[r]]) for r in range(sumk)) and not x_perm[sumk:].any()


def moe_float64(x, ids, probs, mo, d, F):
    """cpu_moe_reference (moe_experts_test.rs:170-275) in float64"""
    xf = f32(x).astype(np.float64)
    w13, w2 = f32(mo.w13).astype(np.float64), f32(mo.w2).astype(np.float64)
    ub, db = f32(mo.up_biases).astype(np.float64), f32(mo.down_biases).astype(np.float64)
    y = np.zeros((x.shape[0], d))
    for t in range(x.shape[0]):
        for kk in range(ids.shape[1]):
            ex = int(ids[t, kk])
            if ex < 0:
                continue
            up = np.clip(w13[ex, :F] @ xf[t] + ub[ex, :F], mo.up_clip[0], mo.up_clip[1])
            gate = np.clip(w13[ex, F:] @ xf[t] + ub[ex, F:], mo.gate_clip[0], mo.gate_clip[1])
            if mo.gating_sel == 2:
                act = gate / (1.0 + np.exp(-mo.silu_alpha * gate))
            else:
                act = 0.5 * gate * (1.0 + np.tanh(0.7978846 * (gate + 0.044715 * gate ** 3)))
            hidden = act * up
            y[t] += float(f32(probs[t, kk:kk + 1])[0]) * (w2[ex] @ hidden + db[ex])
    return y


@pytest.mark.parametrize("gelu,clip,t", [(False, None, 1), (False, 2.0, 9), (True, None, 5)])
def test_block_against_the_reference_tests_float_reference(gelu, clip, t):
    cfg = replace(S.tiny_llama(), moe_experts=8, moe_active=2, moe_hidden=96, moe_gelu=gelu, moe_clip=clip)
    bundle = S.build_model(cfg)
    mo = bundle.layers[0].moe
    d, F, K, E = cfg.model_dim, 96, 2, 8
    rng = np.random.default_rng(5)
    x = bf16(rng.normal(0.0, 1.0, (t, d)))
    ids, probs = run_router(x, mo.router_weights, mo.router_biases, K, True)
    desc = mo.desc()
    fn = O.lib().orc_moe_block
    fn.restype, fn.argtypes = C.c_void_p, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    ptr = fn(C.byref(desc), d, x.ctypes.data, t)
    got = f32(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint16)), shape=(t * d,)).copy().reshape(t, d)).astype(np.float64)
    want = moe_float64(x, ids, probs, mo, d, F)
    # y_partial is rounded to bf16 before the weighted sum and the result again: two bf16 roundings of values of the output's scale
    assert np.abs(got - want).max() <= 1e-2 * max(1.0, np.abs(want).max())
    if clip is not None:  # the clip must bite for the test to mean anything
        xf = f32(x).astype(np.float64)
        ups = np.concatenate([f32(mo.w13[int(e_), :F]).astype(np.float64) @ xf[ti] for ti in range(t) for e_ in ids[ti]])
        assert (np.abs(ups) > clip).any()


def test_finalize_drops_non_finite_terms():
    t, d, k = 2, 8, 3
    tok2row = np.array([[0, 1, -1], [2, 3, 4]], np.int32)
    probs = bf16(np.array([[0.5, np.inf, 0.25], [0.25, 0.5, 0.25]]))
    y_partial = bf16(np.arange(5 * d, dtype=np.float32).reshape(5, d))
    y_partial[3, 2] = bf16(np.array([np.nan]))[0]
    y = np.zeros((t, d), np.uint16)
    O.call("orc_moe_finalize", tok2row, probs, y_partial, y, O.BF16, t, d, k)
    yp = f32(y_partial).astype(np.float32)
    want0 = 0.5 * yp[0]  # slot 1 has a non-finite probability (-> 0), slot 2 no row
    assert np.array_equal(y[0], bf16(want0))
    want1 = np.float32(0.25) * yp[2] + np.float32(0.5) * np.where(np.isfinite(yp[3]), yp[3], 0.0).astype(np.float32) + np.float32(0.25) * yp[4]
    assert np.array_equal(y[1], bf16(want1.astype(np.float32)))


def test_moe_model_round_trips_the_loader_and_differs_from_the_dense_model(tmp_path):
    from uzu_amd import loader as L
    cfg = replace(S.tiny_llama(), moe_experts=8, moe_active=2, moe_hidden=128)
    bundle = S.build_model(cfg)
    prompt = S.synthetic_prompt(12, cfg.vocab_size)

    def stream(b):
        om = O.OracleModel(b)
        tok, lg = om.prefill(prompt, True)
        out = [tok]
        for _ in range(3):
            tok = om.forward([tok])
            out.append(tok)
        om.close()
        return out, lg
    a, la = stream(bundle)
    L.save_model_dir(bundle, str(tmp_path))
    cfg_json = __import__("json").load(open(tmp_path / "config.json"))
    mlp = cfg_json["decoder_config"]["transformer_config"]["layer_configs"][0]["mlp_config"]
    assert mlp["type"] == "MixtureOfExpertsConfig" and mlp["num_routed_experts"] == 8 and mlp["num_active_routed_experts"] == 2 and mlp["routing_function"]["type"] == "SoftmaxRouting"
    b, lb = stream(L.load_model_dir(str(tmp_path)))
    assert a == b and np.array_equal(la, lb)
    dense, ld = stream(S.build_model(S.tiny_llama()))
    assert not np.array_equal(la, ld)
    # MoeBlock::new's refusals are load errors
    import json
    bad = json.loads(json.dumps(cfg_json))
    bad["decoder_config"]["transformer_config"]["layer_configs"][0]["mlp_config"]["num_shared_experts"] = 1
    json.dump(bad, open(tmp_path / "config.json", "w"))
    with pytest.raises(L.UnsupportedModelError, match="shared experts"):
        L.load_model_dir(str(tmp_path))
