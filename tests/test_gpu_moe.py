"""GPU parity tests of the Mixture-of-Experts MLP (csrc/k_moe.hip <- encodable_block/mlp/moe/mod.rs:204-350) through the C ABI against the CPU oracle
(oracle/uzu_oracle_moe.c): router top-k BIT-EXACT (ids and probabilities: the CPU kernel's own accumulation order), counts / offsets, scatter (+ row map), gather
and finalize exact; the expert passes bit-identical in reference-order mode and within the reference test's tolerance with the production kernels; and a MoE
model end to end (prefill + decode: reference-order logits bit-identical to the oracle's, production within the model tests' tolerance)."""
import ctypes as C
from dataclasses import replace

import numpy as np
import pytest

from helpers import bf16, f32
from oracle import oracle as O
from test_gpu_kernels import run
from uzu_amd import _ffi
from uzu_amd import backend as B
from uzu_amd import synthetic as S
from uzu_amd.engine import HipModel

pytestmark = pytest.mark.gpu


def _set_exact(on):
    fn = _ffi.lib().uzu_hip_set_exact
    fn.restype, fn.argtypes = None, [C.c_int32]
    fn(1 if on else 0)


def _moe(gelu=False, clip=None, E=8, K=2, F=96, d_model=None):
    cfg = replace(S.tiny_llama(), moe_experts=E, moe_active=K, moe_hidden=F, moe_gelu=gelu, moe_clip=clip)
    bundle = S.build_model(cfg)
    return cfg, bundle, bundle.layers[0].moe


@pytest.mark.parametrize("t,E,K,renorm", [(1, 8, 2, True), (37, 16, 4, True), (5, 128, 8, False), (3, 512, 128, True)])
def test_router_counts_scatter_gather_finalize_exact(hip_ctx, t, E, K, renorm):
    rng = np.random.default_rng(100 + t)
    d = 256
    x, w, b = bf16(rng.uniform(-1, 1, (t, d))), bf16(rng.uniform(-1, 1, (E, d))), bf16(rng.uniform(-0.5, 0.5, E))
    # oracle chain
    o_ids, o_probs = np.full((t, K), -1, np.int32), np.zeros((t, K), np.uint16)
    O.call("orc_moe_router_topk", x, w, b, o_ids, o_probs, O.BF16, t, d, E, K, int(renorm))
    o_off, o_sumk = np.zeros(E + 1, np.uint32), np.zeros(1, np.uint32)
    O.call("orc_moe_counts_offsets_fused", o_ids, o_off, o_sumk, None, t, E, K)
    o_bid, o_bp, o_t2r = np.zeros(t * K, np.int32), np.zeros(t * K, np.uint16), np.zeros(t * K, np.int32)
    O.call("orc_moe_scatter_buckets", o_ids, o_probs, o_off, o_bid, o_bp, o_t2r, O.BF16, t, E, K)
    o_xp = np.zeros((t * K, d), np.uint16)
    O.call("orc_moe_gather", x, o_bid, o_xp, o_sumk, O.BF16, d, t, K)
    yp = bf16(rng.normal(size=(t * K, d)))
    o_y = np.zeros((t, d), np.uint16)
    O.call("orc_moe_finalize", o_t2r, o_probs, yp, o_y, O.BF16, t, d, K)
    # HIP, through the C ABI
    ctx = hip_ctx
    bx, bw, bb = ctx.buffer_from(x), ctx.buffer_from(w), ctx.buffer_from(b)
    bids, bprobs = ctx.buffer_from(np.full((t, K), -1, np.int32)), ctx.buffer_from(np.zeros((t, K), np.uint16))
    boff, bsum, bpart = ctx.create_buffer((E + 1) * 4), ctx.create_buffer(4), ctx.create_buffer(E * 4)
    bbid, bbp, bt2r, bmap = ctx.create_buffer(t * K * 4), ctx.create_buffer(t * K * 2), ctx.create_buffer(t * K * 4), ctx.create_buffer(t * K * 4)
    bxp, byp, by = ctx.buffer_from(np.zeros((t * K, d), np.uint16)), ctx.buffer_from(yp), ctx.create_buffer(t * d * 2)
    router = B.MoeRouterTopKKernel.new(ctx, B.BF16, 1, 0, 0, 0, 0)
    counts, scatter = B.MoeCountsOffsetsFusedKernel.new(ctx), B.MoeScatterBucketsMapKernel.new(ctx, B.BF16)
    gather, finalize = B.MoeGatherXPermKernel.new(ctx, B.BF16), B.MoeFinalizeKernel.new(ctx, B.BF16)

    def enc(cb):
        router.encode(bx, bw, bb, bids, bprobs, t, d, E, K, renorm, cb)
        counts.encode(bids, boff, bsum, bpart, t, E, K, cb)
        scatter.encode(bids, bprobs, boff, bbid, bbp, t, E, K, bt2r, bmap, cb)
        gather.encode(bx, bbid, bxp, bsum, d, t, K, cb)
        finalize.encode(bt2r, bprobs, byp, by, t, d, K, cb)
    run(ctx, enc)
    assert np.array_equal(bids.download(np.int32).reshape(t, K), o_ids), "top-k ids"
    assert np.array_equal(bprobs.download(np.uint16).reshape(t, K), o_probs), "top-k probabilities (bit-exact: the CPU kernel's accumulation order)"
    assert np.array_equal(boff.download(np.uint32), o_off) and int(bsum.download(np.uint32)[0]) == int(o_sumk[0])
    n = int(o_sumk[0])
    assert np.array_equal(bbid.download(np.int32)[:n], o_bid[:n]) and np.array_equal(bbp.download(np.uint16)[:n], o_bp[:n]) and np.array_equal(bt2r.download(np.int32), o_t2r)
    want_map = np.repeat(np.arange(E, dtype=np.uint32), np.diff(o_off.astype(np.int64)))
    assert np.array_equal(bmap.download(np.uint32)[:n], want_map)
    assert np.array_equal(bxp.download(np.uint16).reshape(t * K, d)[:n], o_xp[:n])
    assert np.array_equal(by.download(np.uint16).reshape(t, d), o_y), "finalize (bit-exact: k terms in slot order)"


@pytest.mark.parametrize("gelu,clip,t", [(False, None, 1), (False, 1.5, 40), (True, None, 7)])
def test_expert_passes_reference_order_bit_identical_production_within_tolerance(hip_ctx, gelu, clip, t):
    cfg, bundle, mo = _moe(gelu=gelu, clip=clip)
    d, F, E, K = cfg.model_dim, 96, 8, 2
    rng = np.random.default_rng(9)
    x = bf16(rng.normal(size=(t, d)))
    ids, probs = np.zeros((t, K), np.int32), np.zeros((t, K), np.uint16)
    O.call("orc_moe_router_topk", x, mo.router_weights, mo.router_biases, ids, probs, O.BF16, t, d, E, K, 1)
    off, sumk = np.zeros(E + 1, np.uint32), np.zeros(1, np.uint32)
    O.call("orc_moe_counts_offsets_fused", ids, off, sumk, None, t, E, K)
    bid, bp, t2r = np.zeros(t * K, np.int32), np.zeros(t * K, np.uint16), np.zeros(t * K, np.int32)
    O.call("orc_moe_scatter_buckets", ids, probs, off, bid, bp, t2r, O.BF16, t, E, K)
    xp = np.zeros((t * K, d), np.uint16)
    O.call("orc_moe_gather", x, bid, xp, sumk, O.BF16, d, t, K)
    o_hidden = np.zeros((t * K, F), np.float32)
    O.call("orc_moe_experts_pass_a", xp, off, mo.w13, mo.up_biases, o_hidden, O.BF16, d, F, E, float(mo.gate_clip[0]), float(mo.gate_clip[1]), float(mo.up_clip[0]), float(mo.up_clip[1]),
           float(mo.silu_alpha), mo.gating_sel)
    rmap = np.repeat(np.arange(E, dtype=np.uint32), np.diff(off.astype(np.int64)))
    o_y = np.zeros((t * K, d), np.uint16)
    O.call("orc_moe_experts_down", o_hidden, rmap, mo.w2, mo.down_biases, o_y, O.BF16, t * K, d, F, E)
    ctx = hip_ctx
    bxp, bmap, bsum = ctx.buffer_from(xp), ctx.buffer_from(rmap), ctx.buffer_from(sumk)
    bw13, bub, bw2, bdb = ctx.buffer_from(mo.w13), ctx.buffer_from(mo.up_biases), ctx.buffer_from(mo.w2), ctx.buffer_from(mo.down_biases)
    pass_a, down = B.MoeExpertsPassAKernel.new(ctx, B.BF16, mo.gating_sel), B.MoeExpertsDownKernel.new(ctx, B.BF16)
    for exact in (True, False):
        _set_exact(exact)
        try:
            bh, by = ctx.buffer_from(np.zeros((t * K, F), np.float32)), ctx.buffer_from(np.zeros((t * K, d), np.uint16))

            def enc(cb):
                pass_a.encode(bxp, bmap, bsum, bw13, bub, bh, d, F, mo.gate_clip[0], mo.gate_clip[1], mo.up_clip[0], mo.up_clip[1], mo.silu_alpha, t * K, cb)
                down.encode(bh, bmap, bsum, bw2, bdb, by, d, F, t * K, cb)
            run(ctx, enc)
        finally:
            _set_exact(False)
        h, y = bh.download(np.float32).reshape(t * K, F), by.download(np.uint16).reshape(t * K, d)
        if exact and not gelu:  # (GEGLU: the device tanhf is not glibc's -- tolerance class in both modes, like the dense MLP's GELU)
            assert np.array_equal(h.view(np.uint32), o_hidden.view(np.uint32)), "pass A hidden values (reference-order mode)"
            assert np.array_equal(y, o_y), "pass B rows (reference-order mode)"
        else:
            assert np.abs(h - o_hidden).max() <= 2e-3 * max(1.0, np.abs(o_hidden).max())
            assert np.abs(f32(y).astype(np.float64) - f32(o_y)).max() <= 1e-2 * max(1.0, np.abs(f32(o_y)).max())  # moe_experts_test.rs's tolerance for the bf16 rows


@pytest.mark.parametrize("exact", [True, False])
def test_moe_model_end_to_end(hip_ctx, exact):
    cfg = replace(S.tiny_llama(), moe_experts=8, moe_active=2, moe_hidden=128)
    bundle = S.build_model(cfg)
    prompt = ((S.synthetic_prompt(140, cfg.vocab_size).astype(np.int64) * 5 + 11) % cfg.vocab_size).astype(np.uint32)  # >= 128 rows: the prefill GEMM paths around the MoE block
    om = O.OracleModel(bundle)
    o_tok, o_lg = om.prefill(prompt, True)
    _set_exact(exact)
    try:
        hm = HipModel(hip_ctx, bundle)
        h_tok = hm.prefill(prompt)
        h_lg = hm.read_logits()
        m = S.readout_row_multipliers(cfg).astype(np.float64)

        def check(h_bits, o_bits, where):
            if exact:
                assert np.array_equal(h_bits, o_bits), f"{where}: logits differ from the oracle's in reference-order mode"
            else:
                w_o, w_h = f32(o_bits).astype(np.float64), f32(h_bits).astype(np.float64)
                sigma = (w_o / m).std()
                assert (np.abs(w_h - w_o) / m).max() <= 0.25 * sigma, f"{where}: {(np.abs(w_h - w_o) / m).max() / sigma:.3f} sigma"
        check(h_lg, o_lg, "prefill")
        tok = o_tok
        for step in range(4):  # teacher-forced on the oracle's tokens
            hm.set_next_token(tok)
            t_h, _ = hm.decode(1)
            tok, o_lg = om.forward([tok], True)
            check(hm.read_logits(), o_lg, f"decode step {step}")
            if exact:
                assert int(t_h[0]) == tok
        hm.close()
    finally:
        _set_exact(False)
    if exact:
        assert h_tok == o_tok
