"""GPU parity tests, kernel level: every HIP kernel called THROUGH THE C ABI (uzu_amd.backend mirrors the
reference's `XxxKernel::{new,encode}`) against the CPU oracle on the same seeded inputs.

Pattern = the reference's own kernel tests (crates/backend-uzu/tests/unit/backends/common/kernel/**):
procedural inputs -> CPU kernel as oracle -> compare.  Bars:
  * element-wise kernels (rope/KV scatter, activations, gather, conv update, tensor ops, argmax, pass-2 merge):
    BIT-EXACT;
  * kernels with a parallel reduction (matmul, norms, attention, delta-rule): tolerance stated per test in
    bf16 ulps -- far tighter than the reference's own CPU-vs-Metal tolerances (gemv_test.rs:149,
    quant_dispatch_test.rs:124, attention_single_pass_test.rs:132-136), which are quoted next to each.
  * matmul in reference-order mode (uzu_hip_set_exact_matmul): BIT-EXACT.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from helpers import MASK_VARIANTS, bf16, f32, mask_case, quant_matrix, ref_attention_inputs, trie_from_parents, ulp_diff_bf16
from oracle import oracle as O
from uzu_amd import _ffi
from uzu_amd import backend as B

pytestmark = pytest.mark.gpu


def run(ctx, fn):
    cb = ctx.create_command_buffer("test").start_encoding()
    fn(cb)
    cb.end_encoding().submit().wait_until_completed()
    return cb


# ------------------------------------------------------------------------------------------ matmul
def oracle_matmul(a_bits, q, m, bias=None, ab_scale=1.0, accumulate_d=None, soft_cap=None, gather=None, n_out=None):
    n = n_out if n_out is not None else q["n"]
    d = np.zeros((m, n), dtype=np.uint16) if accumulate_d is None else accumulate_d.copy()
    args = O.MatmulArgs()
    args.a, args.a_dtype = a_bits.ctypes.data, O.BF16
    args.b = q["weights"].ctypes.data
    args.scales = q["scales"].ctypes.data
    args.biases = q["biases"].ctypes.data if q["biases"] is not None else None
    args.zero_points = q["zero_points"].ctypes.data if q["zero_points"] is not None else None
    args.w_dtype, args.method, args.bits, args.group_size = O.BF16, q["method"], q["bits"], q["group_size"]
    args.signed_codes = int(q.get("signed_codes", False))
    args.b_transpose = 1
    args.d, args.d_dtype = d.ctypes.data, O.BF16
    args.ab_scale, args.accumulate = ab_scale, int(accumulate_d is not None)
    args.bias = bias.ctypes.data if bias is not None else None
    args.has_soft_cap, args.soft_cap = int(soft_cap is not None), soft_cap or 0.0
    args.gather_indices = gather.ctypes.data if gather is not None else None
    args.m, args.n, args.k = m, n, q["k"]
    O.lib().orc_matmul(C.byref(args))
    return d


def hip_matmul(ctx, a_bits, q, m, bias=None, ab_scale=1.0, accumulate_d=None, soft_cap=None, gather=None, n_out=None):
    n = n_out if n_out is not None else q["n"]
    kern = B.MatmulKernel.new(ctx, B.BF16, B.BF16, B.BF16)
    ba, bw, bs = ctx.buffer_from(a_bits), ctx.buffer_from(q["weights"]), ctx.buffer_from(q["scales"])
    bb = ctx.buffer_from(q["biases"]) if q["biases"] is not None else None
    bz = ctx.buffer_from(q["zero_points"]) if q["zero_points"] is not None else None
    bd = ctx.buffer_from(accumulate_d if accumulate_d is not None else np.zeros((m, n), dtype=np.uint16))
    bbias = ctx.buffer_from(bias) if bias is not None else None
    bg = ctx.buffer_from(gather) if gather is not None else None
    kind = {0: B.B_SCALE_BIAS, 1: B.B_SCALE_ZERO_POINT, 2: B.B_SCALE_SYMMETRIC}[q["method"]]
    run(ctx, lambda cb: kern.encode(cb, a=ba, b=bw, d=bd, m=m, n=n, k=q["k"], b_kind=kind, scales=bs, biases=bb, zero_points=bz,
                                     mode=B.QMODE_U4 if q["bits"] == 4 else B.QMODE_U8, group_size=q["group_size"],
                                     signed_codes=q.get("signed_codes", False), ab_scale=ab_scale, accumulate=accumulate_d is not None,
                                     bias=bbias, soft_cap=soft_cap, gather_indices=bg))
    return bd.download(np.uint16, m * n).reshape(m, n)


def activations(rng, m, k):
    return bf16(rng.uniform(-1.0, 1.0, size=(m, k)))


# Shapes: the reference's Qwen3.5-0.8B layer table (crates/backend-uzu/src/tests/matmul/shape.rs:82-100)
QWEN_SHAPES = [(3072, 1024), (1024, 2048), (2048, 1024), (7168, 1024), (1024, 3584), (8224, 1024)]


@pytest.mark.parametrize("n,k", QWEN_SHAPES)
def test_gemv_int4_scale_bias_qwen_shapes(hip_ctx, n, k):
    """int4 ScaleBias g=128, M=1 (decode GEMV).  Reference tolerance bf16 rel 0.05 / abs 0.4
    (quant_dispatch_test.rs:124); ours: <= 1 bf16 ulp per element, >= 99% of elements bit-identical."""
    rng = np.random.default_rng(n * 7 + k)
    q = quant_matrix(rng, n, k, 4, 128, 0)
    a = activations(rng, 1, k)
    want, got = oracle_matmul(a, q, 1), hip_matmul(hip_ctx, a, q, 1)
    ulps = ulp_diff_bf16(want, got)
    assert ulps.max() <= 1.0, f"max {ulps.max()} bf16 ulps"
    assert (want == got).mean() >= 0.99


@pytest.mark.parametrize("n,k,bits,method", [(2048, 16384, 4, 0), (1184, 28672, 4, 1), (4096, 4096, 4, 0), (6144, 8192, 4, 2), (2304, 16384, 8, 0),
                                             (5001, 17408, 4, 1)])
def test_gemv_bandwidth_regime(hip_ctx, n, k, bits, method):
    """M = 1 over >= 16 MB of weights: the persistent-grid plan of the decode GEMV -- K > 8192 on the LDS-resident activation row with
    16-wave workgroups and batches drawn from an LDS counter (int4), the register-resident paths for K <= 8192, int8 on 4-wave
    workgroups -- and an 8 MB matrix just below it; 5001 x 17408 is the Qwen3-14B-class down-projection (20 rows per 16-wave workgroup:
    the plan takes two rows per lane group).  Which wave computes a row does not change the row: <= 1 bf16 ulp, >= 99 % bit-identical
    to the CPU restatement, ragged row counts included."""
    rng = np.random.default_rng(n + k + bits)
    q = quant_matrix(rng, n, k, bits, 128, method)
    a = activations(rng, 1, k)
    want, got = oracle_matmul(a, q, 1), hip_matmul(hip_ctx, a, q, 1)
    ulps = ulp_diff_bf16(want, got)
    assert ulps.max() <= 1.0, f"max {ulps.max()} bf16 ulps"
    assert (want == got).mean() >= 0.99


@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("method", [0, 1, 2])
@pytest.mark.parametrize("group_size", [32, 64, 128])
@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 8])
def test_gemv_quant_variants(hip_ctx, bits, method, group_size, m):
    """bits x {ScaleBias, ScaleZeroPoint, ScaleSymmetric} x group x M<=8 (quant_dispatch_test.rs:103-168)."""
    rng = np.random.default_rng(bits * 1000 + method * 100 + group_size + m)
    n, k = 384, 768
    q = quant_matrix(rng, n, k, bits, group_size, method)
    a = activations(rng, m, k)
    want, got = oracle_matmul(a, q, m), hip_matmul(hip_ctx, a, q, m)
    ulps = ulp_diff_bf16(want, got)
    assert ulps.max() <= 1.0, f"max {ulps.max()} bf16 ulps"
    assert (want == got).mean() >= 0.98


@pytest.mark.parametrize("method", [0, 1, 2])
@pytest.mark.parametrize("n,k,m", [(8224, 1024, 16), (1024, 3584, 16), (7168, 1024, 9), (1024, 2048, 2), (1000, 1024, 5), (31040, 1024, 16), (4096, 4096, 13)])
def test_matmul_few_rows_on_the_matrix_cores(hip_ctx, n, k, m, method):
    """2 <= M <= 16 activation rows, int4 g128 (csrc/k_gemv_rows.hip: the weights as the A operand of v_mfma_f32_16x16x32_bf16 straight
    from memory, the rows as its B operand, one quant group per four MFMAs): the Qwen3.5-0.8B linears at the row counts of a speculative
    verify pass, a ragged N, a slice of the read-out, all three quantisation schemes, with the Linear bias.  <= 1 bf16 ulp against the CPU
    kernel, >= 97 % bit-identical (the matrix core sums in another order, like the prefill GEMMs)."""
    rng = np.random.default_rng(n + k + m + method)
    q = quant_matrix(rng, n, k, 4, 128, method)
    a = activations(rng, m, k)
    bias = bf16(rng.uniform(-0.5, 0.5, size=(n,)))
    want, got = oracle_matmul(a, q, m, bias=bias), hip_matmul(hip_ctx, a, q, m, bias=bias)
    ulps = ulp_diff_bf16(want, got)
    assert ulps.max() <= 1.0, f"max {ulps.max()} bf16 ulps"
    assert (want == got).mean() >= 0.97


def test_gemv_epilogue_and_ragged(hip_ctx):
    """ab_scale + accumulate + bias + soft-cap epilogue, N not a multiple of the tile, signed codes, gather."""
    rng = np.random.default_rng(5)
    n, k, m = 1000, 512, 3
    q = quant_matrix(rng, n, k, 4, 64, 2)
    q["signed_codes"] = True
    a = activations(rng, m, k)
    bias = bf16(rng.uniform(-0.5, 0.5, size=(n,)))
    d0 = bf16(rng.uniform(-1, 1, size=(m, n)))
    want = oracle_matmul(a, q, m, bias=bias, ab_scale=0.5, accumulate_d=d0, soft_cap=2.0)
    got = hip_matmul(hip_ctx, a, q, m, bias=bias, ab_scale=0.5, accumulate_d=d0, soft_cap=2.0)
    assert ulp_diff_bf16(want, got).max() <= 1.0
    # gather (embedding.rs encode_readout_sparse): out[r][j] = dense[r][ids[r][j]]
    ids = rng.integers(0, n, size=(m, 17)).astype(np.uint32)
    want = oracle_matmul(a, q, m, gather=ids, n_out=17)
    got = hip_matmul(hip_ctx, a, q, m, gather=ids, n_out=17)
    assert ulp_diff_bf16(want, got).max() <= 1.0


@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("method", [0, 1, 2])
@pytest.mark.parametrize("group_size", [64, 128])
@pytest.mark.parametrize("m", [16, 37, 128, 200])
def test_gemm_mfma_variants(hip_ctx, bits, method, group_size, m):
    """Prefill-sized M goes to the matrix-core kernel (k_gemm.hip): codes exact in bf16, f32 group scaling.
    Ragged M and N (N = 200 is not a multiple of the 128 tile); same tolerance as the GEMV path."""
    rng = np.random.default_rng(bits * 1000 + method * 100 + group_size + m)
    n, k = 200, 512
    q = quant_matrix(rng, n, k, bits, group_size, method)
    a = activations(rng, m, k)
    want, got = oracle_matmul(a, q, m), hip_matmul(hip_ctx, a, q, m)
    ulps = ulp_diff_bf16(want, got)
    assert ulps.max() <= 1.0, f"max {ulps.max()} bf16 ulps"
    assert (want == got).mean() >= 0.97


@pytest.mark.parametrize("n,k", QWEN_SHAPES)
def test_gemm_mfma_qwen_shapes(hip_ctx, n, k):
    """int4 ScaleBias g=128 at M = 64 on the reference's Qwen3.5-0.8B layer table (prefill GEMM)."""
    rng = np.random.default_rng(n * 3 + k)
    m = 64
    q = quant_matrix(rng, n, k, 4, 128, 0)
    a = activations(rng, m, k)
    want, got = oracle_matmul(a, q, m), hip_matmul(hip_ctx, a, q, m)
    ulps = ulp_diff_bf16(want, got)
    assert ulps.max() <= 1.0, f"max {ulps.max()} bf16 ulps"
    assert (want == got).mean() >= 0.97


@pytest.mark.parametrize("bits,method,group_size", [(4, 0, 128), (4, 1, 256), (4, 2, 64), (8, 0, 64), (8, 1, 128), (8, 2, 256)])
@pytest.mark.parametrize("splits", ["1", "2", ""])
def test_gemm_mfma_large_tile(hip_ctx, bits, method, group_size, splits, monkeypatch):
    """M >= 128 takes the 128 x 128 tile kernel (k_gemm128.hip): weights straight from global memory into the MFMA
    operand, the offset term as extra k-steps of bf16 pieces, optional split-K.  Ragged M (300) and N (520)."""
    if splits:
        monkeypatch.setenv("UZU_HIP_TUNE", f"gemm_splits={splits}")
    else:
        monkeypatch.delenv("UZU_HIP_TUNE", raising=False)
    rng = np.random.default_rng(bits * 1000 + method * 100 + group_size)
    n, k, m = 520, 2048, 300
    q = quant_matrix(rng, n, k, bits, group_size, method)
    a = activations(rng, m, k)
    want, got = oracle_matmul(a, q, m), hip_matmul(hip_ctx, a, q, m)
    ulps = ulp_diff_bf16(want, got)
    assert ulps.max() <= 1.0, f"max {ulps.max()} bf16 ulps"
    assert (want == got).mean() >= 0.97


@pytest.mark.parametrize("bits,method,group_size,n,k,m", [(4, 0, 128, 520, 2048, 300), (4, 1, 64, 1160, 1024, 1000), (4, 2, 256, 264, 1024, 131),
                                                          (8, 0, 128, 520, 1024, 300), (8, 1, 64, 392, 512, 257)])
@pytest.mark.parametrize("splits", ["1", "2"])
@pytest.mark.parametrize("form", ["1", "2"])
def test_gemm_512_thread_forms_are_bit_identical_to_the_256_thread_form(hip_ctx, bits, method, group_size, n, k, m, splits, form, monkeypatch):
    """UZU_HIP_TUNE=gemm_form=1 (k_gemm128.hip, ping-pong: a 128 x 256 tile per 512-thread workgroup, convert / MFMA phases half a k-step apart) and
    gemm_form=2 (wave-specialised: four consumer waves on the matrix cores, four producer waves converting / staging).  Both run the
    256-thread form's arithmetic in the same order, so the outputs are equal byte for byte -- ragged M / N, an odd tile count (one half
    idle), split-K -- and hold the oracle tolerance."""
    rng = np.random.default_rng(bits * 1000 + method * 100 + group_size + m)
    q = quant_matrix(rng, n, k, bits, group_size, method)
    a = activations(rng, m, k)
    monkeypatch.setenv("UZU_HIP_TUNE", f"gemm_splits={splits},gemm_form=0")
    base = hip_matmul(hip_ctx, a, q, m)
    monkeypatch.setenv("UZU_HIP_TUNE", f"gemm_splits={splits},gemm_form={form}")
    got = hip_matmul(hip_ctx, a, q, m)
    assert np.array_equal(base, got)
    want = oracle_matmul(a, q, m)
    assert ulp_diff_bf16(want, got).max() <= 1.0
    assert (want == got).mean() >= 0.97


@pytest.mark.parametrize("m", [131, 203, 1023])
def test_gemm_mfma_row_count_not_a_multiple_of_four(hip_ctx, m):
    """The offset-term tables are padded to whole quads of rows (k_gemm128.hip pre-pass); 1023 is the second chunk of the
    benchmark's 2047-token prefill."""
    rng = np.random.default_rng(m)
    n, k = 264, 512
    for method in (0, 1):
        q = quant_matrix(rng, n, k, 4, 128, method)
        a = activations(rng, m, k)
        want, got = oracle_matmul(a, q, m), hip_matmul(hip_ctx, a, q, m)
        ulps = ulp_diff_bf16(want, got)
        assert ulps.max() <= 1.0, f"max {ulps.max()} bf16 ulps"
        assert (want == got).mean() >= 0.97


@pytest.mark.parametrize("splits", ["1", ""])
def test_gemm_mfma_f32_output(hip_ctx, splits, monkeypatch):
    """f32 result buffer (the tensor-parallel row-parallel linears hand f32 partial sums to the all-reduce): the
    large-tile kernel's direct-store epilogue and the split-K reduction, against the oracle rounded to bf16."""
    if splits:
        monkeypatch.setenv("UZU_HIP_TUNE", f"gemm_splits={splits}")
    else:
        monkeypatch.delenv("UZU_HIP_TUNE", raising=False)
    rng = np.random.default_rng(77)
    n, k, m = 520, 1024, 200
    q = quant_matrix(rng, n, k, 4, 128, 0)
    a = activations(rng, m, k)
    want = oracle_matmul(a, q, m)
    kern = B.MatmulKernel.new(hip_ctx, B.BF16, B.BF16, B.F32)
    ba, bw, bs, bb = hip_ctx.buffer_from(a), hip_ctx.buffer_from(q["weights"]), hip_ctx.buffer_from(q["scales"]), hip_ctx.buffer_from(q["biases"])
    bd = hip_ctx.buffer_from(np.zeros((m, n), dtype=np.float32))
    run(hip_ctx, lambda cb: kern.encode(cb, a=ba, b=bw, d=bd, m=m, n=n, k=k, b_kind=B.B_SCALE_BIAS, scales=bs, biases=bb, zero_points=None,
                                         mode=B.QMODE_U4, group_size=128, signed_codes=False, ab_scale=1.0, accumulate=False, bias=None,
                                         soft_cap=None, gather_indices=None))
    got = bf16(bd.download(np.float32, m * n).reshape(m, n))
    ulps = ulp_diff_bf16(want, got)
    assert ulps.max() <= 1.0, f"max {ulps.max()} bf16 ulps"
    assert (want == got).mean() >= 0.97


def test_gemm_mfma_epilogue(hip_ctx):
    """ab_scale + accumulate + bias + soft-cap epilogue and signed codes on the matrix-core path."""
    rng = np.random.default_rng(6)
    n, k, m = 300, 256, 50
    q = quant_matrix(rng, n, k, 4, 64, 2)
    q["signed_codes"] = True
    a = activations(rng, m, k)
    bias = bf16(rng.uniform(-0.5, 0.5, size=(n,)))
    d0 = bf16(rng.uniform(-1, 1, size=(m, n)))
    want = oracle_matmul(a, q, m, bias=bias, ab_scale=0.5, accumulate_d=d0, soft_cap=2.0)
    got = hip_matmul(hip_ctx, a, q, m, bias=bias, ab_scale=0.5, accumulate_d=d0, soft_cap=2.0)
    assert ulp_diff_bf16(want, got).max() <= 1.0
    m = 160  # the 128 x 128 tile kernel, with and without split-K
    a = activations(rng, m, k)
    d0 = bf16(rng.uniform(-1, 1, size=(m, n)))
    want = oracle_matmul(a, q, m, bias=bias, ab_scale=0.5, accumulate_d=d0, soft_cap=2.0)
    got = hip_matmul(hip_ctx, a, q, m, bias=bias, ab_scale=0.5, accumulate_d=d0, soft_cap=2.0)
    assert ulp_diff_bf16(want, got).max() <= 1.0


def test_matmul_reference_order_is_bit_exact(hip_ctx):
    """With the reference-order kernel the GPU reproduces the CPU path bit for bit (any shape, incl. K % 32 != 0)."""
    rng = np.random.default_rng(11)
    _ffi.lib().uzu_hip_set_exact_matmul(1)
    try:
        for (n, k, bits, g, method, m) in [(96, 200, 8, 40, 0, 2), (130, 256, 4, 64, 1, 3), (64, 1024, 4, 128, 2, 1)]:
            q = quant_matrix(rng, n, k, bits, g, method)
            a = activations(rng, m, k)
            want, got = oracle_matmul(a, q, m), hip_matmul(hip_ctx, a, q, m)
            assert np.array_equal(want, got)
    finally:
        _ffi.lib().uzu_hip_set_exact_matmul(0)


def test_matmul_errors(hip_ctx):
    """MatmulError paths become status codes (kernel/matmul/error.rs:9-30)."""
    with pytest.raises(B.UzuHipError) as e:
        B.MatmulKernel.new(hip_ctx, B.F16, B.BF16, B.BF16)
    assert e.value.status == 2
    kern = B.MatmulKernel.new(hip_ctx, B.BF16, B.BF16, B.BF16)
    buf = hip_ctx.create_buffer(1024)
    cb = hip_ctx.create_command_buffer("err").start_encoding()
    with pytest.raises(B.UzuHipError) as e:  # quantized B without scales
        kern.encode(cb, a=buf, b=buf, d=buf, m=1, n=4, k=32, b_kind=B.B_SCALE_BIAS, group_size=32)
    assert e.value.status == 1
    with pytest.raises(B.UzuHipError) as e:  # output-RHT D op: n must be a whole number of 32-wide Hadamard blocks
        kern.encode(cb, a=buf, b=buf, d=buf, m=1, n=4, k=32, b_kind=B.B_SCALE_SYMMETRIC, scales=buf, group_size=32, rht_factors=buf)
    assert e.value.status == 1
    cb.end_encoding().submit().wait_until_completed()
    with pytest.raises(B.UzuHipError) as e:  # typestate: encode after end_encoding
        kern.encode(cb, a=buf, b=buf, d=buf, m=1, n=4, k=32, b_kind=B.B_SCALE_SYMMETRIC, scales=buf, group_size=32)
    assert e.value.status == 5


# ------------------------------------------------------------------------------------------ normalization
@pytest.mark.parametrize("full_layer", [0, 1])
@pytest.mark.parametrize("mode", ["none", "copy", "add"])
@pytest.mark.parametrize("dim,rows", [(1024, 1), (4096, 3), (1000, 5)])
def test_normalization(hip_ctx, full_layer, mode, dim, rows):
    """RMSNorm with the shortcut protocol (normalization_test.rs:87-128; reference tol 1e-2 bf16).
    Ours: shortcut write-back bit-exact, output <= 1 bf16 ulp."""
    rng = np.random.default_rng(dim + rows)
    x = bf16(rng.normal(0, 1.5, size=(rows, dim)))
    sc = bf16(rng.normal(0, 1.5, size=(rows, dim)))
    scales = rng.uniform(-0.2, 0.2, size=(dim,)).astype(np.float32)
    copy, add = mode != "none", mode == "add"
    want, want_sc = np.zeros_like(x), sc.copy()
    args = O.NormArgs(x.ctypes.data, scales.ctypes.data, None, want.ctypes.data, want_sc.ctypes.data if copy else None, O.BF16, O.F32,
                      rows, dim, 1e-6, 1.0, 1.0, 0, full_layer, int(copy), int(add), 0, 0)
    O.lib().orc_normalization(C.byref(args))
    kern = B.NormalizationKernel.new(hip_ctx, B.BF16, B.F32, B.BF16, B.F32, 0, 0, full_layer, int(copy), int(add), 0, 0, 0, 0, 1)
    bx, bs, bo, bsc = hip_ctx.buffer_from(x), hip_ctx.buffer_from(scales), hip_ctx.create_buffer(x.nbytes), hip_ctx.buffer_from(sc)
    run(hip_ctx, lambda cb: kern.encode(bx, bs, None, bo, bsc if copy else None, None, rows, dim, 1e-6, 1.0, 1.0, cb))
    got = bo.download(np.uint16, rows * dim).reshape(rows, dim)
    if copy:
        assert np.array_equal(bsc.download(np.uint16, rows * dim).reshape(rows, dim), want_sc)
    # OnlyNormalization rounds `normalized` to bf16 BEFORE the scale multiply: a 1-ulp-f32 difference that
    # crosses a bf16 rounding boundary becomes 1 bf16 ulp of the factor and up to 2 of the product
    assert ulp_diff_bf16(want, got).max() <= (1.0 if full_layer else 2.0)
    assert (want == got).mean() >= 0.99


@pytest.mark.parametrize("dim", [1024, 2048, 4096, 5120])
@pytest.mark.parametrize("mode,full_layer", [("add", 1), ("add", 0), ("copy", 1), ("none", 1)])
def test_normalization_prefill_rows_kernel_is_bit_identical_to_the_row_at_a_time_kernel(hip_ctx, dim, mode, full_layer):
    """Batches of >= 16 bf16 rows of 1024 NV elements take normalization_rows_kernel (row in registers between the two passes, 8-byte
    accesses); its arithmetic and reduction order are the general kernel's: the same rows normalised one call per row (batch 1: the
    general kernel) give the same bits -- output and shortcut.  Against the oracle: the tolerance of test_normalization."""
    rng = np.random.default_rng(dim + len(mode) + full_layer)
    rows = 37
    x, sc = bf16(rng.normal(0, 1.5, size=(rows, dim))), bf16(rng.normal(0, 1.5, size=(rows, dim)))
    scales = rng.uniform(-0.2, 0.2, size=(dim,)).astype(np.float32)
    copy, add = mode != "none", mode == "add"
    kern = B.NormalizationKernel.new(hip_ctx, B.BF16, B.F32, B.BF16, B.F32, 0, 0, full_layer, int(copy), int(add), 0, 0, 0, 0, 1)
    bs = hip_ctx.buffer_from(scales)
    bx, bo, bsc = hip_ctx.buffer_from(x), hip_ctx.create_buffer(x.nbytes), hip_ctx.buffer_from(sc)
    run(hip_ctx, lambda cb: kern.encode(bx, bs, None, bo, bsc if copy else None, None, rows, dim, 1e-6, 1.0, 1.0, cb))
    got, got_sc = bo.download(np.uint16, rows * dim).reshape(rows, dim), bsc.download(np.uint16, rows * dim).reshape(rows, dim)
    one, one_sc = np.zeros_like(x), np.zeros_like(x)
    for r in range(0, rows, 6):  # a sample of the rows, one launch each
        bx1, bo1, bsc1 = hip_ctx.buffer_from(x[r:r + 1]), hip_ctx.create_buffer(dim * 2), hip_ctx.buffer_from(sc[r:r + 1])
        run(hip_ctx, lambda cb: kern.encode(bx1, bs, None, bo1, bsc1 if copy else None, None, 1, dim, 1e-6, 1.0, 1.0, cb))
        assert np.array_equal(bo1.download(np.uint16, dim), got[r])
        if copy:
            assert np.array_equal(bsc1.download(np.uint16, dim), got_sc[r])
    want, want_sc = np.zeros_like(x), sc.copy()
    args = O.NormArgs(x.ctypes.data, scales.ctypes.data, None, want.ctypes.data, want_sc.ctypes.data if copy else None, O.BF16, O.F32,
                      rows, dim, 1e-6, 1.0, 1.0, 0, full_layer, int(copy), int(add), 0, 0)
    O.lib().orc_normalization(C.byref(args))
    if copy:
        assert np.array_equal(got_sc, want_sc)
    assert ulp_diff_bf16(want, got).max() <= (1.0 if full_layer else 2.0)


def test_qkv_norm(hip_ctx):
    """Per-head RMSNorm in place on packed QKV (qkv_norm_test.rs:52-186)."""
    rng = np.random.default_rng(3)
    batch, nq, nkv, hd = 3, 8, 2, 256
    total = nq + 2 * nkv
    qkv = bf16(rng.normal(0, 2, size=(batch, total * hd)))
    scales = rng.uniform(-0.2, 0.2, size=(hd,)).astype(np.float32)
    want = qkv.copy()
    O.call("orc_qkv_norm", want, O.BF16, scales, batch, total, hd, 1e-6, 1.0, nq, nkv, 1)
    kern = B.QKVNormKernel.new(hip_ctx, B.BF16, B.F32, B.BF16, B.F32, 1, 1)
    b, bs = hip_ctx.buffer_from(qkv), hip_ctx.buffer_from(scales)
    run(hip_ctx, lambda cb: kern.encode(None, bs, b, batch, total, hd, 1e-6, 1.0, nq, nkv, 1, cb))
    got = b.download(np.uint16, qkv.size).reshape(qkv.shape)
    assert np.array_equal(got[:, : nq * hd], qkv[:, : nq * hd])  # heads outside the range untouched
    assert ulp_diff_bf16(want, got).max() <= 1.0


# ------------------------------------------------------------------------------------------ bit-exact element-wise kernels
def test_attention_prepare_bit_exact(hip_ctx):
    """Split + half-rotation RoPE (partial rotary) + KV scatter at kv_token_offset."""
    rng = np.random.default_rng(4)
    batch, nq, nkv, hd, rope_dim, off, cap = 5, 8, 2, 256, 64, 7, 32
    qkv = bf16(rng.normal(0, 1, size=(batch, (nq + 2 * nkv) * hd)))
    cos = rng.uniform(-1, 1, size=(batch, rope_dim)).astype(np.float32)
    sin = rng.uniform(-1, 1, size=(batch, rope_dim)).astype(np.float32)
    wq = np.zeros((nq, batch, hd), np.uint16)
    wk = np.full((cap, nkv * hd), 0x1234, np.uint16)
    wv = wk.copy()
    O.call("orc_attention_prepare", qkv, wq, wk, wv, cos, sin, nq, nkv, hd, rope_dim, off, batch, 1)
    kern = B.AttentionPrepareKernel.new(hip_ctx, B.BF16, B.F32, 1, 1)
    bq, bk, bv = hip_ctx.create_buffer(wq.nbytes), hip_ctx.buffer_from(np.full((cap, nkv * hd), 0x1234, np.uint16)), hip_ctx.buffer_from(np.full((cap, nkv * hd), 0x1234, np.uint16))
    bqkv, bc, bs = hip_ctx.buffer_from(qkv), hip_ctx.buffer_from(cos), hip_ctx.buffer_from(sin)
    run(hip_ctx, lambda cb: kern.encode(bqkv, bq, bk, bv, bc, bs, nq, nkv, hd, rope_dim, off, batch, cb))
    assert np.array_equal(bq.download(np.uint16, wq.size).reshape(wq.shape), wq)
    assert np.array_equal(bk.download(np.uint16, wk.size).reshape(wk.shape), wk)
    assert np.array_equal(bv.download(np.uint16, wv.size).reshape(wv.shape), wv)


def test_gated_act_mul_reference_kat_and_bit_exact(hip_ctx):
    """(a) the one literal known-answer test of the reference tree (gated_act_mul_test.rs:139-160):
    non-interleaved, IDENTITY, gate [1,2,3,4], value rows with offset 2 / stride 6 -> [10,40,150,240];
    (b) SiLU interleaved (the MLP path) bit-exact vs the oracle on random data."""
    gate = np.array([1, 2, 3, 4], np.float32)
    value = np.array([0, 0, 10, 20, 30, 40, 0, 0, 50, 60, 70, 80], np.float32)
    kern = B.GatedActMulKernel.new(hip_ctx, B.F32, 0, 0, 0, 0, 0)
    bg, bv, bo = hip_ctx.buffer_from(gate), hip_ctx.buffer_from(value), hip_ctx.create_buffer(16)
    run(hip_ctx, lambda cb: kern.encode(bg, bv, bo, None, None, None, None, 2, 2, 2, 6, B.IDENTITY, cb))
    assert bo.download(np.float32, 4).tolist() == [10.0, 40.0, 150.0, 240.0]
    rng = np.random.default_rng(6)
    batch, h = 3, 3584
    fused = bf16(rng.normal(0, 2, size=(batch, 2 * h)))
    want = np.zeros((batch, h), np.uint16)
    O.call("orc_gated_act_mul", fused, None, want, O.BF16, h, batch, 0, 0, B.SILU, 1)
    kern = B.GatedActMulKernel.new(hip_ctx, B.BF16, 0, 1, 0, 0, 0)
    bf, bo = hip_ctx.buffer_from(fused), hip_ctx.create_buffer(want.nbytes)
    run(hip_ctx, lambda cb: kern.encode(bf, None, bo, None, None, None, None, h, batch, 0, 0, B.SILU, cb))
    assert np.array_equal(bo.download(np.uint16, want.size).reshape(want.shape), want)


def test_sigmoid_gate_and_tensor_ops_bit_exact(hip_ctx):
    rng = np.random.default_rng(8)
    n = 5000
    g, o = bf16(rng.normal(0, 3, size=n)), bf16(rng.normal(0, 1, size=n))
    want = o.copy()
    O.call("orc_sigmoid_gate", g, want, O.BF16, n)
    kern = B.SigmoidGateKernel.new(hip_ctx, B.BF16)
    bg, bo = hip_ctx.buffer_from(g), hip_ctx.buffer_from(o)
    run(hip_ctx, lambda cb: kern.encode(bg, bo, n, cb))
    assert np.array_equal(bo.download(np.uint16, n), want)
    # TensorAddSwap / TensorAddBias / TensorAddScale / TensorCopy / LogitTransform
    a, b = bf16(rng.normal(0, 1, size=n)), bf16(rng.normal(0, 1, size=n))
    wa, wb = a.copy(), b.copy()
    O.call("orc_tensor_add_swap", wa, wb, O.BF16, n)
    ba, bb = hip_ctx.buffer_from(a), hip_ctx.buffer_from(b)
    k1 = B.TensorAddSwapKernel.new(hip_ctx, B.BF16)
    run(hip_ctx, lambda cb: k1.encode(ba, bb, n, cb))
    assert np.array_equal(ba.download(np.uint16, n), wa) and np.array_equal(bb.download(np.uint16, n), wb)
    cols = 50
    bias = bf16(rng.normal(0, 1, size=cols))
    want = np.zeros(n, np.uint16)
    O.call("orc_tensor_add_bias", a, bias, want, O.BF16, O.BF16, cols, n)
    k2 = B.TensorAddBiasKernel.new(hip_ctx, B.BF16, B.BF16, 0)
    bi, bbias, bo = hip_ctx.buffer_from(a), hip_ctx.buffer_from(bias), hip_ctx.create_buffer(n * 2)
    run(hip_ctx, lambda cb: k2.encode(bi, bbias, bo, cols, n, cb))
    assert np.array_equal(bo.download(np.uint16, n), want)
    O.call("orc_tensor_add_scale", a, bias, want, O.BF16, cols, n, 0.37)
    k3 = B.TensorAddScaleKernel.new(hip_ctx, B.BF16, 0)
    run(hip_ctx, lambda cb: k3.encode(bi, bbias, bo, cols, n, 0.37, cb))
    assert np.array_equal(bo.download(np.uint16, n), want)
    k4 = B.TensorCopyKernel.new(hip_ctx, B.BF16)
    run(hip_ctx, lambda cb: k4.encode(bi, bo, n, cb))
    assert np.array_equal(bo.download(np.uint16, n), a)
    want = a.copy()
    O.call("orc_logit_transform", want, O.BF16, n, 0.6, 0.0, 0)
    k5 = B.LogitTransformKernel.new(hip_ctx, B.BF16, 0)
    bl = hip_ctx.buffer_from(a)
    run(hip_ctx, lambda cb: k5.encode(bl, n, 0.6, 0.0, cb))
    assert np.array_equal(bl.download(np.uint16, n), want)


@pytest.mark.parametrize("bits,method", [(4, 0), (4, 1), (4, 2), (8, 0), (8, 1)])
def test_quantized_embedding_lookup_bit_exact(hip_ctx, bits, method):
    """Row gather with dequant, out-of-range ids -> zeros (quant_embedding_test.rs)."""
    rng = np.random.default_rng(9 + bits + method)
    vocab, dim, g = 300, 256, 64
    q = quant_matrix(rng, vocab, dim, bits, g, method, scale_mag=1.0)
    ids = np.array([0, 299, 17, 300, 5, 4000000000], np.uint32)
    want = np.zeros((ids.size, dim), np.uint16)
    O.call("orc_quantized_embedding_lookup", ids, q["weights"], q["scales"], q["zero_points"], q["biases"], want, O.BF16, ids.size, vocab, dim,
           1.5, g, bits, method)
    kern = B.QuantizedEmbeddingLookupKernel.new(hip_ctx, B.BF16, g, B.QMODE_U4 if bits == 4 else B.QMODE_U8, method, 0)
    bid, bw, bs = hip_ctx.buffer_from(ids), hip_ctx.buffer_from(q["weights"]), hip_ctx.buffer_from(q["scales"])
    bz = hip_ctx.buffer_from(q["zero_points"]) if q["zero_points"] is not None else None
    bb = hip_ctx.buffer_from(q["biases"]) if q["biases"] is not None else None
    bo = hip_ctx.create_buffer(want.nbytes)
    run(hip_ctx, lambda cb: kern.encode(bid, bw, bs, bz, bb, bo, None, ids.size, vocab, dim, 1.5, cb))
    assert np.array_equal(bo.download(np.uint16, want.size).reshape(want.shape), want)


def test_argmax_exact_with_ties(hip_ctx):
    """Greedy UnifiedSampling: ties -> lowest index (unified_sampling.rs:90-95); vocab not a multiple of anything."""
    rng = np.random.default_rng(10)
    vocab, batch = 248320, 3
    logits = bf16(rng.normal(0, 2, size=(batch, vocab)))
    top = bf16(np.array([20.0]))[0]
    logits[0, [100000, 777, 200001]] = top  # three-way tie -> 777
    logits[1, vocab - 1] = top
    logits[2, 0] = top
    want = np.zeros(batch, np.uint32)
    O.call("orc_argmax", logits, O.BF16, want, vocab, batch)
    assert want.tolist() == [777, vocab - 1, 0]
    kern = B.UnifiedSamplingKernel.new(hip_ctx, B.BF16, 0, 0, 0, 0, 0, 0)
    bl, bo = hip_ctx.buffer_from(logits), hip_ctx.create_buffer(batch * 4)
    run(hip_ctx, lambda cb: kern.encode(bl, bo, None, None, None, None, None, None, vocab, batch, cb))
    assert bo.download(np.uint32, batch).tolist() == want.tolist()
    with pytest.raises(B.UzuHipError):  # a stochastic kernel needs its seeds
        stoch = B.UnifiedSamplingKernel.new(hip_ctx, B.BF16, 1, 0, 0, 0, 0, 0)
        run(hip_ctx, lambda cb: stoch.encode(bl, bo, None, None, None, None, None, None, vocab, batch, cb))



# ------------------------------------------------------------------------------------------ UnifiedSampling, every specialisation
def hip_sample(hip_ctx, logits, seeds=None, bitmask=None, temperature=None, top_k=None, top_p=None, min_p=None, graph=False, kern=None):
    batch, vocab = logits.shape
    kern = kern or B.UnifiedSamplingKernel.new(hip_ctx, B.BF16, int(seeds is not None), int(bitmask is not None), int(temperature is not None),
                                               int(top_k is not None), int(top_p is not None), int(min_p is not None))
    bl, bo = hip_ctx.buffer_from(logits), hip_ctx.create_buffer(batch * 4)
    bs = hip_ctx.buffer_from(np.ascontiguousarray(seeds, dtype=np.uint64)) if seeds is not None else None
    bm = hip_ctx.buffer_from(np.ascontiguousarray(bitmask, dtype=np.uint32)) if bitmask is not None else None
    if graph:
        cb = hip_ctx.create_command_buffer("sampling", graph=True).start_encoding()
        try:
            kern.encode(bl, bo, bs, bm, temperature, top_k, top_p, min_p, vocab, batch, cb)
        except B.UzuHipError:
            _ffi.lib().uzu_hip_cmdbuf_destroy(cb._h)  # ends the stream capture the failed encode left open
            cb._h = C.c_void_p()
            raise
        cb.end_encoding().submit().wait_until_completed()
    else:
        run(hip_ctx, lambda cb: kern.encode(bl, bo, bs, bm, temperature, top_k, top_p, min_p, vocab, batch, cb))
    return bo.download(np.uint32, batch)


def test_unified_sampling_committed_fixture(hip_ctx):
    """tests/golden/sampling.json: tokens of every specialisation (stochastic, temperature, top-k, top-p, min-p, all filters +
    grammar bitmask, greedy + bitmask) for fixed logits / seeds, generated by the CPU restatement of unified_sampling.rs:33-98."""
    from golden.make_sampling_golden import CASES, inputs
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sampling.json")))
    logits, seeds, mask = inputs()
    assert gold["vocab"] == logits.shape[1] and gold["batch"] == logits.shape[0]
    for case, want in zip(CASES, gold["cases"]):
        assert case["name"] == want["name"]
        got = hip_sample(hip_ctx, logits, seeds=seeds if case["stochastic"] else None, bitmask=mask if case["mask"] else None,
                         temperature=case["temperature"], top_k=case["top_k"], top_p=case["top_p"], min_p=case["min_p"])
        assert got.tolist() == want["tokens"], f"{case['name']}: hip {got.tolist()} != fixture {want['tokens']}"


@pytest.mark.parametrize("vocab,spread", [(248320, 2.5), (128256, 0.6), (50000, 6.0), (4097, 0.05)])
@pytest.mark.parametrize("setting", [
    dict(), dict(stochastic=True), dict(stochastic=True, temperature=0.8), dict(stochastic=True, top_k=50), dict(stochastic=True, top_k=1),
    dict(stochastic=True, top_p=0.9), dict(stochastic=True, top_p=0.999), dict(stochastic=True, temperature=1.3, min_p=0.02),
    dict(stochastic=True, temperature=0.7, top_k=200, top_p=0.95, min_p=0.001, mask=True), dict(top_k=5, mask=True),
])
def test_unified_sampling_matches_cpu_restatement(hip_ctx, vocab, spread, setting):
    """Real vocabulary sizes (Qwen3.5 248 320, Llama-3 128 256), peaked and nearly flat bf16 logit rows (flat rows have
    thousands of exactly tied logits: the cut of top-k / top-p then falls inside a tie group, which the reference's sort orders by
    index), every filter combination: identical tokens."""
    rng = np.random.default_rng(vocab + len(setting))
    batch = 3
    logits = bf16(rng.normal(0.0, spread, (batch, vocab)))
    seeds = rng.integers(0, 2 ** 63, batch, dtype=np.uint64) if setting.get("stochastic") else None
    mask = rng.integers(0, 2 ** 32, (batch, (vocab + 31) // 32), dtype=np.uint64).astype(np.uint32) if setting.get("mask") else None
    from test_oracle_sampling import sample
    kw = dict(temperature=setting.get("temperature"), top_k=setting.get("top_k"), top_p=setting.get("top_p"), min_p=setting.get("min_p"))
    want = sample(logits, seeds=seeds, bitmask=mask, **kw)
    got = hip_sample(hip_ctx, logits, seeds=seeds, bitmask=mask, **kw)
    assert got.tolist() == want.tolist()


def test_unified_sampling_in_a_graph_captured_command_buffer(hip_ctx):
    """A UZU_CMDBUF_GRAPH command buffer replays the sampling kernels with the kernel-owned scratch block (no stream-ordered
    allocation inside the capture): same tokens as the eager encode; a batch that would need a larger block than the kernel
    owns is refused inside a capture and accepted after one eager encode."""
    rng = np.random.default_rng(5)
    logits = bf16(rng.normal(0.0, 2.0, (4, 30000)))
    seeds = rng.integers(0, 2 ** 63, 4, dtype=np.uint64)
    for kw in (dict(), dict(seeds=seeds, temperature=0.9), dict(seeds=seeds, top_k=20, top_p=0.9)):
        assert hip_sample(hip_ctx, logits, graph=True, **kw).tolist() == hip_sample(hip_ctx, logits, **kw).tolist()
    big = bf16(rng.normal(0.0, 2.0, (80, 2000)))  # 80 rows > the 64 the kernel object is created for
    kern = B.UnifiedSamplingKernel.new(hip_ctx, B.BF16, 0, 0, 1, 0, 0, 0)
    with pytest.raises(B.UzuHipError):
        hip_sample(hip_ctx, big, temperature=0.5, graph=True, kern=kern)
    eager = hip_sample(hip_ctx, big, temperature=0.5, kern=kern)
    assert hip_sample(hip_ctx, big, temperature=0.5, graph=True, kern=kern).tolist() == eager.tolist()
    assert eager.tolist() == [int(np.argmax(f32(r))) for r in big]


def test_kv_cache_update(hip_ctx):
    rng = np.random.default_rng(12)
    rows, dim = 40, 512
    k, v = bf16(rng.normal(size=(rows, dim))), bf16(rng.normal(size=(rows, dim)))
    copies = [(30, 2), (31, 3), (35, 4)]
    wk, wv = k.copy(), v.copy()
    arr = (O.C.c_uint32 * 6)(*[x for c in copies for x in c])
    O.lib().orc_kv_cache_update(O.p(wk), O.p(wv), O.C.c_uint32(O.BF16), arr, O.C.c_uint32(3), O.C.c_uint32(dim))
    kern = B.KVCacheUpdateKernel.new(hip_ctx, B.BF16)
    bk, bv = hip_ctx.buffer_from(k), hip_ctx.buffer_from(v)
    run(hip_ctx, lambda cb: kern.encode(bk, bv, copies, 3, dim, cb))
    assert np.array_equal(bk.download(np.uint16, k.size).reshape(k.shape), wk)
    assert np.array_equal(bv.download(np.uint16, v.size).reshape(v.shape), wv)


# ------------------------------------------------------------------------------------------ attention
def attention_case(rng, heads, kv_heads, hd, seq, suffix, cap):
    q = bf16(rng.normal(0, 1, size=(heads, suffix, hd)))
    k = bf16(rng.normal(0, 1, size=(cap, kv_heads * hd)))
    v = bf16(rng.normal(0, 1, size=(cap, kv_heads * hd)))
    a = O.AttentionArgs(q.ctypes.data, k.ctypes.data, v.ctypes.data, O.BF16, hd, heads // kv_heads, seq, hd, kv_heads * hd, hd, kv_heads * hd,
                        0, 0, 0, 1.0 / np.sqrt(hd), 0, 0, None, heads, suffix, 1)
    return q, k, v, a


@pytest.mark.parametrize("heads,kv_heads,hd", [(8, 2, 256), (32, 8, 128), (4, 4, 64), (10, 2, 128)])
@pytest.mark.parametrize("seq,suffix", [(1, 1), (100, 1), (1024, 1), (300, 7)])
def test_attention_single_pass(hip_ctx, heads, kv_heads, hd, seq, suffix):
    """Reference tolerance 1e-2 (bf16) (attention_single_pass_test.rs:132-136); ours <= 2 bf16 ulps."""
    rng = np.random.default_rng(heads * hd + seq)
    q, k, v, a = attention_case(rng, heads, kv_heads, hd, seq, suffix, seq + 8)
    want = np.zeros((suffix, heads, hd), np.uint16)
    O.lib().orc_attention_single_pass(C.byref(a), O.p(want))
    kern = B.AttentionSinglePassKernel.new(hip_ctx, B.BF16, hd, 0, 0, 1, 0, 0)
    bq, bk, bv, bo = hip_ctx.buffer_from(q), hip_ctx.buffer_from(k), hip_ctx.buffer_from(v), hip_ctx.create_buffer(want.nbytes)
    run(hip_ctx, lambda cb: kern.encode(bq, bk, bv, bo, heads // kv_heads, seq, hd, kv_heads * hd, hd, kv_heads * hd, None, 1.0 / np.sqrt(hd),
                                        None, None, None, heads, suffix, cb))
    got = bo.download(np.uint16, want.size).reshape(want.shape)
    err = np.abs(f32(want) - f32(got))
    assert err.max() <= 1e-2
    assert ulp_diff_bf16(want, got).max() <= 2.0 or err.max() <= 2e-3


@pytest.mark.parametrize("heads,kv_heads,hd", [(8, 2, 256), (32, 8, 128), (4, 4, 64), (4, 2, 64), (16, 2, 128), (10, 2, 128), (6, 2, 64), (12, 2, 128),
                                               (7, 1, 64)])
@pytest.mark.parametrize("seq,suffix", [(16, 16), (64, 64), (300, 64), (1500, 37), (1200, 200)])
def test_attention_prefill_matrix_core_path(hip_ctx, heads, kv_heads, hd, seq, suffix):
    """Prefill-sized suffixes take the flash-attention kernel on the matrix cores (k_attention_mfma.hip): causal mask
    over prefix + suffix, GQA factors 1 / 2 / 3 / 4 / 5 (Qwen3-14B: 40 q / 8 kv heads) / 6 / 7 / 8 -- a workgroup's four waves
    are four consecutive (query tile, head) tasks --, ragged query tiles.  Exact q.k products, probabilities as
    hi + lo bf16 pairs: same tolerance as the VALU kernels (<= 2 bf16 ulps or 2e-3 absolute)."""
    rng = np.random.default_rng(heads * hd + seq + suffix)
    q, k, v, a = attention_case(rng, heads, kv_heads, hd, seq, suffix, seq + 8)
    want = np.zeros((suffix, heads, hd), np.uint16)
    O.lib().orc_attention_single_pass(C.byref(a), O.p(want))
    kern = B.AttentionSinglePassKernel.new(hip_ctx, B.BF16, hd, 0, 0, 1, 0, 0)
    bq, bk, bv, bo = hip_ctx.buffer_from(q), hip_ctx.buffer_from(k), hip_ctx.buffer_from(v), hip_ctx.create_buffer(want.nbytes)
    run(hip_ctx, lambda cb: kern.encode(bq, bk, bv, bo, heads // kv_heads, seq, hd, kv_heads * hd, hd, kv_heads * hd, None, 1.0 / np.sqrt(hd),
                                        None, None, None, heads, suffix, cb))
    got = bo.download(np.uint16, want.size).reshape(want.shape)
    err = np.abs(f32(want) - f32(got))
    assert err.max() <= 1e-2
    assert ulp_diff_bf16(want, got).max() <= 2.0 or err.max() <= 2e-3
    assert (want == got).mean() >= 0.97


@pytest.mark.parametrize("heads,kv_heads,hd,seq,suffix", [(32, 8, 128, 1100, 1024), (40, 8, 128, 700, 640), (16, 4, 128, 800, 768), (24, 8, 64, 1030, 1024),
                                                          (8, 2, 256, 1100, 1024), (8, 2, 256, 2048, 1024)])
def test_attention_prefill_matrix_core_path_full_chunks(hip_ctx, heads, kv_heads, hd, seq, suffix):
    """Chunk-sized suffixes: grids large enough for four wave tasks per workgroup without help (Llama-3-8B: 32 q / 8 kv; Qwen3-14B: 40 / 8;
    GQA factor 3), and the few-head shape of the benchmark model (Qwen3.5-0.8B: 8 q / 2 kv heads of 256) where the keys are split over 4
    workgroups per task group and merged by attention_prefill_merge_kernel -- as in every small case above.  Same bar as there."""
    rng = np.random.default_rng(heads + hd + suffix)
    q, k, v, a = attention_case(rng, heads, kv_heads, hd, seq, suffix, seq + 8)
    want = np.zeros((suffix, heads, hd), np.uint16)
    O.lib().orc_attention_single_pass(C.byref(a), O.p(want))
    kern = B.AttentionSinglePassKernel.new(hip_ctx, B.BF16, hd, 0, 0, 1, 0, 0)
    bq, bk, bv, bo = hip_ctx.buffer_from(q), hip_ctx.buffer_from(k), hip_ctx.buffer_from(v), hip_ctx.create_buffer(want.nbytes)
    run(hip_ctx, lambda cb: kern.encode(bq, bk, bv, bo, heads // kv_heads, seq, hd, kv_heads * hd, hd, kv_heads * hd, None, 1.0 / np.sqrt(hd),
                                        None, None, None, heads, suffix, cb))
    got = bo.download(np.uint16, want.size).reshape(want.shape)
    err = np.abs(f32(want) - f32(got))
    assert err.max() <= 1e-2
    assert ulp_diff_bf16(want, got).max() <= 2.0 or err.max() <= 2e-3
    assert (want == got).mean() >= 0.97


@pytest.mark.parametrize("heads,kv_heads,hd,seq,suffix", [(8, 2, 256, 2048, 1), (32, 8, 128, 1500, 1), (8, 2, 256, 1100, 3), (4, 2, 64, 40, 1)])
def test_attention_two_pass(hip_ctx, heads, kv_heads, hd, seq, suffix):
    """Split-KV: pass 1 partials have the reference's meaning (block b = keys b, b+32, ...), pass 2 is bit-exact
    given identical partials, end-to-end <= 2 bf16 ulps (attention_two_pass_test.rs)."""
    rng = np.random.default_rng(seq + hd)
    q, k, v, a = attention_case(rng, heads, kv_heads, hd, seq, suffix, seq + 8)
    rows = suffix * heads
    wp, ws, wm = np.zeros((rows, 32, hd), np.float32), np.zeros((rows, 32), np.float32), np.zeros((rows, 32), np.float32)
    O.lib().orc_attention_two_pass1(C.byref(a), O.p(wp), O.p(ws), O.p(wm))
    want = np.zeros((suffix, heads, hd), np.uint16)
    O.call("orc_attention_two_pass2", wp, ws, wm, want, O.BF16, hd, heads, suffix)
    k1 = B.AttentionTwoPass1Kernel.new(hip_ctx, B.BF16, hd, 0, 0, 1, 0, 0)
    k2 = B.AttentionTwoPass2Kernel.new(hip_ctx, B.BF16, hd)
    bq, bk, bv = hip_ctx.buffer_from(q), hip_ctx.buffer_from(k), hip_ctx.buffer_from(v)
    bp, bs, bm, bo = hip_ctx.create_buffer(wp.nbytes), hip_ctx.create_buffer(ws.nbytes), hip_ctx.create_buffer(wm.nbytes), hip_ctx.create_buffer(want.nbytes)

    def enc(cb):
        k1.encode(bq, bk, bv, bp, bs, bm, heads // kv_heads, seq, hd, kv_heads * hd, hd, kv_heads * hd, None, 1.0 / np.sqrt(hd), heads, suffix,
                  None, None, None, cb)
        k2.encode(bp, bs, bm, bo, heads, suffix, cb)
    run(hip_ctx, enc)
    gm = bm.download(np.float32, wm.size).reshape(wm.shape)
    gs = bs.download(np.float32, ws.size).reshape(ws.shape)
    np.testing.assert_allclose(gm, wm, rtol=1e-5, atol=1e-5)     # per-block maxima (empty blocks: -1e9 both)
    np.testing.assert_allclose(gs, ws, rtol=1e-4, atol=1e-6)
    got = bo.download(np.uint16, want.size).reshape(want.shape)
    assert np.abs(f32(want) - f32(got)).max() <= 1e-2
    assert ulp_diff_bf16(want, got).max() <= 2.0 or np.abs(f32(want) - f32(got)).max() <= 2e-3
    # pass 2 alone on the ORACLE's partials: bit-exact
    bp2, bs2, bm2 = hip_ctx.buffer_from(wp), hip_ctx.buffer_from(ws), hip_ctx.buffer_from(wm)
    run(hip_ctx, lambda cb: k2.encode(bp2, bs2, bm2, bo, heads, suffix, cb))
    assert np.array_equal(bo.download(np.uint16, want.size).reshape(want.shape), want)


@pytest.mark.parametrize("variant", sorted(MASK_VARIANTS))
@pytest.mark.parametrize("heads,kv_heads,hd,seq,suffix", [(4, 4, 64, 16, 1), (4, 4, 64, 8, 4), (8, 2, 128, 70, 3), (8, 2, 256, 45, 1)])
def test_attention_single_pass_mask_variants(hip_ctx, variant, heads, kv_heads, hd, seq, suffix):
    """mask.rs:3-61 beyond the plain causal mask -- non-causal, sliding window (causal and centred), ring-buffer KV
    positions (full and partially filled), attention sinks -- on the reference test's own inputs and head-major K/V
    strides (attention_single_pass_test.rs:34-131).  Reference tolerance 1e-2; ours <= 2 bf16 ulps or 2e-3."""
    is_causal, window, ring_params, sinks = mask_case(variant, seq, suffix, heads)
    has_sinks = sinks is not None
    q, k, v = ref_attention_inputs(heads, kv_heads, seq, suffix, hd)
    a = O.AttentionArgs(q.ctypes.data, k.ctypes.data, v.ctypes.data, O.BF16, hd, heads // kv_heads, seq, seq * hd, hd, seq * hd, hd,
                        1 if ring_params else 0, ring_params[0] if ring_params else 0, ring_params[1] if ring_params else 0, 1.0 / np.sqrt(hd),
                        1 if window else 0, window or 0, sinks.ctypes.data if has_sinks else None, heads, suffix, is_causal)
    want = np.zeros((suffix, heads, hd), np.uint16)
    O.lib().orc_attention_single_pass(C.byref(a), O.p(want))
    kern = B.AttentionSinglePassKernel.new(hip_ctx, B.BF16, hd, int(has_sinks), int(ring_params is not None), is_causal, 0, int(window is not None))
    bq, bk, bv, bo = hip_ctx.buffer_from(q), hip_ctx.buffer_from(k), hip_ctx.buffer_from(v), hip_ctx.create_buffer(want.nbytes)
    bs = hip_ctx.buffer_from(sinks) if has_sinks else None
    run(hip_ctx, lambda cb: kern.encode(bq, bk, bv, bo, heads // kv_heads, seq, seq * hd, hd, seq * hd, hd, ring_params, 1.0 / np.sqrt(hd),
                                        None, window, bs, heads, suffix, cb))
    got = bo.download(np.uint16, want.size).reshape(want.shape)
    err = np.abs(f32(want) - f32(got))
    assert err.max() <= 1e-2
    assert ulp_diff_bf16(want, got).max() <= 2.0 or err.max() <= 2e-3


@pytest.mark.parametrize("variant", sorted(MASK_VARIANTS))
@pytest.mark.parametrize("heads,kv_heads,hd,seq,suffix", [(8, 2, 256, 1300, 1), (8, 2, 128, 1100, 3), (4, 4, 64, 40, 2)])
def test_attention_two_pass_mask_variants(hip_ctx, variant, heads, kv_heads, hd, seq, suffix):
    """The same mask variants through the split-KV kernels (attention_two_pass.rs:41-190): pass-1 maxima / sums per
    block agree with the reference's blocks (sinks enter block 0 only), merged output <= 2 bf16 ulps or 2e-3."""
    is_causal, window, ring_params, sinks = mask_case(variant, seq, suffix, heads, window_scale=40 if seq > 400 else 1, ring_scale=37)
    has_sinks = sinks is not None
    rng = np.random.default_rng(seq + hd + len(variant))
    q, k, v, _ = attention_case(rng, heads, kv_heads, hd, seq, suffix, seq + 8)
    a = O.AttentionArgs(q.ctypes.data, k.ctypes.data, v.ctypes.data, O.BF16, hd, heads // kv_heads, seq, hd, kv_heads * hd, hd, kv_heads * hd,
                        1 if ring_params else 0, ring_params[0] if ring_params else 0, ring_params[1] if ring_params else 0, 1.0 / np.sqrt(hd),
                        1 if window else 0, window or 0, sinks.ctypes.data if has_sinks else None, heads, suffix, is_causal)
    rows = suffix * heads
    wp, ws, wm = np.zeros((rows, 32, hd), np.float32), np.zeros((rows, 32), np.float32), np.zeros((rows, 32), np.float32)
    O.lib().orc_attention_two_pass1(C.byref(a), O.p(wp), O.p(ws), O.p(wm))
    want = np.zeros((suffix, heads, hd), np.uint16)
    O.call("orc_attention_two_pass2", wp, ws, wm, want, O.BF16, hd, heads, suffix)
    # the single-pass restatement must agree with the two-pass one on the same mask (cross-check of the oracle itself)
    want1 = np.zeros((suffix, heads, hd), np.uint16)
    O.lib().orc_attention_single_pass(C.byref(a), O.p(want1))
    assert np.abs(f32(want) - f32(want1)).max() <= 1e-2
    k1 = B.AttentionTwoPass1Kernel.new(hip_ctx, B.BF16, hd, int(has_sinks), int(ring_params is not None), is_causal, 0, int(window is not None))
    k2 = B.AttentionTwoPass2Kernel.new(hip_ctx, B.BF16, hd)
    bq, bk, bv = hip_ctx.buffer_from(q), hip_ctx.buffer_from(k), hip_ctx.buffer_from(v)
    bp, bsum, bm, bo = hip_ctx.create_buffer(wp.nbytes), hip_ctx.create_buffer(ws.nbytes), hip_ctx.create_buffer(wm.nbytes), hip_ctx.create_buffer(want.nbytes)
    bs = hip_ctx.buffer_from(sinks) if has_sinks else None

    def enc(cb):
        k1.encode(bq, bk, bv, bp, bsum, bm, heads // kv_heads, seq, hd, kv_heads * hd, hd, kv_heads * hd, ring_params, 1.0 / np.sqrt(hd), heads, suffix,
                  None, window, bs, cb)
        k2.encode(bp, bsum, bm, bo, heads, suffix, cb)
    run(hip_ctx, enc)
    np.testing.assert_allclose(bm.download(np.float32, wm.size).reshape(wm.shape), wm, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(bsum.download(np.float32, ws.size).reshape(ws.shape), ws, rtol=1e-4, atol=1e-6)
    got = bo.download(np.uint16, want.size).reshape(want.shape)
    err = np.abs(f32(want) - f32(got))
    assert err.max() <= 1e-2
    assert ulp_diff_bf16(want, got).max() <= 2.0 or err.max() <= 2e-3


TRIE_PARENTS = {
    "chain": [-1, 0, 1, 2, 3],
    "two_branches": [-1, 0, 1, 1, 0, 4],
    "bushy": [-1, 0, 1, 2, 1, 0, 5, 5, 7, -1, 9],
}


@pytest.mark.parametrize("tree", sorted(TRIE_PARENTS))
@pytest.mark.parametrize("window", [None, 6])
@pytest.mark.parametrize("heads,kv_heads,hd,prefix", [(8, 2, 128, 70), (4, 4, 64, 0), (8, 2, 256, 1200)])
def test_attention_trie_mask(hip_ctx, tree, window, heads, kv_heads, hd, prefix):
    """is_trie (mask.rs:21-29, attention_single_pass.rs:55-61): the tokens of a speculated tree as the suffix, through the single-pass and the
    split-KV kernels against the restatement (pinned by tests/test_oracle_kernels.py::test_attention_trie_mask_against_float64)."""
    parents = TRIE_PARENTS[tree]
    suffix = len(parents)
    seq = prefix + suffix
    trie = trie_from_parents(parents)
    rng = np.random.default_rng(prefix + hd + suffix)
    q, k, v, _ = attention_case(rng, heads, kv_heads, hd, seq, suffix, seq + 8)
    scale = 1.0 / np.sqrt(hd)
    w = (window * 40 if prefix > 400 else window) if window else None
    a = O.AttentionArgs(q.ctypes.data, k.ctypes.data, v.ctypes.data, O.BF16, hd, heads // kv_heads, seq, hd, kv_heads * hd, hd, kv_heads * hd,
                        0, 0, 0, scale, 1 if w else 0, w or 0, None, heads, suffix, 1, trie.ctypes.data)
    want = np.zeros((suffix, heads, hd), np.uint16)
    O.lib().orc_attention_single_pass(C.byref(a), O.p(want))
    bq, bk, bv, bt = hip_ctx.buffer_from(q), hip_ctx.buffer_from(k), hip_ctx.buffer_from(v), hip_ctx.buffer_from(trie)
    bo = hip_ctx.create_buffer(want.nbytes)
    kern = B.AttentionSinglePassKernel.new(hip_ctx, B.BF16, hd, 0, 0, 1, 1, int(w is not None))
    with pytest.raises(B.UzuHipError):  # is_trie without the trie buffer
        run(hip_ctx, lambda cb: kern.encode(bq, bk, bv, bo, heads // kv_heads, seq, hd, kv_heads * hd, hd, kv_heads * hd, None, scale, None, w, None, heads, suffix, cb))
    run(hip_ctx, lambda cb: kern.encode(bq, bk, bv, bo, heads // kv_heads, seq, hd, kv_heads * hd, hd, kv_heads * hd, None, scale, bt, w, None, heads, suffix, cb))
    got = bo.download(np.uint16, want.size).reshape(want.shape)
    assert ulp_diff_bf16(want, got).max() <= 2.0 or np.abs(f32(want) - f32(got)).max() <= 2e-3
    # split-KV
    rows = suffix * heads
    k1 = B.AttentionTwoPass1Kernel.new(hip_ctx, B.BF16, hd, 0, 0, 1, 1, int(w is not None))
    k2 = B.AttentionTwoPass2Kernel.new(hip_ctx, B.BF16, hd)
    bp, bsum, bm, bo2 = hip_ctx.create_buffer(rows * 32 * hd * 4), hip_ctx.create_buffer(rows * 32 * 4), hip_ctx.create_buffer(rows * 32 * 4), hip_ctx.create_buffer(want.nbytes)

    def enc(cb):
        k1.encode(bq, bk, bv, bp, bsum, bm, heads // kv_heads, seq, hd, kv_heads * hd, hd, kv_heads * hd, None, scale, heads, suffix, bt, w, None, cb)
        k2.encode(bp, bsum, bm, bo2, heads, suffix, cb)
    run(hip_ctx, enc)
    got2 = bo2.download(np.uint16, want.size).reshape(want.shape)
    assert ulp_diff_bf16(want, got2).max() <= 2.0 or np.abs(f32(want) - f32(got2)).max() <= 2e-3
    # reference-order mode: bit-identical
    set_exact = _ffi.lib().uzu_hip_set_exact
    set_exact.argtypes, set_exact.restype = [C.c_uint32], None
    set_exact(1)
    try:
        run(hip_ctx, lambda cb: kern.encode(bq, bk, bv, bo, heads // kv_heads, seq, hd, kv_heads * hd, hd, kv_heads * hd, None, scale, bt, w, None, heads, suffix, cb))
    finally:
        set_exact(0)
    assert np.array_equal(bo.download(np.uint16, want.size).reshape(want.shape), want)


def test_trie_verify_then_compaction_equals_linear_decode(hip_ctx):
    """The speculative-decoding round trip at the kernel boundary (f4 starter): the K / V rows of a speculated tree sit behind the prefix in
    depth-first order; trie-masked attention gives every node the output a LINEAR sequence ending in its root path would give (same keys, same
    positions prefix + height); after the verifier accepts one root path, KVCacheUpdate moves the accepted rows to prefix + depth
    (kv_cache_update.rs; the copies the reference derives from the accepted trie nodes) and the compacted cache is the cache linear decoding
    would have built: the next token's plain causal attention over it equals the linear one bit for bit."""
    rng = np.random.default_rng(77)
    heads, kv_heads, hd, prefix = 8, 2, 128, 50
    parents = TRIE_PARENTS["bushy"]
    suffix = len(parents)
    trie = trie_from_parents(parents)
    accepted = [0, 5, 7, 8]  # a root path: 0 -> 5 -> 7 -> 8
    cap = prefix + suffix + 4
    q, k, v, _ = attention_case(rng, heads, kv_heads, hd, prefix + suffix, suffix, cap)
    scale = 1.0 / np.sqrt(hd)
    bq, bk, bv, bt = hip_ctx.buffer_from(q), hip_ctx.buffer_from(k), hip_ctx.buffer_from(v), hip_ctx.buffer_from(trie)
    out_tree = hip_ctx.create_buffer(suffix * heads * hd * 2)
    kt = B.AttentionSinglePassKernel.new(hip_ctx, B.BF16, hd, 0, 0, 1, 1, 0)
    run(hip_ctx, lambda cb: kt.encode(bq, bk, bv, out_tree, heads // kv_heads, prefix + suffix, hd, kv_heads * hd, hd, kv_heads * hd, None, scale, bt, None, None, heads, suffix, cb))
    tree_out = out_tree.download(np.uint16, suffix * heads * hd).reshape(suffix, heads, hd)
    # the linear sequence of the accepted path: prefix rows + the accepted nodes' rows, queries of the accepted nodes
    n_acc = len(accepted)
    k_lin, v_lin = k.copy(), v.copy()
    for d, node in enumerate(accepted):
        k_lin[prefix + d], v_lin[prefix + d] = k[prefix + node], v[prefix + node]
    q_lin = np.ascontiguousarray(q[:, accepted, :])
    bql, bkl, bvl = hip_ctx.buffer_from(q_lin), hip_ctx.buffer_from(k_lin), hip_ctx.buffer_from(v_lin)
    out_lin = hip_ctx.create_buffer(n_acc * heads * hd * 2)
    kl = B.AttentionSinglePassKernel.new(hip_ctx, B.BF16, hd, 0, 0, 1, 0, 0)
    run(hip_ctx, lambda cb: kl.encode(bql, bkl, bvl, out_lin, heads // kv_heads, prefix + n_acc, hd, kv_heads * hd, hd, kv_heads * hd, None, scale, None, None, None, heads, n_acc, cb))
    lin_out = out_lin.download(np.uint16, n_acc * heads * hd).reshape(n_acc, heads, hd)
    # same keys and values in a different visiting order: the online softmax may differ in the last bit
    assert ulp_diff_bf16(lin_out, tree_out[accepted]).max() <= 1.0
    # compaction: accepted node at suffix index i -> row prefix + depth (skipping rows already in place)
    copies = [(prefix + node, prefix + d) for d, node in enumerate(accepted) if node != d]
    ku = B.KVCacheUpdateKernel.new(hip_ctx, B.BF16)
    run(hip_ctx, lambda cb: ku.encode(bk, bv, copies, len(copies), kv_heads * hd, cb))
    rows = (prefix + n_acc) * kv_heads * hd
    assert np.array_equal(bk.download(np.uint16, rows), k_lin.reshape(-1)[:rows])
    assert np.array_equal(bv.download(np.uint16, rows), v_lin.reshape(-1)[:rows])
    # the next token over the compacted cache == over the linear cache, bit for bit
    qn = bf16(rng.normal(size=(heads, 1, hd)))
    bqn = hip_ctx.buffer_from(qn)
    o1, o2 = hip_ctx.create_buffer(heads * hd * 2), hip_ctx.create_buffer(heads * hd * 2)

    def enc(cb):
        kl.encode(bqn, bk, bv, o1, heads // kv_heads, prefix + n_acc, hd, kv_heads * hd, hd, kv_heads * hd, None, scale, None, None, None, heads, 1, cb)
        kl.encode(bqn, bkl, bvl, o2, heads // kv_heads, prefix + n_acc, hd, kv_heads * hd, hd, kv_heads * hd, None, scale, None, None, None, heads, 1, cb)
    run(hip_ctx, enc)
    assert np.array_equal(o1.download(np.uint16, heads * hd), o2.download(np.uint16, heads * hd))


# ------------------------------------------------------------------------------------------ gated delta net
def test_delta_net_conv_update_bit_exact(hip_ctx):
    rng = np.random.default_rng(13)
    conv_dim, ks = 6144, 4
    w = rng.uniform(-0.6, 0.6, size=(conv_dim, ks)).astype(np.float32)
    bias = rng.uniform(-0.1, 0.1, size=(conv_dim,)).astype(np.float32)
    x = bf16(rng.normal(0, 1, size=(conv_dim + 40,)))
    st = rng.normal(0, 1, size=(conv_dim, ks - 1)).astype(np.float32)
    wx, wst = x.copy(), st.copy()
    O.call("orc_delta_net_conv_update", w, bias, wx, wst, ks, conv_dim, ks - 1)
    kern = B.DeltaNetConvUpdateKernel.new(hip_ctx, B.BF16, 1)
    bw, bb, bx, bs = hip_ctx.buffer_from(w), hip_ctx.buffer_from(bias), hip_ctx.buffer_from(x), hip_ctx.buffer_from(st)
    run(hip_ctx, lambda cb: kern.encode(bw, bb, bx, bs, ks, conv_dim, ks - 1, cb))
    assert np.array_equal(bx.download(np.uint16, x.size), wx)
    assert np.array_equal(bs.download(np.float32, st.size).reshape(st.shape), wst)


def test_delta_net_update(hip_ctx):
    """Delta-rule decode step, Qwen3.5 shape (Hv = Hk = 16, Dk = Dv = 128); gdn/delta_net_test.rs compares
    CPU vs Metal with 1e-2-class tolerances.  Ours: state <= 1e-5 abs, output <= 2 bf16 ulps."""
    rng = np.random.default_rng(14)
    Hv, Hk, Dk, Dv = 16, 16, 128, 128
    key_dim, value_dim = Hk * Dk, Hv * Dv
    total = 2 * key_dim + 2 * value_dim + 2 * Hv
    in_proj = bf16(rng.normal(0, 1, size=(total,)))
    a_log, dt_bias = rng.uniform(-1, 1, Hv).astype(np.float32), rng.uniform(-1, 1, Hv).astype(np.float32)
    nw = (1 + rng.uniform(-0.1, 0.1, Dv)).astype(np.float32)
    state = rng.normal(0, 0.3, size=(Hv, Dv, Dk)).astype(np.float32)
    wst, want = state.copy(), np.zeros(value_dim, np.uint16)
    O.call("orc_delta_net_update", in_proj, a_log, dt_bias, nw, wst, want, Hv, Hk, Dk, Dv, key_dim, value_dim, 1e-6)
    kern = B.DeltaNetUpdateKernel.new(hip_ctx, B.BF16, 128)
    bi, ba, bd, bn, bs, bo = (hip_ctx.buffer_from(z) for z in (in_proj, a_log, dt_bias, nw, state, want))
    run(hip_ctx, lambda cb: kern.encode(bi, ba, bd, bn, bs, bo, Hv, Hk, Dv, key_dim, value_dim, 1e-6, cb))
    gst = bs.download(np.float32, state.size).reshape(state.shape)
    np.testing.assert_allclose(gst, wst, rtol=1e-5, atol=1e-5)
    got = bo.download(np.uint16, value_dim)
    assert ulp_diff_bf16(want, got).max() <= 2.0


@pytest.mark.parametrize("T", [37, 64, 200, 513, 1000, 2043])
def test_delta_net_prefill_path(hip_ctx, T):
    """conv_pack -> conv_scan -> prefill_prep -> prefill -> norm_gate against the oracle (GQA-style Hv = 2 Hk).
    T = 37 runs the one/two-token recurrence, T >= 64 the chunked form (32-token chunks, ragged last chunk), from 16 chunks (T = 513, 1000, 2043) as
    two concurrent segments + the S_mid fix-up (k_deltanet_chunk.hip: ScanSplit; T = 513: 9 + 8 chunks, the second segment ends on a 1-token chunk)."""
    rng = np.random.default_rng(15)
    Hv, Hk, Dk, Dv, ks = 4, 2, 128, 128, 4
    key_dim, value_dim = Hk * Dk, Hv * Dv
    conv_dim = 2 * key_dim + value_dim
    total = conv_dim + value_dim + 2 * Hv
    in_proj = bf16(rng.normal(0, 1, size=(T, total)))
    conv_w = rng.uniform(-0.6, 0.6, size=(conv_dim, ks)).astype(np.float32)
    conv_state = rng.normal(0, 1, size=(conv_dim, ks - 1)).astype(np.float32)
    a_log, dt_bias = rng.uniform(-1, 1, Hv).astype(np.float32), rng.uniform(-1, 1, Hv).astype(np.float32)
    nw = (1 + rng.uniform(-0.1, 0.1, Dv)).astype(np.float32)
    state = rng.normal(0, 0.3, size=(Hv, Dv, Dk)).astype(np.float32)
    # oracle
    w_in, w_cs, w_st = in_proj.copy(), conv_state.copy(), state.copy()
    padded = np.zeros((T + ks - 1, total), np.float32)
    O.call("orc_conv1d_pack", w_cs, w_in, padded, ks - 1, total, T, conv_dim)
    O.call("orc_delta_net_conv_scan", padded, conv_w, None, w_in, w_cs, T, ks, total, ks - 1, conv_dim, total)
    qn, kn = np.zeros((T, key_dim), np.float32), np.zeros((T, key_dim), np.float32)
    beta, decay = np.zeros((T, Hv), np.float32), np.zeros((T, Hv), np.float32)
    O.call("orc_delta_net_prefill_prep", w_in, a_log, dt_bias, qn, kn, beta, decay, Hv, Hk, Dk, key_dim, value_dim, T)
    w_out = np.zeros((T, value_dim), np.uint16)
    O.call("orc_delta_net_prefill", qn, kn, beta, decay, w_in, w_st, w_out, Hv, Hk, Dk, Dv, key_dim, value_dim, T)
    O.call("orc_delta_net_norm_gate", w_out, w_in, nw, Hv, Dv, value_dim, conv_dim, total, 1e-6, T)
    # hip
    k_pack = B.Conv1dPackKernel.new(hip_ctx, B.F32, B.BF16)
    k_scan = B.DeltaNetConvScanKernel.new(hip_ctx, B.BF16, 0)
    k_prep = B.DeltaNetPrefillPrepKernel.new(hip_ctx, B.BF16, B.F32, 128, 0, 0)
    k_pre = B.DeltaNetPrefillKernel.new(hip_ctx, B.BF16, 128)
    k_ng = B.DeltaNetNormGateKernel.new(hip_ctx, B.BF16)
    b_in, b_cw, b_cs, b_al, b_dt, b_nw, b_st = (hip_ctx.buffer_from(z) for z in (in_proj, conv_w, conv_state, a_log, dt_bias, nw, state))
    b_pad = hip_ctx.create_buffer(padded.nbytes)
    b_qn, b_kn, b_be, b_de = (hip_ctx.create_buffer(z.nbytes) for z in (qn, kn, beta, decay))
    b_out = hip_ctx.create_buffer(w_out.nbytes)

    def enc(cb):
        k_pack.encode(b_cs, b_in, b_pad, ks - 1, total, T, conv_dim, cb)
        k_scan.encode(b_pad, b_cw, None, b_in, b_cs, T, ks, total, ks - 1, conv_dim, total, cb)
        k_prep.encode(b_in, b_al, b_dt, b_qn, b_kn, None, b_be, b_de, Hv, Hk, key_dim, value_dim, T, cb)
        k_pre.encode(b_qn, b_kn, b_be, b_de, b_in, b_st, b_out, Hv, Hk, Dv, key_dim, value_dim, T, (Dv + 15) // 16, cb)
        k_ng.encode(b_out, b_in, b_nw, Hv, Dv, value_dim, conv_dim, total, 1e-6, T, cb)
    run(hip_ctx, enc)
    assert np.array_equal(b_in.download(np.uint16, in_proj.size).reshape(in_proj.shape), w_in)      # conv + SiLU: bit-exact
    assert np.array_equal(b_cs.download(np.float32, conv_state.size).reshape(conv_state.shape), w_cs)
    np.testing.assert_allclose(b_qn.download(np.float32, qn.size).reshape(qn.shape), qn, rtol=2e-6, atol=1e-7)
    np.testing.assert_array_equal(b_be.download(np.float32, beta.size).reshape(beta.shape), beta)   # exp/log: glibc-exact
    np.testing.assert_array_equal(b_de.download(np.float32, decay.size).reshape(decay.shape), decay)
    np.testing.assert_allclose(b_st.download(np.float32, state.size).reshape(state.shape), w_st, rtol=1e-4, atol=1e-4)
    got = b_out.download(np.uint16, w_out.size).reshape(w_out.shape)
    # the chunked form sums in another order than the one-token recurrence (f32, ~1e-6 relative): after the bf16 rounding of the
    # output every element is within 2e-2 absolute or 1 bf16 ulp (|x| >= 4 has an ulp of 0.031), 99 % within 2 ulps
    err, ulps = np.abs(f32(got) - f32(w_out)), ulp_diff_bf16(w_out, got)
    assert ((err <= 2e-2) | (ulps <= 1)).all() and (ulps <= 2).mean() > 0.99


# ------------------------------------------------------------------------------------------ command buffer
def test_command_buffer_typestate_copy_fill_graph(hip_ctx):
    """CommandBuffer states, encode_copy / encode_fill, gpu_execution_time, and graph replay (re-submit)."""
    a = np.arange(1024, dtype=np.uint8)
    ba, bb = hip_ctx.buffer_from(a), hip_ctx.create_buffer(1024)
    cb = hip_ctx.create_command_buffer("typestate")
    with pytest.raises(B.UzuHipError):
        cb.encode_fill(bb, 16, 1)  # not encoding yet
    cb.start_encoding()
    cb.push_debug_group("copy+fill")
    cb.encode_copy((ba, 100), (bb, 0), 200)
    cb.encode_fill((bb, 200), 824, 0xAB)
    cb.pop_debug_group()
    cb.encode_barrier()
    with pytest.raises(B.UzuHipError):
        cb.submit()  # not executable yet
    cb.end_encoding().submit().wait_until_completed()
    assert cb.gpu_execution_time() >= 0.0
    got = bb.download(np.uint8, 1024)
    assert np.array_equal(got[:200], a[100:300]) and (got[200:] == 0xAB).all()
    # graph command buffer: encoded once, submitted three times
    counter = hip_ctx.buffer_from(np.zeros(4, np.float32))
    one = hip_ctx.buffer_from(np.ones(4, np.float32))
    k = B.TensorAddBiasKernel.new(hip_ctx, B.F32, B.F32, 1)
    g = hip_ctx.create_command_buffer("graph", graph=True).start_encoding()
    k.encode(None, one, counter, 4, 4, g)
    g.end_encoding()
    for _ in range(3):
        g.submit().wait_until_completed()
    assert counter.download(np.float32, 4).tolist() == [3.0] * 4
    assert hip_ctx.peak_memory_usage() > 0 and "gfx950" in hip_ctx.device_name()


# ------------------------------------------------------------------------------------------ boundary: sparse buffers, capture, capabilities
def test_sparse_buffer_map_unmap_and_kernels_on_mapped_pages(hip_ctx):
    """Context::create_sparse_buffer + SparseBuffer::{map, unmap, page_size_bytes} (context.rs:31-34, buffer/sparse.rs:5-19) the
    way the reference's KV cache uses them (mixer/attention/state.rs:144-170): address space first, pages as the context grows;
    a kernel (KVCacheUpdate row copy) runs on the mapped part; data survives mapping further pages; unmapped ranges are refused."""
    assert hip_ctx.device_capabilities() & B.DEVICE_CAP_SPARSE_BUFFERS, "MI355X exposes the HIP virtual-memory API"
    sb = hip_ctx.create_sparse_buffer(10 * (1 << 20) + 5)
    page = sb.page_size_bytes()
    assert page >= 4096 and (page & (page - 1)) == 0 and sb.size() % page == 0 and sb.size() >= 10 * (1 << 20) + 5
    assert sb.total_pages() == sb.size() // page
    gpu_ptr = sb.gpu_ptr()
    before = hip_ctx.peak_memory_usage()
    sb.map(0, 1)
    sb.map(0, 1)  # idempotent
    rows = np.arange(4 * 64, dtype=np.uint16).reshape(4, 64)
    sb.upload(rows)
    other = hip_ctx.buffer_from(rows)
    kern = B.KVCacheUpdateKernel.new(hip_ctx, B.BF16)
    run(hip_ctx, lambda cb: kern.encode(sb, other, [(0, 3), (1, 2)], 2, 64, cb))
    got = sb.download(np.uint16, rows.size).reshape(rows.shape)
    assert np.array_equal(got[3], rows[0]) and np.array_equal(got[2], rows[1]) and np.array_equal(got[:2], rows[:2])
    if sb.total_pages() > 1:
        with pytest.raises(B.UzuHipError):
            sb.download(np.uint16, 16, offset=page)  # second page not mapped yet
        sb.map(1, sb.total_pages())
        sb.upload(rows, offset=page)
        assert np.array_equal(sb.download(np.uint16, rows.size, offset=page).reshape(rows.shape), rows)
        assert np.array_equal(sb.download(np.uint16, rows.size).reshape(rows.shape), got)  # first page untouched
        assert hip_ctx.peak_memory_usage() >= before  # mapped pages are accounted like any allocation
        sb.unmap(1, sb.total_pages())
        with pytest.raises(B.UzuHipError):
            sb.upload(rows, offset=page)
    assert sb.gpu_ptr() == gpu_ptr
    with pytest.raises(B.UzuHipError):
        sb.map(0, sb.total_pages() + 1)
    with pytest.raises(B.UzuHipError):
        call_cpu_ptr = C.c_void_p()
        B.call("uzu_hip_buffer_cpu_ptr", sb._h, C.byref(call_cpu_ptr))


def test_capture_writes_command_buffer_records(hip_ctx, tmp_path):
    """Context::{enable_capture, start_capture, stop_capture} (context.rs:38-45): the trace file lists every command buffer
    completed in between with its debug groups and GPU time; misuse is UZU_ERR_STATE."""
    B.Context.enable_capture()
    with pytest.raises(B.UzuHipError):
        hip_ctx.stop_capture()
    path = tmp_path / "trace.json"
    hip_ctx.start_capture(str(path))
    with pytest.raises(B.UzuHipError):
        hip_ctx.start_capture(str(path))
    a, b = hip_ctx.buffer_from(np.arange(1 << 16, dtype=np.uint8)), hip_ctx.create_buffer(1 << 16)
    for name in ("first", "second"):
        cb = hip_ctx.create_command_buffer(name).start_encoding()
        cb.push_debug_group("copies")
        cb.encode_copy(a, b, 1 << 16)
        cb.pop_debug_group()
        cb.end_encoding().submit().wait_until_completed()
    hip_ctx.stop_capture()
    trace = json.load(open(path))
    assert [r["name"] for r in trace["command_buffers"]] == ["first", "second"]
    assert all(r["debug_groups"] == ["copies"] and r["gpu_time_ns"] > 0 for r in trace["command_buffers"])
    assert "gfx950" in trace["device"]


@pytest.mark.parametrize("heads,kv_heads,hd,prefix,suffix", [(8, 2, 256, 700, 300), (32, 8, 128, 0, 128), (4, 4, 64, 33, 5)])
def test_attention_gemm_core(hip_ctx, heads, kv_heads, hd, prefix, suffix):
    """AttentionGemmCore (attention_gemm/kernel.rs:8-24): is_supported answers the static question AttentionCores::new asks
    (core/mod.rs:53-61); encode on a Full KV cache equals the CPU single-pass kernel with suffix_length = M (the
    CPU backend has no GEMM core, its prefill IS the single-pass kernel)."""
    args = B.AttentionCoreArguments(hd, kv_heads, heads, 0, 0, 1, 0, 0, 0, 0, 0.0, B.BF16)
    assert B.AttentionGemmCore.is_supported(hip_ctx, args)
    for bad in (dict(is_causal=0), dict(is_trie=1), dict(is_kv_cache_ring=1), dict(has_sliding_window=1, sliding_window_size=0), dict(head_dim=96)):
        a2 = B.AttentionCoreArguments(hd, kv_heads, heads, 0, 0, 1, 0, 0, 0, 0, 0.0, B.BF16)
        for k_, v_ in bad.items():
            setattr(a2, k_, v_)
        assert not B.AttentionGemmCore.is_supported(hip_ctx, a2)
        with pytest.raises(B.UzuHipError):
            B.AttentionGemmCore.new(hip_ctx, a2)
    rng = np.random.default_rng(hd + suffix)
    seq = prefix + suffix
    q, k, v, a = attention_case(rng, heads, kv_heads, hd, seq, suffix, seq + 8)
    want = np.zeros((suffix, heads, hd), np.uint16)
    O.lib().orc_attention_single_pass(C.byref(a), O.p(want))
    core = B.AttentionGemmCore.new(hip_ctx, args)
    bq, bk, bv, bo = hip_ctx.buffer_from(q), hip_ctx.buffer_from(k), hip_ctx.buffer_from(v), hip_ctx.create_buffer(want.nbytes)
    run(hip_ctx, lambda cb: core.encode(bq, bk, bv, bo, prefix, suffix, cb))
    got = bo.download(np.uint16, want.size).reshape(want.shape)
    err = np.abs(f32(want) - f32(got))
    assert err.max() <= 1e-2
    assert ulp_diff_bf16(want, got).max() <= 2.0 or err.max() <= 2e-3


@pytest.mark.parametrize("variant", ["sliding_causal", "ring_full", "ring_partial", "ring_sliding", "sinks", "sinks_sliding"])
@pytest.mark.parametrize("heads,kv_heads,hd,prefix,suffix", [(8, 2, 256, 700, 300), (32, 8, 128, 96, 160), (4, 4, 64, 333, 70), (8, 2, 128, 0, 64)])
def test_attention_gemm_core_sliding_window_ring_and_sinks(hip_ctx, variant, heads, kv_heads, hd, prefix, suffix):
    """The prefill core with the masks attention_gemm.metal takes besides the causal one (AttnParams, gpu_types/attention.rs:12-36): a
    sliding window, a ring KV prefix (AttentionStateType::Ring: slot i holds position (prefix + i - offset) mod prefix, live below
    ring_length), attention sinks -- on the matrix cores, against the CPU single-pass kernel with the same arguments (mask.rs:3-61).
    A ring comes with its window (state.rs:69-136), so the ring-only variants run with a window as wide as the ring."""
    is_causal, window, ring_params, sinks = mask_case(variant, prefix + suffix, suffix, heads, window_scale=16, ring_scale=37)
    if prefix == 0:
        ring_params = None if variant.startswith("ring") else ring_params
        if variant.startswith("ring"):
            pytest.skip("an empty ring prefix is the plain causal case")
    if ring_params is not None and window is None:
        window = prefix
    rng = np.random.default_rng(hd + suffix + len(variant))
    seq = prefix + suffix
    q, k, v, _ = attention_case(rng, heads, kv_heads, hd, seq, suffix, seq + 8)
    a = O.AttentionArgs(q.ctypes.data, k.ctypes.data, v.ctypes.data, O.BF16, hd, heads // kv_heads, seq, hd, kv_heads * hd, hd, kv_heads * hd,
                        1 if ring_params else 0, ring_params[0] if ring_params else 0, ring_params[1] if ring_params else 0, 1.0 / np.sqrt(hd),
                        1 if window else 0, window or 0, sinks.ctypes.data if sinks is not None else None, heads, suffix, 1)
    want = np.zeros((suffix, heads, hd), np.uint16)
    O.lib().orc_attention_single_pass(C.byref(a), O.p(want))
    args = B.AttentionCoreArguments(hd, kv_heads, heads, int(sinks is not None), int(ring_params is not None), 1, 0, int(window is not None), window or 0, 0, 0.0, B.BF16)
    assert B.AttentionGemmCore.is_supported(hip_ctx, args)
    core = B.AttentionGemmCore.new(hip_ctx, args)
    bq, bk, bv, bo = hip_ctx.buffer_from(q), hip_ctx.buffer_from(k), hip_ctx.buffer_from(v), hip_ctx.create_buffer(want.nbytes)
    bs = hip_ctx.buffer_from(sinks) if sinks is not None else None
    state = ("ring", ring_params[0], ring_params[1], prefix) if ring_params else ("full", prefix)
    run(hip_ctx, lambda cb: core.encode_state(bq, bk, bv, bs, bo, state, suffix, cb))
    got = bo.download(np.uint16, want.size).reshape(want.shape)
    err = np.abs(f32(want) - f32(got))
    assert err.max() <= 1e-2
    assert ulp_diff_bf16(want, got).max() <= 2.0 or err.max() <= 2e-3


# ------------------------------------------------------------------------------------------ RHT / A8
@pytest.mark.parametrize("op", [B.ATX_INPUT_RHT, B.ATX_OUTPUT_RHT])
@pytest.mark.parametrize("rows,cols,in_place", [(1, 1024, True), (37, 3584, False), (5, 96, True)])
def test_activation_transform_rht_bit_exact(hip_ctx, op, rows, cols, in_place):
    """ActivationTransform InputRht / OutputRht (activation_transform.rs:43-136): every addition of the 32-point butterfly is the
    reference's own, so the result is bit-identical to the CPU kernel; in place and out of place."""
    rng = np.random.default_rng(rows + cols + op)
    x = bf16(rng.normal(0, 1.5, (rows, cols)))
    factors = rng.choice(np.array([-1, 1], np.int32), cols)
    want = x.copy() if in_place else np.zeros_like(x)
    O.lib().orc_activation_transform(None if in_place else O.p(x), O.p(want), None, None, None, O.p(factors), O.BF16, rows, cols, op, 0, 0)
    kern = B.ActivationTransformKernel.new(hip_ctx, B.BF16, op, int(in_place), 0, 0)
    bx, bf_ = hip_ctx.buffer_from(x), hip_ctx.buffer_from(factors)
    bo = bx if in_place else hip_ctx.create_buffer(x.nbytes)
    run(hip_ctx, lambda cb: kern.encode(None if in_place else bx, bo, None, None, None, bf_, rows, cols, cb))
    assert np.array_equal(bo.download(np.uint16, x.size).reshape(x.shape), want)


@pytest.mark.parametrize("scale_group,sum_group", [(32, 32), (64, 128), (128, 64), (128, 128)])
def test_activation_transform_quantize_bit_exact(hip_ctx, scale_group, sum_group):
    """Quantize / QuantizeWithGroupSums: int8 codes, f32 divisors and i32 group sums identical to the CPU kernel (the divisor is one
    IEEE division of the exact group maximum; codes are round-half-away of one IEEE division)."""
    rng = np.random.default_rng(scale_group * 7 + sum_group)
    rows, cols = 19, 2048
    x = bf16(rng.normal(0, 2.0, (rows, cols)))
    x[3, :256] = 0
    factors = rng.choice(np.array([-1, 1], np.int32), cols)
    for op in (B.ATX_QUANTIZE, B.ATX_QUANTIZE_WITH_GROUP_SUMS):
        wq, wsc, wgs = np.zeros((rows, cols), np.int8), np.zeros((rows, cols // scale_group), np.float32), np.zeros((rows, cols // sum_group), np.int32)
        O.lib().orc_activation_transform(O.p(x), None, O.p(wq), O.p(wsc), O.p(wgs), O.p(factors), O.BF16, rows, cols, op, scale_group, sum_group)
        kern = B.ActivationTransformKernel.new(hip_ctx, B.BF16, op, 0, scale_group, sum_group)
        bx, bf_ = hip_ctx.buffer_from(x), hip_ctx.buffer_from(factors)
        bq, bs, bg = hip_ctx.create_buffer(wq.nbytes), hip_ctx.create_buffer(wsc.nbytes), hip_ctx.create_buffer(wgs.nbytes)
        sums = bg if op == B.ATX_QUANTIZE_WITH_GROUP_SUMS else None
        run(hip_ctx, lambda cb: kern.encode(bx, None, bq, bs, sums, bf_, rows, cols, cb))
        assert np.array_equal(bs.download(np.float32, wsc.size).reshape(wsc.shape), wsc)
        assert np.array_equal(bq.download(np.int8, wq.size).reshape(wq.shape), wq)
        if sums is not None:
            assert np.array_equal(bg.download(np.int32, wgs.size).reshape(wgs.shape), wgs)


@pytest.mark.parametrize("ops,scale_group,sum_group", [(0, 0, 0), (1, 64, 0), (2, 128, 32), (2, 32, 256)])
@pytest.mark.parametrize("interleaved", [0, 1])
def test_gated_act_mul_rht_variants_bit_exact(hip_ctx, ops, scale_group, sum_group, interleaved):
    """GatedActMul with use_hadamard (gated_act_mul.rs:47-118): gated products rounded to bf16, sign factors, 32-point butterflies, then
    bf16 (FullPrecision) or int8 codes + f32 divisors (+ i32 group sums): identical to the CPU restatement, SiLU and GELU."""
    rng = np.random.default_rng(ops * 10 + scale_group + interleaved)
    rows, dim = 7, 1536
    for act in (0, 1):
        if interleaved:
            act_op, val_op, voff, vstride = bf16(rng.normal(0, 1.5, (rows, 2 * dim))), None, 0, 0
        else:
            act_op, val_op, voff, vstride = bf16(rng.normal(0, 1.5, (rows, dim))), bf16(rng.normal(0, 1.5, (rows, dim + 64))), 64, dim + 64
        factors = rng.choice(np.array([-1, 1], np.int32), dim)
        wf = np.zeros((rows, dim), np.uint16)
        wq, wsc = np.zeros((rows, dim), np.int8), np.zeros((rows, dim // max(scale_group, 1)), np.float32)
        wgs = np.zeros((rows, dim // max(sum_group, 1)), np.int32)
        O.lib().orc_gated_act_mul_rht(O.p(act_op), O.p(val_op) if val_op is not None else None, O.p(wf) if ops == 0 else None, O.p(wq) if ops else None,
                                      O.p(wsc) if ops else None, O.p(wgs) if ops == 2 else None, O.p(factors), O.BF16, dim, rows, voff, vstride, act,
                                      interleaved, ops, scale_group, sum_group)
        kern = B.GatedActMulKernel.new(hip_ctx, B.BF16, ops, interleaved, 1, scale_group, sum_group)
        ba, bv, bfac = hip_ctx.buffer_from(act_op), hip_ctx.buffer_from(val_op) if val_op is not None else None, hip_ctx.buffer_from(factors)
        bo = hip_ctx.create_buffer(wf.nbytes) if ops == 0 else None
        bq, bs = (hip_ctx.create_buffer(wq.nbytes), hip_ctx.create_buffer(wsc.nbytes)) if ops else (None, None)
        bg = hip_ctx.create_buffer(wgs.nbytes) if ops == 2 else None
        run(hip_ctx, lambda cb: kern.encode(ba, bv, bo, bq, bs, bg, bfac, dim, rows, voff, vstride, act, cb))
        if ops == 0:
            assert np.array_equal(bo.download(np.uint16, wf.size).reshape(wf.shape), wf)
        else:
            assert np.array_equal(bs.download(np.float32, wsc.size).reshape(wsc.shape), wsc)
            assert np.array_equal(bq.download(np.int8, wq.size).reshape(wq.shape), wq)
        if ops == 2:
            assert np.array_equal(bg.download(np.int32, wgs.size).reshape(wgs.shape), wgs)
    with pytest.raises(B.UzuHipError):  # gated_act_mul.rs:39: quantized gate activation requires RHT
        B.GatedActMulKernel.new(hip_ctx, B.BF16, 1, 0, 0, 64, 0)


@pytest.mark.parametrize("bits,method,group_size,a_group", [(4, 0, 128, 128), (4, 1, 64, 32), (4, 2, 32, 64), (8, 0, 128, 64), (8, 1, 64, 128),
                                                            (4, 1, 128, 128), (4, 2, 64, 128), (8, 2, 128, 128)])
@pytest.mark.parametrize("m,n,k", [(1, 1024, 1024), (70, 200, 512), (256, 3072, 1024), (300, 520, 2048)])
def test_matmul_int8_symmetric_activations(hip_ctx, bits, method, group_size, a_group, m, n, k):
    """MatmulA::Int8Symmetric (matmul_a.rs:9-14; CPU semantics kernel.rs:190-200) against the CPU restatement: the integer part of
    every step is exact, the f32 scaling is summation-order class: <= 1 bf16 ulp (the reference's own CPU-vs-GPU bar for
    quantised matmuls is rel 0.05 / abs 0.4, quant_dispatch_test.rs:124).  M >= 128 with min(activation group, weight group) in
    {64, 128} runs on the int8 matrix cores (gemm_a8_mfma_kernel: v_mfma_i32_32x32x32_i8, one f32 fold per 64 / 128-element stage;
    ragged M = 300 / N = 520 included), everything else on the packed-dot VALU kernel."""
    rng = np.random.default_rng(bits * 100 + method * 10 + m)
    q = quant_matrix(rng, n, k, bits, group_size, method)
    x = bf16(rng.normal(0, 1.0, (m, k)))
    factors = rng.choice(np.array([-1, 1], np.int32), k)
    a_q, a_s = np.zeros((m, k), np.int8), np.zeros((m, k // a_group), np.float32)
    O.lib().orc_activation_transform(O.p(x), None, O.p(a_q), O.p(a_s), None, O.p(factors), O.BF16, m, k, 2, a_group, 0)
    bias = bf16(rng.normal(0, 0.2, n))
    want = np.zeros((m, n), np.uint16)
    args = O.MatmulArgs()
    args.a, args.a_dtype = None, O.BF16
    args.a_q, args.a_scales, args.a_group_size = a_q.ctypes.data, a_s.ctypes.data, a_group
    args.b, args.scales = q["weights"].ctypes.data, q["scales"].ctypes.data
    args.biases = q["biases"].ctypes.data if q["biases"] is not None else None
    args.zero_points = q["zero_points"].ctypes.data if q["zero_points"] is not None else None
    args.w_dtype, args.method, args.bits, args.group_size = O.BF16, method, bits, group_size
    args.d, args.d_dtype, args.ab_scale, args.bias = want.ctypes.data, O.BF16, 1.0, bias.ctypes.data
    args.m, args.n, args.k = m, n, k
    O.lib().orc_matmul(C.byref(args))
    kern = B.MatmulKernel.new(hip_ctx, B.BF16, B.BF16, B.BF16)
    ba, bs_, bw, bsc, bd, bb = (hip_ctx.buffer_from(a_q), hip_ctx.buffer_from(a_s), hip_ctx.buffer_from(q["weights"]), hip_ctx.buffer_from(q["scales"]),
                                hip_ctx.create_buffer(want.nbytes), hip_ctx.buffer_from(bias))
    bbi = hip_ctx.buffer_from(q["biases"]) if q["biases"] is not None else None
    bzp = hip_ctx.buffer_from(q["zero_points"]) if q["zero_points"] is not None else None
    run(hip_ctx, lambda cb: kern.encode(cb, a=ba, b=bw, d=bd, m=m, n=n, k=k, b_kind=method + 1, scales=bsc, biases=bbi, zero_points=bzp,
                                        mode=B.QMODE_U4 if bits == 4 else B.QMODE_U8, group_size=group_size, bias=bb, a_int8_scales=bs_, a_group_size=a_group))
    got = bd.download(np.uint16, want.size).reshape(want.shape)
    ulps = ulp_diff_bf16(want, got)
    assert ulps.max() <= 1.0, f"max {ulps.max()} bf16 ulps"
    assert (ulps == 0).mean() >= 0.97


@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("m,n,k", [(1024, 7168, 1024), (4096, 14336, 4096)])
def test_matmul_int8_activations_throughput_report(hip_ctx, m, n, k, bits, capsys):
    """Not a parity test (that is test_matmul_int8_symmetric_activations): the int8-MFMA A8 GEMM next to the bf16-activation GEMM on the
    same int4 g128 weights, GPU time of one command buffer of 5 launches each (uzu_hip_cmdbuf_gpu_execution_time_ns) -- the figure DESIGN.md
    quotes.  Asserts only that both ran and that the two results agree to the activation-quantisation error (int8 codes: ~1 % rms)."""
    rng = np.random.default_rng(m + n)
    q = quant_matrix(rng, n, k, bits, 128, 0)
    mode = B.QMODE_U4 if bits == 4 else B.QMODE_U8
    x = bf16(rng.normal(0, 1.0, (m, k)))
    factors = np.ones(k, np.int32)
    kern = B.MatmulKernel.new(hip_ctx, B.BF16, B.BF16, B.BF16)
    bw, bsc, bbi = hip_ctx.buffer_from(q["weights"]), hip_ctx.buffer_from(q["scales"]), hip_ctx.buffer_from(q["biases"])
    bx, bd16, bd8 = hip_ctx.buffer_from(x), hip_ctx.create_buffer(m * n * 2), hip_ctx.create_buffer(m * n * 2)
    bq, bs_ = hip_ctx.create_buffer(m * k), hip_ctx.create_buffer(m * (k // 128) * 4)
    tk = B.ActivationTransformKernel.new(hip_ctx, B.BF16, 2, 0, 128, 0)
    bf = hip_ctx.buffer_from(factors)
    run(hip_ctx, lambda cb: tk.encode(bx, None, bq, bs_, None, bf, m, k, cb))
    # Quantize includes the Hadamard transform: the bf16 leg gets the same rows transformed (InputRht), as RHTLinearWrapper feeds either
    bxh = hip_ctx.create_buffer(m * k * 2)
    th = B.ActivationTransformKernel.new(hip_ctx, B.BF16, B.ATX_INPUT_RHT, 0, 0, 0)
    run(hip_ctx, lambda cb: th.encode(bx, bxh, None, None, None, bf, m, k, cb))
    bx = bxh

    def bf16_path(cb):
        for _ in range(5):
            kern.encode(cb, a=bx, b=bw, d=bd16, m=m, n=n, k=k, b_kind=B.B_SCALE_BIAS, scales=bsc, biases=bbi, zero_points=None, mode=mode, group_size=128)

    def a8_path(cb):
        for _ in range(5):
            kern.encode(cb, a=bq, b=bw, d=bd8, m=m, n=n, k=k, b_kind=B.B_SCALE_BIAS, scales=bsc, biases=bbi, zero_points=None, mode=mode, group_size=128,
                        a_int8_scales=bs_, a_group_size=128)
    for fn in (bf16_path, a8_path):
        run(hip_ctx, fn)  # warm-up (workspace growth, code load)
    t16 = run(hip_ctx, bf16_path).gpu_execution_time() / 5
    t8 = run(hip_ctx, a8_path).gpu_execution_time() / 5
    flops = 2.0 * m * n * k
    with capsys.disabled():
        print(f"\nA8 report int{bits} weights {m}x{n}x{k}: bf16-activation GEMM {t16 * 1e6:.1f} us = {flops / t16 / 1e12:.0f} TFLOP/s; int8-MFMA A8 GEMM {t8 * 1e6:.1f} us = {flops / t8 / 1e12:.0f} TOP/s")
    d16, d8 = f32(bd16.download(np.uint16, m * n)), f32(bd8.download(np.uint16, m * n))
    assert t16 > 0 and t8 > 0
    assert np.sqrt(np.mean((d16 - d8) ** 2)) <= 0.03 * np.sqrt(np.mean(d16 ** 2))


def test_matmul_output_rht_then_bias_and_the_rht_linear_chain(hip_ctx):
    """MatmulDOps::rht_factors (d_ops.rs:3-9; kernel.rs:296-303): the output RHT runs in place on D after the store and the bias is
    added after it.  Then the whole RHTLinearWrapper chain of the reference (linear/rht_wrapper.rs:215-298) through the C ABI:
    InputRht in place on the activations -> quantised matmul -> OutputRht -> bias, against the same chain of CPU kernels;
    and the A8 form of it (Quantize -> Int8Symmetric matmul -> OutputRht -> bias)."""
    rng = np.random.default_rng(3)
    m, n, k, g = 9, 1024, 2048, 128
    q = quant_matrix(rng, n, k, 4, g, 0)
    x = bf16(rng.normal(0, 1.0, (m, k)))
    fin, fout = rng.choice(np.array([-1, 1], np.int32), k), rng.choice(np.array([-1, 1], np.int32), n)
    bias = bf16(rng.normal(0, 0.2, n))

    def oracle_chain(a8):
        args = O.MatmulArgs()
        xr = x.copy()
        if a8:
            a_q, a_s = np.zeros((m, k), np.int8), np.zeros((m, k // g), np.float32)
            O.lib().orc_activation_transform(O.p(x), None, O.p(a_q), O.p(a_s), None, O.p(fin), O.BF16, m, k, 2, g, 0)
            args.a, args.a_q, args.a_scales, args.a_group_size = None, a_q.ctypes.data, a_s.ctypes.data, g
            keep = (a_q, a_s)
        else:
            O.lib().orc_activation_transform(None, O.p(xr), None, None, None, O.p(fin), O.BF16, m, k, 0, 0, 0)
            args.a = xr.ctypes.data
            keep = (xr,)
        want = np.zeros((m, n), np.uint16)
        args.a_dtype = O.BF16
        args.b, args.scales, args.biases = q["weights"].ctypes.data, q["scales"].ctypes.data, q["biases"].ctypes.data
        args.w_dtype, args.method, args.bits, args.group_size = O.BF16, 0, 4, g
        args.d, args.d_dtype, args.ab_scale, args.bias, args.rht_factors = want.ctypes.data, O.BF16, 1.0, bias.ctypes.data, fout.ctypes.data
        args.m, args.n, args.k = m, n, k
        O.lib().orc_matmul(C.byref(args))
        del keep
        return want

    mat = B.MatmulKernel.new(hip_ctx, B.BF16, B.BF16, B.BF16)
    bw, bsc, bbi, bb = hip_ctx.buffer_from(q["weights"]), hip_ctx.buffer_from(q["scales"]), hip_ctx.buffer_from(q["biases"]), hip_ctx.buffer_from(bias)
    bfin, bfout = hip_ctx.buffer_from(fin), hip_ctx.buffer_from(fout)
    # full-precision chain
    bx, bd = hip_ctx.buffer_from(x), hip_ctx.create_buffer(m * n * 2)
    rht = B.ActivationTransformKernel.new(hip_ctx, B.BF16, B.ATX_INPUT_RHT, 1, 0, 0)

    def enc(cb):
        rht.encode(None, bx, None, None, None, bfin, m, k, cb)
        mat.encode(cb, a=bx, b=bw, d=bd, m=m, n=n, k=k, b_kind=B.B_SCALE_BIAS, scales=bsc, biases=bbi, mode=B.QMODE_U4, group_size=g, bias=bb, rht_factors=bfout)
    run(hip_ctx, enc)
    want = oracle_chain(False)
    got = bd.download(np.uint16, m * n).reshape(m, n)
    # the transform adds 32 matmul outputs (each within 1 bf16 ulp of ITS magnitude) with signs: where they cancel, the sum's own
    # ulp is smaller than the carried error -- a few ulps at worst, the bulk stays within one
    ulps = ulp_diff_bf16(want, got)
    assert ulps.max() <= 4.0 and (ulps <= 1.0).mean() >= 0.98
    # A8 chain
    bx2, bq, bs_ = hip_ctx.buffer_from(x), hip_ctx.create_buffer(m * k), hip_ctx.create_buffer(m * (k // g) * 4)
    quant = B.ActivationTransformKernel.new(hip_ctx, B.BF16, B.ATX_QUANTIZE, 0, g, 0)

    def enc8(cb):
        quant.encode(bx2, None, bq, bs_, None, bfin, m, k, cb)
        mat.encode(cb, a=bq, b=bw, d=bd, m=m, n=n, k=k, b_kind=B.B_SCALE_BIAS, scales=bsc, biases=bbi, mode=B.QMODE_U4, group_size=g, bias=bb, rht_factors=bfout,
                   a_int8_scales=bs_, a_group_size=g)
    run(hip_ctx, enc8)
    want8 = oracle_chain(True)
    got8 = bd.download(np.uint16, m * n).reshape(m, n)
    ulps8 = ulp_diff_bf16(want8, got8)
    assert ulps8.max() <= 4.0 and (ulps8 <= 1.0).mean() >= 0.98
    # the two chains agree with each other within the int8 quantisation noise of the activations (~1 %)
    assert np.abs(f32(want8) - f32(want)).max() <= 0.05 * np.abs(f32(want)).max()


# ------------------------------------------------------------------------------------------------ weight-streaming engine
def _set_stream(mode):
    """csrc/k_stream.hip: -1 environment / default (bandwidth regime only), 0 never, 2 every supported shape."""
    fn = _ffi.lib().uzu_hip_debug_set_decode_stream
    fn.restype, fn.argtypes = None, [C.c_int]
    fn(mode)


def _stream_error():
    fn = _ffi.lib().uzu_hip_debug_decode_stream_error
    fn.restype, fn.argtypes = C.c_uint32, []
    return int(fn())


STREAM_SHAPES = [  # (n, k, group): steps per lane 1 (two and one rows per wave), 2, 3, 4, 7, 9; ragged last slots; less than one slot per CU
    (6144, 4096, 128), (4096, 14336, 128), (1000, 1024, 128), (777, 2048, 64), (515, 5120, 128), (300, 17408, 128), (97, 8192, 32), (8224, 1024, 128),
    (33, 4096, 128), (2, 512, 64),
]


@pytest.mark.parametrize("n,k,group", STREAM_SHAPES)
def test_stream_gemv_is_bit_identical_to_register_gemv(hip_ctx, n, k, group):
    """The LDS-staged weight stream (loader wave + LDS-DMA ring + consumer waves, csrc/k_stream.hip) against the register GEMV of
    k_decode.hip on the same matrix: same lane mapping and arithmetic, so every output is BIT-identical whichever wave / slot / kernel
    computes the row -- and both are within 1 bf16 ulp of the CPU restatement (kernel.rs:190-293).  No bounded wait may give up."""
    rng = np.random.default_rng(n * 3 + k)
    q = quant_matrix(rng, n, k, 4, group, 0)
    a = activations(rng, 1, k)
    bias = bf16(rng.uniform(-0.5, 0.5, size=(n,)))
    try:
        _set_stream(0)
        base = hip_matmul(hip_ctx, a, q, 1, bias=bias)
        _set_stream(2)
        got = hip_matmul(hip_ctx, a, q, 1, bias=bias)
        again = hip_matmul(hip_ctx, a, q, 1, bias=bias)
    finally:
        _set_stream(-1)
    assert _stream_error() == 0, "a bounded wait of the streaming kernel gave up"
    assert np.array_equal(got, again), "the streaming kernel is not deterministic"
    assert np.array_equal(base, got), f"{(base != got).sum()} of {n} outputs differ from the register GEMV"
    want = oracle_matmul(a, q, 1, bias=bias)
    assert ulp_diff_bf16(want, got).max() <= 1.0


# ------------------------------------------------------------------------------------------ HybridSpec: Hadamard hoisted into the norm / the embedding
def test_normalization_with_hoisted_input_hadamard(hip_ctx):
    """Normalization { use_hadamard } (normalization.metal:134-140): the input RHT of the linear behind the norm applied to the rounded
    result.  The reference's own CPU kernel is `unimplemented!` here (normalization.rs:46-48), so the expectation is the Metal source's
    composition restated with the oracle's pieces: orc_normalization, then ActivationTransform::InputRht on its output -- in
    reference-order mode BIT-identical, in production mode within the norm's own tolerance class."""
    rng = np.random.default_rng(77)
    rows, dim = 5, 1024
    x, sc = bf16(rng.normal(0, 1.5, size=(rows, dim))), bf16(rng.normal(0, 1.5, size=(rows, dim)))
    scales = rng.uniform(-0.2, 0.2, size=(dim,)).astype(np.float32)
    signs = rng.choice(np.array([-1, 1], np.int32), dim).astype(np.int32)
    normed, want_sc = np.zeros_like(x), sc.copy()
    args = O.NormArgs(x.ctypes.data, scales.ctypes.data, None, normed.ctypes.data, want_sc.ctypes.data, O.BF16, O.F32, rows, dim, 1e-6, 1.0, 1.0, 0, 1, 1, 1, 0, 0)
    O.lib().orc_normalization(C.byref(args))
    want = np.zeros_like(x)
    O.call("orc_activation_transform", normed, want, None, None, None, signs, O.BF16, rows, dim, 0, 0, 0)
    kern = B.NormalizationKernel.new(hip_ctx, B.BF16, B.F32, B.BF16, B.F32, 0, 0, 1, 1, 1, 1, 0, 0, 0, 1)  # use_hadamard = 1
    fn = _ffi.lib().uzu_hip_set_exact
    fn.restype, fn.argtypes = None, [C.c_int32]
    for exact in (1, 0):
        fn(exact)
        try:
            bx, bs, bo, bsc, bh = hip_ctx.buffer_from(x), hip_ctx.buffer_from(scales), hip_ctx.create_buffer(x.nbytes), hip_ctx.buffer_from(sc), hip_ctx.buffer_from(signs)
            run(hip_ctx, lambda cb: kern.encode(bx, bs, None, bo, bsc, bh, rows, dim, 1e-6, 1.0, 1.0, cb))
            got = bo.download(np.uint16, rows * dim).reshape(rows, dim)
        finally:
            fn(0)
        assert np.array_equal(bsc.download(np.uint16, rows * dim).reshape(rows, dim), want_sc)
        if exact:
            assert np.array_equal(got, want)
        else:  # a 1-ulp difference of a normalised element spreads over its 32-element Hadamard block
            w, g = f32(want).astype(np.float64), f32(got).astype(np.float64)
            assert np.abs(w - g).max() <= 0.02 * np.abs(w).max()
    with pytest.raises(B.UzuHipError):  # the combination the Metal kernel orders differently is refused, not approximated
        B.NormalizationKernel.new(hip_ctx, B.BF16, B.F32, B.BF16, B.F32, 0, 0, 1, 1, 1, 1, 0, 1, 0, 1)


@pytest.mark.parametrize("bits,method", [(4, 0), (8, 1)])
def test_quantized_embedding_lookup_with_output_hadamard(hip_ctx, bits, method):
    """QuantizedEmbeddingLookup { use_hadamard } (quant_embedding.metal:92-98; `unimplemented!` in the reference's CPU kernel): the
    dequantised, rounded row through ActivationTransform::OutputRht -- bit-exact against that composition of the oracle's kernels."""
    rng = np.random.default_rng(19 + bits)
    vocab, dim, g = 300, 256, 64
    q = quant_matrix(rng, vocab, dim, bits, g, method, scale_mag=1.0)
    ids = np.array([0, 299, 17, 5, 123], np.uint32)
    signs = rng.choice(np.array([-1, 1], np.int32), dim).astype(np.int32)
    rows = np.zeros((ids.size, dim), np.uint16)
    O.call("orc_quantized_embedding_lookup", ids, q["weights"], q["scales"], q["zero_points"], q["biases"], rows, O.BF16, ids.size, vocab, dim, 1.5, g, bits, method)
    want = rows.copy()
    O.call("orc_activation_transform", None, want, None, None, None, signs, O.BF16, ids.size, dim, 1, 0, 0)
    kern = B.QuantizedEmbeddingLookupKernel.new(hip_ctx, B.BF16, g, B.QMODE_U4 if bits == 4 else B.QMODE_U8, method, 1)
    bid, bw, bs = hip_ctx.buffer_from(ids), hip_ctx.buffer_from(q["weights"]), hip_ctx.buffer_from(q["scales"])
    bz = hip_ctx.buffer_from(q["zero_points"]) if q["zero_points"] is not None else None
    bb = hip_ctx.buffer_from(q["biases"]) if q["biases"] is not None else None
    bo, bh = hip_ctx.create_buffer(want.nbytes), hip_ctx.buffer_from(signs)
    run(hip_ctx, lambda cb: kern.encode(bid, bw, bs, bz, bb, bo, bh, ids.size, vocab, dim, 1.5, cb))
    assert np.array_equal(bo.download(np.uint16, want.size).reshape(want.shape), want)


def test_matmul_activation_format_policy(hip_ctx):
    """MatmulKernel::{a8_activation_plan, select_activation_format} (kernel.rs:28-42) through the C ABI: the plan exists for every
    quantised B the A8 matmul takes (activation group = ACTIVATION_SCALE_GROUP_SIZE = 128; a sum group for the prologues with an offset
    term: metal/kernel/matmul/mod.rs:154-165), not for full-precision B / full-precision-only kernels / k off the group grid; the selected
    format is Bf16 for decode and prefill shapes alike (the int8 matrix-core GEMM measured at the bf16 GEMM's rate: DESIGN.md section 3)."""
    kern = B.MatmulKernel.new(hip_ctx, B.BF16, B.BF16, B.BF16)
    shape = lambda **kw: B.MatmulShape(**{**dict(m=1024, n=7168, k=1024, b_transpose=1, b_kind=B.B_SCALE_BIAS, b_bits=4, b_group_size=128), **kw})
    assert kern.a8_activation_plan(shape()) == (128, 128)
    assert kern.a8_activation_plan(shape(b_group_size=64, b_kind=B.B_SCALE_ZERO_POINT, b_bits=8)) == (128, 64)
    assert kern.a8_activation_plan(shape(b_kind=B.B_SCALE_SYMMETRIC)) == (128, None)
    assert kern.a8_activation_plan(shape(b_kind=B.B_FULL_PRECISION)) is None
    assert kern.a8_activation_plan(shape(k=1024 + 64)) is None
    assert kern.a8_activation_plan(shape(a_full_precision=1)) is None
    assert B.MatmulKernel.new(hip_ctx, B.F32, B.F32, B.F32).a8_activation_plan(shape()) is None
    for m in (1, 16, 128, 4096):
        assert kern.select_activation_format(shape(m=m, a_full_precision=1)) == B.ACTIVATION_FORMAT_BF16
