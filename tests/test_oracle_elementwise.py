"""CPU tests pinning the oracle's element-wise / index kernels by independent NumPy re-derivations.

Companion of tests/test_oracle_kernels.py (which covers matmul, norms, attention, DeltaNet, the model): same
policy -- float64 math, rounded to bf16 where the reference rounds (half::bf16::from_f32, RNE), so the expected
values here share no code with oracle/*.c.  Edge cases follow the reference's own tests: empty inputs, a single
element, out-of-range token ids, aliasing in-place forms, overlapping cache copies.
References: BU/cpu/kernel/attention/{qkv_norm,sigmoid_gate,kv_cache_update}.rs, logit_transform/logit_transform.rs,
tensor_add_bias / tensor_add_scale / tensor_add_swap, embedding/full_precision_embedding.rs,
common/gpu_types/activation_type.rs:16-65.
"""
import ctypes as C
import math

import numpy as np
import pytest

from helpers import bf16, f32
from oracle import oracle as O


def ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


def bf16_round(x):
    """float64 -> value of the nearest bf16 (through f32, as the reference does: f32 op, then T::from)."""
    return f32(bf16(np.asarray(x, dtype=np.float32))).astype(np.float64)


@pytest.mark.parametrize("act,fn", [
    (0, lambda x: x / (1.0 + np.exp(-x))),                                                        # SiLU
    (1, lambda x: 0.5 * x * (1.0 + np.tanh(np.float32(0.7978846) * (x + np.float32(0.044715) * x ** 3)))),  # GELU (tanh form)
    (2, lambda x: 0.5 * x * (1.0 + np.vectorize(math.erf)(x * 0.70710678118654752440))),           # GELU (erf form)
    (3, lambda x: x),                                                                             # identity
    (4, lambda x: np.where(x > 20.0, x, np.log1p(np.exp(np.minimum(x, 20.0))))),                   # softplus, linear above 20
])
def test_activation_types_against_float64(act, fn):
    lib = O.lib()
    lib.orc_activate.restype = C.c_float
    xs = f32(bf16(np.concatenate([np.linspace(-12, 12, 481), [0.0, -0.0, 20.0, 20.125, 30.0, -30.0, 1e-3, -1e-3]])))
    got = np.array([lib.orc_activate(C.c_uint32(act), C.c_float(float(x)), C.c_uint32(O.BF16)) for x in xs], np.float64)
    want = fn(xs.astype(np.float64))
    if act == 3:
        assert np.array_equal(got, xs.astype(np.float64))  # identity does not round
        return
    if act == 4:
        big = xs > 20.0
        assert np.array_equal(got[big], xs[big].astype(np.float64))  # x > 20: returned as is (no rounding)
    # the result is a bf16 value within one bf16 ulp of the float64 function (f32 libm + one rounding)
    assert np.array_equal(bf16_round(got), got) or act == 4
    # One bf16 rounding (half an ulp) plus the f32 cancellation noise of the reference's formulas: `1 + tanh`, `1 + erf` and
    # `1 + exp` are formed in f32, so the result carries ~2^-23 * max(1, |x|) of absolute error where they cancel.
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(want), 2.0 ** -126))) - 7)
    tol = 0.51 * ulp + 2.0 ** -23 * np.maximum(1.0, np.abs(xs.astype(np.float64)))
    assert np.all(np.abs(got - want) <= tol), np.max(np.abs(got - want) / tol)


def test_sigmoid_gate_in_place_product():
    rng = np.random.default_rng(1)
    for total in (0, 1, 7, 4096):
        gate = bf16(rng.normal(0, 3, total))
        out = bf16(rng.normal(0, 2, total))
        want = bf16_round(f32(out).astype(np.float64) / (1.0 + np.exp(-f32(gate).astype(np.float64))))
        O.lib().orc_sigmoid_gate(ptr(gate), ptr(out), O.BF16, total)
        got = f32(out).astype(np.float64)
        # f32 expf + one f32 product + one rounding: identical to the float64 result except at rounding ties
        ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(want), 2.0 ** -126))) - 7)
        assert np.all(np.abs(got - want) <= ulp), total
        assert total == 0 or (got == want).mean() > 0.99


def test_logit_transform_scale_and_soft_cap():
    rng = np.random.default_rng(2)
    x = bf16(rng.normal(0, 20, 1000))
    for scale, cap in ((1.0, None), (0.5, None), (1.0, 30.0), (0.0625, 50.0)):
        buf = x.copy()
        O.lib().orc_logit_transform(ptr(buf), O.BF16, buf.size, C.c_float(scale), C.c_float(cap or 0.0), 1 if cap else 0)
        v = f32(x).astype(np.float64) * scale
        if cap:
            v = np.tanh(v / cap) * cap
        want = bf16_round(v)
        got = f32(buf).astype(np.float64)
        ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(want), 2.0 ** -126))) - 7)
        assert np.all(np.abs(got - want) <= ulp)
        if not cap:
            assert np.array_equal(got, want)  # a power-of-two scale is exact: no tie can differ
    empty = np.zeros(0, np.uint16)
    O.lib().orc_logit_transform(ptr(empty), O.BF16, 0, C.c_float(2.0), C.c_float(0.0), 0)


def test_tensor_add_bias_scale_swap_copy():
    rng = np.random.default_rng(3)
    rows, cols = 5, 37
    x = bf16(rng.normal(0, 2, (rows, cols)))
    b = bf16(rng.normal(0, 1, cols))
    lib = O.lib()
    # add_bias: out = T(x + bias[col]); the in-place form (input = NULL) gives the same
    out = np.zeros_like(x)
    lib.orc_tensor_add_bias(ptr(x), ptr(b), ptr(out), O.BF16, O.BF16, cols, x.size)
    want = bf16(f32(x).astype(np.float64) + f32(b).astype(np.float64)[None, :])  # one exact f64 sum, one rounding
    assert np.array_equal(out, want)
    inplace = x.copy()
    lib.orc_tensor_add_bias(None, ptr(b), ptr(inplace), O.BF16, O.BF16, cols, x.size)
    assert np.array_equal(inplace, want)
    # add_scale: out = T((x + bias[col]) * scale): the f32 sum of two bf16 values is exact, then one f32 product, one rounding
    for scale in (0.5, 1.25, 3.0):
        lib.orc_tensor_add_scale(ptr(x), ptr(b), ptr(out), O.BF16, cols, x.size, C.c_float(scale))
        s = (f32(x).astype(np.float32) + f32(b).astype(np.float32)[None, :]).astype(np.float32)
        assert np.array_equal(out, bf16((s * np.float32(scale)).astype(np.float32)))
    # add_swap: both buffers end up holding T(skip + main)
    skip, main = x.copy(), bf16(rng.normal(0, 2, (rows, cols)))
    want = bf16(f32(skip).astype(np.float64) + f32(main).astype(np.float64))
    lib.orc_tensor_add_swap(ptr(skip), ptr(main), O.BF16, skip.size)
    assert np.array_equal(skip, want) and np.array_equal(main, want)
    # copy, bf16 and f32, including length 0
    dst = np.zeros_like(x)
    lib.orc_tensor_copy(ptr(x), ptr(dst), O.BF16, x.size)
    assert np.array_equal(dst, x)
    xf = rng.normal(0, 1, 33).astype(np.float32)
    df = np.zeros_like(xf)
    lib.orc_tensor_copy(ptr(xf), ptr(df), O.F32, xf.size)
    assert np.array_equal(df, xf)
    lib.orc_tensor_copy(ptr(x), ptr(dst), O.BF16, 0)


def test_kv_cache_update_copies_rows_in_both_caches():
    """KVCacheUpdate (kv_cache_update.rs:9-28): per element, the copies run in list order -- a later copy may read a row an
    earlier one wrote (speculative-accept compaction relies on ascending destinations)."""
    rng = np.random.default_rng(4)
    rows, dim = 12, 24
    keys0, vals0 = bf16(rng.normal(0, 1, (rows, dim))), bf16(rng.normal(0, 1, (rows, dim)))

    class Copy(C.Structure):
        _fields_ = [("source", C.c_uint32), ("destination", C.c_uint32)]

    for pairs in ([], [(5, 2)], [(3, 1), (7, 2), (9, 3)], [(1, 2), (2, 3)]):  # last: chained (row 3 receives the NEW row 2 = old row 1)
        keys, vals = keys0.copy(), vals0.copy()
        arr = (Copy * max(len(pairs), 1))(*[Copy(s, d) for s, d in pairs])
        O.lib().orc_kv_cache_update(ptr(keys), ptr(vals), O.BF16, arr, len(pairs), dim)
        wk, wv = keys0.copy(), vals0.copy()
        for s, d in pairs:
            wk[d], wv[d] = wk[s].copy(), wv[s].copy()
        assert np.array_equal(keys, wk) and np.array_equal(vals, wv), pairs


@pytest.mark.parametrize("mode", ["none", "full_layer", "only_normalization"])
def test_qkv_norm_head_ranges_against_float64(mode):
    """QKVNorm (qkv_norm.rs:32-77): per-head RMS norm in place on a head range of the packed [batch, heads, hd] rows;
    heads outside [head_offset, head_offset + head_count) are untouched."""
    rng = np.random.default_rng(5)
    batch, heads, hd, off, cnt = 3, 7, 64, 2, 3
    x = bf16(rng.normal(0, 1.5, (batch, heads, hd)))
    scales = (1.0 + 0.1 * rng.normal(0, 1, hd)).astype(np.float32)
    eps, scale_offset = 1e-6, 0.25
    buf = x.copy()
    O.lib().orc_qkv_norm(ptr(buf), O.BF16, ptr(scales) if mode != "none" else None, batch, heads, hd, C.c_float(eps), C.c_float(scale_offset), off, cnt,
                         1 if mode == "full_layer" else 0)
    xf = f32(x).astype(np.float64)
    want = xf.copy()
    sel = xf[:, off:off + cnt, :]
    normalized = sel / np.sqrt((sel * sel).mean(axis=-1, keepdims=True) + eps)
    if mode == "none":
        res = bf16_round(normalized)
    elif mode == "full_layer":
        res = bf16_round(normalized * (scales.astype(np.float64) + scale_offset))
    else:
        res = bf16_round(bf16_round(normalized) * bf16_round(scales.astype(np.float64) + scale_offset))
    want[:, off:off + cnt, :] = res
    got = f32(buf).astype(np.float64)
    assert np.array_equal(got[:, :off], xf[:, :off]) and np.array_equal(got[:, off + cnt:], xf[:, off + cnt:])
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(want), 2.0 ** -126))) - 7)
    assert np.all(np.abs(got - want) <= ulp)  # f32 sum order vs float64: at most a rounding tie apart
    assert (got == want).mean() > 0.99


def test_full_precision_embedding_lookup_and_out_of_range_ids():
    rng = np.random.default_rng(6)
    vocab, dim = 50, 48
    table = bf16(rng.normal(0, 1, (vocab, dim)))
    ids = np.array([0, 49, 7, 50, 2 ** 31, 7], np.uint32)  # two ids outside the vocabulary: rows of zeros (full_precision_embedding.rs:20-24)
    for input_scale in (1.0, 0.5, 1.7):
        out = np.full((ids.size, dim), 0x7FC0, np.uint16)
        O.lib().orc_full_precision_embedding_lookup(ptr(ids), ptr(table), ptr(out), O.BF16, ids.size, vocab, dim, C.c_float(input_scale))
        s = f32(bf16(np.float32(input_scale))).astype(np.float64)  # the scale is rounded to T first
        for r, t in enumerate(ids):
            want = np.zeros(dim) if t >= vocab else bf16_round(f32(table[t]).astype(np.float64) * s)
            got = f32(out[r]).astype(np.float64)
            ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(want), 2.0 ** -126))) - 7)
            assert np.all(np.abs(got - want) <= ulp), (input_scale, r)
            if input_scale in (1.0, 0.5):
                assert np.array_equal(got, want)
