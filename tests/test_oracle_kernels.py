"""CPU tests that PIN THE ORACLE (oracle/ = C restatement of the reference's CPU kernels).

The reference cannot be built here and ships no golden vectors for this path, so the oracle is
pinned by:
  (1) the one literal known-answer test in the reference tree (gated_act_mul_test.rs:139-160);
  (2) independent float64 NumPy implementations of the same math, in the style of the reference's own
      `reference_attention` (tests/unit/encodable_block/attention_test.rs:26-124), at the reference's
      tolerances (bf16: 1e-2 attention/norm, quant matmul rel 0.05 / abs 0.4 -- we assert far tighter);
  (3) structural identities the reference's kernels must satisfy (two-pass == single-pass attention,
      DeltaNet prefill path == repeated decode steps, matmul of M rows == M single-row matmuls);
  (4) committed golden fixtures produced by this oracle (tests/golden/, regression protection).
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from helpers import MASK_VARIANTS, attention_float64, bf16, dequantize, f32, mask_case, quant_matrix, ref_attention_inputs, trie_from_parents
from oracle import oracle as O
from uzu_amd import synthetic as S

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_bf16_conversion_matches_half_crate_rules():
    """half::bf16::from_f32: round to nearest even, NaN quieted, +-inf preserved, subnormals kept."""
    lib = O.lib()
    cases = {0x3F800000: 0x3F80, 0x3F808000: 0x3F80, 0x3F818000: 0x3F82, 0x3F808001: 0x3F81, 0x7F800000: 0x7F80, 0xFF800000: 0xFF80,
             0x7FC00001: 0x7FC0, 0x7F800001: 0x7FC0, 0x00008000: 0x0000, 0x00018000: 0x0002, 0x80000000: 0x8000, 0x7F7FFFFF: 0x7F80}
    for bits, want in cases.items():
        f = np.array([bits], np.uint32).view(np.float32)[0]
        assert lib.orc_f32_to_bf16(C.c_float(f)) == want, hex(bits)
    x = np.random.default_rng(0).normal(0, 100, 100000).astype(np.float32)
    ours = S.f32_to_bf16_bits(x)
    theirs = np.array([lib.orc_f32_to_bf16(C.c_float(v)) for v in x[:2000]], np.uint16)
    assert np.array_equal(ours[:2000], theirs)


def oracle_matmul(a_bits, q, m, **kw):
    n = q["n"]
    d = np.zeros((m, n), np.uint16)
    args = O.MatmulArgs()
    args.a, args.a_dtype, args.b, args.scales = a_bits.ctypes.data, O.BF16, q["weights"].ctypes.data, q["scales"].ctypes.data
    args.biases = q["biases"].ctypes.data if q["biases"] is not None else None
    args.zero_points = q["zero_points"].ctypes.data if q["zero_points"] is not None else None
    args.w_dtype, args.method, args.bits, args.group_size, args.b_transpose = O.BF16, q["method"], q["bits"], q["group_size"], 1
    args.d, args.d_dtype, args.ab_scale = d.ctypes.data, O.BF16, 1.0
    args.m, args.n, args.k = m, n, q["k"]
    O.lib().orc_matmul(C.byref(args))
    return d


@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("method", [0, 1, 2])
@pytest.mark.parametrize("group_size", [32, 64, 128])
def test_matmul_against_float64(bits, method, group_size):
    """Nibble order, group indexing, zero-point packing and the three dequant forms, vs an independent f64
    dequantise-then-matmul (reference tolerance: rel 0.05 / abs 0.4; asserted: 1 bf16 ulp + f32 noise)."""
    rng = np.random.default_rng(bits * 100 + method * 10 + group_size)
    n, k, m = 200, 384, 3
    q = quant_matrix(rng, n, k, bits, group_size, method)
    a = bf16(rng.uniform(-1, 1, size=(m, k)))
    got = f32(oracle_matmul(a, q, m)).astype(np.float64)
    want = f32(a).astype(np.float64) @ dequantize(q).T
    tol = np.abs(want) * 2.0 ** -8 + 1e-4 * np.abs(want).max()
    assert (np.abs(got - want) <= tol).all()
    # M rows == M independent single-row matmuls, bit for bit
    for r in range(m):
        assert np.array_equal(oracle_matmul(np.ascontiguousarray(a[r:r + 1]), q, 1)[0], oracle_matmul(a, q, m)[r])


def test_gated_act_mul_reference_known_answer():
    """The literal KAT of the reference tree (gated_act_mul_test.rs:139-160)."""
    gate = np.array([1, 2, 3, 4], np.float32)
    value = np.array([0, 0, 10, 20, 30, 40, 0, 0, 50, 60, 70, 80], np.float32)
    out = np.zeros(4, np.float32)
    O.call("orc_gated_act_mul", gate, value, out, O.F32, 2, 2, 2, 6, 3, 0)
    assert out.tolist() == [10.0, 40.0, 150.0, 240.0]


def test_silu_gate_rounding_points():
    """bf16 path: silu evaluated in f32, rounded to bf16, multiplied in bf16 (gated_act_mul/mod.rs:5-12)."""
    rng = np.random.default_rng(1)
    h = 64
    fused = bf16(rng.normal(0, 2, size=(2, 2 * h)))
    out = np.zeros((2, h), np.uint16)
    O.call("orc_gated_act_mul", fused, None, out, O.BF16, h, 2, 0, 0, 0, 1)
    up, gate = f32(fused[:, :h]).astype(np.float64), f32(fused[:, h:]).astype(np.float64)
    silu_b = f32(bf16(gate / (1 + np.exp(-gate))))
    want = bf16(up * silu_b)
    # f64 exp vs f32 expf may differ in the last f32 bit before rounding: allow 1 bf16 ulp on < 2% of elements
    diff = np.abs(out.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1 and (diff == 0).mean() > 0.98


@pytest.mark.parametrize("full_layer", [0, 1])
def test_normalization_against_float64(full_layer):
    rng = np.random.default_rng(2)
    rows, dim = 4, 512
    x, sc = bf16(rng.normal(0, 1.5, (rows, dim))), bf16(rng.normal(0, 1.5, (rows, dim)))
    scales = rng.uniform(-0.2, 0.2, dim).astype(np.float32)
    out, sc_out = np.zeros_like(x), sc.copy()
    args = O.NormArgs(x.ctypes.data, scales.ctypes.data, None, out.ctypes.data, sc_out.ctypes.data, O.BF16, O.F32, rows, dim, 1e-6, 1.0, 1.0,
                      0, full_layer, 1, 1, 0, 0)
    O.lib().orc_normalization(C.byref(args))
    res = f32(bf16(f32(x) + f32(sc))).astype(np.float64)  # residual add rounds to bf16 and is written back
    assert np.array_equal(sc_out, bf16(res))
    want = res / np.sqrt((res ** 2).mean(axis=1, keepdims=True) + 1e-6) * (scales.astype(np.float64) + 1.0)
    assert np.abs(f32(out) - want).max() <= 1e-2 * max(1.0, np.abs(want).max())  # reference tolerance (normalization_test.rs:104-110)
    # FullLayer rounds once; OnlyNormalization rounds the normalised value, the scale and the product (3 roundings)
    assert (np.abs(f32(out) - want) <= np.abs(want) * (2.0 ** -7 if full_layer else 2.0 ** -6) + 1e-6).all()


def numpy_attention(q, k, v, heads, kv_heads, hd, seq, suffix, scale):
    """Independent f64 causal softmax attention over a [tokens, kv_heads, hd] cache (cf. attention_test.rs:26-124)."""
    qf = f32(q).astype(np.float64).reshape(heads, suffix, hd)
    kf = f32(k).astype(np.float64)[:seq].reshape(seq, kv_heads, hd)
    vf = f32(v).astype(np.float64)[:seq].reshape(seq, kv_heads, hd)
    out = np.zeros((suffix, heads, hd))
    prefix = seq - suffix
    for h in range(heads):
        g = h // (heads // kv_heads)
        for s in range(suffix):
            n_keys = prefix + s + 1
            sc = (kf[:n_keys, g] @ qf[h, s]) * scale
            p = np.exp(sc - sc.max())
            out[s, h] = (p / p.sum()) @ vf[:n_keys, g]
    return out


@pytest.mark.parametrize("heads,kv_heads,hd,seq,suffix", [(8, 2, 64, 50, 1), (4, 2, 128, 90, 5), (4, 1, 64, 1200, 2)])
def test_attention_single_and_two_pass_against_float64(heads, kv_heads, hd, seq, suffix):
    rng = np.random.default_rng(seq)
    q = bf16(rng.normal(0, 1, (heads, suffix, hd)))
    k, v = bf16(rng.normal(0, 1, (seq + 4, kv_heads * hd))), bf16(rng.normal(0, 1, (seq + 4, kv_heads * hd)))
    scale = 1.0 / np.sqrt(hd)
    a = O.AttentionArgs(q.ctypes.data, k.ctypes.data, v.ctypes.data, O.BF16, hd, heads // kv_heads, seq, hd, kv_heads * hd, hd, kv_heads * hd,
                        0, 0, 0, scale, 0, 0, None, heads, suffix, 1)
    single = np.zeros((suffix, heads, hd), np.uint16)
    O.lib().orc_attention_single_pass(C.byref(a), O.p(single))
    rows = suffix * heads
    parts, sums, maxs = np.zeros((rows, 32, hd), np.float32), np.zeros((rows, 32), np.float32), np.zeros((rows, 32), np.float32)
    O.lib().orc_attention_two_pass1(C.byref(a), O.p(parts), O.p(sums), O.p(maxs))
    two = np.zeros_like(single)
    O.call("orc_attention_two_pass2", parts, sums, maxs, two, O.BF16, hd, heads, suffix)
    want = numpy_attention(q, k, v, heads, kv_heads, hd, seq, suffix, scale)
    assert np.abs(f32(single) - want).max() <= 1e-2  # reference tolerance (attention_single_pass_test.rs:132-136)
    assert np.abs(f32(two) - want).max() <= 1e-2
    assert (np.abs(f32(single) - want) <= np.abs(want) * 2.0 ** -7 + 2e-3).all()
    assert np.abs(f32(single) - f32(two)).max() <= 4e-3  # same math, different summation order


def test_attention_sliding_window_and_sinks_masks():
    """mask.rs: causal sliding window keeps keys with 0 <= q_pos - k_pos < window; sinks add exp(sink) to the denominator."""
    rng = np.random.default_rng(3)
    heads, hd, seq, window = 2, 64, 40, 8
    q, k, v = bf16(rng.normal(0, 1, (heads, 1, hd))), bf16(rng.normal(0, 1, (seq, hd))), bf16(rng.normal(0, 1, (seq, hd)))
    sinks = bf16(np.array([0.5, -1.0]))
    scale = 0.125
    a = O.AttentionArgs(q.ctypes.data, k.ctypes.data, v.ctypes.data, O.BF16, hd, heads, seq, hd, hd, hd, hd, 0, 0, 0, scale, 1, window,
                        sinks.ctypes.data, heads, 1, 1)
    out = np.zeros((1, heads, hd), np.uint16)
    O.lib().orc_attention_single_pass(C.byref(a), O.p(out))
    qf, kf, vf = f32(q).astype(np.float64), f32(k).astype(np.float64), f32(v).astype(np.float64)
    for h in range(heads):
        keys = np.arange(seq - window, seq)
        sc = (kf[keys] @ qf[h, 0]) * scale
        m = max(sc.max(), float(f32(sinks)[h]))
        p = np.exp(sc - m)
        want = (p @ vf[keys]) / (p.sum() + np.exp(float(f32(sinks)[h]) - m))
        assert np.abs(f32(out)[0, h] - want).max() <= 1e-2


@pytest.mark.parametrize("variant", sorted(MASK_VARIANTS))
@pytest.mark.parametrize("heads,kv_heads,hd,seq,suffix", [(4, 4, 64, 16, 1), (4, 4, 64, 8, 4), (4, 2, 64, 45, 3)])
def test_attention_mask_variants_against_float64(variant, heads, kv_heads, hd, seq, suffix):
    """Pins the restatement of mask.rs:3-61 (non-causal, sliding window causal / centred, ring positions full and
    partially filled, sinks) with an independent float64 statement, on the reference test's procedural inputs and
    head-major strides (attention_single_pass_test.rs:34-131); single-pass and two-pass restatements agree."""
    is_causal, window, ring_params, sinks = mask_case(variant, seq, suffix, heads)
    q, k, v = ref_attention_inputs(heads, kv_heads, seq, suffix, hd)
    scale = 1.0 / np.sqrt(hd)
    a = O.AttentionArgs(q.ctypes.data, k.ctypes.data, v.ctypes.data, O.BF16, hd, heads // kv_heads, seq, seq * hd, hd, seq * hd, hd,
                        1 if ring_params else 0, ring_params[0] if ring_params else 0, ring_params[1] if ring_params else 0, scale,
                        1 if window else 0, window or 0, sinks.ctypes.data if sinks is not None else None, heads, suffix, is_causal)
    got = np.zeros((suffix, heads, hd), np.uint16)
    O.lib().orc_attention_single_pass(C.byref(a), O.p(got))
    want = attention_float64(q, k, v, heads, kv_heads, hd, seq, suffix, seq * hd, hd, np.float32(scale), is_causal, window, ring_params, sinks)
    assert np.abs(f32(got) - want).max() <= 4e-3  # bf16 output rounding of values <= 0.5
    rows = suffix * heads
    wp, ws, wm = np.zeros((rows, 32, hd), np.float32), np.zeros((rows, 32), np.float32), np.zeros((rows, 32), np.float32)
    O.lib().orc_attention_two_pass1(C.byref(a), O.p(wp), O.p(ws), O.p(wm))
    got2 = np.zeros((suffix, heads, hd), np.uint16)
    O.call("orc_attention_two_pass2", wp, ws, wm, got2, O.BF16, hd, heads, suffix)
    assert np.abs(f32(got2) - want).max() <= 4e-3


def test_rope_table_and_attention_prepare():
    """Host RoPE table (rope.rs:13-114): Llama-3 scaling branches + half-rotation pairing and KV scatter."""
    rope = S.D.RopeConfig(kind=S.D.ROPE_LLAMA, head_dim=16, max_sequence_length=8192, base=500000.0, scaling_factor=8.0,
                          original_context_length=64, low_frequency_factor=1.0, high_frequency_factor=4.0).desc()
    pos = np.array([0, 1, 77, 1000], np.uint32)
    cos, sin = np.zeros((4, 16), np.float32), np.zeros((4, 16), np.float32)
    O.lib().orc_rope_tables(C.byref(rope), O.p(pos), C.c_uint32(4), O.p(cos), O.p(sin))
    inv = 1.0 / 500000.0 ** (np.arange(0, 16, 2) / 16.0)
    wl = 2 * np.pi / inv
    smooth = np.clip((64.0 / wl - 1.0) / (4.0 - 1.0), None, None)
    scaled = np.where(wl < 64.0 / 4.0, inv, np.where(wl > 64.0 / 1.0, inv / 8.0, smooth * inv + (1 - smooth) * inv / 8.0))
    want = np.outer(pos.astype(np.float64), scaled)
    assert np.abs(cos[:, :8] - np.cos(want)).max() < 2e-4 and np.abs(sin[:, :8] - np.sin(want)).max() < 2e-4
    assert np.array_equal(cos[:, :8], cos[:, 8:]) and np.array_equal(sin[:, :8], sin[:, 8:])
    # prepare: rotate first rope_dim dims of Q and K only; V untouched; K/V land at kv_token_offset
    rng = np.random.default_rng(4)
    batch, nq, nkv, hd = 4, 2, 1, 32
    qkv = bf16(rng.normal(0, 1, (batch, (nq + 2 * nkv) * hd)))
    queries, keys, values = np.zeros((nq, batch, hd), np.uint16), np.zeros((10, nkv * hd), np.uint16), np.zeros((10, nkv * hd), np.uint16)
    O.call("orc_attention_prepare", qkv, queries, keys, values, cos, sin, nq, nkv, hd, 16, 3, batch, 1)
    x = f32(qkv).reshape(batch, nq + 2 * nkv, hd).astype(np.float64)
    rot = x.copy()
    c, s_ = cos.astype(np.float64)[:, None, :], sin.astype(np.float64)[:, None, :]
    rot[..., :8] = x[..., :8] * c[..., :8] - x[..., 8:16] * s_[..., :8]
    rot[..., 8:16] = x[..., 8:16] * c[..., 8:] + x[..., :8] * s_[..., 8:]
    assert np.abs(f32(queries).transpose(1, 0, 2) - rot[:, :nq]).max() <= 2e-2
    assert np.abs(f32(keys)[3:7].reshape(batch, nkv, hd) - rot[:, nq:nq + nkv]).max() <= 2e-2
    assert np.array_equal(values[3:7].reshape(batch, nkv, hd), qkv.reshape(batch, nq + 2 * nkv, hd)[:, nq + nkv:])
    assert not keys[:3].any() and not keys[7:].any()


def numpy_delta_step(in_proj, a_log, dt_bias, nw, state, Hv, Hk, Dk, Dv, eps):
    """Independent f64 gated delta rule (one token): S <- decay*S + k (beta (v - decay S k))^T, o = S_new q."""
    x = f32(in_proj).astype(np.float64)
    key_dim, value_dim = Hk * Dk, Hv * Dv
    conv_dim = 2 * key_dim + value_dim
    out = np.zeros(value_dim)
    new_state = state.astype(np.float64).copy()
    for hv in range(Hv):
        hk = hv // (Hv // Hk)
        q, k = x[hk * Dk:(hk + 1) * Dk], x[key_dim + hk * Dk: key_dim + (hk + 1) * Dk]
        q = q / np.sqrt((q ** 2).sum() + 1e-6) / np.sqrt(Dk)
        k = k / np.sqrt((k ** 2).sum() + 1e-6)
        v = x[2 * key_dim + hv * Dv: 2 * key_dim + (hv + 1) * Dv]
        z = x[conv_dim + hv * Dv: conv_dim + (hv + 1) * Dv]
        beta = 1 / (1 + np.exp(-x[conv_dim + value_dim + hv]))
        sp = np.log1p(np.exp(x[conv_dim + value_dim + Hv + hv] + dt_bias[hv]))
        decay = np.exp(-np.exp(a_log[hv]) * sp)
        Sd = decay * new_state[hv]                      # [Dv, Dk]
        delta = beta * (v - Sd @ k)
        Sn = Sd + np.outer(delta, k)
        new_state[hv] = Sn
        o = Sn @ q
        o = o / np.sqrt((o ** 2).mean() + eps) * nw * (z / (1 + np.exp(-z)))
        out[hv * Dv:(hv + 1) * Dv] = o
    return out, new_state


def test_delta_net_update_against_float64():
    rng = np.random.default_rng(5)
    Hv, Hk, Dk, Dv = 4, 2, 128, 128
    key_dim, value_dim = Hk * Dk, Hv * Dv
    total = 2 * key_dim + 2 * value_dim + 2 * Hv
    in_proj = bf16(rng.normal(0, 1, total))
    a_log, dt_bias = rng.uniform(-1, 1, Hv).astype(np.float32), rng.uniform(-1, 1, Hv).astype(np.float32)
    nw = (1 + rng.uniform(-0.1, 0.1, Dv)).astype(np.float32)
    state = rng.normal(0, 0.3, (Hv, Dv, Dk)).astype(np.float32)
    st, out = state.copy(), np.zeros(value_dim, np.uint16)
    O.call("orc_delta_net_update", in_proj, a_log, dt_bias, nw, st, out, Hv, Hk, Dk, Dv, key_dim, value_dim, 1e-6)
    want, want_state = numpy_delta_step(in_proj, a_log, dt_bias, nw, state, Hv, Hk, Dk, Dv, 1e-6)
    assert np.abs(st - want_state).max() <= 1e-5
    assert (np.abs(f32(out) - want) <= np.abs(want) * 2.0 ** -7 + 1e-3).all()


def test_delta_net_prefill_path_equals_repeated_decode_steps():
    """conv_pack+conv_scan+prefill_prep+prefill+norm_gate over T tokens == T x (conv_update + update): the two code
    paths of delta_net.rs:505-636 must agree (up to f32 association: `(decay*s)*k` vs `decay*(s.k)`)."""
    rng = np.random.default_rng(6)
    Hv, Hk, Dk, Dv, ks, T = 4, 2, 128, 128, 4, 9
    key_dim, value_dim = Hk * Dk, Hv * Dv
    conv_dim = 2 * key_dim + value_dim
    total = conv_dim + value_dim + 2 * Hv
    in_proj = bf16(rng.normal(0, 1, (T, total)))
    conv_w = rng.uniform(-0.6, 0.6, (conv_dim, ks)).astype(np.float32)
    a_log, dt_bias = rng.uniform(-1, 1, Hv).astype(np.float32), rng.uniform(-1, 1, Hv).astype(np.float32)
    nw = (1 + rng.uniform(-0.1, 0.1, Dv)).astype(np.float32)
    # decode path
    cs, ss, outs = np.zeros((conv_dim, ks - 1), np.float32), np.zeros((Hv, Dv, Dk), np.float32), []
    for t in range(T):
        row, o = in_proj[t].copy(), np.zeros(value_dim, np.uint16)
        O.call("orc_delta_net_conv_update", conv_w, None, row, cs, ks, conv_dim, ks - 1)
        O.call("orc_delta_net_update", row, a_log, dt_bias, nw, ss, o, Hv, Hk, Dk, Dv, key_dim, value_dim, 1e-6)
        outs.append(o)
    # prefill path
    cs2, ss2, ip = np.zeros((conv_dim, ks - 1), np.float32), np.zeros((Hv, Dv, Dk), np.float32), in_proj.copy()
    padded = np.zeros((T + ks - 1, total), np.float32)
    O.call("orc_conv1d_pack", cs2, ip, padded, ks - 1, total, T, conv_dim)
    O.call("orc_delta_net_conv_scan", padded, conv_w, None, ip, cs2, T, ks, total, ks - 1, conv_dim, total)
    qn, kn = np.zeros((T, key_dim), np.float32), np.zeros((T, key_dim), np.float32)
    beta, decay = np.zeros((T, Hv), np.float32), np.zeros((T, Hv), np.float32)
    O.call("orc_delta_net_prefill_prep", ip, a_log, dt_bias, qn, kn, beta, decay, Hv, Hk, Dk, key_dim, value_dim, T)
    out2 = np.zeros((T, value_dim), np.uint16)
    O.call("orc_delta_net_prefill", qn, kn, beta, decay, ip, ss2, out2, Hv, Hk, Dk, Dv, key_dim, value_dim, T)
    O.call("orc_delta_net_norm_gate", out2, ip, nw, Hv, Dv, value_dim, conv_dim, total, 1e-6, T)
    assert np.array_equal(cs, cs2)
    assert np.abs(ss - ss2).max() <= 1e-5
    assert np.abs(f32(np.stack(outs)) - f32(out2)).max() <= 3e-2  # prefill rounds o to bf16 before the norm; decode does not


def test_delta_net_prefill_splits_into_concurrent_segments():
    """The identity behind the split scan of the HIP prefill (k_deltanet_chunk.hip: ScanSplit), on the reference's own recurrence (orc_delta_net_prefill =
    gdn/prefill.rs): the state update is affine in the state with the same linear part for every row, so tokens [mid, T) can run before S_mid is known --
    once from a zero state with the real values (Z, outputs Z_t q_t) and once, as Dk more value rows per head, from the identity with zero values
    (H, outputs u_t = H_t q_t) -- and  S_T = S_mid H_T + Z_T,  o_t = Z_t q_t + S_mid u_t.  The state pieces come from the oracle kernel (f32 states); the
    output identity is checked in float64 (the kernel rounds its outputs to bf16)."""
    rng = np.random.default_rng(21)
    Hv, Hk, Dk, Dv, T, mid = 4, 2, 128, 128, 96, 56
    key_dim, value_dim = Hk * Dk, Hv * Dv
    total = 2 * key_dim + 2 * value_dim + 2 * Hv
    in_proj = bf16(rng.normal(0, 1, (T, total)))
    kn = rng.normal(0, 1, (T, key_dim)).astype(np.float32)
    kn /= np.linalg.norm(kn.reshape(T, Hk, Dk), axis=2).repeat(Dk, axis=1).reshape(T, key_dim)
    qn = (rng.normal(0, 1, (T, key_dim)) / np.sqrt(Dk)).astype(np.float32)
    beta, decay = rng.uniform(0.05, 0.95, (T, Hv)).astype(np.float32), rng.uniform(0.6, 1.0, (T, Hv)).astype(np.float32)
    s0 = rng.normal(0, 0.3, (Hv, Dv, Dk)).astype(np.float32)

    def run(state, rows, ip, dv, vdim):
        st, out = state.copy(), np.zeros((rows.stop - rows.start, vdim), np.uint16)
        O.call("orc_delta_net_prefill", qn[rows].copy(), kn[rows].copy(), beta[rows].copy(), decay[rows].copy(), ip[rows].copy(), st, out, Hv, Hk, Dk, dv, key_dim, vdim,
               rows.stop - rows.start)
        return st

    s_full = run(s0, slice(0, T), in_proj, Dv, value_dim)
    s_mid = run(s0, slice(0, mid), in_proj, Dv, value_dim)
    z_end = run(np.zeros_like(s0), slice(mid, T), in_proj, Dv, value_dim)
    zero_v = np.zeros((T, 2 * key_dim + 2 * Hv * Dk + 2 * Hv), np.uint16)  # the homogeneous rows: Dk "value columns" per head, all values zero
    h_end = run(np.broadcast_to(np.eye(Dk, dtype=np.float32), (Hv, Dk, Dk)).copy(), slice(mid, T), zero_v, Dk, Hv * Dk)
    recombined = np.einsum("hij,hjk->hik", s_mid.astype(np.float64), h_end.astype(np.float64)) + z_end
    assert np.abs(recombined - s_full).max() <= 2e-5 * max(1.0, np.abs(s_full).max())

    # outputs, float64, one head: o_t = Z_t q_t + S_mid (H_t q_t)
    hv, hk = 1, 0
    v = f32(in_proj[:, 2 * key_dim + hv * Dv:2 * key_dim + (hv + 1) * Dv]).astype(np.float64)
    k, q = kn[:, hk * Dk:(hk + 1) * Dk].astype(np.float64), qn[:, hk * Dk:(hk + 1) * Dk].astype(np.float64)

    def scan(state, vals, t0, t1):
        st, outs = state.copy(), []
        for t in range(t0, t1):
            d = beta[t, hv] * (vals[t] - decay[t, hv] * (st @ k[t]))
            st = decay[t, hv] * st + np.outer(d, k[t])
            outs.append(st @ q[t])
        return st, np.array(outs)

    sm, _ = scan(s0[hv].astype(np.float64), v, 0, mid)
    _, o_full = scan(s0[hv].astype(np.float64), v, 0, T)
    _, o_zero = scan(np.zeros((Dv, Dk)), v, mid, T)
    _, u = scan(np.eye(Dk), np.zeros((T, Dk)), mid, T)
    assert np.abs(o_zero + u @ sm.T - o_full[mid:]).max() <= 1e-10


def test_embedding_lookup_and_argmax_rules():
    rng = np.random.default_rng(7)
    vocab, dim = 50, 128
    q = quant_matrix(rng, vocab, dim, 4, 32, 1, scale_mag=1.0)
    ids = np.array([3, 49, 50], np.uint32)
    out = np.zeros((3, dim), np.uint16)
    O.call("orc_quantized_embedding_lookup", ids, q["weights"], q["scales"], q["zero_points"], None, out, O.BF16, 3, vocab, dim, 2.0, 32, 4, 1)
    want = dequantize(q)[[3, 49]] * 2.0
    assert np.abs(f32(out[:2]) - want).max() <= np.abs(want).max() * 2.0 ** -7
    assert not out[2].any()  # out-of-range id -> zeros
    logits = bf16(np.array([[1, 5, 5, 2], [7, 7, 7, 7], [-1, -2, -0.5, -0.5]], np.float32))
    toks = np.zeros(3, np.uint32)
    O.call("orc_argmax", logits, O.BF16, toks, 4, 3)
    assert toks.tolist() == [1, 0, 2]  # ties -> lowest index (unified_sampling.rs:90-95)


@pytest.mark.parametrize("preset", ["tiny-qwen", "tiny-llama"])
def test_model_matches_committed_golden(preset):
    """tests/golden/tiny_models.json (make_golden.py): tokens, logit digests, last-layer digest."""
    gold = json.load(open(os.path.join(GOLDEN, "tiny_models.json")))[preset]
    cfg = S.PRESETS[preset]()
    bundle = S.build_model(cfg)
    for threads in (1, 3):  # results must not depend on the OpenMP team size
        O.set_threads(threads)
        m = O.OracleModel(bundle)
        tok, logits = m.prefill(S.synthetic_prompt(gold["prompt_len"], cfg.vocab_size), True)
        tokens, digests = [tok], [hashlib.sha256(logits.tobytes()).hexdigest()[:16]]
        for _ in range(len(gold["tokens"]) - 1):
            tok, logits = m.forward([tokens[-1]], True)
            tokens.append(tok)
            digests.append(hashlib.sha256(logits.tobytes()).hexdigest()[:16])
        last = hashlib.sha256(m.layer_output(len(bundle.layers) - 1).tobytes()).hexdigest()[:16]
        assert tokens == gold["tokens"] and digests == gold["logit_sha256_16"] and last == gold["last_layer_sha256_16"]
    O.set_threads(O.default_threads())


def test_model_chunked_prefill_equals_token_by_token_for_attention_only_model():
    """Prefill of N tokens in one chunk and decode-style one-token passes give the same KV cache semantics: the
    final greedy token after the same 12 tokens agrees (attention-only model; same kernels, same order per row)."""
    cfg = S.tiny_llama()
    bundle = S.build_model(cfg)
    prompt = S.synthetic_prompt(12, cfg.vocab_size)
    m1, m2 = O.OracleModel(bundle), O.OracleModel(bundle)
    t1, l1 = m1.prefill(prompt, True)
    for t in prompt:
        t2, l2 = m2.forward([int(t)], True)
    assert t1 == t2 and np.array_equal(l1, l2)


def test_byte_accounting_matches_survey():
    """Qwen3.5-0.8B int4 g128 streams ~399 MB of weights and ~0.46 GB per token at context 2048."""
    cfg = S.qwen35_0p8b()
    # shapes only: build a skeletal bundle without allocating 400 MB by using the formula on the config
    d, h, v = cfg.model_dim, cfg.hidden_dim, cfg.vocab_size
    per_w = 0.5 + 4.0 / 128
    attn = (3072 + 2048 + 1024 * 2) * 1024  # qkv + gate + out (2048x1024)
    mlp = 3 * h * d
    dn = (8224 + 2048) * 1024
    weights = (6 * (attn + mlp) + 18 * (dn + mlp) + v * d) * per_w
    assert abs(weights - 399.4e6) / 399.4e6 < 0.01


def test_rope_yarn_and_longrope_tables_against_float64():
    """rope.rs:21-27,60-89: YaRN (NTK-by-parts ramp between the dimensions whose wavelengths fit beta_fast / beta_slow times
    into the original context, magnitude factor 0.1 ln(s) + 1) and LongRoPE (per-pair rescale factors, magnitude
    sqrt(1 + ln s / ln L)), each against a float64 statement of the published formulas (YaRN: Peng et al. 2023 eq. 17-23;
    LongRoPE: Ding et al. 2024)."""
    pos = np.array([0, 3, 500, 40000], np.uint32)
    hd, base, s_, L = 64, 10000.0, 4.0, 4096
    # YaRN
    for truncate in (True, False):
        rope = S.D.RopeConfig(kind=S.D.ROPE_YARN, head_dim=hd, max_sequence_length=16384, base=base, scaling_factor=s_,
                              original_context_length=L, beta_fast=32.0, beta_slow=1.0, truncate=truncate)
        d = rope.desc()
        cos, sin = np.zeros((4, hd), np.float32), np.zeros((4, hd), np.float32)
        O.lib().orc_rope_tables(C.byref(d), O.p(pos), C.c_uint32(4), O.p(cos), O.p(sin))
        i = np.arange(hd // 2, dtype=np.float64)
        inv = 1.0 / base ** (2 * i / hd)

        def corr(beta):
            return hd * np.log(L / (beta * 2 * np.pi)) / (2 * np.log(base))
        low, high = corr(32.0), corr(1.0)
        if truncate:
            low, high = np.floor(low), np.ceil(high)
        low, high = max(low, 0.0), min(high, hd - 1.0)
        ramp = np.clip((i - low) / (high - low), 0.0, 1.0)
        freq = (inv / s_) * ramp + inv * (1.0 - ramp)
        mag = 0.1 * np.log(s_) + 1.0
        ang = np.outer(pos.astype(np.float64), freq)
        # f32 angle products at position 40000: |d angle| ~ 40000 * 2^-24 * freq
        tol = 3e-3
        assert np.abs(cos[:, :hd // 2] - mag * np.cos(ang)).max() < tol and np.abs(sin[:, :hd // 2] - mag * np.sin(ang)).max() < tol
        assert np.array_equal(cos[:, :hd // 2], cos[:, hd // 2:])
    # LongRoPE: long factors when the model's max sequence exceeds the original context, short factors otherwise
    rng = np.random.default_rng(0)
    short, long_ = rng.uniform(1.0, 1.2, hd // 2).astype(np.float32), rng.uniform(1.0, 8.0, hd // 2).astype(np.float32)
    for max_seq, factors in ((131072, long_), (4096, short)):
        rope = S.D.RopeConfig(kind=S.D.ROPE_LONGROPE, head_dim=hd, max_sequence_length=max_seq, base=base, scaling_factor=32.0,
                              original_context_length=L, short_factor=short, long_factor=long_)
        d = rope.desc()
        cos, sin = np.zeros((4, hd), np.float32), np.zeros((4, hd), np.float32)
        O.lib().orc_rope_tables(C.byref(d), O.p(pos), C.c_uint32(4), O.p(cos), O.p(sin))
        inv = 1.0 / base ** (2 * np.arange(hd // 2) / hd) / factors.astype(np.float64)
        mag = np.sqrt(1.0 + np.log(32.0) / np.log(L))
        ang = np.outer(pos.astype(np.float64), inv)
        assert np.abs(cos[:, :hd // 2] - mag * np.cos(ang)).max() < 3e-3 and np.abs(sin[:, :hd // 2] - mag * np.sin(ang)).max() < 3e-3


# ------------------------------------------------------------------------------------------ RHT / A8 (row f1)
def hadamard32_np(v):
    """Sylvester-ordered Walsh-Hadamard transform of 32 values / sqrt(32) -- the closed form of mod.rs:27-47's butterfly."""
    h = np.array([[1.0]])
    while h.shape[0] < 32:
        h = np.block([[h, h], [h, -h]])
    return h @ v / np.sqrt(32.0)


def activation_transform_oracle(x_bits, factors, op, scale_group=0, sum_group=0, in_place=False):
    rows, cols = x_bits.shape
    fp = x_bits.copy() if in_place else np.zeros_like(x_bits)
    q = np.zeros((rows, cols), np.int8)
    sc = np.zeros((rows, max(cols // scale_group, 1) if scale_group else 1), np.float32)
    gs = np.zeros((rows, max(cols // sum_group, 1) if sum_group else 1), np.int32)
    O.lib().orc_activation_transform(None if in_place else O.p(x_bits), O.p(fp), O.p(q), O.p(sc), O.p(gs), O.p(factors), O.BF16, rows, cols, op,
                                     scale_group, sum_group)
    return fp, q, sc, gs


@pytest.mark.parametrize("op", [0, 1])
def test_activation_transform_rht_against_float64(op):
    """InputRht: H (x * s); OutputRht: (H x) * s with the 32-point Sylvester Hadamard matrix / sqrt(32) (activation_transform.rs:
    80-104).  The butterfly's f32 sums against the float64 closed form: <= 1 bf16 ulp after the final rounding; the transform is its
    own inverse up to the sign placement (InputRht then OutputRht with the same factors is NOT the identity; H is)."""
    rng = np.random.default_rng(op)
    rows, cols = 5, 96
    x = bf16(rng.normal(0, 1, (rows, cols)))
    factors = rng.choice(np.array([-1, 1], np.int32), cols)
    fp, _, _, _ = activation_transform_oracle(x, factors, op)
    xf = f32(x).astype(np.float64)
    for r in range(rows):
        for s0 in range(0, cols, 32):
            blk, sg = xf[r, s0:s0 + 32], factors[s0:s0 + 32].astype(np.float64)
            want = hadamard32_np(blk * sg) if op == 0 else hadamard32_np(blk) * sg
            assert np.abs(f32(fp[r, s0:s0 + 32]) - want).max() <= np.abs(want).max() * 2.0 ** -7
    # in place == out of place
    fp2, _, _, _ = activation_transform_oracle(x, factors, op, in_place=True)
    assert np.array_equal(fp, fp2)


@pytest.mark.parametrize("scale_group,sum_group", [(32, 32), (64, 128), (128, 64)])
def test_activation_transform_quantize_against_numpy(scale_group, sum_group):
    """Quantize / QuantizeWithGroupSums (activation_transform.rs:11-41, mod.rs:9-25): divisor = max |v| / 127 per activation group (1.0
    for an all-zero group), codes = round-half-away(v / divisor) clamped to +-127, i32 code sums per sum group."""
    rng = np.random.default_rng(scale_group)
    rows, cols = 4, 256
    x = bf16(rng.normal(0, 2, (rows, cols)))
    x[1, :128] = 0  # an all-zero activation group keeps divisor 1.0
    factors = rng.choice(np.array([-1, 1], np.int32), cols)
    _, q, sc, gs = activation_transform_oracle(x, factors, 3, scale_group, sum_group)
    _, q2, sc2, _ = activation_transform_oracle(x, factors, 2, scale_group, sum_group)
    assert np.array_equal(q, q2) and np.array_equal(sc, sc2)
    xf = f32(x).astype(np.float64)
    t = np.zeros((rows, cols))
    for r in range(rows):
        for s0 in range(0, cols, 32):
            t[r, s0:s0 + 32] = hadamard32_np(xf[r, s0:s0 + 32] * factors[s0:s0 + 32])
    for r in range(rows):
        for g in range(cols // scale_group):
            blk = t[r, g * scale_group:(g + 1) * scale_group]
            mag = np.abs(blk).max()
            div = mag / 127.0 if mag > 0 else 1.0
            assert abs(sc[r, g] - div) <= 1e-6 * max(div, 1e-30) + (0 if mag > 0 else 0)
            codes = q[r, g * scale_group:(g + 1) * scale_group].astype(np.int64)
            assert np.abs(codes).max() <= 127
            # the f32 butterfly can sit one f32 ulp off the float64 value: allow a code to differ by 1 only on an exact .5 boundary
            ideal = blk / float(sc[r, g])
            assert np.abs(codes - ideal).max() <= 0.5 + 1e-3
    for r in range(rows):
        assert np.array_equal(gs[r], q[r].astype(np.int32).reshape(-1, sum_group).sum(axis=1))
    assert sc[1, 0] == 1.0 and not q[1, :scale_group].any()


@pytest.mark.parametrize("ops", [0, 1, 2])
@pytest.mark.parametrize("interleaved", [0, 1])
def test_gated_act_mul_rht_is_gated_product_then_activation_transform(ops, interleaved):
    """gated_act_mul.rs:47-118 (use_hadamard): the restatement of that function equals GatedActMul (FullPrecision) followed by
    ActivationTransform {InputRht, Quantize, QuantizeWithGroupSums} on the bf16 products -- the identity the HIP boundary relies on
    (capi_kernels.hip runs exactly these two kernels) -- and the gated products themselves match float64 to bf16 rounding."""
    rng = np.random.default_rng(ops + 3 * interleaved)
    rows, dim, sg, gg = 5, 256, 64, 128
    if interleaved:
        act_op, val_op, voff, vstride = bf16(rng.normal(0, 1.5, (rows, 2 * dim))), None, 0, 0
        value, gate = act_op[:, :dim], act_op[:, dim:]
    else:
        act_op, val_op, voff, vstride = bf16(rng.normal(0, 1.5, (rows, dim))), bf16(rng.normal(0, 1.5, (rows, dim + 32))), 32, dim + 32
        value, gate = val_op[:, 32:], act_op
    factors = rng.choice(np.array([-1, 1], np.int32), dim)
    wf, wq = np.zeros((rows, dim), np.uint16), np.zeros((rows, dim), np.int8)
    wsc, wgs = np.zeros((rows, dim // sg), np.float32), np.zeros((rows, dim // gg), np.int32)
    O.lib().orc_gated_act_mul_rht(O.p(act_op), O.p(val_op) if val_op is not None else None, O.p(wf) if ops == 0 else None, O.p(wq) if ops else None,
                                  O.p(wsc) if ops else None, O.p(wgs) if ops == 2 else None, O.p(factors), O.BF16, dim, rows, voff, vstride, 0, interleaved,
                                  ops, sg, gg)
    prod = np.zeros((rows, dim), np.uint16)
    O.lib().orc_gated_act_mul(O.p(act_op), O.p(val_op) if val_op is not None else None, O.p(prod), O.BF16, dim, rows, voff, vstride, 0, interleaved)
    g64, v64 = f32(gate).astype(np.float64), f32(value).astype(np.float64)
    ideal = v64 * (g64 / (1.0 + np.exp(-g64)))
    assert np.abs(f32(prod) - ideal).max() <= 2.0 ** -7 * np.abs(ideal).max()  # two bf16 roundings (SiLU, product)
    fp, q, sc, gs = activation_transform_oracle(prod, factors, {0: 0, 1: 2, 2: 3}[ops], sg, gg)
    if ops == 0:
        assert np.array_equal(wf, fp)
    else:
        assert np.array_equal(wq, q) and np.array_equal(wsc, sc)
    if ops == 2:
        assert np.array_equal(wgs, gs)


@pytest.mark.parametrize("bits,method", [(4, 0), (4, 1), (8, 2)])
def test_matmul_int8_activations_and_output_rht_against_float64(bits, method):
    """MatmulA::Int8Symmetric (kernel.rs:190-200): A = q * scale per activation group; MatmulDOps::rht_factors (kernel.rs:296-303):
    OutputRht on D, then the bias.  Against float64 over the dequantised operands."""
    rng = np.random.default_rng(bits + method)
    m, n, k, ag = 3, 64, 256, 64
    qm = quant_matrix(rng, n, k, bits, 128, method)
    a_q = rng.integers(-127, 128, (m, k), dtype=np.int8)
    a_s = rng.uniform(0.001, 0.02, (m, k // ag)).astype(np.float32)
    bias = bf16(rng.normal(0, 0.1, n))
    factors = rng.choice(np.array([-1, 1], np.int32), n)
    d = np.zeros((m, n), np.uint16)
    args = O.MatmulArgs()
    args.a, args.a_dtype = None, O.BF16
    args.a_q, args.a_scales, args.a_group_size = a_q.ctypes.data, a_s.ctypes.data, ag
    args.b, args.scales = qm["weights"].ctypes.data, qm["scales"].ctypes.data
    args.biases = qm["biases"].ctypes.data if qm["biases"] is not None else None
    args.zero_points = qm["zero_points"].ctypes.data if qm["zero_points"] is not None else None
    args.w_dtype, args.method, args.bits, args.group_size = O.BF16, method, bits, 128
    args.d, args.d_dtype, args.ab_scale = d.ctypes.data, O.BF16, 1.0
    args.bias, args.rht_factors = bias.ctypes.data, factors.ctypes.data
    args.m, args.n, args.k = m, n, k
    O.lib().orc_matmul(C.byref(args))
    A = a_q.astype(np.float64) * np.repeat(a_s.astype(np.float64), ag, axis=1)
    plain = A @ dequantize(qm).T
    plain = f32(bf16(plain.astype(np.float32))).astype(np.float64)  # the matmul stores D before the transform
    want = np.zeros((m, n))
    for r in range(m):
        for s0 in range(0, n, 32):
            want[r, s0:s0 + 32] = hadamard32_np(plain[r, s0:s0 + 32]) * factors[s0:s0 + 32]
    want = f32(bf16(want.astype(np.float32))).astype(np.float64) + f32(bias).astype(np.float64)
    assert np.abs(f32(d) - want).max() <= 0.03 * np.abs(want).max()


TRIE_PARENTS = {
    # speculated trees as parent indices in depth-first order (-1 = child of the last accepted token)
    "chain": [-1, 0, 1, 2, 3],
    "two_branches": [-1, 0, 1, 1, 0, 4],
    "bushy": [-1, 0, 1, 2, 1, 0, 5, 5, 7, -1, 9],
}


@pytest.mark.parametrize("tree", sorted(TRIE_PARENTS))
@pytest.mark.parametrize("window", [None, 6])
@pytest.mark.parametrize("heads,kv_heads,hd,prefix", [(4, 2, 64, 19), (4, 4, 64, 0)])
def test_attention_trie_mask_against_float64(tree, window, heads, kv_heads, hd, prefix):
    """mask.rs:21-29 + attention_single_pass.rs:55-61 (is_trie): a speculated tree's tokens as the suffix -- a query sees the prefix and the
    suffix keys on its own root path (trie_start <= q <= trie_end), positions are prefix + height (what the sliding window measures).  The
    independent statement works from parent pointers, not from the flat {start, end, height} encoding; a chain equals the plain causal mask."""
    parents = TRIE_PARENTS[tree]
    suffix = len(parents)
    seq = prefix + suffix
    trie = trie_from_parents(parents)
    q, k, v = ref_attention_inputs(heads, kv_heads, seq, suffix, hd)
    scale = 1.0 / np.sqrt(hd)

    def args(with_trie):
        return O.AttentionArgs(q.ctypes.data, k.ctypes.data, v.ctypes.data, O.BF16, hd, heads // kv_heads, seq, seq * hd, hd, seq * hd, hd, 0, 0, 0, scale,
                               1 if window else 0, window or 0, None, heads, suffix, 1, trie.ctypes.data if with_trie else None)
    a = args(True)
    got = np.zeros((suffix, heads, hd), np.uint16)
    O.lib().orc_attention_single_pass(C.byref(a), O.p(got))
    want = attention_float64(q, k, v, heads, kv_heads, hd, seq, suffix, seq * hd, hd, np.float32(scale), 1, window, None, None, parents=parents)
    assert np.abs(f32(got) - want).max() <= 4e-3
    rows = suffix * heads
    wp, ws, wm = np.zeros((rows, 32, hd), np.float32), np.zeros((rows, 32), np.float32), np.zeros((rows, 32), np.float32)
    O.lib().orc_attention_two_pass1(C.byref(a), O.p(wp), O.p(ws), O.p(wm))
    got2 = np.zeros((suffix, heads, hd), np.uint16)
    O.call("orc_attention_two_pass2", wp, ws, wm, got2, O.BF16, hd, heads, suffix)
    assert np.abs(f32(got2) - want).max() <= 4e-3
    if tree == "chain":
        plain = np.zeros_like(got)
        b = args(False)
        O.lib().orc_attention_single_pass(C.byref(b), O.p(plain))
        assert np.array_equal(plain, got)
    else:
        linear = attention_float64(q, k, v, heads, kv_heads, hd, seq, suffix, seq * hd, hd, np.float32(scale), 1, window, None, None)
        assert np.abs(linear - want).max() > 1e-2  # the tree mask is not the linear one
