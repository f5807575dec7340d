"""Shared helpers for the parity tests: bf16 conversion, quantised-matrix generators in the reference's
layouts, and error metrics expressed in bf16 ulps."""
import numpy as np

from uzu_amd.synthetic import bf16_bits_to_f32, f32_to_bf16_bits  # noqa: F401


def bf16(x):
    """float array -> bf16 bit patterns (uint16), half::bf16::from_f32 rounding."""
    return f32_to_bf16_bits(np.asarray(x, dtype=np.float32))


def f32(b):
    return bf16_bits_to_f32(np.asarray(b, dtype=np.uint16))


def ulp_diff_bf16(a_bits, b_bits, floor=None):
    """|a-b| measured in bf16 ulps of the larger magnitude (0 when bit-identical).

    `floor`: magnitude below which the ulp is not shrunk any further.  Outputs of a reduction that happen to
    cancel to ~0 carry the absolute f32 accumulation noise of the whole sum, so their error is judged against
    the ulp of a typical output (default: rms(a)/8), not against their own tiny ulp."""
    a, b = f32(a_bits).astype(np.float64), f32(b_bits).astype(np.float64)
    if floor is None:
        floor = float(np.sqrt(np.mean(a * a))) / 8.0 if a.size else 0.0
    mag = np.maximum(np.maximum(np.abs(a), np.abs(b)), floor)
    exp = np.floor(np.log2(np.maximum(mag, 1e-30)))
    ulp = 2.0 ** (exp - 7)
    return np.abs(a - b) / ulp


def quant_matrix(rng, n, k, bits, group_size, method, scale_mag=None):
    """Random quantised [n,k] matrix in the reference layout; returns dict of numpy arrays.
    Value ranges follow crates/backend-uzu/src/tests/matmul/quant.rs:58-110 (rescaled by 1/sqrt(k))."""
    groups = (k + group_size - 1) // group_size
    if bits == 4:
        codes = rng.integers(0, 256, size=(n, k // 2), dtype=np.uint8)
    else:
        codes = rng.integers(0, 256, size=(n, k), dtype=np.uint8)
    mag = scale_mag if scale_mag is not None else 1.0 / np.sqrt(k)
    scales = bf16(rng.uniform(0.01, 0.3, size=(n, groups)) * mag * 8)
    out = dict(n=n, k=k, bits=bits, group_size=group_size, method=method, weights=codes, scales=scales, biases=None, zero_points=None)
    if method == 0:
        out["biases"] = bf16(-((1 << (bits - 1)) - 0.5) * f32(scales) + rng.uniform(-0.03, 0.03, size=(n, groups)) * mag)
    elif method == 1:
        if bits == 4:
            zp = rng.integers(0, 16, size=(n, (groups + 1) // 2 * 2), dtype=np.uint8)
            if groups % 2:
                zp[:, -1] = 0
            out["zero_points"] = np.ascontiguousarray(zp[:, 0::2] | (zp[:, 1::2] << 4)).astype(np.uint8)
        else:
            out["zero_points"] = rng.integers(0, 256, size=(n, groups), dtype=np.uint8)
    return out


def dequantize(q):
    """float64 dense [n,k] matrix of a quant_matrix() dict (independent of the oracle's code path)."""
    n, k, bits, g = q["n"], q["k"], q["bits"], q["group_size"]
    if bits == 4:
        w = q["weights"]
        codes = np.empty((n, k), dtype=np.float64)
        codes[:, 0::2] = w & 0x0F
        codes[:, 1::2] = w >> 4
    else:
        codes = q["weights"].astype(np.float64)
    groups = (k + g - 1) // g
    s = np.repeat(f32(q["scales"]).astype(np.float64), g, axis=1)[:, :k]
    if q["method"] == 0:
        b = np.repeat(f32(q["biases"]).astype(np.float64), g, axis=1)[:, :k]
    elif q["method"] == 1:
        if bits == 4:
            zp = np.empty((n, ((groups + 1) // 2) * 2), dtype=np.float64)
            zp[:, 0::2] = q["zero_points"] & 0x0F
            zp[:, 1::2] = q["zero_points"] >> 4
            zp = zp[:, :groups]
        else:
            zp = q["zero_points"].astype(np.float64)
        b = -np.repeat(f32(q["scales"]).astype(np.float64) * zp, g, axis=1)[:, :k]
    else:
        b = -s * (1 << (bits - 1))
    return s * codes + b
