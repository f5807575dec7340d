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


def ref_attention_inputs(heads, kv_heads, seq, suffix, hd):
    """The reference test's own procedural inputs and layout (attention_single_pass_test.rs:34-79): queries
    [heads * suffix, hd] = sin(0.13 i + 0.5) / 2, keys / values HEAD-major [kv_heads, seq, hd] = cos(0.07 i + 1) / 2 and
    sin(0.11 i + 2) / 2, evaluated in f32 and rounded to bf16; k_head_stride = seq * hd, k_seq_stride = hd."""
    def gen(n, a, b, fn):
        i = np.arange(n, dtype=np.float32)
        return bf16(fn(i * np.float32(a) + np.float32(b)).astype(np.float32) * np.float32(0.5))
    q = gen(heads * suffix * hd, 0.13, 0.5, np.sin)
    k = gen(kv_heads * seq * hd, 0.07, 1.0, np.cos)
    v = gen(kv_heads * seq * hd, 0.11, 2.0, np.sin)
    return q, k, v


MASK_VARIANTS = {
    # name: (is_causal, sliding window or None, (ring_offset, ring_length) or None, sinks)
    "non_causal": (0, None, None, False),
    "causal": (1, None, None, False),
    "sliding_causal": (1, 9, None, False),
    "sliding_non_causal": (0, 10, None, False),
    "ring_full": (1, None, (5, None), False),        # ring_length = prefix length (every slot holds a token)
    "ring_partial": (1, None, (0, 0.6), False),      # 60 % of the slots filled
    "ring_sliding": (1, 12, (7, None), False),
    "sinks": (1, None, None, True),
    "sinks_sliding": (1, 9, None, True),
}


def mask_case(variant, seq, suffix, heads, window_scale=1, ring_scale=1):
    """(is_causal, window, ring_params, sinks_bits) of a MASK_VARIANTS entry for a concrete shape."""
    is_causal, window, ring, has_sinks = MASK_VARIANTS[variant]
    prefix = seq - suffix
    ring_params = None
    if ring is not None and prefix > 0:
        length = prefix if ring[1] is None else max(1, int(prefix * ring[1]))
        ring_params = ((ring[0] * ring_scale) % prefix, length)
    sinks = bf16(np.linspace(-1.0, 2.0, heads)) if has_sinks else None
    return is_causal, (window * window_scale if window else None), ring_params, sinks


def trie_from_parents(parents):
    """Flat trie buffer (gpu_types/trie.rs: {trie_start, trie_end, height} per node) of a speculated tree given as parent indices in
    depth-first order (parents[i] < i, -1 = a root): a node's subtree is the index range [i, trie_end]."""
    n = len(parents)
    height = [0] * n
    end = list(range(n))
    for i in range(n):
        assert parents[i] < i
        if parents[i] >= 0:
            height[i] = height[parents[i]] + 1
    for i in range(n - 1, -1, -1):
        if parents[i] >= 0:
            assert end[parents[i]] <= end[i] or end[parents[i]] >= i, "parents must be in depth-first order"
            end[parents[i]] = max(end[parents[i]], end[i])
    return np.array([[i, end[i], height[i]] for i in range(n)], np.uint32)


def attention_float64(q, k, v, heads, kv_heads, hd, seq, suffix, k_head_stride, k_seq_stride, scale, is_causal, window, ring_params, sinks, parents=None):
    """Independent float64 statement of attention_single_pass.rs:37-127 + mask.rs:3-61: returns [suffix, heads, hd].  `parents` (a speculated
    tree as parent indices): a query sees the suffix keys on its own root path, every token sits at prefix + its depth."""
    depth, anc = None, None
    if parents is not None:
        depth = [0] * suffix
        anc = [set([i]) for i in range(suffix)]
        for i in range(suffix):
            if parents[i] >= 0:
                depth[i] = depth[parents[i]] + 1
                anc[i] |= anc[parents[i]]
    qf, kf, vf = f32(q).astype(np.float64), f32(k).astype(np.float64), f32(v).astype(np.float64)
    prefix = seq - suffix
    suffix_position = ring_params[1] if ring_params else prefix
    out = np.zeros((suffix, heads, hd))
    gqa = heads // kv_heads
    for h in range(heads):
        kvh = h // gqa
        for qi in range(suffix):
            qv = qf.reshape(-1)[(h * suffix + qi) * hd:(h * suffix + qi + 1) * hd] * scale
            query_position = suffix_position + (depth[qi] if parents is not None else qi)
            scores, vals = [], []
            for i in range(seq):
                use = True
                if i >= prefix and parents is not None:
                    key_position = suffix_position + depth[i - prefix]
                    if is_causal:
                        use &= (i - prefix) in anc[qi]
                elif i >= prefix:
                    key_position = suffix_position + (i - prefix)
                    if is_causal:
                        use &= (i - prefix) <= qi
                else:
                    if ring_params:
                        key_position = (prefix + i - ring_params[0]) % prefix
                        use &= key_position < ring_params[1]
                    else:
                        key_position = i
                if window:
                    if is_causal:
                        use &= key_position <= query_position and (query_position - key_position) < window
                    elif key_position <= query_position:
                        use &= (query_position - key_position) <= window // 2
                    else:
                        use &= (key_position - query_position) <= window // 2
                if not use:
                    continue
                base = kvh * k_head_stride + i * k_seq_stride
                scores.append(float(qv @ kf.reshape(-1)[base:base + hd]))
                vals.append(vf.reshape(-1)[base:base + hd])
            sink = float(f32(sinks)[h]) if sinks is not None else None
            if not scores and sink is None:
                continue
            m = max(scores + ([sink] if sink is not None else []))
            p = np.exp(np.array(scores) - m) if scores else np.zeros(0)
            denom = p.sum() + (np.exp(sink - m) if sink is not None else 0.0)
            out[qi, h] = (p @ np.array(vals)) / denom if scores else 0.0
    return out


class OracleTarget:
    """The `target` duck type of uzu_amd.speculator over the CPU oracle (the GPU tests pair it with HipModel + HipDrafter)."""

    def __init__(self, om, layer_ids):
        self.om, self.layer_ids = om, list(layer_ids)
        om.capture_features(True)

    @property
    def context_length(self):
        return self.om.context_length

    def prefill(self, tokens):
        return self.om.prefill(tokens)

    def hidden_features(self):
        return [self.om.hidden_feature(l) for l in self.layer_ids]

    def verify_tree(self, token_ids, nodes, seeds=None):
        return self.om.verify_tree(token_ids, nodes)

    def final_hidden_rows(self):
        return self.om.final_hidden_rows()

    def accept(self, indices):
        self.om.accept(indices)

    # OracleDFlash.draft wants the OracleModel
    @property
    def _h(self):
        return self.om._h


