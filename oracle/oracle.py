"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (uzu_amd/) must never import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # idle OpenMP workers sleep instead of spinning

BF16, F32 = 0, 2


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("uzu_oracle_kernels.c", "uzu_oracle_tree_verify.c", "uzu_oracle_speculator.c", "uzu_oracle_model.c", "uzu_oracle.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "uzu_model_desc.h"))
    stale = not os.path.exists(_LIB_PATH) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


class MatmulArgs(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("a_dtype", C.c_uint32),
        ("b", C.c_void_p), ("scales", C.c_void_p), ("biases", C.c_void_p), ("zero_points", C.c_void_p),
        ("w_dtype", C.c_uint32), ("method", C.c_uint32), ("bits", C.c_uint32), ("group_size", C.c_uint32),
        ("signed_codes", C.c_uint32), ("b_transpose", C.c_uint32), ("b_leading_dimension", C.c_uint32),
        ("d", C.c_void_p), ("d_dtype", C.c_uint32), ("ab_scale", C.c_float), ("accumulate", C.c_uint32),
        ("bias", C.c_void_p), ("has_soft_cap", C.c_uint32), ("soft_cap", C.c_float),
        ("gather_indices", C.c_void_p), ("m", C.c_uint32), ("n", C.c_uint32), ("k", C.c_uint32),
        ("a_q", C.c_void_p), ("a_scales", C.c_void_p), ("a_group_size", C.c_uint32), ("rht_factors", C.c_void_p),
    ]


class NormArgs(C.Structure):
    _fields_ = [
        ("input", C.c_void_p), ("scales", C.c_void_p), ("biases", C.c_void_p), ("output", C.c_void_p),
        ("shortcut", C.c_void_p), ("io_dtype", C.c_uint32), ("affine_dtype", C.c_uint32),
        ("batch_size", C.c_uint32), ("element_count", C.c_uint32),
        ("epsilon", C.c_float), ("scale_offset", C.c_float), ("post_layer_scalar", C.c_float),
        ("subtract_mean", C.c_uint32), ("full_layer", C.c_uint32), ("copy_to_shortcut", C.c_uint32),
        ("residual_add", C.c_uint32), ("scale_residual_sum", C.c_uint32), ("scale_output", C.c_uint32),
    ]


class AttentionArgs(C.Structure):
    _fields_ = [
        ("queries", C.c_void_p), ("keys", C.c_void_p), ("values", C.c_void_p), ("dtype", C.c_uint32),
        ("head_dim", C.c_uint32), ("gqa_factor", C.c_uint32), ("sequence_length", C.c_uint32),
        ("k_head_stride", C.c_uint32), ("k_seq_stride", C.c_uint32), ("v_head_stride", C.c_uint32),
        ("v_seq_stride", C.c_uint32),
        ("is_kv_cache_ring", C.c_uint32), ("ring_offset", C.c_uint32), ("ring_length", C.c_uint32),
        ("scale", C.c_float), ("is_sliding_window", C.c_uint32), ("sliding_window_size", C.c_uint32),
        ("sinks", C.c_void_p), ("num_heads", C.c_uint32), ("suffix_length", C.c_uint32), ("is_causal", C.c_uint32),
        ("trie", C.c_void_p),  # uint32 [suffix, 3] = {trie_start, trie_end, height} per suffix token, or None
    ]


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_model_create.restype = C.c_void_p
        _lib.orc_model_create.argtypes = [C.c_void_p]
        _lib.orc_model_destroy.argtypes = [C.c_void_p]
        _lib.orc_model_reset.argtypes = [C.c_void_p]
        _lib.orc_model_context_length.argtypes = [C.c_void_p]
        _lib.orc_model_context_length.restype = C.c_uint32
        _lib.orc_model_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        _lib.orc_model_forward.restype = C.c_uint32
        _lib.orc_model_layer_output.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        _lib.orc_model_layer_output.restype = C.c_void_p
        _lib.orc_model_final_hidden.argtypes = [C.c_void_p]
        _lib.orc_model_final_hidden.restype = C.c_void_p
        _lib.orc_f32_to_bf16.argtypes = [C.c_float]
        _lib.orc_f32_to_bf16.restype = C.c_uint16
        _lib.orc_activate.argtypes = [C.c_uint32, C.c_float, C.c_uint32]
        _lib.orc_activate.restype = C.c_float
        _lib.orc_get_max_threads.restype = C.c_int
        _lib.orc_set_threads(C.c_int(default_threads()))
    return _lib


def default_threads() -> int:
    """Threads the oracle may use: CPU affinity, capped by the cgroup CPU quota and by 16.  (A GPU box can
    expose 256 logical CPUs to a container that is only allowed a few: an OpenMP team of 256 spinning threads
    then makes every parallel region crawl.)  Results never depend on the thread count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def set_threads(n: int) -> None:
    lib().orc_set_threads(C.c_int(n))


def p(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle needs contiguous arrays"
    return C.c_void_p(a.ctypes.data)


def call(name: str, *args):
    """Call a void oracle kernel; numpy arrays are passed as pointers, ints as u32, floats as f32."""
    conv = []
    for a in args:
        if isinstance(a, np.ndarray):
            conv.append(p(a))
        elif a is None:
            conv.append(None)
        elif isinstance(a, (float, np.floating)):
            conv.append(C.c_float(float(a)))
        elif isinstance(a, (bool, int, np.integer)):
            conv.append(C.c_uint32(int(a)))
        else:
            conv.append(a)
    getattr(lib(), name)(*conv)


class OracleModel:
    """orc_model_* : Decoder::encode + greedy sampling + accept for one sequence."""

    def __init__(self, bundle):
        self.bundle = bundle
        self._desc = bundle.desc()
        self._h = lib().orc_model_create(C.byref(self._desc))
        self.vocab_size = bundle.vocab_size
        self.model_dim = bundle.model_dim

    def close(self):
        if self._h:
            lib().orc_model_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def reset(self):
        lib().orc_model_reset(self._h)

    @property
    def context_length(self) -> int:
        return lib().orc_model_context_length(self._h)

    def fill_synthetic_context(self, n: int):
        """Timing aid: jump to context length n with synthetic KV rows (no prefill); see uzu_oracle_model.c."""
        lib().orc_model_fill_synthetic_context.argtypes = [C.c_void_p, C.c_uint32]
        lib().orc_model_fill_synthetic_context(self._h, C.c_uint32(n))

    def forward(self, tokens, want_logits: bool = False):
        tokens = np.ascontiguousarray(tokens, dtype=np.uint32)
        logits = np.empty(self.vocab_size, dtype=np.uint16) if want_logits else None
        tok = lib().orc_model_forward(self._h, p(tokens), C.c_uint32(tokens.size), p(logits))
        return (int(tok), logits) if want_logits else int(tok)

    def prefill(self, tokens, want_logits: bool = False):
        """LanguageModelStream::new: chunks of <= 1024 tokens (stream.rs:194-195)."""
        tokens = np.ascontiguousarray(tokens, dtype=np.uint32)
        out = None
        for s in range(0, tokens.size, 1024):
            out = self.forward(tokens[s:s + 1024], want_logits)
        return out

    def verify_tree(self, tokens, trie, want_logits: bool = False):
        """One NOT-accepted pass over a speculated tree (DFS order; trie uint32 [n, 3] = {start, end, height} per node):
        -> greedy token of every node [, logits bf16 [n, vocab]].  Follow with accept()."""
        tokens = np.ascontiguousarray(tokens, dtype=np.uint32)
        trie = np.ascontiguousarray(trie, dtype=np.uint32).reshape(tokens.size, 3)
        sampled = np.empty(tokens.size, dtype=np.uint32)
        logits = np.empty((tokens.size, self.vocab_size), dtype=np.uint16) if want_logits else None
        fn = lib().orc_model_verify_tree
        fn.restype, fn.argtypes = None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        fn(self._h, p(tokens), p(trie), C.c_uint32(tokens.size), p(sampled), p(logits))
        return (sampled, logits) if want_logits else sampled

    def accept(self, indices):
        """TransformerState::encode_accept with a root path of the pending tree (FlatTrie::accept's indices)."""
        indices = np.ascontiguousarray(indices, dtype=np.uint32)
        fn = lib().orc_model_accept
        fn.restype, fn.argtypes = None, [C.c_void_p, C.c_void_p, C.c_uint32]
        fn(self._h, p(indices), C.c_uint32(indices.size))

    def layer_output(self, layer: int) -> np.ndarray:
        rows = C.c_uint32(0)
        ptr = lib().orc_model_layer_output(self._h, C.c_uint32(layer), C.byref(rows))
        n = rows.value * self.model_dim
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint16)), shape=(n,)).copy().reshape(rows.value, -1)

    # ---- speculator taps (stream.rs:213-214,632-633)
    def capture_features(self, on: bool = True):
        lib().orc_model_capture_features.argtypes = [C.c_void_p, C.c_uint32]
        lib().orc_model_capture_features(self._h, C.c_uint32(1 if on else 0))

    def hidden_feature(self, layer: int) -> np.ndarray:
        """capture_residual(shortcut, hidden) of `layer` for every row of the last forward: bf16 bits [rows, model_dim]"""
        fn = lib().orc_model_hidden_feature
        fn.restype, fn.argtypes = C.c_void_p, [C.c_void_p, C.c_uint32, C.c_void_p]
        rows = C.c_uint32(0)
        ptr = fn(self._h, C.c_uint32(layer), C.byref(rows))
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint16)), shape=(rows.value * self.model_dim,)).copy().reshape(rows.value, -1)

    def final_hidden_rows(self) -> np.ndarray:
        """output norm of every output row of the last forward (DecoderEncodeOutput::final_hidden): bf16 bits [rows, model_dim]"""
        fn = lib().orc_model_final_hidden_rows
        fn.restype, fn.argtypes = C.c_void_p, [C.c_void_p, C.c_void_p]
        rows = C.c_uint32(0)
        ptr = fn(self._h, C.byref(rows))
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint16)), shape=(rows.value * self.model_dim,)).copy().reshape(rows.value, -1)

    def final_hidden(self) -> np.ndarray:
        ptr = lib().orc_model_final_hidden(self._h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint16)), shape=(self.model_dim,)).copy()


class OracleDFlash:
    """orc_dflash_*: the DFlash draft model (encodable_block/dflash.rs:41-346) next to an OracleModel target."""

    def __init__(self, bundle):
        self.bundle = bundle
        self._desc = bundle.desc()
        fn = lib().orc_dflash_create
        fn.restype, fn.argtypes = C.c_void_p, [C.c_void_p]
        self._h = fn(C.byref(self._desc))
        self.block_size, self.model_dim, self.vocab_size = bundle.block_size, bundle.model_dim, bundle.vocab_size
        self.target_layer_ids = list(bundle.target_layer_ids)

    def close(self):
        if self._h:
            lib().orc_dflash_destroy.argtypes = [C.c_void_p]
            lib().orc_dflash_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def reset(self):
        lib().orc_dflash_reset.argtypes = [C.c_void_p]
        lib().orc_dflash_reset(self._h)

    @property
    def context_length(self) -> int:
        fn = lib().orc_dflash_context_length
        fn.restype, fn.argtypes = C.c_uint32, [C.c_void_p]
        return int(fn(self._h))

    def accept(self, target_features, accepted_indices):
        """DFlash::encode_accept: target_features = one bf16-bits array [rows, model_dim] per tapped target layer (in target_layer_ids order)"""
        feats = [np.ascontiguousarray(f, dtype=np.uint16) for f in target_features]
        assert len(feats) == len(self.target_layer_ids)
        idx = np.ascontiguousarray(accepted_indices, dtype=np.uint32)
        assert all(f.ndim == 2 and f.shape[1] == self.model_dim and (idx.size == 0 or int(idx.max()) < f.shape[0]) for f in feats)
        ptrs = (C.c_void_p * len(feats))(*[f.ctypes.data for f in feats])
        fn = lib().orc_dflash_accept
        fn.restype, fn.argtypes = None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        fn(self._h, ptrs, p(idx), C.c_uint32(idx.size))

    def draft(self, target: "OracleModel", target_output_token: int, batch_size: int):
        """DFlash::encode_draft + greedy tokens of the lookahead rows -> (draft_hidden bf16 bits [batch, d], logits f32 [batch - 1, vocab], tokens [batch - 1])"""
        hidden = np.empty((batch_size, self.model_dim), dtype=np.uint16)
        logits = np.empty((batch_size - 1, self.vocab_size), dtype=np.float32)
        tokens = np.empty(batch_size - 1, dtype=np.uint32)
        fn = lib().orc_dflash_draft
        fn.restype, fn.argtypes = None, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        fn(self._h, target._h, C.c_uint32(int(target_output_token)), C.c_uint32(batch_size), p(hidden), p(logits), p(tokens))
        return hidden, logits, tokens


class OracleWeaver:
    """orc_weaver_*: the Weaver tree constructor (encodable_block/weaver.rs:166-676) next to an OracleModel target."""

    def __init__(self, bundle):
        self.bundle = bundle
        self._desc = bundle.desc()
        fn = lib().orc_weaver_create
        fn.restype, fn.argtypes = C.c_void_p, [C.c_void_p]
        self._h = fn(C.byref(self._desc))
        self.max_depth = bundle.max_depth

    def close(self):
        if self._h:
            lib().orc_weaver_destroy.argtypes = [C.c_void_p]
            lib().orc_weaver_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def encode_tree(self, target: "OracleModel", target_hidden, draft_hidden, logits, depth_seeds, root_token_id: int, shape):
        """-> (packed_tree u32 [6, slots], frontier u32 [7, slots * expand_width]), or None for WeaverEncodeError::InvalidTreeInput"""
        target_hidden = np.ascontiguousarray(target_hidden, dtype=np.uint16)
        draft_hidden = np.ascontiguousarray(draft_hidden, dtype=np.uint16)
        logits = np.ascontiguousarray(logits, dtype=np.float32)
        seeds = np.ascontiguousarray(depth_seeds, dtype=np.uint64)
        slots = shape.slot_count()
        packed = np.zeros((6, slots), dtype=np.uint32)
        frontier = np.zeros((7, slots * max(shape.expand_width, 1)), dtype=np.uint32)
        fn = lib().orc_weaver_encode_tree
        fn.restype = C.c_int32
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        rc = fn(self._h, target._h, p(target_hidden), p(draft_hidden), p(logits), p(seeds), C.c_uint32(seeds.size), C.c_uint32(int(root_token_id)), C.byref(shape), p(packed), p(frontier))
        return None if rc else (packed, frontier)
