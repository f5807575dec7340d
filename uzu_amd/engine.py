"""Python handle on the C++ model driver (include/uzu_hip_engine.h): HipModel.

Plays the role of ``LanguageModel`` + ``LanguageModelState`` + ``LanguageModelStream`` for one sequence
(crates/backend-uzu/src/engine/language_model/{mod.rs,state.rs,stream/stream.rs}): prefill in chunks of
<= 1024 tokens, then chained greedy decode.  All compute runs in libuzu_hip.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _ffi
from ._ffi import call
from .backend import Context
from .desc import ModelBundle

MODEL_DEFAULT, MODEL_NO_GRAPH, MODEL_NO_FUSION, MODEL_DEBUG_TAPS = 0, 1, 2, 4


def check_desc_abi():
    """The description structs are part of the ABI (uzu_layer_desc is an array element and has grown in place): the ctypes mirrors of uzu_amd/desc.py must have the
    sizes the loaded library was built with (uzu_hip_desc_abi)."""
    from . import desc as D
    got = (C.c_uint32 * 6)()
    fn = _ffi.lib().uzu_hip_desc_abi
    fn.restype, fn.argtypes = None, [C.c_void_p]
    fn(got)
    want = [C.sizeof(t) for t in (D.LinearDesc, D.NormDesc, D.RopeDesc, D.LayerDesc, D.ModelDesc, D.DFlashDesc)]
    if list(got) != want:
        raise RuntimeError(f"libuzu_hip.so was built with description structs of {list(got)} bytes, uzu_amd/desc.py mirrors {want}: rebuild the library (one header, one library)")


def MODEL_BATCH(n: int) -> int:
    """flags bits 8..15: sequences one batched prefill pass may carry (UZU_MODEL_BATCH)."""
    return (int(n) & 0xFF) << 8


class HipState:
    """One sequence's state (LanguageModelState, engine/language_model/state.rs:9-16): KV caches, DeltaNet conv / SSM
    states, token history, decode graphs.  HipModel.bind(state) makes the model work on it."""

    def __init__(self, model: "HipModel"):
        self.model = model
        self._h = C.c_void_p()
        call("uzu_hip_state_create", model._h, C.byref(self._h))

    def close(self):
        if self._h and self.model._h:
            fn = _ffi.lib().uzu_hip_state_destroy
            fn.restype, fn.argtypes = None, [C.c_void_p]
            fn(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        call("uzu_hip_state_reset", self._h)

    def copy_from(self, other: "HipState"):
        """self <- other (uzu_hip_state_copy): caches, recurrent states, token history, context length."""
        call("uzu_hip_state_copy", self._h, other._h)

    @property
    def context_length(self) -> int:
        fn = _ffi.lib().uzu_hip_state_context_length
        fn.restype, fn.argtypes = C.c_uint32, [C.c_void_p]
        return fn(self._h)


class HipModel:
    def __init__(self, ctx: Context, bundle: ModelBundle, flags: int = MODEL_DEFAULT, tp_group=None, vocab_offset: int = 0):
        """`tp_group` (uzu_amd.tp.TpGroup) + `vocab_offset`: `bundle` is this rank's shard from uzu_amd.tp.shard_bundle."""
        self.ctx = ctx
        self.vocab_size = bundle.vocab_size
        self.model_dim = bundle.model_dim
        self.num_layers = len(bundle.layers)
        self.tp_group = tp_group
        check_desc_abi()
        desc = bundle.desc()
        self._h = C.c_void_p()
        if tp_group is None:
            call("uzu_hip_model_create", ctx._h, C.byref(desc), C.c_uint32(flags), C.byref(self._h))
        else:
            call("uzu_hip_model_create_tp", ctx._h, C.byref(desc), C.c_uint32(flags), tp_group._h, C.c_uint32(vocab_offset), C.byref(self._h))
        self.logit_count = int(_ffi.lib().uzu_hip_model_logit_count(self._h))

    def close(self):
        if self._h:
            _ffi.lib().uzu_hip_model_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            if self.ctx._h:
                self.close()
        except Exception:
            pass

    def reset(self):
        call("uzu_hip_model_reset", self._h)

    # ---- sequence states ----
    def new_state(self) -> HipState:
        return HipState(self)

    def bind(self, state: Optional[HipState]):
        """Work on `state` from now on (None: the model's own state)."""
        call("uzu_hip_model_bind_state", self._h, state._h if state is not None else None)

    def prefill_batch(self, states, tokens) -> np.ndarray:
        """`tokens` [len(states), count]: prefill every state with its row in one batched pass per chunk; returns the first
        sampled token of every sequence."""
        tokens = np.ascontiguousarray(tokens, dtype=np.uint32)
        assert tokens.ndim == 2 and tokens.shape[0] == len(states)
        handles = (C.c_void_p * len(states))(*[st._h for st in states])
        first = np.zeros(len(states), dtype=np.uint32)
        call("uzu_hip_model_prefill_batch", self._h, handles, C.c_uint32(len(states)), C.c_void_p(tokens.ctypes.data), C.c_uint32(tokens.shape[1]),
             C.c_void_p(first.ctypes.data))
        return first

    @property
    def context_length(self) -> int:
        return _ffi.lib().uzu_hip_model_context_length(self._h)

    @property
    def weight_bytes(self) -> int:
        return _ffi.lib().uzu_hip_model_weight_bytes(self._h)

    @property
    def decode_launch_count(self) -> int:
        return _ffi.lib().uzu_hip_model_decode_launch_count(self._h)

    def prefill(self, tokens) -> int:
        tokens = np.ascontiguousarray(tokens, dtype=np.uint32)
        first = C.c_uint32()
        call("uzu_hip_model_prefill", self._h, C.c_void_p(tokens.ctypes.data), C.c_uint32(tokens.size), C.byref(first))
        return first.value

    def decode(self, steps: int) -> Tuple[np.ndarray, float]:
        out = np.empty(steps, dtype=np.uint32)
        ms = C.c_float()
        call("uzu_hip_model_decode", self._h, C.c_uint32(steps), C.c_void_p(out.ctypes.data), C.byref(ms))
        return out, ms.value

    def decode_enqueue(self, steps: int):
        call("uzu_hip_model_decode_enqueue", self._h, C.c_uint32(steps))

    def read_tokens(self, first_position: int, count: int) -> np.ndarray:
        out = np.empty(count, dtype=np.uint32)
        call("uzu_hip_model_read_tokens", self._h, C.c_uint32(first_position), C.c_uint32(count), C.c_void_p(out.ctypes.data))
        return out

    def set_sampling(self, seed: Optional[int] = None, temperature: Optional[float] = None, top_k: Optional[int] = None, top_p: Optional[float] = None,
                     min_p: Optional[float] = None):
        """SamplingMethod::Stochastic { temperature, top_k, top_p, min_p } with PRng::new(seed) (stream.rs:248-258, 598-600);
        seed None = back to greedy."""
        if seed is None:
            call("uzu_hip_model_set_sampling", self._h, None)
            return

        class Cfg(C.Structure):
            _fields_ = [("seed", C.c_uint64), ("has_temperature", C.c_uint32), ("temperature", C.c_float), ("has_top_k", C.c_uint32), ("top_k", C.c_uint32),
                        ("has_top_p", C.c_uint32), ("top_p", C.c_float), ("has_min_p", C.c_uint32), ("min_p", C.c_float)]
        cfg = Cfg(int(seed) & (2 ** 64 - 1), int(temperature is not None), float(temperature or 0.0), int(top_k is not None), int(top_k or 0),
                  int(top_p is not None), float(top_p or 0.0), int(min_p is not None), float(min_p or 0.0))
        call("uzu_hip_model_set_sampling", self._h, C.byref(cfg))

    def set_next_token(self, token: int):
        call("uzu_hip_model_set_next_token", self._h, C.c_uint32(int(token)))

    def read_logits(self) -> np.ndarray:
        out = np.empty(self.logit_count, dtype=np.uint16)
        call("uzu_hip_model_read_logits", self._h, C.c_void_p(out.ctypes.data))
        return out

    # ---- speculative decoding (stream.rs:380-470, 556-628; host trie: uzu_amd/trie.py) ----
    def verify_tree(self, token_ids, trie_nodes, seeds=None) -> np.ndarray:
        """One forward pass over a speculated tree (DFS order; trie_nodes uint32 [n, 3] = {start, end, height}) hanging off the
        sequence; nothing is accepted.  -> the token sampled at every node (greedy, or with set_sampling the node's own draw).
        `seeds` (uint64 [n], FlatTrie.token_seeds(): the seeds the speculator gave its nodes, stream.rs:694) replace the default
        PRng::derive(context + height) per node under stochastic sampling."""
        token_ids = np.ascontiguousarray(token_ids, dtype=np.uint32)
        trie_nodes = np.ascontiguousarray(trie_nodes, dtype=np.uint32).reshape(token_ids.size, 3)
        sampled = np.empty(token_ids.size, dtype=np.uint32)
        if seeds is not None:
            seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
            assert seeds.size == token_ids.size
        call("uzu_hip_model_verify_tree_seeded", self._h, C.c_void_p(token_ids.ctypes.data), C.c_void_p(trie_nodes.ctypes.data),
             C.c_void_p(seeds.ctypes.data) if seeds is not None else None, C.c_uint32(token_ids.size), C.c_void_p(sampled.ctypes.data))
        self._tree_size = int(token_ids.size)
        return sampled

    @property
    def verify_gpu_ms(self) -> float:
        out = C.c_float()
        call("uzu_hip_model_verify_gpu_ms", self._h, C.byref(out))
        return out.value

    def accept(self, indices):
        """encode_accept with a root path of the pending tree (FlatTrie.accept)."""
        indices = np.ascontiguousarray(indices, dtype=np.uint32)
        call("uzu_hip_model_accept", self._h, C.c_void_p(indices.ctypes.data), C.c_uint32(indices.size))

    def read_tree_logits(self) -> np.ndarray:
        out = np.empty((self._tree_size, self.logit_count), dtype=np.uint16)
        call("uzu_hip_model_read_tree_logits", self._h, C.c_void_p(out.ctypes.data))
        return out

    def speculative_step(self, root_token: int, propose, trie_seeds: bool = False):
        """One round of LanguageModelStream's speculative loop: `propose(root_token)` returns a uzu_amd.trie.TrieNode whose root carries
        `root_token` (the last sampled token); the tree is verified in one pass, the accepted path taken.  -> the tokens gained (the
        tokens sampled along the accepted path: at least one, whatever the proposal).  `trie_seeds`: under stochastic sampling every node
        draws with the seed its TrieNode carries (what the reference stream uploads, stream.rs:690-695) instead of the position-derived one."""
        flat = propose(root_token).linearize()
        sampled = self.verify_tree(flat.token_ids(), flat.nodes(), flat.token_seeds() if trie_seeds else None)
        accepted = flat.accept(sampled)
        self.accept([index for index, _, _ in accepted])
        return [int(out) for _, _, out in accepted]

    # ---- the speculator's hidden-feature taps (stream.rs:213-214,632-633; transformer.rs:160-171,285-293) ----
    def set_feature_layers(self, layer_ids):
        """Every later pass files bf16(shortcut + hidden) of every row after each listed layer (production outputs; [] removes the taps)."""
        ids = np.ascontiguousarray(layer_ids, dtype=np.uint32)
        call("uzu_hip_model_set_feature_layers", self._h, C.c_void_p(ids.ctypes.data) if ids.size else None, C.c_uint32(ids.size))
        self._feature_layers = [int(i) for i in ids]

    def hidden_features(self):
        """-> one bf16-bits array [rows, model_dim] per tapped layer, for the rows of the last pass"""
        out = []
        for i in range(len(getattr(self, "_feature_layers", []))):
            rows = C.c_uint32()
            call("uzu_hip_model_read_features", self._h, C.c_uint32(i), None, C.byref(rows))
            a = np.empty((rows.value, self.model_dim), dtype=np.uint16)
            call("uzu_hip_model_read_features", self._h, C.c_uint32(i), C.c_void_p(a.ctypes.data), C.byref(rows))
            out.append(a)
        return out

    def final_hidden_rows(self) -> np.ndarray:
        """DecoderEncodeOutput::final_hidden of the last prefill (1 row) / tree pass (one row per node): bf16 bits [rows, model_dim]"""
        rows = C.c_uint32()
        call("uzu_hip_model_read_final_hidden", self._h, None, C.c_uint32(0), C.byref(rows))
        a = np.empty((rows.value, self.model_dim), dtype=np.uint16)
        call("uzu_hip_model_read_final_hidden", self._h, C.c_void_p(a.ctypes.data), C.c_uint32(rows.value), C.byref(rows))
        return a

    def read_layer_output(self, layer: int) -> np.ndarray:
        rows, capacity = C.c_uint32(), C.c_uint32()
        call("uzu_hip_model_layer_output_rows", self._h, C.byref(rows), C.byref(capacity))  # a prefill pass holds up to `chunk` rows (2048 by default, UZU_PREFILL_CHUNK)
        out = np.empty(max(capacity.value, rows.value, 1) * self.model_dim, dtype=np.uint16)
        call("uzu_hip_model_read_layer_output", self._h, C.c_uint32(layer), C.c_void_p(out.ctypes.data), C.byref(rows))
        return out[: rows.value * self.model_dim].reshape(rows.value, self.model_dim).copy()

    def profile_decode_step(self, capacity: int = 4096):
        """One eager decode step with HIP events around every kernel -> list of (label, algorithmic_bytes, ms)."""
        names = (C.c_char_p * capacity)()
        nbytes = (C.c_uint64 * capacity)()
        ms = (C.c_float * capacity)()
        count = C.c_uint32()
        call("uzu_hip_model_profile_decode_step", self._h, C.c_uint32(capacity), names, nbytes, ms, C.byref(count))
        return [(names[i].decode(), int(nbytes[i]), float(ms[i])) for i in range(count.value)]


class HipDrafter:
    """The DFlash draft model next to its target (include/uzu_hip_engine.h: uzu_hip_drafter_*; encodable_block/dflash.rs:41-346): the `drafter`
    duck type of uzu_amd.speculator."""

    def __init__(self, ctx: Context, target: HipModel, bundle):
        self.ctx, self.target, self.bundle = ctx, target, bundle
        self.block_size, self.model_dim, self.vocab_size = bundle.block_size, bundle.model_dim, bundle.vocab_size
        self.target_layer_ids = list(bundle.target_layer_ids)
        desc = bundle.desc()
        self._h = C.c_void_p()
        call("uzu_hip_drafter_create", ctx._h, target._h, C.byref(desc), C.byref(self._h))
        target.set_feature_layers(self.target_layer_ids)

    def close(self):
        if self._h:
            fn = _ffi.lib().uzu_hip_drafter_destroy
            fn.restype, fn.argtypes = None, [C.c_void_p]
            fn(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            if self.ctx._h:
                self.close()
        except Exception:
            pass

    def reset(self):
        call("uzu_hip_drafter_reset", self._h)

    @property
    def context_length(self) -> int:
        fn = _ffi.lib().uzu_hip_drafter_context_length
        fn.restype, fn.argtypes = C.c_uint32, [C.c_void_p]
        return int(fn(self._h))

    def accept(self, target_features, accepted_indices):
        """DFlash::encode_accept over rows of the target's LAST pass; `target_features` belongs to the duck type (a host-side drafter would need the rows) and is ignored here:
        the rows never leave the device (the engine reads the target's taps in place)."""
        idx = np.ascontiguousarray(accepted_indices, dtype=np.uint32)
        call("uzu_hip_drafter_accept", self._h, C.c_void_p(idx.ctypes.data) if idx.size else None, C.c_uint32(idx.size))

    def draft(self, target, target_output_token: int, batch_size: int, want_outputs: bool = False):
        """-> (draft_hidden bf16 bits [batch, d] | None, logits f32 [batch - 1, vocab] | None, greedy tokens [batch - 1])"""
        tokens = np.empty(batch_size - 1, dtype=np.uint32)
        call("uzu_hip_drafter_draft", self._h, C.c_uint32(int(target_output_token)), C.c_uint32(batch_size), C.c_void_p(tokens.ctypes.data))
        hidden = logits = None
        if want_outputs:
            hidden = np.empty((batch_size, self.model_dim), dtype=np.uint16)
            logits = np.empty((batch_size - 1, self.target.logit_count), dtype=np.float32)
            call("uzu_hip_drafter_read_draft", self._h, C.c_void_p(hidden.ctypes.data), C.c_void_p(logits.ctypes.data), None)
        return hidden, logits, tokens

    @property
    def gpu_ms(self):
        """(accept_ms, draft_ms) of the last calls: device time between HIP events on the engine's stream"""
        a, d = C.c_float(), C.c_float()
        call("uzu_hip_drafter_gpu_ms", self._h, C.byref(a), C.byref(d))
        return a.value, d.value


class HipWeaver:
    """The Weaver tree constructor on a drafter (include/uzu_hip_engine.h: uzu_hip_weaver_*; encodable_block/weaver.rs:166-676): the `weaver` duck type of
    uzu_amd.speculator.  Close it before its drafter."""

    def __init__(self, ctx: Context, drafter: HipDrafter, bundle):
        self.ctx, self.drafter, self.bundle = ctx, drafter, bundle
        self.max_depth = bundle.max_depth
        desc = bundle.desc()
        self._h = C.c_void_p()
        call("uzu_hip_weaver_create", ctx._h, drafter._h, C.byref(desc), C.byref(self._h))

    def close(self):
        if self._h:
            fn = _ffi.lib().uzu_hip_weaver_destroy
            fn.restype, fn.argtypes = None, [C.c_void_p]
            fn(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            if self.ctx._h and self.drafter._h:
                self.close()
        except Exception:
            pass

    def encode_tree(self, target, target_hidden, draft_hidden, logits, depth_seeds, root_token_id: int, shape):
        """-> (packed_tree u32 [6, slots], frontier u32 [7, slots * expand_width]), or None for WeaverEncodeError::InvalidTreeInput.  `draft_hidden` / `logits`
        belong to the duck type and are ignored: the tree reads the drafter's last draft where it lies in HBM."""
        row = np.ascontiguousarray(np.asarray(target_hidden, dtype=np.uint16).reshape(-1)[: self.drafter.model_dim])
        seeds = np.ascontiguousarray(depth_seeds, dtype=np.uint64)
        slots = shape.slot_count()
        packed = np.zeros((6, slots), dtype=np.uint32)
        frontier = np.zeros((7, slots * max(shape.expand_width, 1)), dtype=np.uint32)
        try:
            call("uzu_hip_weaver_encode_tree", self._h, C.c_void_p(row.ctypes.data), C.c_void_p(seeds.ctypes.data), C.c_uint32(seeds.size), C.c_uint32(int(root_token_id)),
                 C.byref(shape), C.c_void_p(packed.ctypes.data), C.c_void_p(frontier.ctypes.data))
        except _ffi.UzuHipError as err:
            if "invalid Weaver tree input" in str(err):
                return None
            raise
        return packed, frontier

    @property
    def stats(self):
        """(device ms, kernel launches) of the last tree"""
        ms, n = C.c_float(), C.c_uint32()
        call("uzu_hip_weaver_stats", self._h, C.byref(ms), C.byref(n))
        return ms.value, n.value
