"""Tensor-parallel shard planner and group setup (not present in the reference).

One process per GPU.  ``shard_bundle(bundle, rank, size)`` cuts a whole-model ``ModelBundle`` into the shard
rank ``rank`` of ``size`` owns; the engine (``uzu_hip_model_create_tp``) runs the ordinary forward on the shard
and exchanges data at exactly three kinds of places:

  * after every ROW-parallel linear (attention / DeltaNet out-projection, MLP down-projection): all-reduce(sum) of
    the f32 partial rows ``[tokens, model_dim]``, rounded to bf16 once after the sum;
  * after the vocab-sharded read-out: all-reduce(max) of one packed (logit, index) key per sampled token;
  * nothing else: norms, residual stream, RoPE tables and the embedding lookup are replicated.

Partitioning (the quant groups run along k, so any split of the OUTPUT rows is layout-safe; splits of K must fall
on group boundaries):

  qkv / gate            column-parallel by q heads; the kv heads follow their q heads (replicated when
                        size > kv heads)                                                         no collective
  DeltaNet in-proj      column-parallel by k heads / v heads: sections [q | k | v | z | beta | a] are cut per head;
                        conv weights / biases, a_log, dt_bias follow the same channels         no collective
  up (fused up||gate)   column-parallel, each half cut the same way; hidden is zero-padded to a multiple of
                        size * group_size so that every rank owns whole quant groups           no collective
  out-proj / down       row-parallel: K slice = the rank's heads / hidden slice                all-reduce(sum)
  read-out              rows [rank * V/size, (rank+1) * V/size); the full table stays for the embedding lookup
"""
from __future__ import annotations

import ctypes as C
from dataclasses import replace
from typing import Callable, List, Optional, Tuple

import numpy as np

from . import _ffi
from ._ffi import call
from . import desc as D


# ------------------------------------------------------------------------------------------------ slicing helpers
RHT_BLOCK = 32  # HybridSpec incoherence block (rht_wrapper.rs:140-160): the Hadamard factors act on 32 consecutive elements


def _refuse_adapters(w: D.LinearWeights):
    # QLoRA (qlora_wrapper.rs:177-251): a row-parallel shard would need `down` cut along k plus an all-reduce of the rank-r
    # intermediate in front of `up`; not planned -- refuse instead of dropping the adapter term (the helpers below rebuild
    # LinearWeights field by field)
    if w.adapter_down is not None or w.adapter_up is not None:
        raise NotImplementedError("tensor-parallel shards of QLoRA (HybridSpec + LowRankSpec) linears are not planned: the adapters would be dropped")


def _whole_rht_blocks(idx: np.ndarray) -> bool:
    """True iff `idx` is a concatenation of whole, aligned 32-row blocks (then OutputRht of the shard == shard of OutputRht)."""
    if len(idx) % RHT_BLOCK:
        return False
    blocks = np.asarray(idx).reshape(-1, RHT_BLOCK)
    return bool(np.all(blocks[:, 0] % RHT_BLOCK == 0) and np.all(blocks == blocks[:, :1] + np.arange(RHT_BLOCK)))


def take_rows(w: D.LinearWeights, idx: np.ndarray) -> D.LinearWeights:
    """Column-parallel shard: keep output rows `idx` (in that order).  HybridSpec linears (RHTLinearWrapper): the input factors stay
    whole (k is not cut), the output factors follow their rows -- which must then be whole 32-row Hadamard blocks."""
    _refuse_adapters(w)
    cp = lambda a: None if a is None else np.ascontiguousarray(a[idx])
    if w.output_signs is not None and not _whole_rht_blocks(idx):
        raise NotImplementedError("tensor-parallel shard cuts a 32-row OutputRht block of a HybridSpec linear")
    return D.LinearWeights(int(len(idx)), w.k, w.bits, w.group_size, w.method, cp(w.weights), cp(w.scales), cp(w.biases),
                           cp(w.zero_points), cp(w.out_biases), input_signs=w.input_signs, output_signs=cp(w.output_signs))


def take_k(w: D.LinearWeights, k0: int, k1: int, owns_out_bias: bool) -> D.LinearWeights:
    """Row-parallel shard: keep input columns [k0, k1).  k0, k1 must be quant-group boundaries (nibble-packed int4
    zero-points are repacked).  The Linear's output bias is added by one rank only."""
    _refuse_adapters(w)
    # HybridSpec: InputRht is block-diagonal along k, so a 32-aligned k slice takes its own factors; OutputRht (and the bias behind
    # it, rht_wrapper.rs:281-298) runs on the all-reduced rows on EVERY rank, so the output factors and the bias stay whole everywhere
    isg = None
    if w.input_signs is not None:
        if k0 % RHT_BLOCK or k1 % RHT_BLOCK:
            raise NotImplementedError("tensor-parallel K split cuts a 32-element InputRht block of a HybridSpec linear")
        isg = np.ascontiguousarray(w.input_signs[k0:k1])
    if w.output_signs is not None:
        owns_out_bias = True
    if w.method == D.QUANT_NONE:
        assert w.bits == 16
        wt = np.ascontiguousarray(w.weights.reshape(w.n, w.k)[:, k0:k1])
        return D.LinearWeights(w.n, k1 - k0, 16, 0, w.method, wt, None, None, None, w.out_biases if owns_out_bias else None,
                               input_signs=isg, output_signs=w.output_signs)
    g = w.group_size
    assert k0 % g == 0 and k1 % g == 0, f"K split [{k0},{k1}) is not on group boundaries (group {g})"
    g0, g1 = k0 // g, k1 // g
    bpk = w.bits / 8.0
    b0, b1 = int(k0 * bpk), int(k1 * bpk)
    wt = np.ascontiguousarray(w.weights.reshape(w.n, -1)[:, b0:b1])
    sc = np.ascontiguousarray(w.scales.reshape(w.n, -1)[:, g0:g1])
    bi = None if w.biases is None else np.ascontiguousarray(w.biases.reshape(w.n, -1)[:, g0:g1])
    zp = None
    if w.zero_points is not None:
        z = w.zero_points.reshape(w.n, -1)
        if w.bits == 4:  # nibble-packed (even group = low nibble): unpack, cut, repack so that any group start works
            per_group = np.empty((w.n, 2 * z.shape[1]), dtype=np.uint8)
            per_group[:, 0::2], per_group[:, 1::2] = z & 0xF, z >> 4
            cut = per_group[:, g0:g1]
            if cut.shape[1] % 2:
                cut = np.concatenate([cut, np.zeros((w.n, 1), dtype=np.uint8)], axis=1)
            zp = np.ascontiguousarray(cut[:, 0::2] | (cut[:, 1::2] << 4))
        else:
            zp = np.ascontiguousarray(z[:, g0:g1])
    return D.LinearWeights(w.n, k1 - k0, w.bits, g, w.method, wt, sc, bi, zp, w.out_biases if owns_out_bias else None,
                           input_signs=isg, output_signs=w.output_signs)


def _zero_rows(w: D.LinearWeights, count: int) -> D.LinearWeights:
    """`count` extra output rows that dequantise to exactly 0 (scale = bias = 0) -- MLP hidden padding."""
    z = lambda a: None if a is None else np.zeros((count,) + a.shape[1:], dtype=a.dtype)
    ones = None if w.output_signs is None else np.ones(count, dtype=w.output_signs.dtype)  # H * 0 = 0 whatever the factors
    return D.LinearWeights(count, w.k, w.bits, w.group_size, w.method, z(w.weights), z(w.scales), z(w.biases), z(w.zero_points),
                           z(w.out_biases), input_signs=w.input_signs, output_signs=ones)


def _concat_rows(parts: List[D.LinearWeights]) -> D.LinearWeights:
    first = parts[0]
    cat = lambda name: None if getattr(first, name) is None else np.ascontiguousarray(np.concatenate([getattr(p, name) for p in parts], axis=0))
    return D.LinearWeights(sum(p.n for p in parts), first.k, first.bits, first.group_size, first.method, cat("weights"), cat("scales"),
                           cat("biases"), cat("zero_points"), cat("out_biases"), input_signs=first.input_signs, output_signs=cat("output_signs"))


def _pad_k(w: D.LinearWeights, new_k: int) -> D.LinearWeights:
    """Extend the input dimension with columns that contribute exactly 0 (their activations are 0 as well)."""
    if new_k == w.k:
        return w
    _refuse_adapters(w)
    assert w.method != D.QUANT_NONE and (new_k - w.k) % w.group_size == 0
    isg = None if w.input_signs is None else np.concatenate([w.input_signs, np.ones(new_k - w.k, dtype=w.input_signs.dtype)])  # the padded inputs are 0
    extra_groups = (new_k - w.k) // w.group_size
    extra_bytes = (new_k - w.k) * w.bits // 8
    wt = np.concatenate([w.weights.reshape(w.n, -1), np.zeros((w.n, extra_bytes), dtype=np.uint8)], axis=1)
    sc = np.concatenate([w.scales.reshape(w.n, -1), np.zeros((w.n, extra_groups), dtype=np.uint16)], axis=1)
    bi = None if w.biases is None else np.concatenate([w.biases.reshape(w.n, -1), np.zeros((w.n, extra_groups), dtype=np.uint16)], axis=1)
    zp = None
    if w.zero_points is not None:
        groups = new_k // w.group_size
        zp_cols = (groups + 1) // 2 if w.bits == 4 else groups
        z = w.zero_points.reshape(w.n, -1)
        zp = np.concatenate([z, np.zeros((w.n, zp_cols - z.shape[1]), dtype=np.uint8)], axis=1)
    return D.LinearWeights(w.n, new_k, w.bits, w.group_size, w.method, np.ascontiguousarray(wt), np.ascontiguousarray(sc),
                           None if bi is None else np.ascontiguousarray(bi), None if zp is None else np.ascontiguousarray(zp), w.out_biases,
                           input_signs=isg, output_signs=w.output_signs)


def _head_range(total: int, rank: int, size: int) -> Tuple[int, int]:
    """Heads [lo, hi) of `total` owned by `rank`; with fewer heads than ranks the heads are replicated."""
    if total >= size:
        assert total % size == 0, f"{total} heads do not split over {size} ranks"
        per = total // size
        return rank * per, (rank + 1) * per
    assert size % total == 0, f"{size} ranks do not replicate {total} heads evenly"
    h = rank * total // size
    return h, h + 1


def padded_hidden(hidden: int, group: int, size: int, rht: bool = False) -> int:
    """Hidden width every rank can take an equal, whole-quant-group slice of.  `rht`: the MLP linears carry Hadamard factors -- the padding
    between the up and gate halves of the fused matrix must then be whole 32-row blocks too (otherwise the gate half shifts against its
    OutputRht blocks while still passing take_rows' alignment check on the REBUILT matrix) and every rank's slice a multiple of 32."""
    unit = max(group, 1) * size
    if rht:
        unit = int(np.lcm(unit, RHT_BLOCK * size))
        if hidden % RHT_BLOCK:
            raise NotImplementedError("tensor-parallel MLP shard of a HybridSpec linear whose hidden width is not a multiple of the 32-element Hadamard block")
    return (hidden + unit - 1) // unit * unit


# ------------------------------------------------------------------------------------------------ the planner
def shard_layer(l: D.LayerWeights, rank: int, size: int) -> D.LayerWeights:
    out = replace(l)
    # ---- MLP: up || gate column-parallel, down row-parallel
    h = l.hidden_dim
    group = l.down_projection.group_size if l.down_projection.method != D.QUANT_NONE else 1
    rht = any(x is not None for x in (l.up_projection.output_signs, l.up_projection.input_signs, l.down_projection.input_signs, l.down_projection.output_signs))
    hp = padded_hidden(h, group, size, rht)
    per = hp // size
    lo, hi = rank * per, (rank + 1) * per
    up_full, down_full = l.up_projection, l.down_projection
    if hp != h:  # zero rows in both halves of the fused up matrix, zero-contribution columns in down
        up_half, gate_half = take_rows(up_full, np.arange(0, h)), take_rows(up_full, np.arange(h, 2 * h))
        pad = _zero_rows(up_full, hp - h)
        up_full = _concat_rows([up_half, pad, gate_half, pad])
        down_full = _pad_k(down_full, hp)
    out.hidden_dim = per
    out.up_projection = take_rows(up_full, np.concatenate([np.arange(lo, hi), np.arange(hp + lo, hp + hi)]))
    out.down_projection = take_k(down_full, lo, hi, owns_out_bias=rank == 0)

    if l.mixer_kind == D.MIXER_ATTENTION:
        nq, nkv, hd = l.num_heads, l.num_groups, l.head_dim
        assert nq % size == 0, f"{nq} query heads do not split over {size} ranks"
        q_lo, q_hi = rank * nq // size, (rank + 1) * nq // size
        gqa = nq // nkv
        kv_lo, kv_hi = q_lo // gqa, (q_hi - 1) // gqa + 1
        assert (q_hi - q_lo) % (kv_hi - kv_lo) == 0
        rows = lambda first_head, lo_, hi_: np.arange((first_head + lo_) * hd, (first_head + hi_) * hd)
        qkv_rows = np.concatenate([rows(0, q_lo, q_hi), rows(nq, kv_lo, kv_hi), rows(nq + nkv, kv_lo, kv_hi)])
        out.num_heads, out.num_groups = q_hi - q_lo, kv_hi - kv_lo
        out.qkv_projection = take_rows(l.qkv_projection, qkv_rows)
        if l.gate_projection is not None:
            out.gate_projection = take_rows(l.gate_projection, rows(0, q_lo, q_hi))
        out.out_projection = take_k(l.out_projection, q_lo * hd, q_hi * hd, owns_out_bias=rank == 0)
        if l.sinks is not None:
            out.sinks = np.ascontiguousarray(l.sinks[q_lo:q_hi])
    else:
        Hv, Hk, Dk, Dv = l.dn_num_heads, l.dn_num_groups, l.dn_head_dim, l.dn_value_head_dim
        assert Hv % size == 0, f"{Hv} DeltaNet value heads do not split over {size} ranks"
        v_lo, v_hi = rank * Hv // size, (rank + 1) * Hv // size
        gph = Hv // Hk
        k_lo, k_hi = v_lo // gph, (v_hi - 1) // gph + 1
        assert (v_hi - v_lo) % (k_hi - k_lo) == 0
        key_dim, value_dim = Hk * Dk, Hv * Dv
        conv_dim = 2 * key_dim + value_dim
        q_rows = np.arange(k_lo * Dk, k_hi * Dk)
        k_rows = key_dim + q_rows
        v_rows = 2 * key_dim + np.arange(v_lo * Dv, v_hi * Dv)
        z_rows = conv_dim + np.arange(v_lo * Dv, v_hi * Dv)
        beta_rows = conv_dim + value_dim + np.arange(v_lo, v_hi)
        a_rows = conv_dim + value_dim + Hv + np.arange(v_lo, v_hi)
        conv_rows = np.concatenate([q_rows, k_rows, v_rows])
        out.dn_num_heads, out.dn_num_groups = v_hi - v_lo, k_hi - k_lo
        out.dn_in_proj = take_rows(l.dn_in_proj, np.concatenate([conv_rows, z_rows, beta_rows, a_rows]))
        out.dn_out_proj = take_k(l.dn_out_proj, v_lo * Dv, v_hi * Dv, owns_out_bias=rank == 0)
        out.dn_conv_weights = np.ascontiguousarray(l.dn_conv_weights.reshape(conv_dim, -1)[conv_rows])
        if l.dn_conv_biases is not None:
            out.dn_conv_biases = np.ascontiguousarray(l.dn_conv_biases[conv_rows])
        out.dn_a_log = np.ascontiguousarray(l.dn_a_log[v_lo:v_hi])
        out.dn_dt_bias = np.ascontiguousarray(l.dn_dt_bias[v_lo:v_hi])
    return out


def shard_bundle(bundle: D.ModelBundle, rank: int, size: int) -> Tuple[D.ModelBundle, int]:
    """-> (shard bundle, vocab_offset).  size == 1 still produces the untied-readout form the TP engine expects."""
    assert 0 <= rank < size
    # HybridSpec linears shard with their Hadamard factors (take_rows / take_k: 32-aligned cuts only, refused otherwise).  Not planned:
    # QLoRA adapters (refused by the helpers).  RHT embeddings (embedding.rs:126-341; round 5): their sign vectors run along MODEL_DIM -- the
    # lookup's OutputRht on the replicated table, the read-out's InputRht on the normalised row -- which the vocabulary split does not cut:
    # the shard's read-out keeps them whole as its input factors (a tied table's output factors ARE the read-out's input factors,
    # embedding.rs:167-173).  Only the combinations the reference does not have either are refused.
    # Not planned either: the Gemma-family options that only the one-kernel-per-reference-kernel pass implements (the engine refuses them on a shard too)
    if bundle.embedding_norm.present or bundle.ple is not None or any(
            l.post_layer_scalar is not None or l.kv_source_layer_index is not None or l.normalize_values or l.ple is not None for l in bundle.layers):
        raise NotImplementedError("tensor-parallel shards of models with post-layer scalars, an embedding norm, KV sharing, value normalisation or per-layer embeddings")
    emb, out_emb = bundle.embedding, bundle.output_embedding
    if emb.input_signs is not None or (out_emb is not None and out_emb.output_signs is not None):
        raise NotImplementedError("tensor-parallel shards of RHT embeddings: input factors on the lookup table / output factors on the read-out are not a HybridSpec embedding form")
    for w in (emb, out_emb):
        if w is not None:
            _refuse_adapters(w)
    V = bundle.vocab_size
    assert V % size == 0, f"vocab {V} does not split over {size} ranks"
    lo, hi = rank * V // size, (rank + 1) * V // size
    readout = emb if bundle.tied_embeddings else out_emb
    readout_in_signs = emb.output_signs if bundle.tied_embeddings else out_emb.input_signs
    rows = np.arange(lo, hi)
    cp = lambda a: None if a is None else np.ascontiguousarray(a[rows])
    readout_shard = D.LinearWeights(hi - lo, readout.k, readout.bits, readout.group_size, readout.method, cp(readout.weights), cp(readout.scales), cp(readout.biases),
                                    cp(readout.zero_points), cp(readout.out_biases), input_signs=readout_in_signs, output_signs=None)
    shard = replace(bundle, layers=[shard_layer(l, rank, size) for l in bundle.layers], tied_embeddings=False,
                    output_embedding=readout_shard, _keep=[])
    shard.name = f"{bundle.name}[tp {rank}/{size}]"
    return shard, lo


# ------------------------------------------------------------------------------------------------ group setup
class TpGroup:
    """RCCL communicator of the engine (uzu_hip_tp_comm).  `broadcast` ships rank 0's 128-byte id to every rank:
    a callable  bytes|None -> bytes  (rank 0 passes the id, the others None)."""

    def __init__(self, ctx, rank: int, size: int, broadcast: Optional[Callable[[Optional[bytes]], bytes]] = None):
        self.ctx, self.rank, self.size = ctx, rank, size
        ident = None
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            call("uzu_hip_tp_unique_id", buf)
            ident = bytes(buf)
        if size > 1:
            assert broadcast is not None, "a multi-rank TpGroup needs a broadcast callable"
            ident = broadcast(ident)
        arr = (C.c_uint8 * 128).from_buffer_copy(ident)
        self._h = C.c_void_p()
        call("uzu_hip_tp_comm_create", ctx._h, arr, C.c_int32(rank), C.c_int32(size), C.byref(self._h))

    @classmethod
    def local(cls, ctx, rank: int, size: int) -> "TpGroup":
        """A group without an RCCL communicator: peer-to-peer exchanges only (uzu_hip_tp_comm_create_local)."""
        self = cls.__new__(cls)
        self.ctx, self.rank, self.size = ctx, rank, size
        self._h = C.c_void_p()
        call("uzu_hip_tp_comm_create_local", ctx._h, C.c_int32(rank), C.c_int32(size), C.byref(self._h))
        return self

    def enable_p2p(self, all_gather: Callable[[bytes], List[bytes]]):
        """One-shot peer-to-peer all-reduce for the decode-sized messages (csrc/tp.hip): export this rank's mailbox, hand the
        64-byte IPC handle to `all_gather` (returns every rank's handle, in rank order), open the peers' mailboxes."""
        mine = (C.c_uint8 * 64)()
        call("uzu_hip_tp_p2p_export", self.ctx._h, self._h, mine)
        handles = all_gather(bytes(mine))
        assert len(handles) == self.size and all(len(h) == 64 for h in handles)
        flat = (C.c_uint8 * (64 * self.size)).from_buffer_copy(b"".join(handles))
        call("uzu_hip_tp_p2p_connect", self.ctx._h, self._h, flat)

    def disable_p2p(self):
        fn = _ffi.lib().uzu_hip_tp_p2p_disable
        fn.restype, fn.argtypes = None, [C.c_void_p]
        fn(self._h)

    def p2p_error(self) -> int:
        out = C.c_uint32()
        call("uzu_hip_tp_p2p_error", self._h, C.byref(out))
        return out.value

    def stats(self) -> Tuple[int, int, int]:
        """(ranks the RCCL communicator reports -- 0 for a local group --, collectives enqueued through RCCL, exchanges through the mailboxes)"""
        ranks, rccl, p2p = C.c_uint32(), C.c_uint64(), C.c_uint64()
        call("uzu_hip_tp_comm_stats", self._h, C.byref(ranks), C.byref(rccl), C.byref(p2p))
        return ranks.value, rccl.value, p2p.value

    def all_reduce_sum_f32(self, buf, count: int, offset_bytes: int = 0):
        call("uzu_hip_tp_all_reduce_sum_f32", self.ctx._h, self._h, buf._h, C.c_size_t(offset_bytes), C.c_size_t(count))

    def all_reduce_max_u64(self, buf, count: int, offset_bytes: int = 0):
        call("uzu_hip_tp_all_reduce_max_u64", self.ctx._h, self._h, buf._h, C.c_size_t(offset_bytes), C.c_size_t(count))

    def close(self):
        if self._h:
            _ffi.lib().uzu_hip_tp_comm_destroy(self._h)
            self._h = C.c_void_p()


def torch_broadcast(dist, device=None) -> Callable[[Optional[bytes]], bytes]:
    """Broadcast helper over an initialised torch.distributed process group (gloo on CPU, nccl = RCCL on GPU)."""
    import torch

    def bcast(ident: Optional[bytes]) -> bytes:
        t = torch.zeros(128, dtype=torch.uint8, device=device)
        if ident is not None:
            t = torch.tensor(list(ident), dtype=torch.uint8, device=device)
        dist.broadcast(t, src=0)
        return bytes(t.cpu().tolist())

    return bcast


def torch_all_gather_bytes(dist) -> Callable[[bytes], List[bytes]]:
    """all_gather of one bytes object per rank over an initialised torch.distributed group (for TpGroup.enable_p2p)."""
    def gather(mine: bytes) -> List[bytes]:
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, mine)
        return out
    return gather


# ------------------------------------------------------------------------------------------------ exchange protocol
def pack_argmax_key(logit: float, global_index: int) -> int:
    """Host mirror of tp.hip's packed greedy key: (order-preserving u32 of the f32 logit) << 32 | ~index.
    max() over the ranks' keys = highest logit, ties -> lowest index (unified_sampling.rs:90-95)."""
    bits = int(np.float32(logit).view(np.uint32))
    if bits & 0x7FFFFFFF == 0:
        bits = 0  # -0.0 == +0.0: same key, the index decides
    orderable = (~bits & 0xFFFFFFFF) if bits & 0x80000000 else (bits | 0x80000000)
    return (orderable << 32) | (0xFFFFFFFF - int(global_index))


def unpack_argmax_key(key: int) -> Tuple[float, int]:
    orderable, inv = (key >> 32) & 0xFFFFFFFF, key & 0xFFFFFFFF
    bits = (orderable & 0x7FFFFFFF) if orderable & 0x80000000 else (~orderable & 0xFFFFFFFF)
    return float(np.uint32(bits).view(np.float32)), 0xFFFFFFFF - inv
