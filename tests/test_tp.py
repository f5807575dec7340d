"""Tensor-parallel path: shard planner algebra, the exchange protocol over world_size-2 gloo
on CPU, and (GPU, 1 device) the engine's TP code path with a one-rank RCCL communicator.

The reference has no multi-device path, so the check is self-consistency: a sharded forward must reproduce the
unsharded one -- column-parallel shards are row selections, row-parallel shards sum to the full product."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from uzu_amd import desc as D, synthetic as S, tp as TP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bf16_to_f64(bits):
    return (bits.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def dequant(w: D.LinearWeights) -> np.ndarray:
    """float64 [n, k] per the reference dequantisation (matmul/kernel.rs:236-275)."""
    if w.method == D.QUANT_NONE:
        return bf16_to_f64(w.weights.reshape(w.n, w.k))
    raw = w.weights.reshape(w.n, -1)
    if w.bits == 4:
        q = np.empty((w.n, w.k), dtype=np.float64)
        q[:, 0::2], q[:, 1::2] = raw & 0xF, raw >> 4
    else:
        q = raw.astype(np.float64)
    g = w.group_size
    groups = w.k // g
    sc = np.repeat(bf16_to_f64(w.scales.reshape(w.n, groups)), g, axis=1)
    if w.method == D.QUANT_SCALE_BIAS:
        return sc * q + np.repeat(bf16_to_f64(w.biases.reshape(w.n, groups)), g, axis=1)
    if w.method == D.QUANT_SCALE_ZERO_POINT:
        z = w.zero_points.reshape(w.n, -1)
        if w.bits == 4:
            zp = np.empty((w.n, 2 * z.shape[1]), dtype=np.float64)
            zp[:, 0::2], zp[:, 1::2] = z & 0xF, z >> 4
            zp = zp[:, :groups]
        else:
            zp = z.astype(np.float64)
        return sc * (q - np.repeat(zp, g, axis=1))
    return sc * (q - float(1 << (w.bits - 1)))


def close(a, b):  # BLAS sums a row subset in a different blocking: equal up to f64 rounding
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-13)


def silu(x):
    return x / (1.0 + np.exp(-x))


def small_cfg(method=D.QUANT_SCALE_BIAS, hidden=384, group=64):
    # hidden 384 at group 64 does not split over 4 ranks without padding (384/4 = 96): exercises the zero padding
    return S.tiny_qwen(hidden_dim=hidden, group_size=group, method=method, num_heads=4, num_groups=2, dn_num_heads=4, dn_num_groups=2)


def _rht(v, signs):
    """float64 InputRht / OutputRht on whole 32-element blocks: H32 (Sylvester, orthonormal) applied to sign-flipped values
    (activation_transform.rs:43-136: input mode multiplies by the factors first, output mode after -- H is symmetric)."""
    H = np.array([[1.0]])
    while H.shape[0] < 32:
        H = np.block([[H, H], [H, -H]])
    H /= np.sqrt(32.0)
    return v.reshape(-1, 32), H, signs.reshape(-1, 32).astype(np.float64)


def in_rht(x, signs):
    b, H, s = _rht(x, signs)
    return ((b * s) @ H).reshape(-1)


def out_rht(y, signs):
    b, H, s = _rht(y, signs)
    return ((b @ H) * s).reshape(-1)


@pytest.mark.parametrize("size", [2, 4])
def test_rht_linears_shard_with_their_hadamard_factors(size):
    """HybridSpec linears: y = OutRht(W InRht(x)) (rht_wrapper.rs:215-298).  Column-parallel shards are the matching rows of y; the
    row-parallel shards' partial products sum to W InRht(x) and the (whole) output factors are applied once behind the sum."""
    cfg = S.tiny_llama(rht=True, hidden_dim=256, group_size=32)
    bundle = S.build_model(cfg)
    shards = [TP.shard_bundle(bundle, r, size)[0] for r in range(size)]
    rng = np.random.default_rng(5)
    x = rng.normal(size=(cfg.model_dim,))
    for li, full in enumerate(bundle.layers):
        parts = [s.layers[li] for s in shards]
        up = full.up_projection
        assert up.input_signs is not None and up.output_signs is not None
        want = out_rht(dequant(up) @ in_rht(x, up.input_signs), up.output_signs)
        h = full.hidden_dim
        for r, p in enumerate(parts):
            hp = p.hidden_dim
            got = out_rht(dequant(p.up_projection) @ in_rht(x, p.up_projection.input_signs), p.up_projection.output_signs)
            close(got[:hp], want[r * hp:(r + 1) * hp])
            close(got[hp:], want[h + r * hp:h + (r + 1) * hp])
        y = rng.normal(size=(h,))
        down = full.down_projection
        want = out_rht(dequant(down) @ in_rht(y, down.input_signs), down.output_signs)
        acc = np.zeros(cfg.model_dim)
        for r, p in enumerate(parts):
            hp = p.hidden_dim
            acc += dequant(p.down_projection) @ in_rht(y[r * hp:(r + 1) * hp], p.down_projection.input_signs)
            assert np.array_equal(p.down_projection.output_signs, down.output_signs)
            assert (p.down_projection.out_biases is None) == (down.out_biases is None)  # the bias sits behind OutputRht: on every rank
        np.testing.assert_allclose(out_rht(acc, down.output_signs), want, rtol=1e-10, atol=1e-11)


def test_planner_refuses_what_it_would_get_wrong():
    # QLoRA adapters would be dropped by the slicing helpers; a DeltaNet in-proj's beta / a rows cut a 32-row OutputRht block
    with pytest.raises(NotImplementedError, match="QLoRA"):
        TP.shard_bundle(S.build_model(S.tiny_llama(qlora_rank=4)), 0, 2)
    up = S.build_model(S.tiny_llama(rht=True)).layers[0].up_projection
    with pytest.raises(NotImplementedError, match="OutputRht block"):
        TP.take_rows(up, np.arange(8, 40))
    assert np.array_equal(TP.take_rows(up, np.arange(32, 96)).output_signs, up.output_signs[32:96])
    TP.shard_bundle(S.build_model(S.tiny_qwen(rht=True)), 0, 2)  # (the synthetic in-proj, n % 32 != 0, carries no factors)
    # RHT embeddings (round 5): the sign vectors run along model_dim, which the vocabulary split does not cut -- the shard's read-out takes them whole as
    # its input factors (a tied table's output factors are the read-out's input factors, embedding.rs:167-173), the lookup table stays replicated
    for cfg in (S.tiny_llama(rht_embeddings=True), S.tiny_qwen(rht_embeddings=True)):
        full = S.build_model(cfg)
        for r in range(2):
            sh, off = TP.shard_bundle(full, r, 2)
            want = full.embedding.output_signs if full.tied_embeddings else full.output_embedding.input_signs
            assert not sh.tied_embeddings and sh.output_embedding.output_signs is None and np.array_equal(sh.output_embedding.input_signs, want)
            assert np.array_equal(sh.embedding.output_signs, full.embedding.output_signs) and sh.output_embedding.n == cfg.vocab_size // 2 and off == r * cfg.vocab_size // 2
            src = full.embedding if full.tied_embeddings else full.output_embedding
            assert np.array_equal(sh.output_embedding.weights, src.weights[off:off + cfg.vocab_size // 2])
    bad = S.build_model(S.tiny_llama(rht_embeddings=True))
    bad.output_embedding.output_signs = np.ones(bad.output_embedding.n, np.int32)
    with pytest.raises(NotImplementedError, match="RHT embeddings"):
        TP.shard_bundle(bad, 0, 2)
    # hidden padding of an RHT MLP: the zero rows between the up and gate halves must be whole 32-row Hadamard blocks (advisor finding, round 4:
    # with an unquantised down projection, group 1, the gate half could shift against its OutputRht blocks and still pass take_rows' check)
    assert TP.padded_hidden(224, 1, 4) == 224 and TP.padded_hidden(224, 1, 4, rht=True) == 256
    assert TP.padded_hidden(224, 16, 2, rht=True) == 256 and (TP.padded_hidden(224, 16, 2, rht=True) - 224) % 32 == 0
    assert TP.padded_hidden(3584, 128, 8, rht=True) == 4096
    with pytest.raises(NotImplementedError, match="Hadamard block"):
        TP.padded_hidden(240, 16, 4, rht=True)


@pytest.mark.parametrize("size", [2, 4])
@pytest.mark.parametrize("method", [D.QUANT_SCALE_BIAS, D.QUANT_SCALE_ZERO_POINT, D.QUANT_SCALE_SYMMETRIC])
def test_shards_reassemble_the_full_layer(size, method):
    cfg = small_cfg(method)
    bundle = S.build_model(cfg)
    shards = [TP.shard_bundle(bundle, r, size)[0] for r in range(size)]
    rng = np.random.default_rng(3)
    d = cfg.model_dim
    x = rng.normal(size=(d,))
    for li, full in enumerate(bundle.layers):
        parts = [s.layers[li] for s in shards]
        # ---- MLP: sum over ranks of down_r(act(up_r x)) == down(act(up x)); padded hidden rows contribute exactly 0
        up = dequant(full.up_projection) @ x
        h = full.hidden_dim
        want = dequant(full.down_projection) @ (up[:h] * silu(up[h:]))
        got = np.zeros(d)
        for p in parts:
            u = dequant(p.up_projection) @ x
            hp = p.hidden_dim
            assert p.up_projection.n == 2 * hp and p.down_projection.k == hp and hp % max(cfg.group_size, 1) == 0
            got += dequant(p.down_projection) @ (u[:hp] * silu(u[hp:]))
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
        if full.mixer_kind == D.MIXER_ATTENTION:
            hd, nq, nkv = full.head_dim, full.num_heads, full.num_groups
            qkv = dequant(full.qkv_projection) @ x
            gate = dequant(full.gate_projection) @ x
            y = rng.normal(size=(nq * hd,))  # stand-in for the attention output (head-major)
            want = dequant(full.out_projection) @ y
            got = np.zeros(d)
            for r, p in enumerate(parts):
                q_lo = r * nq // size
                local = dequant(p.qkv_projection) @ x
                lq, lkv = p.num_heads, p.num_groups
                assert local.size == (lq + 2 * lkv) * hd
                close(local[: lq * hd], qkv[q_lo * hd:(q_lo + lq) * hd])
                kv_lo = q_lo // (nq // nkv)
                close(local[lq * hd:(lq + lkv) * hd], qkv[(nq + kv_lo) * hd:(nq + kv_lo + lkv) * hd])
                close(local[(lq + lkv) * hd:], qkv[(nq + nkv + kv_lo) * hd:(nq + nkv + kv_lo + lkv) * hd])
                close(dequant(p.gate_projection) @ x, gate[q_lo * hd:(q_lo + lq) * hd])
                got += dequant(p.out_projection) @ y[q_lo * hd:(q_lo + lq) * hd]
            np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
        else:
            Hv, Hk, Dk, Dv = full.dn_num_heads, full.dn_num_groups, full.dn_head_dim, full.dn_value_head_dim
            key_dim, value_dim = Hk * Dk, Hv * Dv
            conv_dim = 2 * key_dim + value_dim
            proj = dequant(full.dn_in_proj) @ x
            y = rng.normal(size=(value_dim,))
            want = dequant(full.dn_out_proj) @ y
            got = np.zeros(d)
            for r, p in enumerate(parts):
                lv, lk = p.dn_num_heads, p.dn_num_groups
                v_lo, k_lo = r * Hv // size, (r * Hv // size) // (Hv // Hk)
                local = dequant(p.dn_in_proj) @ x
                lkd, lvd = lk * Dk, lv * Dv
                assert local.size == 2 * lkd + 2 * lvd + 2 * lv
                sections = [(0, lkd, k_lo * Dk), (lkd, lkd, key_dim + k_lo * Dk), (2 * lkd, lvd, 2 * key_dim + v_lo * Dv),
                            (2 * lkd + lvd, lvd, conv_dim + v_lo * Dv), (2 * lkd + 2 * lvd, lv, conv_dim + value_dim + v_lo),
                            (2 * lkd + 2 * lvd + lv, lv, conv_dim + value_dim + Hv + v_lo)]
                for off, n, src in sections:
                    close(local[off:off + n], proj[src:src + n])
                cw = full.dn_conv_weights.reshape(conv_dim, -1)
                np.testing.assert_array_equal(p.dn_conv_weights.reshape(2 * lkd + lvd, -1)[2 * lkd:], cw[2 * key_dim + v_lo * Dv: 2 * key_dim + (v_lo + lv) * Dv])
                np.testing.assert_array_equal(p.dn_a_log, full.dn_a_log[v_lo:v_lo + lv])
                got += dequant(p.dn_out_proj) @ y[v_lo * Dv:(v_lo + lv) * Dv]
            np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
    # read-out shards tile the vocabulary in order
    full_ro = dequant(bundle.embedding)
    offs = [TP.shard_bundle(bundle, r, size)[1] for r in range(size)]
    assert offs == [r * cfg.vocab_size // size for r in range(size)]
    np.testing.assert_array_equal(np.concatenate([dequant(s.output_embedding) for s in shards]), full_ro)
    assert all(not s.tied_embeddings and s.embedding.n == cfg.vocab_size for s in shards)


@pytest.mark.parametrize("heads,kv_heads,size,hidden", [(32, 8, 4, 448), (40, 8, 8, 544), (8, 2, 8, 224)])
def test_attention_only_models_shard_like_baseline_configs(heads, kv_heads, size, hidden):
    """BASELINE configs C4 (Llama-3-8B: 32 q / 8 kv heads, TP 4) and C5 (Qwen3-14B class: 40 / 8, TP 8) at toy widths, plus
    kv heads replicated over more ranks than heads: untied read-out, no gate projection, ZeroPoint int4 groups of 32
    (hidden sizes that need zero padding)."""
    hd = 32  # = the quant group: a rank's slice of the out-projection's K (its heads) falls on group boundaries, as with hd 128 / group 128
    cfg = S.tiny_llama(num_heads=heads, num_groups=kv_heads, head_dim=hd, model_dim=128, hidden_dim=hidden, vocab_size=64 * size,
                       rope=D.RopeConfig(kind=D.ROPE_UNSCALED, head_dim=hd, max_sequence_length=256, base=10000.0), layer_kinds=[D.MIXER_ATTENTION] * 2)
    bundle = S.build_model(cfg)
    shards = [TP.shard_bundle(bundle, r, size) for r in range(size)]
    rng = np.random.default_rng(9)
    x, y = rng.normal(size=(cfg.model_dim,)), rng.normal(size=(heads * hd,))
    for li, full in enumerate(bundle.layers):
        qkv = dequant(full.qkv_projection) @ x
        want_attn = dequant(full.out_projection) @ y
        up = dequant(full.up_projection) @ x
        want_mlp = dequant(full.down_projection) @ (up[:hidden] * silu(up[hidden:]))
        got_attn, got_mlp = np.zeros(cfg.model_dim), np.zeros(cfg.model_dim)
        for r, (shard, _) in enumerate(shards):
            p = shard.layers[li]
            lq, lkv = p.num_heads, p.num_groups
            assert lq == heads // size and lkv == max(kv_heads // size, 1) and lq % lkv == 0
            q_lo, kv_lo = r * lq, (r * lq) // (heads // kv_heads)
            local = dequant(p.qkv_projection) @ x
            close(local[: lq * hd], qkv[q_lo * hd:(q_lo + lq) * hd])
            close(local[lq * hd:(lq + lkv) * hd], qkv[(heads + kv_lo) * hd:(heads + kv_lo + lkv) * hd])
            close(local[(lq + lkv) * hd:], qkv[(heads + kv_heads + kv_lo) * hd:(heads + kv_heads + kv_lo + lkv) * hd])
            got_attn += dequant(p.out_projection) @ y[q_lo * hd:(q_lo + lq) * hd]
            u = dequant(p.up_projection) @ x
            hp = p.hidden_dim
            assert hp % cfg.group_size == 0 and hp * size >= hidden
            got_mlp += dequant(p.down_projection) @ (u[:hp] * silu(u[hp:]))
        np.testing.assert_allclose(got_attn, want_attn, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(got_mlp, want_mlp, rtol=1e-12, atol=1e-12)
    assert [o for _, o in shards] == [r * cfg.vocab_size // size for r in range(size)]
    np.testing.assert_array_equal(np.concatenate([dequant(s.output_embedding) for s, _ in shards]), dequant(bundle.output_embedding))
    assert all(np.array_equal(dequant(s.embedding), dequant(bundle.embedding)) for s, _ in shards)  # input embedding replicated


def test_qwen35_0p8b_shard_shapes_at_8_ranks():
    """The headline model at TP=8 without building its weights: 8 q / 2 kv heads -> 1 q head + a replicated kv head,
    16 DeltaNet heads -> 2, hidden 3584 -> padded to 4096 (512 per rank = 4 groups of 128), vocab 248320 -> 31040."""
    assert TP.padded_hidden(3584, 128, 8) == 4096 and TP.padded_hidden(3584, 128, 4) == 3584 and TP.padded_hidden(3584, 128, 2) == 3584
    assert TP._head_range(2, 5, 8) == (1, 2) and TP._head_range(16, 3, 8) == (6, 8) and 248320 % 8 == 0


def test_argmax_key_orders_like_the_reference_tie_rule():
    """max() over packed keys = highest logit, ties -> lowest index; the key order is the float order."""
    vals = [(-3.5, 7), (0.0, 2), (2.25, 9), (2.25, 4), (1e-30, 3), (-1e30, 0), (-1e-30, 5)]
    keys = [TP.pack_argmax_key(v, i) for v, i in vals]
    assert TP.unpack_argmax_key(max(keys)) == (2.25, 4)
    by_key = [vals[j][0] for j in sorted(range(len(vals)), key=lambda j: keys[j])]
    assert by_key == sorted(by_key)
    for v, i in vals:
        assert TP.unpack_argmax_key(TP.pack_argmax_key(v, i)) == (float(np.float32(v)), i)


WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["UZU_ROOT"]); sys.path.insert(0, os.path.join(os.environ["UZU_ROOT"], "tests"))
from uzu_amd import desc as D, synthetic as S, tp as TP
from test_tp import dequant, silu, small_cfg

dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, size = dist.get_rank(), dist.get_world_size()
cfg = small_cfg()
bundle = S.build_model(cfg)
shard, vocab_offset = TP.shard_bundle(bundle, rank, size)
rng = np.random.default_rng(11)            # same stream on every rank: replicated activations
x = rng.normal(size=(cfg.model_dim,))
# one decoder MLP with the row-parallel exchange: partial rows in f32, all-reduce(sum), then one rounding
full, part = bundle.layers[0], shard.layers[0]
u = dequant(part.up_projection) @ x
partial = (dequant(part.down_projection) @ (u[:part.hidden_dim] * silu(u[part.hidden_dim:]))).astype(np.float32)
t = torch.from_numpy(partial.copy())
dist.all_reduce(t, op=dist.ReduceOp.SUM)
up = dequant(full.up_projection) @ x
want = dequant(full.down_projection) @ (up[:full.hidden_dim] * silu(up[full.hidden_dim:]))
assert np.allclose(t.numpy(), want, rtol=1e-5, atol=1e-5), (rank, np.abs(t.numpy() - want).max())
# vocab-sharded greedy sampling: every rank must end with the same (global) token
h = rng.normal(size=(cfg.model_dim,))
local_logits = (dequant(shard.output_embedding) @ h).astype(np.float32)
li = int(np.argmax(local_logits))
key = TP.pack_argmax_key(float(local_logits[li]), li + vocab_offset)
k = torch.tensor([key - (1 << 63)], dtype=torch.int64)   # gloo has no uint64: shift to keep the unsigned order
dist.all_reduce(k, op=dist.ReduceOp.MAX)
value, token = TP.unpack_argmax_key(int(k.item()) + (1 << 63))
all_logits = (dequant(bundle.embedding) @ h).astype(np.float32)
assert token == int(np.argmax(all_logits)) and value == float(all_logits[token]), (rank, token, int(np.argmax(all_logits)))
# the id broadcast helper used to set up the RCCL communicator
ident = TP.torch_broadcast(dist)(bytes(range(128)) if rank == 0 else None)
assert ident == bytes(range(128))
dist.barrier()
dist.destroy_process_group()
print(f"rank {rank} ok")
'''


def test_exchange_protocol_world2_gloo(tmp_path):
    """world_size 2 on CPU (gloo): row-parallel partial sums + all-reduce(sum), packed arg-max key + all-reduce(max)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), UZU_ROOT=ROOT,
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {rank} ok" in out, out[-2000:]


# ------------------------------------------------------------------------------------------------ GPU (1 device)
@pytest.mark.gpu
@pytest.mark.parametrize("preset,flags", [("tiny-qwen", 0), ("tiny-qwen", 1), ("tiny-qwen", 2), ("tiny-llama", 0)])
def test_tp_engine_path_world1_matches_single_gpu(hip_ctx, preset, flags):
    """The engine's TP code path (f32 partials -> RCCL all-reduce -> bf16, packed arg-max key) with a one-rank
    communicator is bit-identical to the plain path: tokens and logits.  flags: graph / no graph / no fusion."""
    from uzu_amd.engine import HipModel
    cfg = S.PRESETS[preset]()
    bundle = S.build_model(cfg)
    prompt = S.synthetic_prompt(40, cfg.vocab_size)
    plain = HipModel(hip_ctx, bundle, flags)
    t0 = plain.prefill(prompt)
    l0 = plain.read_logits()
    toks0, _ = plain.decode(12)
    l1 = plain.read_logits()
    group = TP.TpGroup(hip_ctx, 0, 1)
    shard, off = TP.shard_bundle(bundle, 0, 1)
    assert off == 0
    tp_model = HipModel(hip_ctx, shard, flags, tp_group=group, vocab_offset=off)
    assert tp_model.prefill(prompt) == t0
    assert np.array_equal(tp_model.read_logits(), l0)
    toks1, _ = tp_model.decode(12)
    assert np.array_equal(toks0, toks1) and np.array_equal(tp_model.read_logits(), l1)
    tp_model.close()
    group.close()
    plain.close()


@pytest.mark.gpu
@pytest.mark.parametrize("size", [2, 4, 8])
def test_full_size_shard_shapes_execute(hip_ctx, size):
    """Every kernel of the engine at the SHARD shapes of Qwen3.5-0.8B (tp 2 / 4 / 8: 1-4 q heads, 2-8 DeltaNet heads,
    hidden 1792 / 896 / 512 with padding, K slices 1024 / 512 / 256) -- on one GPU with a one-rank communicator, so the
    partial sums are not completed and the tokens mean nothing; the point is that no launch is rejected or faults,
    in prefill (matrix-core GEMM, chunked DeltaNet, flash attention) and in graph-replayed decode."""
    from uzu_amd.engine import HipModel
    cfg = S.qwen35_0p8b(max_context_length=512)
    bundle = S.build_model(cfg)
    shard, off = TP.shard_bundle(bundle, size - 1, size)
    group = TP.TpGroup(hip_ctx, 0, 1)
    model = HipModel(hip_ctx, shard, 0, tp_group=group, vocab_offset=off)
    first = model.prefill(S.synthetic_prompt(130, cfg.vocab_size))
    toks, _ = model.decode(6)
    assert off <= first < cfg.vocab_size and all(off <= int(t) < off + shard.output_embedding.n for t in toks)
    assert model.logit_count == cfg.vocab_size // size
    model.close()
    group.close()


@pytest.mark.gpu
def test_p2p_all_reduce_two_processes_one_gpu(tmp_path):
    """The one-shot peer-to-peer all-reduce of the decode-sized messages (csrc/tp.hip): two PROCESSES share GPU 0, export their
    mailboxes as hipIpc handles, open each other's, and run the exchange kernels concurrently -- f32 sums in rank order
    (bit-identical on both ranks, equal to the host's rank-order sum), the u64 arg-max key, and a captured exchange replayed
    as a hipGraph (the sequence number advances on the device)."""
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "p2p_worker.py"), str(r), "2", str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("p2p workers timed out")
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


@pytest.mark.gpu
def test_p2p_failure_reaches_every_rank(tmp_path):
    """A rank that arrives later than the bounded wait: the punctual rank's exchange gives up and writes the failing sequence number into
    EVERY mailbox (csrc/tp.hip), so the late rank -- which finds all the flags it waits for -- starts its next exchange poisoned as well:
    both ranks end with a non-zero error word and NaN sums, i.e. every host raises at its next sync point instead of one rank sitting in a
    host barrier while the other carries on (the hang an 8-GPU run would otherwise risk)."""
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), UZU_TP_TIMEOUT_MS="150")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "p2p_worker.py"), str(r), "2", str(tmp_path), "skew"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("p2p workers timed out")
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--no-graph"]])
def test_bench_tensor_parallel_dry_run_two_ranks_one_gpu(extra):
    """`bench.py --gpus 2 --share-gpu`: the multi-rank code path of the benchmark itself -- rank spawn on 127.0.0.1, gloo rendezvous, shard
    planner, hipIpc mailboxes, the TP decode step replayed as a hipGraph (default) and launched eagerly (--no-graph), max-over-ranks timing,
    ONE JSON line from rank 0 -- on the one GPU of the test box (both ranks on device 0; a plumbing run, not a scaling point: the line says
    so).  Every rank must have committed the same tokens."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "4", "--warmup", "1", "--context", "96", "--no-cpu-baseline"] + extra
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["physical_gpus"] == 1 and line["steps"] == 4 and line["scaling"] == "strong"
    assert line["all_ranks_tokens_agree"] is True
    assert line["config"]["graph"] == ("--no-graph" not in extra)
    assert line["tp"]["all_reduces_per_token"] >= 2 * 24 and line["value"] > 0


# ------------------------------------------------------------------------------------------------ N ranks on ONE GPU
TP_CASES = [
    # (preset, kwargs, prompt_len, steps, logit tolerance in sigma, near-tie gap)
    ("tiny-qwen", {}, 40, 8, 0.25, 0.05),
    ("tiny-llama", {}, 40, 8, 0.25, 0.05),
    ("llama-3-8b", {"max_context_length": 256, "layer_kinds": ["MIXER_ATTENTION", "MIXER_ATTENTION"], "seed": 7}, 48, 5, 0.25, 0.05),
    # HybridSpec (RHT) linears under TP: the Hadamard factors are cut with their rows / k slices (tp.py::take_rows / take_k), OutputRht of a
    # row-parallel linear runs behind the all-reduce (rht_wrapper.rs:215-298)
    ("tiny-llama", {"rht": True}, 40, 8, 0.25, 0.05),
    # ... and at a width the FUSED decode step covers (model_dim % 1024 == 0): the transforms ride in the GEMV prologues, the OutputRht of the
    # row-parallel projections behind the one-hop exchange
    ("tiny-llama", {"rht": True, "model_dim": 1024}, 40, 8, 0.25, 0.05),
    # RHT embeddings under TP (round 5): untied (Output mode on the lookup table, Input mode on the sharded read-out) and tied (the table's output factors
    # are the read-out's input factors)
    ("tiny-llama", {"rht_embeddings": True}, 40, 8, 0.25, 0.05),
    ("tiny-qwen", {"rht_embeddings": True}, 40, 8, 0.25, 0.05),
]


@pytest.mark.gpu
@pytest.mark.parametrize("size", [2, 4])
@pytest.mark.parametrize("case", range(len(TP_CASES)))
def test_tp_sharded_forward_n_ranks_one_gpu(tmp_path, case, size):
    """BASELINE configs[3] / [4] are tensor-parallel configurations: here the SHARDED forward really runs with `size` ranks -- `size`
    processes that share GPU 0, each with its shard from shard_bundle, every exchange through the hipIpc mailboxes (prefill rows in
    32 KB slices, decode rows and the arg-max key in one hop, the decode step replayed as a hipGraph).  Against the CPU oracle of the
    WHOLE model: (1) teacher-forced, the concatenation of the ranks' vocab-shard logits matches the oracle's logits within the stated
    tolerance at every step and the committed token is the oracle's wherever the oracle's top-2 gap is not a near-tie; (2) all ranks
    commit bit-identical tokens, teacher-forced and chained (rank-order sums, one reduced key: unified_sampling.rs:90-95 tie rule);
    (3) no bounded wait gave up.  Tolerance class: the all-reduce adds `size` partial sums where the single-GPU dot product adds
    once -- same class as a reduction-order change."""
    from helpers import f32
    from test_gpu_model import logits_close, top2_gap
    from oracle import oracle as O
    preset, kwargs, prompt_len, steps, tol, tie = TP_CASES[case]
    kw = dict(kwargs)
    if "layer_kinds" in kw:
        kw["layer_kinds"] = [getattr(D, k) for k in kw["layer_kinds"]]
    cfg = S.PRESETS[preset](**kw)
    bundle = S.build_model(cfg)
    row_mult = S.readout_row_multipliers(cfg)
    prompt = S.synthetic_prompt(prompt_len, cfg.vocab_size)
    om = O.OracleModel(bundle)
    tok, lg = om.prefill(prompt, True)
    o_tokens, o_logits = [tok], [lg]
    for _ in range(steps):
        tok, lg = om.forward([o_tokens[-1]], True)
        o_tokens.append(tok)
        o_logits.append(lg)
    om.close()
    spec = tmp_path / "spec.json"
    spec.write_text(json.dumps({"preset": preset, "kwargs": kwargs, "prompt_len": prompt_len, "steps": steps, "teacher": [int(t) for t in o_tokens]}))
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS="4")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tp_worker.py"), str(r), str(size), str(tmp_path), str(spec)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(size)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("tp workers timed out")
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    res = [np.load(tmp_path / f"out_{r}.npz") for r in range(size)]
    assert all(int(r["p2p_error"]) == 0 for r in res)
    assert [int(r["vocab_offset"]) for r in res] == [r_ * cfg.vocab_size // size for r_ in range(size)]
    # (2) ranks agree bit for bit
    for r in res[1:]:
        assert np.array_equal(r["tf_tokens"], res[0]["tf_tokens"]) and np.array_equal(r["chained"], res[0]["chained"]), "ranks disagree on the committed tokens"
    # (1) against the oracle of the whole model
    worst = 0.0
    for step in range(steps + 1):
        full = np.concatenate([r["tf_logits"][step] for r in res])
        ok = logits_close(o_logits[step], full, row_mult, tol)
        worst = max(worst, logits_close.worst)
        assert ok.all(), f"{preset} tp{size} step {step}: logits off by {logits_close.worst:.3f} sigma (tolerance {tol})"
        got, want, gap = int(res[0]["tf_tokens"][step]), o_tokens[step], top2_gap(o_logits[step])
        assert got == int(np.argmax(f32(full))), f"step {step}: committed token {got} is not the arg-max (ties -> lowest index) of the gathered logits"
        assert got == want or gap < tie, f"{preset} tp{size} step {step}: oracle {want}, tp {got}, oracle top-2 gap {gap:.4f} sigma"
    # chained: identical to the oracle up to the first near-tie
    chained = [int(t) for t in res[0]["chained"]]
    common = 0
    while common < len(o_tokens) and chained[common] == o_tokens[common]:
        common += 1
    if common < len(o_tokens):
        assert top2_gap(o_logits[common]) < tie or any(top2_gap(o_logits[i]) < tie for i in range(common + 1)), \
            f"{preset} tp{size}: chained stream leaves the oracle's at step {common} without a near-tie\noracle {o_tokens}\ntp     {chained}"
    print(f"{preset} tp{size}: worst logit error {worst:.3f} sigma, chained stream identical for {common}/{len(o_tokens)} tokens, {int(res[0]['launches'])} launches per decode step")


@pytest.mark.gpu
@pytest.mark.parametrize("preset", ["tiny-qwen", "tiny-llama"])
def test_tp_stochastic_sampling_and_tree_verify_n_ranks_one_gpu(tmp_path, preset):
    """The two engine features BASELINE configs[3] / [4] (tensor-parallel configurations) could not use before round 5, on 2 ranks sharing
    GPU 0: (a) SamplingMethod::Stochastic over the vocab-sharded read-out -- every rank gathers the whole logit row (tp::gather_logits: its
    shard filed into a zeroed f32 row, all-reduce(sum): exact) and draws with the same derived seed: every token must be EXACTLY what the CPU
    restatement of UnifiedSampling draws from the concatenation of the ranks' own logit shards of that step (unified_sampling.rs:13-99,
    stream.rs:248-258, 598-600), identical on all ranks, in the eager steps and the graph replays; (b) uzu_hip_model_verify_tree /
    _accept on a shard (stream.rs:556-628, 380-470): the token of every node = arg-max (ties -> lowest index) of the gathered per-node
    logits through one all-reduce(max) of per-node keys, under stochastic sampling the node's own seed PRng::derive(context + height);
    the accept compacts each shard's KV rows / advances its DeltaNet heads, and the ranks keep committing identical tokens afterwards."""
    import ctypes as C
    from helpers import f32
    from oracle import oracle as O
    from test_gpu_model import prng_derive
    size, prompt_len, steps = 2, 33, 6
    cfg = S.PRESETS[preset]()
    settings = dict(temperature=25.0, top_k=40)
    seed = 0x0BADC0FFEE123
    spec = tmp_path / "spec.json"
    spec.write_text(json.dumps({"preset": preset, "kwargs": {}, "prompt_len": prompt_len, "steps": steps, "teacher": [1] * (steps + 1),
                                "stochastic": {"seed": seed, "settings": settings}, "tree": True}))
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS="4")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tp_worker.py"), str(r), str(size), str(tmp_path), str(spec)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(size)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("tp workers timed out")
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    res = [np.load(tmp_path / f"out_{r}.npz") for r in range(size)]
    assert all(int(r["p2p_error"]) == 0 for r in res)
    for key in ("st_tokens", "st_many", "st_tree_sampled", "tree_sampled", "tree_accepted", "tree_after"):
        for r in res[1:]:
            assert np.array_equal(r[key], res[0][key]), f"ranks disagree on {key}: {[list(x[key]) for x in res]}"

    def draw(row, position):
        out = np.zeros(1, np.uint32)
        seeds = np.array([prng_derive(seed, position)], np.uint64)
        O.lib().orc_unified_sampling(O.p(np.ascontiguousarray(row)), O.BF16, O.p(out), O.p(seeds), None, 1, C.c_float(settings["temperature"]), 1, settings["top_k"],
                                     0, C.c_float(0.0), 0, C.c_float(0.0), cfg.vocab_size, 1)
        return int(out[0])

    # (a) stochastic stream: position of the sampled row = prompt_len - 1 + step
    st = [int(t) for t in res[0]["st_tokens"]]
    for step in range(steps + 1):
        full = np.concatenate([r["st_logits"][step] for r in res])
        assert st[step] == draw(full, prompt_len - 1 + step), f"{preset}: stochastic step {step}"
    assert len(set(st)) > 2, f"a stochastic stream that never moves: {st}"
    # ... and a stochastic tree pass: node i draws with derive(context + height_i) from ITS gathered row
    ctx_len = int(res[0]["st_tree_ctx"])
    for i, h in enumerate(res[0]["st_tree_heights"]):
        full = np.concatenate([r["st_tree_logits"][i] for r in res])
        assert int(res[0]["st_tree_sampled"][i]) == draw(full, ctx_len + int(h)), f"{preset}: stochastic tree node {i}"
    # (b) greedy tree pass: every node's token is the arg-max of the gathered row, lowest index on ties
    for i in range(len(res[0]["tree_sampled"])):
        full = f32(np.concatenate([r["tree_logits"][i] for r in res]))
        assert int(res[0]["tree_sampled"][i]) == int(np.argmax(full)), f"{preset}: tree node {i}"
    assert len(res[0]["tree_accepted"]) >= 1 and res[0]["tree_accepted"][0] == 0
    print(f"{preset} tp{size}: stochastic stream {st}, tree sampled {list(res[0]['tree_sampled'])}, accepted {list(res[0]['tree_accepted'])}, then {list(res[0]['tree_after'])}")


@pytest.mark.gpu
def test_bench_falls_back_to_rccl_when_a_p2p_exchange_fails():
    """bench.py's recovery branch for first contact with a multi-GPU node (p2p + graph TP decode fails -> every all-reduce on RCCL, eager
    launches, the line says so) executed for real: one rank through the N > 1 code path (--force-dist: nccl process group, RCCL
    communicator, mailbox of its own), with mailbox exchange number 30 failing the way a timed-out wait does (UZU_TP_INJECT_TIMEOUT_AT, the
    sticky group-wide error of csrc/tp.hip).  The line must carry the note, report RCCL as the exchange, the rank count the RCCL communicator
    itself returns (ncclCommCount) and a non-zero number of RCCL collectives, and still hold a positive value."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "4", "--warmup", "1", "--context", "96", "--no-cpu-baseline"]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), UZU_TP_INJECT_TIMEOUT_AT="30",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert "rerun on RCCL, eager" in line["config"].get("note", ""), line["config"]
    assert line["tp"]["exchange"] == "RCCL" and line["config"]["graph"] is False
    assert line["tp"]["rccl_ranks_seen"] == 1 and line["tp"]["rccl_collectives_enqueued"] > 0 and line["tp"]["p2p_exchanges_enqueued"] >= 30
    assert line["value"] > 0 and line["steps"] == 4
