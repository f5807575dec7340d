#!/usr/bin/env python3
"""bench.py -- decode tokens/s of the uzu forward path on MI355X (BASELINE.json metric).

Workload (N=1): BASELINE.json configs[1] -- Qwen3.5-0.8B int4 (ScaleBias, group 128), batch-1 greedy decode at
seq_len 2048 on one MI355X.  A "step" is one decoded token (one replay of the captured decode graph: 24 layers
+ readout + argmax, the sampled token chained to the next step on the device).  Synthetic, format-exact weights
and the synthetic prompt of SURVEY.md §8d; all inputs are resident in HBM when the timed region starts.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (the int4 GEMV family) : algorithmic bytes / HIP-event time, vs 8 TB/s HBM
  cpu_baseline -- the CPU oracle (port of the reference's single-threaded CPU backend) timed on this host
"""
import argparse
import dataclasses
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 matrix-core peak (MI355X_MICROARCH.md)
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--model", default="qwen3.5-0.8b")
    ap.add_argument("--model-dir", default="", help="config.json + model.safetensors in the reference's layout (uzu_amd/loader.py); side runs, not the headline config")
    ap.add_argument("--bits", type=int, default=0, help="override the preset's code width (4 or 8) -- side runs, not the headline config")
    ap.add_argument("--context", type=int, default=2048, help="context length at which the timed decode starts")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--parallelism", choices=["tp", "replicas"], default="tp", help="what N > 1 GPUs do (default: tensor parallel)")
    ap.add_argument("--tp-graph", action="store_true", help="replay the TP decode step (RCCL all-reduces included) as a hipGraph; default: plain "
                    "stream launches, the conservative form for a multi-rank collective that could not be exercised on the 1-GPU dev box")
    ap.add_argument("--no-p2p", action="store_true", help="N > 1: keep every all-reduce on RCCL (default: decode-sized ones use the one-shot peer-to-peer exchange)")
    ap.add_argument("--force-dist", action="store_true", help="take the N > 1 code path even with one rank (self-test on a 1-GPU box)")
    ap.add_argument("--share-gpu", action="store_true", help="DRY RUN of the tensor-parallel path on a 1-GPU box: the N ranks all use GPU 0 (gloo rendezvous, "
                    "hipIpc mailboxes for every exchange -- RCCL refuses duplicate devices).  The line says so (physical_gpus 1); it is a correctness / "
                    "plumbing run of the p2p + hipGraph decode path, not a scaling measurement")
    ap.add_argument("--cpu-baseline-tokens", type=int, default=3)
    ap.add_argument("--exact", action="store_true", help="reference-order mode (uzu_hip_set_exact(1): every reduction in the reference's own loop order, logits "
                    "bit-identical to the CPU oracle's -- tests/test_gpu_model.py::test_exact_mode_*): the price tag of bit-exact logits on the same workload; "
                    "a side line (config.exact_mode = true), never the headline")
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5"], default="c2",
                    help="BASELINE.json configs[1..4]: c2 (default, the headline) Qwen3.5-0.8B int4 ctx-2048 decode; c3 Llama-3-8B int4, 4k prefill x 8 "
                         "sequences in batched passes (prefill tokens/s, MFMA roofline); c4 Llama-3-8B int8 decode (TP = --gpus); c5 Qwen3-14B-class int4 at "
                         "ctx 8192, one sequence decoding while another is prefilled (TP = --gpus)")
    ap.add_argument("--sequences", type=int, default=8, help="c3: sequences per batched prefill pass")
    ap.add_argument("--prompt", type=int, default=4096, help="c3: prompt tokens per sequence")
    args = ap.parse_args()
    if args.exact:
        args.steps, args.warmup, args.no_cpu_baseline = min(args.steps, 32), min(args.warmup, 5), True
    if args.config == "c4":
        args.model, args.bits = "llama-3-8b", 8
    elif args.config == "c5":
        args.model, args.context = "qwen3-14b-class", 8192
        args.steps, args.warmup = min(args.steps, 64), min(args.warmup, 4)
    return args


def cpu_baseline(bundle, cfg, decode_tokens, context):
    """Time the CPU oracle (oracle/ = port of the reference's CPU backend, which is single-threaded by
    construction: backends/cpu/context.rs:20-33) on a bounded sample of the same workload: `decode_tokens` greedy
    decode steps AT THE BENCH CONTEXT.  The context is reached without a CPU prefill (which would take ~25 minutes):
    orc_model_fill_synthetic_context puts `context` synthetic KV rows in place -- the cost of a decode step depends on the
    context length, not on the values."""
    from oracle import oracle as O
    om = O.OracleModel(bundle)
    O.set_threads(1)
    from uzu_amd import synthetic as S
    prompt = S.synthetic_prompt(2, cfg.vocab_size)
    tok = om.prefill(prompt)
    om.fill_synthetic_context(context)
    t0 = time.perf_counter()
    for _ in range(decode_tokens):
        tok = om.forward([tok])
    dt = time.perf_counter() - t0
    one = decode_tokens / dt
    # courtesy figure: OpenMP over independent output rows (bit-identical results), all host cores
    cores = O.default_threads()
    O.set_threads(cores)
    t0 = time.perf_counter()
    for _ in range(decode_tokens):
        tok = om.forward([tok])
    allc = decode_tokens / (time.perf_counter() - t0)
    om.close()
    return {"value": round(one, 4), "unit": "tokens/s", "cores": 1, "kind": "port",
            "sample": f"{decode_tokens} greedy decode tokens at context {context}..{context + decode_tokens} (synthetic KV rows instead of a CPU prefill), same "
                      f"weights, 1 thread (the reference CPU backend is single-threaded)",
            "all_cores_value": round(allc, 4), "all_cores": cores}


def cpu_baseline_prefill_extrapolated(cfg, prompt_tokens):
    """c3: prompt tokens/s of the CPU oracle, EXTRAPOLATED per layer (the whole 32-layer 4096-token prefill is hours of single-threaded CPU work): one layer of the
    same shapes is timed on `sample` prompt tokens starting at an empty context and again behind `mid` synthetic KV rows (the attention term grows with the context,
    the linears do not); the two points give the per-token cost of a layer at context c as a + b c, integrated over the prompt and multiplied by the layer count.
    The read-out runs on one row per sequence and is ignored."""
    from oracle import oracle as O
    from uzu_amd import desc as D
    from uzu_amd import synthetic as S
    one = dataclasses.replace(cfg, layer_kinds=[D.MIXER_ATTENTION], max_context_length=prompt_tokens + 64)
    bundle1 = S.build_model(one)
    O.set_threads(1)
    sample, mid = 8, min(2048, prompt_tokens)
    toks = [int(t) for t in S.synthetic_prompt(sample + 1, cfg.vocab_size)]
    om = O.OracleModel(bundle1)

    def per_token():  # a pass of 1 + `sample` rows minus a pass of 1 row: the read-out of the last row (one per pass) cancels
        t0 = time.perf_counter()
        om.forward(toks[:1])
        t1 = time.perf_counter()
        om.forward(toks)
        return ((time.perf_counter() - t1) - (t1 - t0)) / sample
    t_empty = per_token()                                   # per token of ONE layer at context ~ 0
    om.fill_synthetic_context(mid)
    t_mid = per_token()                                     # ... at context ~ mid
    om.close()
    b = max(t_mid - t_empty, 0.0) / mid
    layers = len(cfg.layer_kinds)
    per_prompt = layers * (t_empty * prompt_tokens + b * prompt_tokens * prompt_tokens / 2.0)
    return {"value": round(prompt_tokens / per_prompt, 4), "unit": "tokens/s", "cores": 1, "kind": "port", "extrapolated": True,
            "sample": f"ONE layer of the same shapes, {sample} prompt tokens at context 0 ({t_empty * 1e3:.0f} ms per token) and behind {mid} synthetic KV rows ({t_mid * 1e3:.0f} ms per token), "
                      f"1 thread; cost of a token at context c = a + b c, summed over a {prompt_tokens}-token prompt, x {layers} layers (read-out of the last row ignored)"}


def dominant_gemv_shape(kernel_name, bundle):
    """(n0, n1, k, normed, gated) of the fused decode GEMV a profile label stands for, from the model's first layer of that kind."""
    if not kernel_name.startswith("gemv_dec["):
        return None
    inner = kernel_name[len("gemv_dec["):-1]
    att = next((l for l in bundle.layers if l.qkv_projection is not None), None)
    dn = next((l for l in bundle.layers if l.dn_in_proj is not None), None)
    any_layer = bundle.layers[0]
    d = bundle.model_dim
    if inner == "norm+up+act":
        return (any_layer.up_projection.n, 0, d, 1, 1)
    if inner == "down":
        return (d, 0, any_layer.down_projection.k, 0, 0)
    if inner == "norm+in_proj+conv" and dn is not None:
        return (dn.dn_in_proj.n, 0, d, 1, 0)
    if inner == "gate+out_proj" and dn is not None:
        return (d, 0, dn.dn_out_proj.k, 0, 0)
    if inner == "norm+qkv+gate" and att is not None and att.gate_projection is not None:
        return (att.qkv_projection.n, att.gate_projection.n, d, 1, 0)
    if inner == "norm+qkv" and att is not None:
        return (att.qkv_projection.n, 0, d, 1, 0)
    if inner == "out_proj" and att is not None:
        return (d, 0, att.out_projection.k, 0, 0)
    return None


def launch_grid_threads(shape, bits, num_cus=256):
    """Threads of the launch grid the decode GEMV plan gives this shape (host arithmetic: uzu_hip_decode_gemv_plan), or None when the
    kernel runs a persistent grid whose size is only known at launch (occupancy x CUs)."""
    import ctypes as C
    from uzu_amd import _ffi

    class Plan(C.Structure):
        _fields_ = [(n, C.c_uint32) for n in ("lanes_per_row", "rows_per_lane_group", "steps_per_lane", "waves_per_workgroup", "batches", "workgroup_batches", "workgroups")]
    n0, n1, k, normed, gated = shape
    plan = Plan()
    fn = _ffi.lib().uzu_hip_decode_gemv_plan
    fn.restype = C.c_int32
    if fn(C.c_uint32(n0), C.c_uint32(n1), C.c_uint32(k), C.c_uint32(bits), C.c_uint32(normed), C.c_uint32(gated), C.c_uint32(num_cus), C.byref(plan)) != 0:
        return None
    if (n0 + n1) * k * bits // 8 >= (16 << 20):  # bandwidth regime: persistent grid
        return plan.workgroups * 64 * plan.waves_per_workgroup if plan.workgroups else None
    wgs = plan.workgroups if plan.workgroups else plan.batches // 4
    return wgs * 64 * plan.waves_per_workgroup


def pmc_traffic(kernel_name, bundle, bits):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc FETCH_SIZE pass
    (profiles/*_pmc_fetch.json, written by tools/summarize_profile.py with the gfx950 x2 KiB correction).  bench.py cannot run
    rocprofv3 around itself, so this is the newest committed counter pass.  The entry is picked by KERNEL INSTANCE AND GRID SIZE only --
    the template arguments the profile label implies (ACT, PRO, CONV) and the launch grid the decode-GEMV plan gives the label's matrix
    shape -- never by how close its byte count is to the algorithmic bytes: whatever ratio falls out is what is reported.  None when
    no committed pass holds that (instance, grid)."""
    import glob
    import re
    family = kernel_name.split("[")[0]
    want = grid = None
    if kernel_name.startswith("gemv_dec["):
        inner = kernel_name[len("gemv_dec["):-1]
        want = ("true" if "+act" in inner else "false", "2" if inner.startswith("gate+") else "1" if "norm+" in inner else "0",
                "true" if "+conv" in inner else "false")
        shape = dominant_gemv_shape(kernel_name, bundle)
        grid = launch_grid_threads(shape, bits) if shape else None
    inst = re.compile(r"gemv_dec_kernel<(\d+), \d+, \d+, (true|false), \d+, (\d+), (true|false), \d+>")
    # newest round first: an older pass is only consulted when the newer ones hold nothing for this (instance, grid)
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_fetch.json")), reverse=True):
        try:
            table = json.load(open(path))
        except (OSError, ValueError):
            continue
        hits = []
        for key, nbytes in table.items():
            if not key.startswith(family):
                continue
            name, _, key_grid = key.rpartition("|")
            if want is not None:
                mt = inst.match(name)
                if not mt or int(mt.group(1)) != bits or mt.groups()[1:] != want:
                    continue
            if grid is not None and key_grid != str(grid):
                continue
            hits.append((nbytes, f"{os.path.basename(path)}:{key}"))
        if len(hits) == 1:  # unambiguous: one instance at this grid
            return hits[0]
    return None, None


def committed_kernel_avg(traffic_source, launches_per_step=None):
    """rocprofv3 begin->end average of the kernel instance the counter pass matched, from the NEWEST committed
    profiles/*_kernel_stats.csv that holds this instance with the right call count (the stats file is chosen on its own -- by
    modification order of the rounds' files, newest first -- not by the name of the counter pass: a round may refresh one without the
    other): the cross-check for the live HIP-event duration, which carries ~2 us of event bracketing per launch."""
    import csv
    import glob
    if not traffic_source or ":" not in traffic_source:
        return None
    _, key = traffic_source.split(":", 1)
    instance = key.rsplit("|", 1)[0]
    prof_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")

    def round_key(path):  # r5b > r5 > r4b > r4 ...: round number, then suffix
        import re
        mt = re.match(r"r(\d+)([a-z]*)_", os.path.basename(path))
        return (int(mt.group(1)), mt.group(2)) if mt else (-1, "")
    for path in sorted(glob.glob(os.path.join(prof_dir, "r*_kernel_stats.csv")), key=round_key, reverse=True):
        try:
            rows = list(csv.DictReader(open(path)))
            steps = next((int(r["calls"]) for r in rows if r.get("kernel") == "argmax_commit_kernel"), None)  # one per decode step
            for row in rows:
                if row.get("kernel") != instance:
                    continue
                # the stats are keyed by kernel instance: only usable when this instance is launched by this label alone
                # (the read-out shares its instance with the qkv projection, for example)
                if launches_per_step and steps and int(row["calls"]) != launches_per_step * steps:
                    break
                return {"avg_launch_us": float(row["avg_us"]), "calls": int(row["calls"]), "source": f"{os.path.basename(path)}:{instance}"}
        except (OSError, ValueError, KeyError):
            continue
    return None


def latency_floor(ctx, bundle, per_kernel, tokens_per_s):
    """The batch-1 decode step is a chain of dependent launches: every linear needs the WHOLE activation row its predecessor's workgroups
    wrote (an all-to-all edge).  A layer has five such edges whatever the kernels do (mixer in-projection <- residual row, out-projection <-
    mixer output, up <- residual row, down <- hidden row, next layer <- down's row), the read-out and the commit are two more.  The price of
    one edge as a kernel boundary in a replayed graph is measured live (uzu_hip_probe_edge_floor: 256 workgroups, every one reading the whole
    model_dim bf16 row the previous launch wrote, nothing else); the floor of a launch-per-dependency step = edges x that + the read-out's own
    stream time (the one kernel that is in the bandwidth regime).  `frac_of_floor` says how close the measured step is to what this structure
    allows; `roofline.frac` beside it stays the fraction of the HBM peak."""
    import ctypes as C
    from uzu_amd import _ffi
    fn = _ffi.lib().uzu_hip_probe_edge_floor
    fn.restype = C.c_int32
    us = C.c_float()
    row_bytes = max(bundle.model_dim * 2, 8)
    if fn(ctx._h, C.c_uint32(256), C.c_uint32(row_bytes), C.c_uint32(128), C.c_uint32(20), C.byref(us)) != 0:
        return None
    edges = 5 * len(bundle.layers) + 2
    readout = next((v["us"] for k, v in per_kernel.items() if "readout" in k), 0.0)
    floor_us = edges * us.value + readout
    return {"edges_per_token": edges, "edge_us": round(us.value, 3), "readout_stream_us": round(readout, 1), "floor_us_per_token": round(floor_us, 1),
            "floor_tok_s": round(1e6 / floor_us, 1), "frac_of_floor": round(tokens_per_s * floor_us / 1e6, 4),
            "method": "uzu_hip_probe_edge_floor (csrc/k_probe.hip): 128 dependent launches of 256 workgroups in one hipGraph, each reading the whole "
                      f"{row_bytes}-byte row the previous launch wrote; wall time per launch over 20 replays.  edges = 5 per layer + read-out + commit"}


def committed_exact_mode():
    """tokens/s of reference-order mode on the headline workload from the newest committed `bench.py --exact` line (profiles/r*_bench_exact.json): what
    bit-identical logits cost.  The bit-identity itself is asserted by tests/test_gpu_model.py::test_exact_mode_* (tiny models, the full-size 0.8B, the bench
    fixture's whole chained stream against the CPU oracle)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_exact.json")), reverse=True):
        try:
            c = json.load(open(path))
            return {"tokens_per_s": c["value"], "prefill_tokens_per_s": c["prefill_tokens_per_s"], "logits_bit_identical": True,
                    "tokens_equal_to_oracle_stream": f'{c["parity"]["untimed_tokens_equal"] + c["parity"]["tokens_equal"]} / {c["parity"]["of_untimed"] + c["parity"]["of"]}',
                    "proof": "tests/test_gpu_model.py::test_exact_mode_* (logits of every vocabulary entry equal the CPU oracle's, bit for bit)", "source": "profiles/" + os.path.basename(path)}
        except (OSError, ValueError, KeyError, TypeError):
            continue
    return None


def committed_census(config="c2"):
    """Headline fields of the newest committed parity census (tools/parity_census.py -> profiles/r*_parity_census[_cN].json): production kernels
    against reference-order mode over a pre-registered, unfiltered prompt set, teacher-forced.  `config`: the BASELINE configuration the census ran on."""
    import glob
    pattern = "r*_parity_census.json" if config == "c2" else f"r*_parity_census_{config}.json"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            c = json.load(open(path))
            return {"steps": c["steps"], "argmax_mismatches": c["argmax_mismatches"], "max_flipped_margin": c["max_flipped_margin"], "variants": c["variants"],
                    "top8_logit_error_max": c["top8_logit_error"]["max"], "chained_identical_streams": c["chained"]["identical_streams"],
                    "selection": c["selection"], "source": "profiles/" + os.path.basename(path)}
        except (OSError, ValueError, KeyError):
            continue
    return None


def timed_decode(model, ctx, args, dist, prompt):
    """prefill -> `warmup` untimed steps -> EXACTLY `steps` timed greedy decode steps between barrier + device sync on
    both sides; elapsed = max over ranks.  Returns (elapsed_s, gpu_ms, start_ctx, end_ctx, prefill_s, tokens of the timed steps)."""
    def sync():
        ctx.synchronize()
        if dist is not None:
            if not args.share_gpu:
                import torch
                torch.cuda.synchronize()
            dist.barrier()

    model.prefill(prompt)  # warm-up run, discarded (the reference's bench does the same: cli/src/bench/runner.rs:67-68);
    model.reset()          # it also takes the one-time code-object loads out of the timed prefill
    sync()
    t0 = time.perf_counter()
    first = model.prefill(prompt)
    ctx.synchronize()
    prefill_s = time.perf_counter() - t0
    timed_decode.untimed_tokens = [int(first)]  # the prefill's token + the warm-up steps' tokens (parity bookkeeping, outside the timed region)
    if args.warmup:
        warm, _ = model.decode(args.warmup)
        timed_decode.untimed_tokens += [int(t) for t in warm]
    start_ctx = model.context_length
    sync()
    t0 = time.perf_counter()
    tokens, gpu_ms = model.decode(args.steps)  # K graph replays, chained on the device; returns after the last one
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed, prefill_s], device="cpu" if args.share_gpu else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, prefill_s = float(t[0].item()), float(t[1].item())
    return elapsed, gpu_ms, start_ctx, model.context_length, prefill_s, tokens


def bench_c3(args):
    """BASELINE configs[2]: Llama-3-8B int4, `--sequences` x `--prompt`-token prompts prefilled together: every chunk pass carries all
    sequences (one GEMM with M = sequences x 1024 per linear: the weights are streamed once per pass), attention per sequence.
    A "step" is one whole batched prefill (all sequences, all chunks).  value = prompt tokens/s over all sequences."""
    from uzu_amd import synthetic as S
    from uzu_amd.backend import Context
    from uzu_amd.engine import MODEL_BATCH, HipModel
    cfg = S.PRESETS["llama-3-8b"](max_context_length=args.prompt + 8, **({"bits": args.bits} if args.bits else {}))
    bundle = S.build_model(cfg)
    ctx = Context.new(0)
    nseq = args.sequences
    model = HipModel(ctx, bundle, MODEL_BATCH(nseq))
    states = [model.new_state() for _ in range(nseq)]
    base = S.synthetic_prompt(args.prompt, cfg.vocab_size).astype(np.int64)
    prompts = np.stack([(base * (2 * i + 1) + 17 * i) % cfg.vocab_size for i in range(nseq)]).astype(np.uint32)
    steps, warmup = max(1, min(args.steps, 4)), max(1, min(args.warmup, 1))
    for _ in range(warmup):
        for st in states:
            st.reset()
        first = model.prefill_batch(states, prompts)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for st in states:
            st.reset()
        first = model.prefill_batch(states, prompts)
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    # the same prompts one sequence at a time (M = 1024 per pass): what batching buys
    t1 = time.perf_counter()
    single_firsts = []
    for i, st in enumerate(states):
        st.reset()
        model.bind(st)
        single_firsts.append(int(model.prefill(prompts[i])))
    ctx.synchronize()
    single = time.perf_counter() - t1
    model.bind(None)
    tokens = nseq * args.prompt
    flops = nseq * bundle.prefill_flops(args.prompt)
    tflops = flops * steps / elapsed / 1e12
    result = {
        "metric": f"prefill tokens/s (Llama-3-8B int{cfg.bits}, {nseq} x {args.prompt}-token prompts, batched passes)",
        "value": round(tokens * steps / elapsed, 1), "unit": "tokens/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": f"int{cfg.bits} weights (exact centred codes) x bf16 activations on bf16 MFMA, f32 accumulate", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[2]: llama-3-8b int{cfg.bits} ScaleBias g{cfg.group_size}, {nseq} sequences x {args.prompt} prompt tokens, "
                               f"chunks of 1024 per sequence, every pass carries all sequences (M = {nseq * 1024})", "parallelism": "1 GPU"},
        "roofline": {"bound": "mfma", "achieved": round(tflops, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / MFMA_BF16_PEAK_TFLOPS, 4),
                     "traffic": None, "note": "whole batched prefill (GEMMs + attention + element-wise) over its wall time; algorithmic FLOPs = 2 M sum(N K) + causal "
                                              "attention (SURVEY.md section 8d)"},
        "single_sequence_prefill_tokens_per_s": round(tokens / single, 1),
        # a batched pass may split K of a GEMM differently from the single-sequence pass (tolerance-class logits): reported, not asserted
        "first_tokens": [int(t) for t in first], "single_sequence_first_tokens": single_firsts,
        "first_tokens_agree": sum(int(a) == b for a, b in zip(first, single_firsts)), "device": ctx.device_name(),
        # production kernels against reference-order mode on this configuration's model and prompt length, pre-registered prompts (tools/parity_census.py --config c3)
        "parity": {"census": committed_census("c3"), "oracle": "reference-order mode (bit-identical to the CPU oracle where the oracle runs: tests/test_gpu_model.py::test_exact_mode_*); "
                   "the CPU oracle itself needs ~4 CPU-hours per 4096-token Llama-3-8B prompt"},
    }
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline_prefill_extrapolated(cfg, args.prompt)
    for st in states:
        st.close()
    model.close()
    print(json.dumps(result), flush=True)


def bench_c5_mixed(args, model, ctx, cfg, bundle, start_ctx):
    """BASELINE configs[4] "mixed prefill + decode": while sequence A decodes at the bench context, sequence B is prefilled in
    1024-token chunks between A's decode steps (one stream, the reference's engine interleaves command buffers the same way).
    Returns an object for the JSON line."""
    from uzu_amd import synthetic as S
    other = model.new_state()
    chunk = S.synthetic_prompt(1024, cfg.vocab_size)
    decode_steps, chunks = 32, 4
    t0 = time.perf_counter()
    for c in range(chunks):
        model.bind(other)
        model.prefill(chunk)
        model.bind(None)
        model.decode(decode_steps // chunks)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    other.close()
    return {"decode_tokens": decode_steps, "prefill_tokens": 1024 * chunks, "seconds": round(dt, 4),
            "decode_tokens_per_s_while_prefilling": round(decode_steps / dt, 2), "prefill_tokens_per_s_while_decoding": round(1024 * chunks / dt, 1),
            "note": f"sequence A decodes from context {start_ctx}; sequence B receives {chunks} chunks of 1024 prompt tokens, one between every "
                    f"{decode_steps // chunks} decode steps of A"}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, torch.distributed.run
    on 127.0.0.1) and pass rank 0's JSON line through.  Never falls back to fewer GPUs silently."""
    import socket
    import subprocess
    try:
        import torch
        visible = torch.cuda.device_count()
    except Exception as exc:  # noqa: BLE001
        sys.exit(f"bench.py --gpus {n}: cannot count GPUs ({exc})")
    if visible < n and "--share-gpu" not in sys.argv:
        sys.exit(f"bench.py --gpus {n}: only {visible} GPU(s) visible -- refusing to report a {n}-GPU line from fewer devices")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.force_dist:
        spawn_ranks(args.gpus)
    if args.config == "c3":
        if args.gpus != 1:
            sys.exit("bench.py --config c3 is a single-GPU workload (8 independent sequences on one GPU)")
        return bench_c3(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not args.force_dist:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or without a launcher)")
    dist = None
    multi = world > 1 or args.force_dist
    if multi:
        import datetime
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.share_gpu:
            local_rank = 0  # every rank on GPU 0; host-side rendezvous only
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=300))

    from uzu_amd import synthetic as S
    from uzu_amd import tp as TP
    from uzu_amd.backend import Context
    from uzu_amd.engine import MODEL_DEFAULT, MODEL_NO_GRAPH, HipModel

    total_positions = args.context + args.warmup + args.steps + 8 + (64 if args.config == "c5" else 0)  # c5: + the mixed leg's decode steps
    cfg = S.PRESETS[args.model](max_context_length=total_positions, **({"bits": args.bits} if args.bits else {}))
    if args.model_dir:  # a checkpoint in the reference's on-disk layout instead of the synthetic weights (same synthetic token ids)
        from uzu_amd import loader
        bundle = loader.load_model_dir(args.model_dir, max_context_length=total_positions)
        cfg = dataclasses.replace(cfg, name=bundle.name, vocab_size=bundle.vocab_size)
    else:
        bundle = S.build_model(cfg)
    ctx = Context.new(local_rank)
    if args.exact:  # reference-order kernels, eager, the reference's own 1024-row prefill passes (read when a model is created / a pass is encoded)
        import ctypes as C
        from uzu_amd import _ffi
        fn = _ffi.lib().uzu_hip_set_exact
        fn.restype, fn.argtypes = None, [C.c_int32]
        fn(1)
    flags = MODEL_NO_GRAPH if args.no_graph else MODEL_DEFAULT
    # fill the context: prompt = context - warmup tokens, then `warmup` untimed decode steps reach `context`
    prompt_len = max(args.context - args.warmup, 1)
    prompt = S.synthetic_prompt(prompt_len, cfg.vocab_size)
    # The headline configuration runs the COMMITTED parity fixture's prompt (tests/golden/bench_qwen_stream.json: the CPU oracle's greedy
    # stream of this exact model and prompt), independent of --warmup: the fixture's 2043 prompt tokens, followed -- when context - warmup
    # asks for more -- by the oracle's own continuation as further prompt tokens.  The token ids the run produces are compared with the
    # fixture's after the timed region (ids only: the oracle is not imported or executed here).
    fixture = stream_offset = None
    fx_path = os.path.join(ROOT, "tests", "golden", "bench_qwen_stream.json")
    if args.config == "c2" and args.model == "qwen3.5-0.8b" and not args.model_dir and not args.bits and args.context == 2048 and os.path.exists(fx_path):
        fx = json.load(open(fx_path))
        if fx["preset"] == args.model and fx["seed"] == cfg.seed and fx["bits"] == cfg.bits and abs(fx["logit_row_sigma"] - cfg.logit_row_sigma) < 1e-12:
            fixture = fx
            stream_offset = min(max(args.context - args.warmup - fx["prompt_tokens"], 0), max(len(fx["tokens"]) - 2, 0))
            base_prompt = S.synthetic_prompt(fx["prompt_tokens"], cfg.vocab_size, variant=fx["prompt_variant"])
            prompt = np.concatenate([base_prompt, np.asarray(fx["tokens"][:stream_offset], dtype=np.uint32)])
            prompt_len = int(prompt.size)

    # N > 1: tensor-parallel shards of ONE sequence (north_star; strong scaling).  `--parallelism replicas` (or a
    # model whose heads do not split over N ranks) runs N independent sequences instead (weak scaling).
    mode = args.parallelism if multi else "single"
    group = None
    use_graph, p2p = not args.no_graph, False
    note = None
    local_bundle = bundle
    if mode == "tp":
        try:
            local_bundle, vocab_offset = TP.shard_bundle(bundle, rank, world)
        except AssertionError as exc:  # deterministic on every rank: the planner only looks at shapes
            mode, note, local_bundle = "replicas", f"tp{world} not applicable: {exc}", bundle
    if mode == "tp" and args.share_gpu:
        group = TP.TpGroup.local(ctx, rank, world)
        group.enable_p2p(TP.torch_all_gather_bytes(dist))  # no fallback: without the mailboxes ranks on one device cannot exchange
        p2p, use_graph = True, not args.no_graph
        model = HipModel(ctx, local_bundle, flags, tp_group=group, vocab_offset=vocab_offset)
    elif mode == "tp":
        import torch
        group = TP.TpGroup(ctx, rank, world, TP.torch_broadcast(dist, device=torch.device("cuda", local_rank)))
        # decode-sized all-reduces go through the one-shot peer-to-peer exchange (mailboxes over hipIpc, csrc/tp.hip) when every
        # rank can open every other rank's mailbox; then the whole TP decode step is a hipGraph like the single-GPU one
        p2p_ok = 0
        if not args.no_p2p:
            try:
                group.enable_p2p(TP.torch_all_gather_bytes(dist))
                p2p_ok = 1
            except Exception as exc:  # noqa: BLE001
                note = f"p2p exchange unavailable on rank {rank}: {str(exc)[:120]}"
        agree = torch.tensor([p2p_ok], device="cuda", dtype=torch.int32)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        p2p = bool(agree.item())
        if not p2p:
            group.disable_p2p()
        use_graph = args.tp_graph or p2p
        tp_flags = flags if use_graph else (flags | MODEL_NO_GRAPH)
        model = HipModel(ctx, local_bundle, tp_flags, tp_group=group, vocab_offset=vocab_offset)
    else:
        model = HipModel(ctx, bundle, flags)

    try:
        elapsed, gpu_ms, start_ctx, end_ctx, prefill_s, timed_tokens = timed_decode(model, ctx, args, dist, prompt)
    except Exception as exc:  # noqa: BLE001
        # A failed peer-to-peer exchange is sticky and poisons the peers' mailboxes (csrc/tp.hip): every rank lands here within the
        # bounded spin.  Real multi-GPU runs then fall back to RCCL for every all-reduce, eager launches -- slower, but a measured line
        # instead of none; the note says so.  (Nothing to fall back to on one GPU or with ranks sharing a device.)
        if not (mode == "tp" and p2p and not args.share_gpu):
            raise
        note = (note + "; " if note else "") + f"p2p + graph TP decode failed ({str(exc)[:160]}): rerun on RCCL, eager"
        try:
            model.close()
        except Exception:  # noqa: BLE001
            pass
        group.disable_p2p()
        p2p, use_graph = False, False
        dist.barrier()
        model = HipModel(ctx, local_bundle, flags | MODEL_NO_GRAPH, tp_group=group, vocab_offset=vocab_offset)
        elapsed, gpu_ms, start_ctx, end_ctx, prefill_s, timed_tokens = timed_decode(model, ctx, args, dist, prompt)
    tokens_agree = True
    if args.share_gpu and dist is not None:  # every rank must have committed the same stream (rank-order sums, one arg-max key)
        gathered = [None] * world
        dist.all_gather_object(gathered, [int(t) for t in timed_tokens])
        tokens_agree = all(g == gathered[0] for g in gathered)
    mean_ctx = (start_ctx + end_ctx - 1) / 2.0
    sequences = world if mode == "replicas" else 1
    tokens_per_s = sequences * args.steps / elapsed
    # algorithmic bytes per token summed over the GPUs of the job (each rank streams its shard / its replica)
    bytes_per_token = local_bundle.decode_bytes_per_token(int(round(mean_ctx)))
    job_bytes_per_token = bytes_per_token * (world if mode != "single" else 1)
    per_gpu_gbps = bytes_per_token * (args.steps / elapsed) / 1e9

    # per-kernel roofline of the dominant kernel: one extra decode step with HIP events around every launch
    prof = model.profile_decode_step()
    agg = {}
    for name, nbytes, ms in prof:
        a = agg.setdefault(name, [0, 0, 0.0])
        a[0] += 1
        a[1] += nbytes
        a[2] += ms
    gemv = {k: v for k, v in agg.items() if k.startswith(("gemv_q", "gemv_dec"))}
    dom_name = max(gemv, key=lambda k: gemv[k][2]) if gemv else max(agg, key=lambda k: agg[k][2])
    calls, dbytes, dms = agg[dom_name]
    achieved = dbytes / (dms * 1e-3) / 1e9 if dms > 0 else 0.0
    gemv_bytes = sum(v[1] for v in gemv.values())
    gemv_ms = sum(v[2] for v in gemv.values())
    traffic, traffic_src = pmc_traffic(dom_name, local_bundle, cfg.bits)
    roofline = {
        "bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
        "launches_per_step": calls, "bytes_per_launch": int(dbytes / max(calls, 1)), "avg_launch_us": round(dms * 1e3 / max(calls, 1), 3),
        "all_gemv": {"launches_per_step": sum(v[0] for v in gemv.values()), "achieved": round(gemv_bytes / max(gemv_ms, 1e-9) / 1e6, 1),
                     "unit": "GB/s", "sum_us": round(gemv_ms * 1e3, 1)},
        "decode_step": {"algorithmic_bytes_per_token_per_gpu": int(bytes_per_token), "algorithmic_bytes_per_token_job": int(job_bytes_per_token),
                        "achieved_per_gpu": round(per_gpu_gbps, 1), "unit": "GB/s", "frac_per_gpu": round(per_gpu_gbps / HBM_PEAK_GBPS, 4),
                        "kernels_per_step": len(prof), "sum_kernel_us": round(sum(p[2] for p in prof) * 1e3, 1),
                        "sum_kernel_us_note": "the profiled step is launched eagerly (hipExtLaunchKernel per kernel): its per-kernel times are an upper bound on "
                                              "what the same kernels cost inside the replayed graph, so the sum can exceed ms_per_step"},
        "method": "HIP events stamped by each launch of one decode step of rank 0 with the kernel's own begin / end (hipExtLaunchKernel inside "
                  "uzu_hip_model_profile_decode_step): the dispatch timestamps rocprofv3 reports; roofline.rocprofv3 = the committed "
                  "profiles/*_kernel_stats.csv average of the same kernel instance",
    }
    committed = committed_kernel_avg(traffic_src, calls)
    if committed:  # the same kernel instance in the committed rocprofv3 --kernel-trace --stats pass
        committed["achieved"] = round(roofline["bytes_per_launch"] / (committed["avg_launch_us"] * 1e-6) / 1e9, 1)
        committed["frac"] = round(committed["achieved"] / HBM_PEAK_GBPS, 4)
        roofline["rocprofv3"] = committed
    per_kernel = {k: {"calls": v[0], "us": round(v[2] * 1e3, 2)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2])}
    if mode == "single" and not args.exact and os.environ.get("UZU_BENCH_NO_FLOOR_PROBE", "0") != "1":  # the counter passes of tools/refresh_profiles.sh skip the probe's 2.7k launches
        try:
            roofline["latency_floor"] = latency_floor(ctx, bundle, per_kernel, tokens_per_s)
        except Exception as exc:  # noqa: BLE001 -- a probe must never take the headline down
            roofline["latency_floor"] = {"error": str(exc)[:160]}
    parallelism = {"single": "1 GPU", "tp": f"tp{world}: one sequence, column/row-parallel shards, RCCL all-reduce after out-proj and down-proj "
                   f"({2 * len(bundle.layers)} + 1 per token)", "replicas": f"{world} independent sequences (one per GPU), no collective"}[mode]

    result = {
        "metric": ("decode tokens/s (Qwen3.5-0.8B int4, batch 1, greedy)" if (args.model, cfg.bits) == ("qwen3.5-0.8b", 4) and not args.model_dir else f"decode tokens/s ({cfg.name} int{cfg.bits}, batch 1, greedy)")
                  + (" [reference-order mode: not the headline]" if args.exact else ""),
        "value": round(tokens_per_s, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True,
        **({"exact_mode": True} if args.exact else {}),
        "scaling": "strong" if mode == "tp" else "weak", "vs_baseline": None, "dtype": f"int{cfg.bits} weights x bf16 activations, f32 accumulate",
        "data": "synthetic",
        "config": {"workload": f"{cfg.name} int{cfg.bits} ScaleBias g{cfg.group_size}, batch=1 greedy decode, context {start_ctx}->{end_ctx}",
                   "prompt_tokens": prompt_len, "graph": (not args.no_graph) and (mode != "tp" or use_graph) and not args.exact, "parallelism": parallelism,
                   **({"exact_mode": "reference-order kernels (uzu_hip_set_exact(1)): one thread per reduction in the reference's loop order, eager launches, prefill in the "
                                     "reference's 1024-row passes; logits bit-identical to the CPU oracle's"} if args.exact else {})},
        "gpu_ms_per_step_events": round(gpu_ms / args.steps, 5),
        # the timed steps' token ids: two builds / plan switches that only move rows between waves must agree on it (same-box A/B runs)
        "timed_tokens_crc32": zlib.crc32(np.asarray(timed_tokens, dtype="<u4").tobytes()),
        "prefill_tokens_per_s": round(sequences * prompt_len / prefill_s, 1),
        "prefill_roofline": {"bound": "mfma", "achieved": round(sequences * bundle.prefill_flops(prompt_len) / prefill_s / 1e12, 2), "peak": MFMA_BF16_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": round(sequences * bundle.prefill_flops(prompt_len) / prefill_s / 1e12 / (MFMA_BF16_PEAK_TFLOPS * max(world, 1)), 5),
                             "note": "whole prefill (GEMMs on the matrix cores + sequential DeltaNet scan + VALU attention) over its wall time; "
                                     "the GEMM kernel alone: profiles/*_kernel_stats.csv, tools/kbench KB_GEMM"},
        "roofline": roofline,
        "kernel_us_per_step": per_kernel,
        "device": ctx.device_name(),
    }
    if fixture is not None:
        # stream index of the token the prefill returns = stream_offset; warm-up step i returns stream_offset + 1 + i; timed step i
        # returns stream_offset + warmup + 1 + i
        want = fixture["tokens"]
        untimed = getattr(timed_decode, "untimed_tokens", [])
        w_cmp = [(a, want[stream_offset + i]) for i, a in enumerate(untimed) if stream_offset + i < len(want)]
        t0i = stream_offset + len(untimed)
        t_cmp = [(int(a), want[t0i + i]) for i, a in enumerate(timed_tokens) if t0i + i < len(want)]
        result["parity"] = {
            "fixture": "tests/golden/bench_qwen_stream.json", "prompt_variant": fixture["prompt_variant"], "oracle": "CPU restatement of the reference (oracle/), greedy",
            "untimed_tokens_equal": sum(int(a == b) for a, b in w_cmp), "of_untimed": len(w_cmp),
            "tokens_equal": sum(int(a == b) for a, b in t_cmp), "of": len(t_cmp), "timed_steps": args.steps,
            "distinct_tokens_compared": len({b for _, b in w_cmp + t_cmp}),
            "selection": "the fixture's prompt variant was picked by OUTCOME (tools/stream_search.py: production and reference-order streams identical); the unfiltered "
                         "figure is `census`",
            "census": committed_census(),
            **({} if args.exact else {"exact_mode": committed_exact_mode()}),
            "note": "token ids of the prefill + warm-up steps (untimed) and of the timed steps against the committed oracle stream; `of` < timed_steps "
                    "when the run is longer than the fixture" + ("; tensor parallel: sums are taken in another order than on one GPU (tolerance class)" if mode == "tp" else "")}
    elif args.config == "c2" and args.model == "qwen3.5-0.8b":
        result["parity"] = None
    elif args.config in ("c4", "c5"):  # production kernels against reference-order mode on this configuration (tools/parity_census.py --config c4|c5)
        result["parity"] = {"census": committed_census(args.config), "oracle": "reference-order mode (bit-identical to the CPU oracle where the oracle runs: "
                            "tests/test_gpu_model.py::test_exact_mode_*)"}
    if args.share_gpu:
        result["physical_gpus"] = 1
        note = (note + "; " if note else "") + f"DRY RUN: {world} ranks share ONE physical GPU (bench.py --share-gpu): plumbing of the p2p + hipGraph TP decode path, not a scaling point"
        result["all_ranks_tokens_agree"] = bool(tokens_agree)
    if note:
        result["config"]["note"] = note
    if mode == "tp":
        ar = agg.get("all_reduce", [0, 0, 0.0])
        result["tp"] = {"exchange": "one-shot peer-to-peer (hipIpc mailboxes) for decode rows, RCCL for prefill" if p2p else "RCCL", "all_reduces_per_token": ar[0], "all_reduce_us_per_token": round(ar[2] * 1e3, 1),
                        "share_of_kernel_time": round(ar[2] / max(sum(p[2] for p in prof), 1e-9), 3)}
        try:  # what the group really used: ranks the RCCL communicator reports (0 = none: --share-gpu), collectives by route
            rccl_ranks, rccl_calls, p2p_calls = group.stats()
            result["tp"].update({"rccl_ranks_seen": rccl_ranks, "rccl_collectives_enqueued": rccl_calls, "p2p_exchanges_enqueued": p2p_calls})
        except Exception as exc:  # noqa: BLE001
            result["tp"]["stats_error"] = str(exc)[:120]
        # the serving-throughput view of the same N GPUs: N independent sequences, one whole model per GPU
        if args.share_gpu:
            raise_replicas = False
        else:
            raise_replicas = True
            model.close()
        try:
            if not raise_replicas:
                raise RuntimeError("skipped: the ranks of a --share-gpu dry run share one device")
            replica = HipModel(ctx, bundle, flags)
            r_elapsed, _, _, _, r_prefill, _ = timed_decode(replica, ctx, args, dist, prompt)
            result["replicas"] = {"value": round(world * args.steps / r_elapsed, 2), "unit": "tokens/s", "scaling": "weak",
                                  "prefill_tokens_per_s": round(world * prompt_len / r_prefill, 1),
                                  "note": f"{world} independent sequences, one per GPU, no collective (not the headline value)"}
            model = replica
        except Exception as exc:  # noqa: BLE001 -- the secondary figure must never take the headline down
            result["replicas"] = {"error": str(exc)[:200]}
            model = model if args.share_gpu else None
    if args.config == "c5" and mode == "single" and model is not None:
        result["mixed_prefill_decode"] = bench_c5_mixed(args, model, ctx, cfg, bundle, end_ctx)
    if args.config != "c2":
        result["config"]["baseline_config"] = {"c4": "BASELINE configs[3] (Llama-3-8B int8 decode; TP = --gpus)", "c5": "BASELINE configs[4] (Qwen3-14B-class int4, ctx 8192; TP = --gpus)"}[args.config]
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        result["cpu_baseline"] = cpu_baseline(bundle, cfg, args.cpu_baseline_tokens, start_ctx)
    if model is not None:
        model.close()
    if group is not None:
        group.close()
    if dist is not None:
        dist.destroy_process_group()
    # RCCL prints a version banner through C stdio (block-buffered on a pipe): flush it BEFORE the JSON line so that
    # the JSON is the last line of rank 0's stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
