#!/usr/bin/env python3
"""GPU box: the reference's own quantised-GEMV microbenchmark grid through the C ABI's MatmulKernel (VERDICT r5 "missing" 5).

Grid (BU/tests/unit/backends/common/kernel/matmul/quant_gemv_bench.rs:57-67, BU/src/tests/matmul/shape.rs:69-80):
  (N, K) in {4096, 14336}^2, M in {1, 2, 4}, seven labels: ScaleBias / ZP x group 32 / 64 / 128 at 4 bits, ZP group 64 at 8 bits; bf16.
Protocol (BU/BENCHMARKS.md:130-134, BU/src/tests/cold_pool.rs, BU/src/tests/matmul/bench.rs:54-74): a cold pool of ceil(256 MiB / weight bytes)
copies of the weight buffers, one command buffer holding `iters` encodes that walk the pool round robin, the figure = GPU execution time of that
command buffer / iters.  Here: uzu_hip_cmdbuf_* with UZU_CMDBUF_EAGER, gpu_execution_time() (HIP events around the submitted work).

GB/s = algorithmic bytes (codes + scales + biases | zero points + the M activation rows + the M output rows) / time; `frac` against 8 TB/s.

  python tools/gemv_grid.py --out gpurun_out/r6_gemv_grid.json [--iters 64] [--quick]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LABELS = [  # (label, group, bits, zero-point?)
    ("ScaleBias_BF16_gs32", 32, 4, False), ("ZP_BF16_gs32", 32, 4, True), ("ScaleBias_BF16_gs64", 64, 4, False), ("ZP_BF16_gs64", 64, 4, True),
    ("ScaleBias_BF16_gs128", 128, 4, False), ("ZP_BF16_gs128", 128, 4, True), ("ZP_BF16_gs64_8b", 64, 8, True)]
NK = [(4096, 4096), (4096, 14336), (14336, 4096), (14336, 14336)]
MS = [1, 2, 4]
COLD_BYTES = 256 << 20


def bf16_bits(x):
    u = np.asarray(x, dtype=np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r6_gemv_grid.json")
    ap.add_argument("--iters", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--quick", action="store_true", help="first shape and two labels only (plumbing check)")
    args = ap.parse_args()
    from uzu_amd import backend as B

    ctx = B.Context.new(0)
    rng = np.random.default_rng(42)
    kern = B.MatmulKernel.new(ctx, B.BF16, B.BF16, B.BF16)
    rows = []
    t_start = time.time()
    for label, group, bits, zp in (LABELS[:2] if args.quick else LABELS):
        for n, k in (NK[:1] if args.quick else NK):
            block = 512 if bits == 4 else 256
            if n % 8 or k % block:
                continue
            groups = k // group
            w_bytes = n * k * bits // 8
            s_bytes = n * groups * 2
            o_bytes = (n * ((groups + 1) // 2 if bits == 4 else groups)) if zp else n * groups * 2
            per_copy = w_bytes + s_bytes + o_bytes
            copies = max(1, -(-COLD_BYTES // per_copy))
            # one host image, uploaded `copies` times (the values do not matter for the time; distinct device buffers do)
            w_host = rng.integers(0, 256, size=w_bytes, dtype=np.uint8)
            s_host = bf16_bits(rng.uniform(0.01, 0.3, size=n * groups) / np.sqrt(k))
            o_host = rng.integers(0, 256, size=o_bytes, dtype=np.uint8) if zp else bf16_bits(rng.uniform(-0.03, 0.03, size=n * groups))
            pool = []
            for _ in range(copies):
                pool.append((ctx.buffer_from(w_host), ctx.buffer_from(s_host), ctx.buffer_from(o_host)))
            for m in MS:
                a = ctx.buffer_from(bf16_bits(rng.uniform(-1, 1, size=m * k)))
                d = ctx.create_buffer(m * n * 2)
                best = None
                for rep in range(args.reps + 1):  # rep 0 = warm-up (code object load, plan)
                    cb = ctx.create_command_buffer("gemv_grid").start_encoding()
                    for i in range(args.iters):
                        wb, sb, ob = pool[i % copies]
                        kw = dict(zero_points=ob) if zp else dict(biases=ob)
                        kern.encode(cb, a=a, b=wb, d=d, m=m, n=n, k=k, b_kind=B.B_SCALE_ZERO_POINT if zp else B.B_SCALE_BIAS, scales=sb,
                                    mode=B.QMODE_U4 if bits == 4 else B.QMODE_U8, group_size=group, **kw)
                    cb.end_encoding().submit().wait_until_completed()
                    t = cb.gpu_execution_time() / args.iters
                    if rep > 0:
                        best = t if best is None else min(best, t)
                    del cb
                algo = per_copy + m * k * 2 + m * n * 2
                rows.append({"label": label, "m": m, "n": n, "k": k, "bits": bits, "group_size": group, "cold_pool_copies": copies, "us": round(best * 1e6, 2),
                             "gbps": round(algo / best / 1e9, 1), "frac_of_8TBps": round(algo / best / 8e12, 4), "algorithmic_bytes": algo})
                print(f"{label:24s} M={m} N={n:5d} K={k:5d}  {best * 1e6:8.2f} us  {algo / best / 1e9:7.1f} GB/s  ({copies} copies)", flush=True)
                del a, d
            del pool
    out = {"grid": "quant_gemv_bench.rs:57-67 x shape.rs:69-80 (bf16 activations / outputs)", "protocol": f"cold pool >= 256 MiB of weight copies walked round robin; {args.iters} encodes per "
           f"command buffer; GPU execution time of the buffer / {args.iters}; best of {args.reps} buffers after one warm-up", "device": ctx.device_name(), "peak_gbps": 8000.0,
           "seconds": round(time.time() - t_start, 1), "rows": rows}
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(f"# wrote {args.out}: {len(rows)} rows in {out['seconds']} s")
    ctx.close()


if __name__ == "__main__":
    main()
