#!/usr/bin/env python3
"""Where does a decode step's time go?  Runs the bench configuration (Qwen3.5-0.8B int4, context 2048, graph
replay) on the UZU_TIMELINE build of the library, which stamps the 100 MHz wall clock per workgroup at kernel entry /
after the prologue / after the row loop / at exit, and prints per launch:
  gap    first entry of this launch - last exit of the previous launch   (the kernel boundary)
  ramp   last entry - first entry                                        (dispatch of the grid)
  pro    median (after prologue - entry)
  body   median (after row loop - after prologue)
  tail   median (exit - after row loop)
  span   last exit - first entry
GPU box only:  UZU_HIP_LIB=uzu_amd/lib_tl/libuzu_hip.so python tools/timeline.py [--context 2048] > gpurun_out/timeline.txt
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

SLOTS = 8  # UZU_TL_SLOTS (kernels_decode.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--context", type=int, default=2048)
    ap.add_argument("--model", default="qwen3.5-0.8b")
    ap.add_argument("--detail", type=float, default=0.0, help="for launches whose span is at least this many us: exit-time distribution by XCD / grid position")
    ap.add_argument("--dump", type=int, nargs="*", default=[], help="launch indices: the phase stamps of the 20 workgroups that finish last, next to the medians")
    args = ap.parse_args()
    from uzu_amd import _ffi
    from uzu_amd import synthetic as S
    from uzu_amd.backend import Context
    from uzu_amd.engine import HipModel
    lib = _ffi.lib()
    assert hasattr(lib, "uzu_hip_debug_set_timeline"), "not the UZU_TIMELINE build (set UZU_HIP_LIB)"
    cfg = S.PRESETS[args.model](max_context_length=args.context + 64)
    bundle = S.build_model(cfg)
    ctx = Context.new(0)
    hm = HipModel(ctx, bundle)
    hm.prefill(S.synthetic_prompt(args.context - 8, cfg.vocab_size))
    max_launches = 256
    buf = ctx.create_buffer(max_launches * 1024 * SLOTS * 8)
    buf.upload(np.zeros(max_launches * 1024 * SLOTS, dtype=np.uint64))
    lib.uzu_hip_debug_set_timeline.argtypes = [C.c_void_p, C.c_uint32]
    lib.uzu_hip_debug_set_timeline.restype = None
    lib.uzu_hip_debug_set_timeline(C.c_void_p(buf.gpu_ptr()), C.c_uint32(max_launches))
    hm.decode(8)  # first call captures the graph (the stamp slots are baked into it); the last replay's stamps remain
    ctx.synchronize()
    t = buf.download(np.uint64).reshape(max_launches, 1024, SLOTS).astype(np.int64)
    prev_end = None
    print(f"# {cfg.name} ctx {hm.context_length}; times in us (10 ns clock); only gemv_dec / delta_dec / attn_dec launches are stamped")
    print("# gap = first entry - previous launch's last exit; ramp = last entry - first entry; the other columns are medians over the workgroups:")
    print("# x = entry -> activation vector consumed; pro = entry -> prologue done; dots = prologue done -> last batch's dot products done;")
    print("# fin = reduction + epilogue of the last batch; span = first entry -> last exit")
    print("# attn_dec rows (256 workgroups, no dots-only phase): x = entry -> context length arrived; pro = entry -> q / new k normalised, roped, appended;")
    print("# dots = prologue done -> K / V rows arrived and folded in; fin = merge of the key groups through LDS + partial stores")
    print("# right of the bar (round 4: the stamps left of it are THREAD 0's): last = entry -> the workgroup's LAST wave done (median); end = first entry -> the last wave")
    print("# of the last workgroup; gap* = first entry - the previous launch's true end = the kernel boundary proper")
    print(f"{'#':>3} {'wgs':>5} {'gap':>6} {'ramp':>6} {'x':>6} {'pro':>6} {'dots':>6} {'fin':>6} {'exit':>6} {'span':>6} | {'last':>6} {'end':>6} {'gap*':>6}")
    tot = dict(gap=0.0, span=0.0)
    first = None
    prev_true_end = None

    def med(a):
        return float(np.median(a)) * 0.01

    for i in range(max_launches):
        live = t[i][:, 0] > 0
        if not live.any():
            continue
        e = t[i][live]
        t0, t_end = e[:, 0], e[:, 4]
        if first is None:
            first = t0.min()
        gap = (t0.min() - prev_end) * 0.01 if prev_end is not None else float("nan")
        ramp = (t0.max() - t0.min()) * 0.01
        nan = float("nan")
        x = med(e[:, 1] - t0) if (e[:, 1] > 0).all() else nan
        pro = med(e[:, 2] - t0) if (e[:, 2] > 0).all() else nan
        dots = med(e[:, 5] - e[:, 2]) if (e[:, 5] > 0).all() and (e[:, 2] > 0).all() else nan
        fin = med(e[:, 6] - e[:, 5]) if (e[:, 6] > 0).all() and (e[:, 5] > 0).all() else nan
        landed = med(e[:, 3] - t0) if (e[:, 3] > 0).all() else nan  # streaming kernels (k_stream.hip): the loader's whole share has landed in LDS
        ex = med(t_end - t0)
        span = (t_end.max() - t0.min()) * 0.01
        # slot 7: when the LAST wave of the workgroup was done (thread 0's stamps say nothing about the other waves)
        has_last = (e[:, 7] > 0).all()
        true_end = max(int(e[:, 7].max()), int(t_end.max())) if has_last else int(t_end.max())
        gap2 = (t0.min() - prev_true_end) * 0.01 if prev_true_end is not None else nan
        last = med(e[:, 7] - t0) if has_last else nan
        end = (true_end - int(t0.min())) * 0.01
        print(f"{i:3d} {int(live.sum()):5d} {gap:6.2f} {ramp:6.2f} {x:6.2f} {pro:6.2f} {dots:6.2f} {fin:6.2f} {ex:6.2f} {span:6.2f} | {last:6.2f} {end:6.2f} {gap2:6.2f}")
        if args.detail and span >= args.detail:
            # who finishes late?  exit times (relative to the first entry) by XCD (workgroup id % 8: dispatch is round-robin over the
            # XCDs) and by position in the grid (quarters of the workgroup-id range: the first workgroups own one batch more)
            wg = np.nonzero(live)[0]
            rel = (t_end - t0.min()) * 0.01
            q = np.percentile(rel, [0, 10, 50, 90, 100])
            print(f"      exit min/p10/p50/p90/max {q[0]:.2f} {q[1]:.2f} {q[2]:.2f} {q[3]:.2f} {q[4]:.2f}")
            print("      per XCD  p50: " + " ".join(f"{np.median(rel[wg % 8 == xc]):6.2f}" for xc in range(8)) +
                  "   max: " + " ".join(f"{rel[wg % 8 == xc].max():6.2f}" for xc in range(8)))
            quarters = np.array_split(np.argsort(wg), 4)
            print("      per quarter of the grid  p50: " + " ".join(f"{np.median(rel[ix]):6.2f}" for ix in quarters) +
                  "   max: " + " ".join(f"{rel[ix].max():6.2f}" for ix in quarters))
            cu = wg // 8 % 32  # consecutive workgroups of one XCD land on consecutive CUs (first wave of the grid)
            late = rel > q[3]
            print(f"      the slowest 10 %: XCD histogram {np.bincount(wg[late] % 8, minlength=8).tolist()}, first-wave CU-slot histogram "
                  f"{np.bincount(cu[late], minlength=32).tolist()}")
        if i in args.dump:
            wg = np.nonzero(live)[0]
            base = int(t0.min())
            fin_t = np.maximum(e[:, 7], t_end) if has_last else t_end
            order = np.argsort(-fin_t)[:20]
            cols = lambda r: (f"{(r[0] - base) * 0.01:6.2f} " + " ".join(f"{(r[k] - r[0]) * 0.01:6.2f}" if r[k] > 0 else "   nan" for k in (1, 2, 5, 6, 4, 7)))
            print("      workgroups finishing last:   wg xcd |  entry      x    pro   dots    fin   exit   last   (entry: after the first entry; the others: after the workgroup's own entry)")
            for j in order:
                print(f"                                 {wg[j]:4d}  {wg[j] % 8}  | {cols(e[j])}")
            medr = np.median(e, axis=0)
            print(f"                                 median   | {cols(medr.astype(np.int64))}")
        if gap == gap:
            tot["gap"] += gap
        if gap2 == gap2:
            tot["gap2"] = tot.get("gap2", 0.0) + gap2
        tot["span"] += span
        tot["end"] = tot.get("end", 0.0) + end
        prev_end = t_end.max()
        prev_true_end = true_end
    print(f"# stamped launches: sum of spans {tot['span']:.1f} us, sum of gaps {tot['gap']:.1f} us (gaps include the un-stamped kernels: attn_merge, commit, embedding), "
          f"first entry -> last exit {(prev_end - first) * 0.01:.1f} us; to the true ends: {tot.get('end', 0.0):.1f} + {tot.get('gap2', 0.0):.1f} us")


if __name__ == "__main__":
    main()
