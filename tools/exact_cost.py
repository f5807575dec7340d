#!/usr/bin/env python3
"""GPU box: what a reference-order (uzu_hip_set_exact) pass costs on a model -- prefill seconds, seconds per decode step, and the
per-kernel breakdown of one decode step (uzu_hip_model_profile_decode_step).  Planning aid for tools/parity_census.py and the
configuration-scale parity tests (reference-order mode is the proxy oracle there).

  python tools/exact_cost.py [--model qwen3.5-0.8b] [--prompt 2043] [--steps 3]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3.5-0.8b")
    ap.add_argument("--prompt", type=int, default=2043)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--layers", type=int, default=0, help="llama / 14B-class presets: keep only this many layers (0 = all)")
    args = ap.parse_args()
    from uzu_amd import _ffi
    from uzu_amd import desc as D
    from uzu_amd import synthetic as S
    from uzu_amd.backend import Context
    from uzu_amd.engine import HipModel

    def set_exact(on):
        fn = _ffi.lib().uzu_hip_set_exact
        fn.restype, fn.argtypes = None, [C.c_int32]
        fn(1 if on else 0)

    kw = dict(max_context_length=args.prompt + args.steps + 8)
    if args.layers:
        kw["layer_kinds"] = [D.MIXER_ATTENTION] * args.layers
    cfg = S.PRESETS[args.model](**kw)
    t0 = time.time()
    bundle = S.build_model(cfg)
    print(f"# built {cfg.name} in {time.time() - t0:.1f} s", flush=True)
    ctx = Context.new(0)
    prompt = S.synthetic_prompt(args.prompt, cfg.vocab_size)
    out = {"model": cfg.name, "prompt": args.prompt}
    for mode in ("production", "exact"):
        set_exact(mode == "exact")
        hm = HipModel(ctx, bundle)
        t0 = time.time()
        hm.prefill(prompt)
        ctx.synchronize()
        out[f"{mode}_prefill_s"] = round(time.time() - t0, 3)
        t0 = time.time()
        hm.decode(args.steps)
        ctx.synchronize()
        out[f"{mode}_decode_s_per_step"] = round((time.time() - t0) / args.steps, 4)
        agg = {}
        for name, _, ms in hm.profile_decode_step():
            a = agg.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += ms
        out[f"{mode}_decode_kernels_ms"] = {k: [v[0], round(v[1], 3)] for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]}
        hm.close()
        set_exact(False)
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
