#!/usr/bin/env python3
"""Parity report: HIP engine vs CPU oracle -- token agreement, logit error statistics, per-layer error growth.
usage: python tools/parity_report.py [preset ...] [--prompt N] [--steps K]   (runs on the GPU box)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import f32, ulp_diff_bf16  # noqa: E402
from oracle import oracle as O  # noqa: E402
from uzu_amd import _ffi  # noqa: E402
from uzu_amd import synthetic as S  # noqa: E402
from uzu_amd.backend import Context  # noqa: E402
from uzu_amd.engine import MODEL_DEBUG_TAPS, HipModel  # noqa: E402


def logit_stats(o_bits, h_bits):
    o, h = f32(o_bits).astype(np.float64), f32(h_bits).astype(np.float64)
    err = np.abs(o - h)
    s = np.sort(o)[::-1]
    ulp_top = 2.0 ** (np.floor(np.log2(abs(s[0]))) - 7)
    return dict(max_abs=err.max(), std=o.std(), max_rel_std=err.max() / o.std(), top=s[0], gap_ulps=(s[0] - s[1]) / ulp_top,
                err_top=err[np.argmax(o)], argmax_equal=int(np.argmax(o) == np.argmax(h)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("presets", nargs="*", default=["tiny-qwen", "tiny-llama"])
    ap.add_argument("--prompt", type=int, default=40)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--taps", action="store_true")
    args = ap.parse_args()
    ctx = Context.new(0)
    print("device:", ctx.device_name())
    for preset in args.presets:
        cfg = S.PRESETS[preset](max_context_length=max(args.prompt + args.steps + 8, 64))
        bundle = S.build_model(cfg)
        prompt = S.synthetic_prompt(args.prompt, cfg.vocab_size)
        for exact in (0, 1):
            if exact and preset not in ("tiny-qwen", "tiny-llama"):
                continue
            _ffi.lib().uzu_hip_set_exact_matmul(exact)
            om = O.OracleModel(bundle)
            hm = HipModel(ctx, bundle, MODEL_DEBUG_TAPS if args.taps else 0)
            t0 = time.time()
            o_tok, o_logits = om.prefill(prompt, True)
            t_or = time.time() - t0
            h_tok = hm.prefill(prompt)
            st = logit_stats(o_logits, hm.read_logits())
            print(f"[{preset} exact_matmul={exact}] prefill {args.prompt}: oracle tok {o_tok} hip tok {h_tok} | max|dlogit| {st['max_abs']:.4f} "
                  f"= {st['max_rel_std']:.4f} std | top {st['top']:.2f} gap {st['gap_ulps']:.1f} ulps | oracle {t_or:.1f}s")
            if args.taps:
                for l in range(len(bundle.layers)):
                    u = ulp_diff_bf16(om.layer_output(l), hm.read_layer_output(l))
                    print(f"   layer {l:2d}: bit-equal {np.mean(u == 0):.4f}  <=1ulp {np.mean(u <= 1):.4f}  <=2ulp {np.mean(u <= 2):.4f}  max {u.max():.1f}")
            o_toks, h_toks, worst, mism = [o_tok], [h_tok], 0.0, 0
            for i in range(args.steps):
                hm.set_next_token(o_toks[-1])  # teacher forced: every step is compared on identical inputs
                o_tok, o_logits = om.forward([o_toks[-1]], True)
                toks, _ = hm.decode(1)
                st = logit_stats(o_logits, hm.read_logits())
                worst = max(worst, st["max_rel_std"])
                if int(toks[0]) != o_tok:
                    mism += 1
                    print(f"   step {i}: token mismatch oracle {o_tok} hip {int(toks[0])}; oracle top-2 gap {st['gap_ulps']:.2f} bf16 ulps, err at top {st['err_top']:.4f}")
                o_toks.append(o_tok)
                h_toks.append(int(toks[0]))
            print(f"   teacher-forced decode x{args.steps}: token mismatches {mism}, worst max|dlogit| {worst:.4f} std")
            hm.close()
            om.close()
        _ffi.lib().uzu_hip_set_exact_matmul(0)


if __name__ == "__main__":
    main()
