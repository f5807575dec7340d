#!/usr/bin/env python3
"""GPU box: pick the prompt of the benchmarked configuration's parity fixture.

What round 4 measured first (profiles/r4_stream_search.txt): a random-weight transformer has no greedy stream that is both VARIED and
robustly decided.  In the units the parity tests use -- margin = min over t != best of (l_best - l_t) / (sigma_n (m_best + m_t)), the
synthetic read-out rows carrying log-normal multipliers m_i -- the per-token margin of varied streams has a median of 0.15 (20 % of the
steps are below 0.045) against a measured production-vs-oracle error of up to 0.15-0.19 per competing logit; making the logits peakier
(logit_row_sigma 0.7 ... 1.0) turns every surviving stream into a fixed point (one token repeated: 750 of 750 survivors).  So a fixture
cannot be chosen by a margin threshold; it is chosen by OUTCOME:

  1. production pass (fused kernels, graph replay -- what bench.py times): prefill the candidate prompt
     (uzu_amd.synthetic.synthetic_prompt(variant=v): all candidates share all but the last 16 token ids), `--steps` chained greedy tokens,
     margins from the logits; keep candidates with >= --min-distinct tokens and no margin below --min-gap;
  2. reference-order pass (uzu_hip_set_exact(1): every reduction in the reference's own loop order, logits bit-identical to the CPU
     oracle's -- tests/test_gpu_model.py::test_exact_mode_*): the same prompt, chained; the candidate survives iff BOTH streams are the same
     tokens, i.e. no step's decision lies inside that step's actual numerical error.

The survivor with the largest worst-step slack is then re-run by the CPU oracle itself (tests/golden/make_bench_stream.py), which is what
the committed fixture holds.

  python tools/stream_search.py --prompt 2043 --steps 32 --min-gap 0.03 --min-distinct 10 --budget-s 300
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3.5-0.8b")
    ap.add_argument("--prompt", type=int, default=2043)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--min-gap", type=float, default=0.03)
    ap.add_argument("--min-distinct", type=int, default=10)
    ap.add_argument("--variants", type=int, default=1000000)
    ap.add_argument("--first-variant", type=int, default=1)
    ap.add_argument("--seed", type=int, default=45)
    ap.add_argument("--sigma", type=float, default=0.6)
    ap.add_argument("--budget-s", type=float, default=300.0, help="wall-clock budget for the whole search")
    ap.add_argument("--max-survivors", type=int, default=6)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from helpers import f32
    from uzu_amd import _ffi
    from uzu_amd import synthetic as S
    from uzu_amd.backend import Context
    from uzu_amd.engine import HipModel

    def set_exact(on):
        fn = _ffi.lib().uzu_hip_set_exact
        fn.restype, fn.argtypes = None, [C.c_int32]
        fn(1 if on else 0)

    ctx = Context.new(0)
    t_start = time.time()
    cfg = S.PRESETS[args.model](max_context_length=args.prompt + args.steps + 8, seed=args.seed, logit_row_sigma=args.sigma)
    bundle = S.build_model(cfg)
    row_mult = S.readout_row_multipliers(cfg).astype(np.float64)
    prod = HipModel(ctx, bundle)
    set_exact(True)
    exact = HipModel(ctx, bundle)
    set_exact(False)

    def margin_of(logit_bits):
        w = f32(logit_bits).astype(np.float64)
        best = int(np.argmax(w))
        sigma_n = (w / row_mult).std()
        d = (w[best] - w) / (sigma_n * (row_mult[best] + row_mult))
        d[best] = np.inf
        return float(d.min())

    tried = filtered = verified = 0
    t_prod = t_exact = 0.0
    survivors = []
    prefix_hist = np.zeros(args.steps + 2, dtype=np.int64)
    for v in range(args.first_variant, args.first_variant + args.variants):
        if time.time() - t_start > args.budget_s or len(survivors) >= args.max_survivors:
            break
        prompt = S.synthetic_prompt(args.prompt, cfg.vocab_size, variant=v)
        # ---- 1. production, chained, with margins; dropped at the first margin below the floor
        t0 = time.time()
        set_exact(False)
        prod.reset()
        toks = [prod.prefill(prompt)]
        margins = [margin_of(prod.read_logits())]
        while margins[-1] >= args.min_gap and len(toks) <= args.steps:
            t, _ = prod.decode(1)
            toks.append(int(t[0]))
            margins.append(margin_of(prod.read_logits()))
        t_prod += time.time() - t0
        tried += 1
        if margins[-1] < args.min_gap or len(set(toks)) < args.min_distinct:
            continue
        filtered += 1
        # ---- 2. reference-order mode, chained: must reproduce the production stream token for token
        t0 = time.time()
        set_exact(True)
        exact.reset()
        e_toks = [exact.prefill(prompt)]
        e_margins = [margin_of(exact.read_logits())]
        while e_toks[-1] == toks[len(e_toks) - 1] and len(e_toks) <= args.steps:
            t, _ = exact.decode(1)
            e_toks.append(int(t[0]))
            e_margins.append(margin_of(exact.read_logits()))
        set_exact(False)
        t_exact += time.time() - t0
        verified += 1
        same = 0
        while same < len(e_toks) and e_toks[same] == toks[same]:
            same += 1
        prefix_hist[min(same, args.steps + 1)] += 1
        if same == args.steps + 1:
            rec = {"seed": args.seed, "sigma": args.sigma, "variant": v, "prompt": args.prompt, "distinct": len(set(toks)), "tokens": toks,
                   "min_margin_production": round(min(margins), 4), "min_margin_exact": round(min(e_margins), 4),
                   "margins_exact": [round(g, 4) for g in e_margins], "margins_production": [round(g, 4) for g in margins]}
            survivors.append(rec)
            print(json.dumps(rec), flush=True)
    print(f"# seed {args.seed} sigma {args.sigma}: {tried} variants through the production pass ({t_prod:.0f} s), {filtered} passed the filter (margin >= {args.min_gap}, "
          f">= {args.min_distinct} distinct tokens in {args.steps + 1}), {verified} through the reference-order pass ({t_exact:.0f} s), {len(survivors)} identical streams", flush=True)
    print("# identical-prefix histogram of the verified candidates (index = tokens before the two streams part): " + " ".join(str(int(x)) for x in prefix_hist), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(survivors, f, indent=1)
    prod.close()
    exact.close()
    ctx.close()


if __name__ == "__main__":
    main()
