#!/usr/bin/env python3
"""GPU box: a parity CENSUS of the benchmarked configuration -- production kernels against reference-order mode over a pre-registered,
UNFILTERED set of prompts (VERDICT r4 "what's weak" 2 / ADVICE r4: the bench line's token parity holds on a prompt picked by outcome;
this is the number for the general case).

Prompts: uzu_amd.synthetic.synthetic_prompt(2043, vocab, variant=v) for v = first .. first + N - 1, in order, none skipped (variant 0 is the
original SURVEY.md section 8d prompt; the committed fixture's variant is inside the range when N > 127).  Per prompt:

  1. reference-order mode (uzu_hip_set_exact(1): every reduction in the reference's own loop order -- logits bit-identical to the CPU oracle's
     where the oracle can run, tests/test_gpu_model.py::test_exact_mode_*): prefill + `steps` chained greedy steps -> reference tokens r_0..r_S,
     the reference logits of every step;
  2. production (what bench.py times: matrix-core prefill, fused decode kernels, graph replay), TEACHER-FORCED with the reference tokens:
     every step is an independent comparison on the same prefix -> production arg-max p_i and logits;
  3. production chained (bench.py's mode exactly): the stream, compared with r as a whole.

Reported (profiles/<round>_parity_census.json; bench.py copies the headline fields into its `parity.census`): the number of teacher-forced
steps whose arg-max differs, for each of them the REFERENCE-ORDER margin between the two tokens involved, the largest margin that ever
flipped, the distribution of the reference's own top-2 margins (how often a step is that close), the production-vs-reference error of the
reference's top-8 logits, and how many chained streams are identical / where they part.  Margins and errors in the units of the parity
tests: logits divided by the synthetic read-out row multiplier m_i, in standard deviations of the normalised row; a margin between tokens
a and b is (l_a - l_b) / (sigma (m_a + m_b)).

  python tools/parity_census.py --variants 200 --steps 32 --budget-s 1500 --out gpurun_out/r5_parity_census.json
(resumable: --resume <file> continues an earlier, budget-cut run)"""
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


def summarise(records, args, seconds):
    steps = sum(len(r["ref_tokens"]) for r in records)
    mism = [m for r in records for m in r["mismatches"]]
    margins = np.array([g for r in records for g in r["ref_margins"]], dtype=np.float64)
    errs = np.array([e for r in records for e in r["top8_err"]], dtype=np.float64)
    max_flip = max((m["reference_margin"] for m in mism), default=0.0)
    identical = sum(1 for r in records if r["chained_common"] == len(r["ref_tokens"]))
    hist = {}
    for r in records:
        if r["chained_common"] < len(r["ref_tokens"]):
            hist[str(r["chained_common"])] = hist.get(str(r["chained_common"]), 0) + 1
    q = lambda a, p: float(np.quantile(a, p)) if a.size else None
    return {
        "config": f"BASELINE configs[{'c2 c3 c4 c5'.split().index(args.config) + 1}]: {args.model} int{args.bits or 4} g128, {args.prompt}-token prompts, reference-order prefill + {args.steps} chained greedy steps; production teacher-forced on the reference's tokens",
        "baseline_config": args.config,
        "selection": f"pre-registered: synthetic_prompt variants {args.first}..{args.first + len(records) - 1}, in order, none filtered or skipped",
        "reference": "reference-order mode (uzu_hip_set_exact(1)): bit-identical to the CPU oracle where the oracle runs (tests/test_gpu_model.py::test_exact_mode_*)",
        "variants": len(records), "steps_per_variant": args.steps + 1, "steps": steps,
        "argmax_mismatches": len(mism), "mismatch_rate": round(len(mism) / max(steps, 1), 5),
        "max_flipped_margin": round(max_flip, 5),
        "mismatches": mism,
        "reference_top2_margin": {"min": q(margins, 0.0), "p01": q(margins, 0.01), "p05": q(margins, 0.05), "p10": q(margins, 0.10), "median": q(margins, 0.5),
                                  "fraction_below_max_flipped": round(float((margins <= max_flip).mean()), 5) if margins.size and mism else 0.0},
        "top8_logit_error": {"median": q(errs, 0.5), "p99": q(errs, 0.99), "max": q(errs, 1.0)},
        "chained": {"identical_streams": identical, "of": len(records), "first_difference_histogram": hist,
                    "note": "production chained greedy (bench.py's mode) against the reference-order chained stream: past the first differing token the streams are on different prefixes"},
        "distinct_reference_tokens": len({t for r in records for t in r["ref_tokens"]}),
        "gpu_seconds": round(seconds, 1),
        "units": "margins / errors: logits divided by the read-out row multiplier, in standard deviations of the normalised row; margin(a, b) = (l_a - l_b) / (sigma (m_a + m_b))",
        "per_variant": [{"variant": r["variant"], "mismatches": len(r["mismatches"]), "chained_common": r["chained_common"], "min_ref_margin": round(min(r["ref_margins"]), 5),
                         "distinct": len(set(r["ref_tokens"]))} for r in records],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3.5-0.8b")
    ap.add_argument("--bits", type=int, default=0)
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5"], default="c2", help="BASELINE configs[1..4]: c3 Llama-3-8B int4, 4096-token prompts (prefill decisions + a few steps); "
                    "c4 Llama-3-8B int8 decode at context 2048; c5 Qwen3-14B-class int4 at context 8192 (round 6: the side lines of bench.py carry their own census)")
    ap.add_argument("--prompt", type=int, default=0)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--variants", type=int, default=200)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--budget-s", type=float, default=1500.0)
    ap.add_argument("--out", default="gpurun_out/parity_census.json")
    ap.add_argument("--resume", default="")
    args = ap.parse_args()
    preset = {"c2": ("qwen3.5-0.8b", 0, 2043), "c3": ("llama-3-8b", 4, 4096), "c4": ("llama-3-8b", 8, 2043), "c5": ("qwen3-14b-class", 4, 8187)}[args.config]
    if args.config != "c2":
        args.model, args.bits = preset[0], preset[1]
    if not args.prompt:
        args.prompt = preset[2]
    from helpers import f32
    from uzu_amd import _ffi
    from uzu_amd import synthetic as S
    from uzu_amd.backend import Context
    from uzu_amd.engine import HipModel

    def set_exact(on):
        fn = _ffi.lib().uzu_hip_set_exact
        fn.restype, fn.argtypes = None, [C.c_int32]
        fn(1 if on else 0)

    t_start = time.time()
    ctx = Context.new(0)
    cfg = S.PRESETS[args.model](max_context_length=args.prompt + args.steps + 8, **({"bits": args.bits} if args.bits else {}))
    bundle = S.build_model(cfg)
    m = S.readout_row_multipliers(cfg).astype(np.float64)
    prod = HipModel(ctx, bundle)
    set_exact(True)
    exact = HipModel(ctx, bundle)
    set_exact(False)
    # The prompts share all but their last 16 tokens: the reference-order pass prefills the common prefix ONCE (passes of 1024 + the rest, the
    # reference's own chunking) and continues a copy of that state per prompt (uzu_hip_state_copy).  Reference-order kernels reduce every row
    # on its own in the reference's loop order, so the split cannot change a bit -- checked below on the first prompt against an unsplit
    # prefill; if it ever did, every prompt would be prefilled whole.
    tail = 16
    set_exact(True)
    snap, work = exact.new_state(), exact.new_state()
    exact.bind(snap)
    exact.prefill(S.synthetic_prompt(args.prompt, cfg.vocab_size)[: args.prompt - tail])
    p0 = S.synthetic_prompt(args.prompt, cfg.vocab_size, variant=args.first)
    work.copy_from(snap)
    exact.bind(work)
    t_split = exact.prefill(p0[args.prompt - tail:])
    l_split = exact.read_logits()
    exact.bind(None)
    exact.reset()
    t_whole = exact.prefill(p0)
    split_ok = bool(t_split == t_whole and np.array_equal(l_split, exact.read_logits()))
    set_exact(False)
    print(f"# reference-order prefill of a copied prefix state + {tail}-token tail is bit-identical to the whole prefill: {split_ok}", flush=True)
    records = []
    if args.resume and os.path.exists(args.resume):
        records = json.load(open(args.resume)).get("_records", [])
    done = {r["variant"] for r in records}

    def analyse(bits):
        w = f32(bits).astype(np.float64)
        sigma = (w / m).std()
        best = int(np.argmax(w))
        d = (w[best] - w) / (sigma * (m[best] + m))
        d[best] = np.inf
        top8 = np.argpartition(w, -8)[-8:]
        return w, sigma, best, float(d.min()), top8

    for v in range(args.first, args.first + args.variants):
        if v in done:
            continue
        if time.time() - t_start > args.budget_s:
            break
        prompt = S.synthetic_prompt(args.prompt, cfg.vocab_size, variant=v)
        # 1. reference-order, chained
        set_exact(True)
        if split_ok:
            work.copy_from(snap)
            exact.bind(work)
            ref_tokens, ref_rows = [exact.prefill(prompt[args.prompt - tail:])], [exact.read_logits()]
        else:
            exact.bind(None)
            exact.reset()
            ref_tokens, ref_rows = [exact.prefill(prompt)], [exact.read_logits()]
        for _ in range(args.steps):
            t, _ms = exact.decode(1)
            ref_tokens.append(int(t[0]))
            ref_rows.append(exact.read_logits())
        set_exact(False)
        # 2. production, teacher-forced on the reference's tokens
        prod.reset()
        rec = {"variant": v, "ref_tokens": ref_tokens, "ref_margins": [], "top8_err": [], "mismatches": []}
        for i in range(args.steps + 1):
            if i == 0:
                p_tok = prod.prefill(prompt)
            else:
                prod.set_next_token(ref_tokens[i - 1])
                t, _ms = prod.decode(1)
                p_tok = int(t[0])
            w_r, sigma, best, margin, top8 = analyse(ref_rows[i])
            assert best == ref_tokens[i]
            w_p = f32(prod.read_logits()).astype(np.float64)
            rec["ref_margins"].append(margin)
            rec["top8_err"].append(float((np.abs(w_p[top8] - w_r[top8]) / m[top8]).max() / sigma))
            if p_tok != ref_tokens[i]:
                flipped = float((w_r[best] - w_r[p_tok]) / (sigma * (m[best] + m[p_tok])))
                rec["mismatches"].append({"variant": v, "step": i, "reference_token": ref_tokens[i], "production_token": p_tok, "reference_margin": round(flipped, 5),
                                          "production_margin": round(float((w_p[p_tok] - w_p[best]) / (sigma * (m[best] + m[p_tok]))), 5)})
        # 3. production chained (bench.py's mode)
        prod.reset()
        chained = [prod.prefill(prompt)]
        toks, _ms = prod.decode(args.steps)
        chained += [int(t) for t in toks]
        common = 0
        while common < len(ref_tokens) and chained[common] == ref_tokens[common]:
            common += 1
        rec["chained_common"] = common
        records.append(rec)
        out = summarise(records, args, time.time() - t_start)
        out["reference_prefix_state_reused"] = split_ok
        out["_records"] = records
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out + ".tmp", "w") as f:
            json.dump(out, f)
        os.replace(args.out + ".tmp", args.out)
        print(f"variant {v}: {len(rec['mismatches'])} mismatches, chained identical for {common}/{len(ref_tokens)}, min reference margin {min(rec['ref_margins']):.4f}, "
              f"worst top-8 error {max(rec['top8_err']):.3f}  [{time.time() - t_start:.0f} s]", flush=True)
    out = summarise(records, args, time.time() - t_start)
    out["reference_prefix_state_reused"] = split_ok
    print(json.dumps({k: v for k, v in out.items() if k not in ("per_variant", "mismatches")}, indent=1))
    prod.close()
    snap.close(), work.close()
    exact.close()
    ctx.close()


if __name__ == "__main__":
    main()
