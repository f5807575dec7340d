#!/usr/bin/env python3
"""GPU box: decode rate of models whose layer linears are all HybridSpec InputOutput (RHT) linears, through the fused decode step (round 4:
the transforms in the GEMV prologues / as the reference's own kernels around the launches that cannot take them) and through the
one-kernel-per-reference-kernel path (UZU_MODEL_NO_FUSION: what these models took before), next to the same model without the transforms.
Synthetic weights.  Prints one JSON object.

  python tools/rht_decode_cost.py > profiles/r5_rht_decode.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="qwen3.5-0.8b,llama-3-8b")
    args = ap.parse_args()
    from uzu_amd import synthetic as S
    from uzu_amd.backend import Context
    from uzu_amd.engine import MODEL_NO_FUSION, HipModel
    ctx = Context.new(0)
    out = {"note": "tokens/s of chained greedy decode (graph replay), batch 1, after a 500-token prompt; synthetic weights", "models": []}
    for preset, steps in (("qwen3.5-0.8b", 128), ("llama-3-8b", 48)):
        if preset not in args.models.split(","):
            continue
        row = {"model": preset}
        tokens = {}
        for label, rht, flags in (("plain_fused", False, 0), ("rht_fused", True, 0), ("rht_unfused", True, MODEL_NO_FUSION)):
            cfg = S.PRESETS[preset](max_context_length=1024, rht=rht)
            hm = HipModel(ctx, S.build_model(cfg), flags)
            hm.prefill(S.synthetic_prompt(500, cfg.vocab_size))
            first = hm.decode(4)
            ctx.synchronize()
            t0 = time.perf_counter()
            rest = hm.decode(steps)
            ctx.synchronize()
            tokens[label] = [int(t) for t in list(first[0] if isinstance(first, tuple) else first) + list(rest[0] if isinstance(rest, tuple) else rest)]
            dt = time.perf_counter() - t0
            row[label] = {"tokens_per_s": round(steps / dt, 1), "us_per_token": round(dt / steps * 1e6, 1), "launches_per_token": hm.decode_launch_count}
            if label == "rht_fused":  # where the step goes: one eager step with events around every launch (an upper bound on the in-graph times)
                agg = {}
                for name, _, ms in hm.profile_decode_step():
                    a = agg.setdefault(name, [0, 0.0])
                    a[0] += 1
                    a[1] += ms * 1e3
                row[label]["kernel_us_per_step"] = {k: {"calls": v[0], "us": round(v[1], 1)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
            hm.close()
        row["rht_over_plain"] = round(row["rht_fused"]["tokens_per_s"] / row["plain_fused"]["tokens_per_s"], 3)
        row["rht_fused_over_unfused"] = round(row["rht_fused"]["tokens_per_s"] / row["rht_unfused"]["tokens_per_s"], 2)
        out["models"].append(row)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
