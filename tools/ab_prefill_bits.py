#!/usr/bin/env python3
"""GPU box: one prefill (+ 8 greedy tokens) of a DeltaNet model, printed as hashes -- run it under different prefill switches of ONE library build
and compare the lines (tools/ab_prefill_switches.sh):

  UZU_HIP_TUNE=conv_apply4=0     conv: one channel per thread          (the 4-channel kernel is meant to be BIT-IDENTICAL: equal hashes)
  UZU_HIP_TUNE=norm_partials=0   split-K reduction and normalisation as two launches   (one launch is meant to be BIT-IDENTICAL: equal hashes, fewer launches)
  UZU_HIP_TUNE=dn_split=0        DeltaNet scan as one chain of chunks  (the two-segment scan sums in another order: logits close, not equal)

  python tools/ab_prefill_bits.py [--model tiny|qwen3.5-0.8b] [--prompt 2043]
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="tiny")
    ap.add_argument("--prompt", type=int, default=700)
    ap.add_argument("--dump", default="", help="write the logits (bf16 bits, .npy) here")
    args = ap.parse_args()
    from uzu_amd import synthetic as S
    from uzu_amd.backend import Context
    from uzu_amd.engine import HipModel
    if args.model == "tiny":  # model_dim 1024: the rows kernels / split-K + normalisation launch need rows of 1024 elements
        cfg = S.tiny_qwen(model_dim=1024, hidden_dim=1536, max_context_length=args.prompt + 64, seed=34)
    else:
        cfg = S.PRESETS[args.model](max_context_length=args.prompt + 64)
    ctx = Context.new(0)
    hm = HipModel(ctx, S.build_model(cfg))
    prompt = S.synthetic_prompt(args.prompt, cfg.vocab_size)
    hm.prefill(prompt)  # warm-up pass (workspace allocation, code objects)
    hm.reset()
    ctx.synchronize()
    t0 = time.perf_counter()
    first = hm.prefill(prompt)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    launches = int(hm.decode_launch_count)
    logits = np.asarray(hm.read_logits())
    toks, _ = hm.decode(8)
    if args.dump:
        np.save(args.dump, logits)
    f = (logits.astype(np.uint32) << 16).view(np.float32)
    print(json.dumps({"model": cfg.name, "prompt": args.prompt, "switches": os.environ.get("UZU_HIP_TUNE", ""),
                      "prefill_launches": launches, "prefill_ms": round(dt * 1e3, 3), "prompt_tokens_per_s": round(args.prompt / dt, 1), "first_token": int(first),
                      "tokens": [int(t) for t in toks], "logits_sha256": hashlib.sha256(logits.tobytes()).hexdigest()[:16],
                      "logits_rms": float(np.sqrt(np.mean(f.astype(np.float64) ** 2)))}))
    hm.close()
    ctx.close()


if __name__ == "__main__":
    main()
