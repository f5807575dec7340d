#!/usr/bin/env python3
"""GPU box: decode tokens/s against the number of key splits of attn_dec (UZU_DEC_SPLITS, read when a model is created), same weights, same box.
  python tools/splits_sweep.py llama-3-8b 2048 16,32,64,128"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    model, context, splits = sys.argv[1], int(sys.argv[2]), [int(x) for x in sys.argv[3].split(",")]
    from uzu_amd import synthetic as S
    from uzu_amd.backend import Context
    from uzu_amd.engine import HipModel
    ctx = Context.new(0)
    cfg = S.PRESETS[model](max_context_length=context + 80)
    bundle = S.build_model(cfg)
    prompt = S.synthetic_prompt(context, cfg.vocab_size)
    for s in [0] + splits + [0]:
        if s:
            os.environ["UZU_DEC_SPLITS"] = str(s)
        else:
            os.environ.pop("UZU_DEC_SPLITS", None)
        hm = HipModel(ctx, bundle)
        hm.prefill(prompt)
        hm.decode(4)
        ctx.synchronize()
        t0 = time.perf_counter()
        toks, _ = hm.decode(48)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        agg = {}
        for name, _, ms in hm.profile_decode_step():
            if name.startswith("attn"):
                a = agg.setdefault(name, [0, 0.0])
                a[0] += 1
                a[1] += ms * 1e3
        print(f"{model} ctx {context} splits {s or 'default'}: {48 / dt:.1f} tok/s   " + "  ".join(f"{k} {v[1] / v[0]:.2f} us" for k, v in agg.items()), flush=True)
        hm.close()


if __name__ == "__main__":
    main()
