#!/usr/bin/env python3
"""GPU box: the headline model's prefill alone (BASELINE configs[1]: Qwen3.5-0.8B int4, the 2043-token prompt), N passes -- the command rocprofv3 is wrapped around
for the per-kernel prefill table (profiles/r6*_prefill_kernel_stats.csv):  rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python tools/prefill_profile.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    model = sys.argv[2] if len(sys.argv) > 2 else "qwen3.5-0.8b"
    tokens = int(sys.argv[3]) if len(sys.argv) > 3 else 2043
    from uzu_amd import synthetic as S
    from uzu_amd.backend import Context
    from uzu_amd.engine import HipModel
    cfg = S.PRESETS[model](max_context_length=tokens + 16)
    bundle = S.build_model(cfg)
    ctx = Context.new(0)
    hm = HipModel(ctx, bundle)
    prompt = S.synthetic_prompt(tokens, cfg.vocab_size)
    hm.prefill(prompt)
    ctx.synchronize()
    times = []
    for _ in range(passes):
        hm.reset()
        t0 = time.perf_counter()
        hm.prefill(prompt)
        ctx.synchronize()
        times.append(time.perf_counter() - t0)
    print(f"{model}: {tokens} prompt tokens, passes {[round(t * 1e3, 3) for t in times]} ms -> best {tokens / min(times):.0f} tok/s, {passes + 1} passes in the trace")
    hm.close()
    ctx.close()


if __name__ == "__main__":
    main()
