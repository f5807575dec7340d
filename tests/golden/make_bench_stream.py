"""The benchmarked configuration's parity fixture, from the CPU oracle (run from the repo root; ~10 CPU-minutes):

  python tests/golden/make_bench_stream.py <variant> [tokens]     # variant: the prompt tools/stream_search.py picked on the GPU box

bench.py's default workload is Qwen3.5-0.8B int4 (seed 45) prefilled with uzu_amd.synthetic.synthetic_prompt(2043, vocab, variant) and
decoded greedily; this script runs exactly that on the oracle -- prefill 2043 tokens (two chunks), then `tokens` - 1 chained greedy steps
-- and writes tests/golden/bench_qwen_stream.json: the token stream, per step the top-8 (token id, logit bits) and the decidability
margin in the parity tests' row-normalised units (min over competitors t of (l_best - l_t) / (sigma_n (m_best + m_t))).

Why a VARIANT of the prompt: a random-weight transformer has no greedy stream that is both varied and free of near-ties
(profiles/r4_stream_search.txt), so the prompt is chosen by outcome -- tools/stream_search.py keeps prompts whose production stream
(fused kernels, graph replay) and reference-order stream (bit-identical to this oracle) are the same tokens; this script is the oracle's
own word on the chosen one, and tests/test_gpu_model.py + bench.py compare the production stream with it token for token.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from uzu_amd import synthetic as S  # noqa: E402

PROMPT_TOKENS = 2043  # = 2048 - 5: the driver's `--warmup 5` reaches context 2048 where the timed steps start


def f32(bits):
    return (np.asarray(bits, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


def row_record(tok, logits, row_mult):
    w = f32(logits).astype(np.float64)
    order = np.lexsort((np.arange(w.size), -w))[:8]  # descending value, ties -> lowest index (unified_sampling.rs:90-95)
    assert int(order[0]) == tok
    sigma_n = (w / row_mult).std()
    d = (w[tok] - w) / (sigma_n * (row_mult[tok] + row_mult))
    d[tok] = np.inf
    return {"token": int(tok), "top8": [[int(i), int(logits[i])] for i in order], "margin": round(float(d.min()), 4)}


def main():
    variant = int(sys.argv[1])
    tokens = int(sys.argv[2]) if len(sys.argv) > 2 else 33
    cfg = S.PRESETS["qwen3.5-0.8b"](max_context_length=PROMPT_TOKENS + tokens + 8)
    t0 = time.time()
    bundle = S.build_model(cfg)
    row_mult = S.readout_row_multipliers(cfg).astype(np.float64)
    m = O.OracleModel(bundle)
    prompt = S.synthetic_prompt(PROMPT_TOKENS, cfg.vocab_size, variant=variant)
    tok, logits = m.prefill(prompt, True)
    rows = [row_record(tok, logits, row_mult)]
    print(f"prefill {PROMPT_TOKENS} tokens: {time.time() - t0:.0f} s, first token {tok}", flush=True)
    for _ in range(tokens - 1):
        tok, logits = m.forward([rows[-1]["token"]], True)
        rows.append(row_record(tok, logits, row_mult))
    stream = [r["token"] for r in rows]
    out = {"preset": "qwen3.5-0.8b", "seed": cfg.seed, "logit_row_sigma": cfg.logit_row_sigma, "bits": cfg.bits, "prompt_tokens": PROMPT_TOKENS,
           "prompt_variant": variant, "tokens": stream, "distinct_tokens": len(set(stream)), "min_margin": min(r["margin"] for r in rows), "rows": rows,
           "generator": "tests/golden/make_bench_stream.py (CPU oracle, OpenMP over output rows: bit-identical to one thread); prompt variant from tools/stream_search.py"}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_qwen_stream.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(f"variant {variant}: {len(set(stream))} distinct tokens of {len(stream)}, min margin {out['min_margin']}, {time.time() - t0:.0f} s -> {path}")
    print(stream)


if __name__ == "__main__":
    main()
