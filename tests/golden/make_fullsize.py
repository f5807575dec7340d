"""Full-size fixtures from the CPU oracle (run from the repo root; minutes to tens of minutes of CPU time each):

  python tests/golden/make_fullsize.py qwen_bench     # BASELINE configs[1] in bench.py's mode: Qwen3.5-0.8B int4, prefill 2040 + 24 greedy steps
  python tests/golden/make_fullsize.py llama_int4     # BASELINE configs[2] weights: Llama-3-8B int4 ScaleBias, all 32 layers, prefill 48 + 8 steps
  python tests/golden/make_fullsize.py llama_int8     # BASELINE configs[3] weights: Llama-3-8B int8 ScaleZeroPoint, all 32 layers, prefill 48 + 8 steps

Each fixture (tests/golden/fullsize_<name>.json) holds, per sampled position: the greedy token, the top-8 (token id,
logit bf16 bits) and the top-2 gap in units of the logit row's standard deviation; plus sha256 digests of every layer's
output for the last prompt row (regression protection of the oracle itself -- the GPU path is tolerance-class per layer).
The weights are NOT stored: both sides regenerate them from (preset, seed) with uzu_amd/synthetic.py.
The reference ships no model-level golden outputs and cannot be built here, so these pin the
ORACLE, which tests/test_oracle_*.py pin kernel by kernel against float64 re-derivations and the reference's one KAT.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from uzu_amd import synthetic as S  # noqa: E402

CASES = {
    # name: (preset, config overrides, prompt tokens, decode steps)
    "qwen_bench": ("qwen3.5-0.8b", dict(max_context_length=2048 + 64), 2040, 24),
    "llama_int4": ("llama-3-8b", dict(max_context_length=128, seed=7), 48, 8),
    "llama_int8": ("llama-3-8b", dict(max_context_length=128, seed=7, bits=8, method=1), 48, 8),
}


def f32(bits):
    return (np.asarray(bits, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


def row_record(tok, logits):
    w = f32(logits).astype(np.float64)
    order = np.lexsort((np.arange(w.size), -w))[:8]  # descending value, ties -> lowest index (unified_sampling.rs:90-95)
    assert int(order[0]) == tok
    return {"token": int(tok), "top8": [[int(i), int(logits[i])] for i in order], "gap_sigma": round(float((w[order[0]] - w[order[1]]) / w.std()), 4)}


def main():
    name = sys.argv[1]
    preset, kw, prompt_len, steps = CASES[name]
    cfg = S.PRESETS[preset](**kw)
    t0 = time.time()
    bundle = S.build_model(cfg)
    m = O.OracleModel(bundle)
    prompt = S.synthetic_prompt(prompt_len, cfg.vocab_size)
    tok, logits = m.prefill(prompt, True)
    rows = [row_record(tok, logits)]
    layer_sha = []
    for layer in range(len(bundle.layers)):
        out = m.layer_output(layer)
        layer_sha.append(hashlib.sha256(out[-1].tobytes()).hexdigest()[:16])
    print(f"prefill {prompt_len} tokens: {time.time() - t0:.0f} s, first token {tok}", flush=True)
    for _ in range(steps):
        tok, logits = m.forward([rows[-1]["token"]], True)
        rows.append(row_record(tok, logits))
    tokens = [r["token"] for r in rows]
    out = {"preset": preset, "config": {k: v for k, v in kw.items()}, "seed": cfg.seed, "logit_row_sigma": cfg.logit_row_sigma, "bits": cfg.bits,
           "prompt_len": prompt_len, "steps": steps, "tokens": tokens, "distinct_tokens": len(set(tokens)),
           "min_gap_sigma": min(r["gap_sigma"] for r in rows), "rows": rows, "last_prompt_row_layer_sha256_16": layer_sha,
           "generator": "tests/golden/make_fullsize.py (CPU oracle, OpenMP over output rows: bit-identical to one thread)"}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"fullsize_{name}.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(f"{name}: {len(set(tokens))} distinct tokens of {len(tokens)}, min top-2 gap {out['min_gap_sigma']} sigma, {time.time() - t0:.0f} s -> {path}")
    print(tokens)


if __name__ == "__main__":
    main()
