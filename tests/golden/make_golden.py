"""Regenerates tests/golden/*.json from the CPU oracle (run from the repo root: python tests/golden/make_golden.py).

The reference ships no golden vectors for this path and cannot be built here (Rust), so these fixtures pin the
ORACLE's behaviour (regression protection for the restatement itself), not the reference's: see oracle/uzu_oracle.h.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from uzu_amd import synthetic as S  # noqa: E402


def main():
    out = {}
    for preset in ("tiny-qwen", "tiny-llama"):
        cfg = S.PRESETS[preset]()
        bundle = S.build_model(cfg)
        m = O.OracleModel(bundle)
        prompt = S.synthetic_prompt(40, cfg.vocab_size)
        tok, logits = m.prefill(prompt, True)
        tokens, digests = [tok], [hashlib.sha256(logits.tobytes()).hexdigest()[:16]]
        for _ in range(24):
            tok, logits = m.forward([tokens[-1]], True)
            tokens.append(tok)
            digests.append(hashlib.sha256(logits.tobytes()).hexdigest()[:16])
        out[preset] = {"prompt_len": 40, "tokens": tokens, "logit_sha256_16": digests,
                       "last_layer_sha256_16": hashlib.sha256(m.layer_output(len(bundle.layers) - 1).tobytes()).hexdigest()[:16]}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny_models.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v["tokens"] for k, v in out.items()}))


if __name__ == "__main__":
    main()
