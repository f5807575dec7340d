#!/usr/bin/env python3
"""Writes tests/golden/sampling.json: tokens drawn by the oracle's UnifiedSampling restatement (oracle/uzu_oracle_kernels.c,
unified_sampling.rs:13-99) for fixed logits, seeds and filter settings.  Regression protection for the oracle and the
known-answer file a future HIP kernel for the stochastic specialisations is checked against (together with the oracle
itself on other inputs).  Run from the repository root:  python tests/golden/make_sampling_golden.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import bf16  # noqa: E402
from test_oracle_sampling import sample  # noqa: E402

VOCAB, BATCH = 1543, 12  # more than one 1024-logit block, not a multiple of anything
CASES = [
    dict(name="stochastic", temperature=None, top_k=None, top_p=None, min_p=None, stochastic=True, mask=False),
    dict(name="temperature", temperature=0.7, top_k=None, top_p=None, min_p=None, stochastic=True, mask=False),
    dict(name="top_k", temperature=1.1, top_k=40, top_p=None, min_p=None, stochastic=True, mask=False),
    dict(name="top_p", temperature=0.9, top_k=None, top_p=0.85, min_p=None, stochastic=True, mask=False),
    dict(name="min_p", temperature=1.0, top_k=None, top_p=None, min_p=0.05, stochastic=True, mask=False),
    dict(name="all_filters_bitmask", temperature=0.8, top_k=64, top_p=0.95, min_p=0.01, stochastic=True, mask=True),
    dict(name="greedy_bitmask", temperature=None, top_k=None, top_p=None, min_p=None, stochastic=False, mask=True),
]


def inputs():
    rng = np.random.default_rng(20260923)
    logits = bf16(rng.normal(0.0, 2.5, (BATCH, VOCAB)))
    seeds = rng.integers(0, 2 ** 63, BATCH, dtype=np.uint64)
    mask = rng.integers(0, 2 ** 32, (BATCH, (VOCAB + 31) // 32), dtype=np.uint64).astype(np.uint32)
    return logits, seeds, mask


def draw(case, logits, seeds, mask):
    return sample(logits, seeds=seeds if case["stochastic"] else None, bitmask=mask if case["mask"] else None, temperature=case["temperature"],
                  top_k=case["top_k"], top_p=case["top_p"], min_p=case["min_p"]).tolist()


if __name__ == "__main__":
    logits, seeds, mask = inputs()
    out = {"vocab": VOCAB, "batch": BATCH, "rng_seed": 20260923, "cases": [dict(c, tokens=draw(c, logits, seeds, mask)) for c in CASES]}
    with open(os.path.join(ROOT, "tests", "golden", "sampling.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote tests/golden/sampling.json")
