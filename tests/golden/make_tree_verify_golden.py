"""tests/golden/tree_verify.json from the CPU oracle (run from the repo root: python tests/golden/make_tree_verify_golden.py; seconds).

Pins the oracle's speculative-verification path (oracle/uzu_oracle_tree_verify.c + orc_model_verify_tree / orc_model_accept): for
tiny-qwen (DeltaNet x3 + gated attention) and tiny-llama, one prompt, two rounds of a speculated tree -- the true continuation as a chain
with wrong siblings -- the token sampled at every node, a digest of every node's logits, the accepted path, and the tokens plain decoding
produces afterwards.  Like the other fixtures this freezes the RESTATEMENT (the reference ships no vectors for this path and cannot be
built here); the kernels themselves are pinned by float64 path recurrences in tests/test_oracle_tree_verify.py.
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
from uzu_amd.trie import TrieNode  # noqa: E402

CASES = {"tiny-qwen": 121, "tiny-llama": 191}  # prompt multipliers whose linear streams have no near-tie (tests/test_oracle_tree_verify.py)


def tree_for(last, want, i, depth, vocab):
    root = TrieNode(last)
    node = root
    for d in range(1, depth + 1):
        wrong = TrieNode((want[i + d] + 17 * d) % vocab)
        if d == 2:
            wrong.add(TrieNode((want[i + d] + 5) % vocab))
        node.add(wrong)
        child = TrieNode(want[i + d])
        node.add(child)
        node = child
    return root


def main():
    out = {}
    for preset, mult in CASES.items():
        cfg = S.PRESETS[preset]()
        bundle = S.build_model(cfg)
        base = S.synthetic_prompt(37, cfg.vocab_size).astype(np.int64)
        prompt = ((base * mult + 11 * mult) % cfg.vocab_size).astype(np.uint32)
        m = O.OracleModel(bundle)
        tok = m.prefill(prompt)
        want = [tok]
        for _ in range(16):
            tok = m.forward([tok])
            want.append(tok)
        m.reset()
        got = [m.prefill(prompt)]
        rounds = []
        for _ in range(2):
            i = len(got) - 1
            flat = tree_for(got[-1], want, i, 4, cfg.vocab_size).linearize()
            sampled, logits = m.verify_tree(flat.token_ids(), flat.nodes(), True)
            accepted = flat.accept(sampled)
            m.accept([a for a, _, _ in accepted])
            got.extend(int(s) for _, _, s in accepted)
            rounds.append({"token_ids": [int(t) for t in flat.token_ids()], "nodes": flat.nodes().tolist(), "sampled": [int(t) for t in sampled],
                           "logit_sha256_16": [hashlib.sha256(logits[n].tobytes()).hexdigest()[:16] for n in range(len(flat))],
                           "accepted": [int(a) for a, _, _ in accepted]})
        after = []
        tok = got[-1]
        for _ in range(4):
            tok = m.forward([tok])
            after.append(tok)
        out[preset] = {"prompt_multiplier": mult, "prompt_len": 37, "linear_stream": want, "rounds": rounds, "decoded_after": after}
        m.close()
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tree_verify.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: [r["accepted"] for r in v["rounds"]] for k, v in out.items()}))


if __name__ == "__main__":
    main()
