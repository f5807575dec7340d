#!/usr/bin/env python3
"""GPU box: what one speculative verify pass costs on the benchmarked configuration (Qwen3.5-0.8B int4, context 2048) for trees of
M = 1, 2, 4, 8, 16 nodes, next to M plain decode steps.  The trees are chains with one wrong sibling per level (the
shape FlatTrie::accept walks); the weights are synthetic, so NO acceptance rate is claimed -- the figure is the cost side of the trade:
verify(M) + accept against M x decode.  Prints one JSON object.

  python tools/verify_cost.py > profiles/r4_verify_cost.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from uzu_amd import synthetic as S
    from uzu_amd.backend import Context
    from uzu_amd.engine import HipModel
    from uzu_amd.trie import TrieNode
    ctx = Context.new(0)
    cfg = S.PRESETS["qwen3.5-0.8b"](max_context_length=2048 + 512)
    hm = HipModel(ctx, S.build_model(cfg))
    tok = hm.prefill(S.synthetic_prompt(2043, cfg.vocab_size))
    toks, _ = hm.decode(5)
    tok = int(toks[-1])
    reps = 12
    # plain decode, graph replay: the baseline a verify pass competes with
    ctx.synchronize()
    t0 = time.perf_counter()
    toks, gpu_ms = hm.decode(64)
    ctx.synchronize()
    decode_us = (time.perf_counter() - t0) / 64 * 1e6
    tok = int(toks[-1])
    out = {"model": "qwen3.5-0.8b int4 g128 (synthetic weights)", "context": hm.context_length, "decode_us_per_token": round(decode_us, 1), "verify": []}
    for m in ([int(v) for v in os.environ["VERIFY_NODES"].split(",")] if os.environ.get("VERIFY_NODES") else (1, 2, 4, 8, 16)):  # VERIFY_NODES=16: one size (profiling runs)
        t_verify = t_accept = t_gpu = 0.0
        launches = 0
        for rep in range(reps + 1):  # rep 0 builds the pass's hipGraph: not timed
            # a chain of (m + 1) // 2 levels with a wrong sibling at each level but the root: m nodes in all
            root = TrieNode(tok)
            node, count, level = root, 1, 1
            while count < m:
                child = TrieNode((tok + 7 * level) % cfg.vocab_size)
                node.add(child)
                count += 1
                if count < m:
                    node.add(TrieNode((tok + 7 * level + 3) % cfg.vocab_size))
                    count += 1
                node, level = child, level + 1
            flat = root.linearize()
            ids, nodes = flat.token_ids(), flat.nodes()
            ctx.synchronize()
            t0 = time.perf_counter()
            sampled = hm.verify_tree(ids, nodes)
            t1 = time.perf_counter()
            launches = hm.decode_launch_count
            accepted = [i for i, _, _ in flat.accept(sampled)]
            t2 = time.perf_counter()
            hm.accept(accepted)
            t3 = time.perf_counter()
            if rep:
                t_verify += t1 - t0
                t_accept += t3 - t2
                t_gpu += hm.verify_gpu_ms * 1e3
            tok = int(sampled[accepted[-1]])
        out["verify"].append({"nodes": m, "verify_us": round(t_verify / reps * 1e6, 1), "verify_gpu_us": round(t_gpu / reps, 1), "accept_us": round(t_accept / reps * 1e6, 1),
                              "launches": int(launches),
                              "m_decode_steps_us": round(decode_us * m, 1),
                              "break_even_accepted_tokens": round((t_verify + t_accept) / reps * 1e6 / decode_us, 2)})
    out["note"] = ("verify_us / accept_us are host wall times around the synchronous C-ABI calls (uploads, one hipGraph replay, one download + sync); verify_gpu_us is the "
                   "device time of the replayed pass (HIP events); "
                   "break_even_accepted_tokens = how many tokens a round must yield on average to beat plain decoding")
    print(json.dumps(out, indent=1))
    hm.close()
    ctx.close()


if __name__ == "__main__":
    main()
