#!/usr/bin/env python3
"""GPU box: what one round of the speculative loop costs on the benchmarked model -- draft + verify + accept (target and drafter), at 4 / 8 / 16 nodes --
against a plain decode step, and the break-even acceptance (VERDICT r5 "next" 3).  The weights are synthetic, so the ACCEPTANCE rate of the draft model is
meaningless here (reported only to show the loop runs); the break-even is what a trained drafter would have to reach.

  python tools/spec_round_cost.py --out gpurun_out/r6_spec_round_cost.json"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3.5-0.8b")
    ap.add_argument("--context", type=int, default=2043)
    ap.add_argument("--draft-layers", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--weaver-dim", type=int, default=512, help="model_dim of the synthetic Weaver (heads of 128)")
    ap.add_argument("--weaver-only", action="store_true")
    ap.add_argument("--out", default="gpurun_out/r6_spec_round_cost.json")
    args = ap.parse_args()
    from uzu_amd import synthetic as S
    from uzu_amd.backend import Context
    from uzu_amd.engine import HipDrafter, HipModel
    from uzu_amd.speculator import DFlashSpeculator, TreeShape
    from uzu_amd.trie import PRng

    cfg = S.PRESETS[args.model](max_context_length=args.context + 1024)
    bundle = S.build_model(cfg)
    ctx = Context.new(0)
    prompt = S.synthetic_prompt(args.context, cfg.vocab_size)
    # plain decode step of the same model (no taps: fused kernels, graph replay)
    plain = HipModel(ctx, bundle)
    plain.prefill(prompt)
    plain.decode(8)
    _, ms = plain.decode(32)
    plain_ms = ms / 32
    plain.close()
    out = {"model": cfg.name, "context": args.context, "draft_layers": args.draft_layers, "plain_decode_ms_per_token": round(plain_ms, 4), "nodes": {}}
    for nodes in (() if args.weaver_only else (4, 8, 16)):
        hm = HipModel(ctx, bundle)
        db = S.build_drafter(cfg, num_layers=args.draft_layers, block_size=16, context_capacity=args.context + 1024)
        hd = HipDrafter(ctx, hm, db)
        spec = DFlashSpeculator(hd)
        prng = PRng(5)
        tok = None
        t0 = time.perf_counter()
        for s in range(0, prompt.size, 1024):
            chunk = prompt[s:s + 1024]
            tok = hm.prefill(chunk)
            hd.accept(None, np.arange(chunk.size))
        prefill_s = time.perf_counter() - t0
        rec = {"draft_ms": [], "verify_ms": [], "drafter_accept_ms": [], "round_wall_ms": [], "accepted": []}
        for r in range(args.rounds):
            t0 = time.perf_counter()
            trie = spec.propose_tree(hm, tok, TreeShape(tree_budget=nodes, dflash_depth_override=nodes), prng)
            flat = trie.linearize()
            sampled = hm.verify_tree(flat.token_ids(), flat.nodes())
            full = flat.accept(sampled)
            idx = np.array([i for i, _, _ in full], dtype=np.uint32)
            hm.accept(idx)
            hd.accept(None, idx)
            ctx.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            a_ms, d_ms = hd.gpu_ms
            if r >= 2:  # the first rounds capture the tree-pass graph
                rec["draft_ms"].append(d_ms), rec["verify_ms"].append(hm.verify_gpu_ms), rec["drafter_accept_ms"].append(a_ms), rec["round_wall_ms"].append(wall)
                rec["accepted"].append(len(full))
            tok = int(full[-1][2])
        med = lambda v: round(float(np.median(v)), 4)
        device = med(rec["draft_ms"]) + med(rec["verify_ms"]) + med(rec["drafter_accept_ms"])
        out["nodes"][str(nodes)] = {
            "draft_ms": med(rec["draft_ms"]), "verify_ms": med(rec["verify_ms"]), "drafter_accept_ms": med(rec["drafter_accept_ms"]), "device_ms_per_round": round(device, 4),
            "wall_ms_per_round": med(rec["round_wall_ms"]), "prefill_with_taps_and_drafter_accepts_s": round(prefill_s, 3),
            "break_even_tokens_per_round_device": round(device / plain_ms, 2), "break_even_tokens_per_round_wall": round(med(rec["round_wall_ms"]) / plain_ms, 2),
            "tokens_per_round_synthetic_weights": med(rec["accepted"]),
            "note": "a round emits (accepted drafted tokens + 1) tokens; break-even = round time / plain ms per token.  wall includes the host legs: trie build, two "
                    "uploads, the sampled-token download, KV compaction uploads (the target's accept is host-timed only)"}
        hd.close(), hm.close()
    # the Weaver construction of the reference's stream (stream.rs:567-590: budget = the speculation batch, max_tree_depth 16, rounds 16 x 4 nodes x 4 children)
    from uzu_amd.engine import HipWeaver
    hm = HipModel(ctx, bundle)
    db = S.build_drafter(cfg, num_layers=args.draft_layers, block_size=16, context_capacity=args.context + 1024)
    hd = HipDrafter(ctx, hm, db)
    wb = S.build_weaver(cfg, model_dim=args.weaver_dim, num_layers=2, num_heads=args.weaver_dim // 128, hidden_dim=2 * args.weaver_dim, max_depth=15, candidate_pool_size=64)
    hw = HipWeaver(ctx, hd, wb)
    spec = DFlashSpeculator(hd, hw)
    prng = PRng(5)
    for s in range(0, prompt.size, 1024):
        chunk = prompt[s:s + 1024]
        tok = hm.prefill(chunk)
        hd.accept(None, np.arange(chunk.size))
    norm_row = hm.final_hidden_rows()[-1:]
    rec = {"draft_ms": [], "weaver_ms": [], "verify_ms": [], "drafter_accept_ms": [], "round_wall_ms": [], "accepted": [], "tree_nodes": []}
    launches = 0
    for r in range(args.rounds):
        t0 = time.perf_counter()
        trie = spec.propose_tree(hm, tok, TreeShape(tree_budget=16, max_tree_depth=16, construction_method="weaver", rounds=16, expand_per_round=4, expand_width=4), prng, norm_row)
        flat = trie.linearize()
        sampled = hm.verify_tree(flat.token_ids(), flat.nodes(), flat.token_seeds())
        full = flat.accept(sampled)
        idx = np.array([i for i, _, _ in full], dtype=np.uint32)
        norm_row = hm.final_hidden_rows()[int(idx[-1]):int(idx[-1]) + 1]
        hm.accept(idx)
        hd.accept(None, idx)
        ctx.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        a_ms, d_ms = hd.gpu_ms
        w_ms, launches = hw.stats
        if r >= 2:
            rec["draft_ms"].append(d_ms), rec["weaver_ms"].append(w_ms), rec["verify_ms"].append(hm.verify_gpu_ms), rec["drafter_accept_ms"].append(a_ms), rec["round_wall_ms"].append(wall)
            rec["accepted"].append(len(full)), rec["tree_nodes"].append(len(flat))
        tok = int(full[-1][2])
    med = lambda v: round(float(np.median(v)), 4)
    device = med(rec["draft_ms"]) + med(rec["weaver_ms"]) + med(rec["verify_ms"]) + med(rec["drafter_accept_ms"])
    out["weaver_16"] = {
        "weaver": f"model_dim {args.weaver_dim}, 2 layers, {args.weaver_dim // 128} heads x 128, hidden {2 * args.weaver_dim}, max_depth 15, candidate pool 64 (synthetic weights); shape rounds 16 x 4 nodes x 4 children",
        "draft_ms": med(rec["draft_ms"]), "weaver_tree_ms": med(rec["weaver_ms"]), "weaver_launches_in_one_graph": int(launches), "verify_ms": med(rec["verify_ms"]),
        "drafter_accept_ms": med(rec["drafter_accept_ms"]), "device_ms_per_round": round(device, 4), "wall_ms_per_round": med(rec["round_wall_ms"]),
        "tree_nodes": med(rec["tree_nodes"]), "break_even_tokens_per_round_device": round(device / plain_ms, 2),
        "break_even_tokens_per_round_wall": round(med(rec["round_wall_ms"]) / plain_ms, 2), "tokens_per_round_synthetic_weights": med(rec["accepted"])}
    hw.close(), hd.close(), hm.close()
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
