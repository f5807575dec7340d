"""Worker of tests/test_tp.py::test_tp_sharded_forward_n_ranks_one_gpu: rank `rank` of `size` PROCESSES that share GPU 0 and run the
tensor-parallel forward on their shards -- every exchange (row-parallel partial sums of prefill and decode, the arg-max key) goes
through the hipIpc mailboxes (csrc/tp.hip; RCCL refuses two ranks on one device).  The parent holds the oracle's tokens and logits.

  python tests/tp_worker.py <rank> <size> <workdir> <spec.json>

spec: {"preset", "kwargs", "prompt_len", "steps", "teacher": [oracle tokens, first = after the prefill]}
Writes <workdir>/out_<rank>.npz: teacher-forced local logits per step (this rank's vocab shard), teacher-forced arg-max tokens,
the chained (graph-replayed) stream, vocab offset.  Exit code 0 = ran to completion."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, size, workdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    spec = json.load(open(sys.argv[4]))
    from uzu_amd import desc as D
    from uzu_amd import synthetic as S
    from uzu_amd import tp as TP
    from uzu_amd.backend import Context
    from uzu_amd.engine import HipModel

    def exchange(tag: str, mine: bytes):
        """all-gather of one bytes object per rank through files (also the host barrier of the group)"""
        tmp = os.path.join(workdir, f"{tag}_{rank}.tmp")
        with open(tmp, "wb") as f:
            f.write(mine)
        os.replace(tmp, os.path.join(workdir, f"{tag}_{rank}"))
        out = []
        for r in range(size):
            path = os.path.join(workdir, f"{tag}_{r}")
            t0 = time.time()
            while not os.path.exists(path):
                if time.time() - t0 > 300:
                    raise TimeoutError(f"rank {rank}: nothing from rank {r} at '{tag}'")
                time.sleep(0.005)
            out.append(open(path, "rb").read())
        return out

    kw = dict(spec["kwargs"])
    if "layer_kinds" in kw:
        kw["layer_kinds"] = [getattr(D, k) for k in kw["layer_kinds"]]
    cfg = S.PRESETS[spec["preset"]](**kw)
    bundle = S.build_model(cfg)
    shard, off = TP.shard_bundle(bundle, rank, size)
    ctx = Context.new(0)
    group = TP.TpGroup.local(ctx, rank, size)
    group.enable_p2p(lambda mine: exchange("handle", mine))
    model = HipModel(ctx, shard, spec.get("flags", 0), tp_group=group, vocab_offset=off)
    prompt = S.synthetic_prompt(spec["prompt_len"], cfg.vocab_size)
    teacher = spec["teacher"]
    steps = spec["steps"]

    # pass 1: teacher-forced (one decode step per call: the shard's logits are read back every step)
    exchange("ready1", b"x")  # host barrier: the bounded device-side waits (UZU_TP_TIMEOUT_MS) start from a common point
    tf_tokens = [model.prefill(prompt)]
    tf_logits = [model.read_logits()]
    for i in range(steps):
        model.set_next_token(teacher[i])
        exchange(f"step{i}", b"x")
        toks, _ = model.decode(1)
        tf_tokens.append(int(toks[0]))
        tf_logits.append(model.read_logits())
    # pass 2: chained greedy decode, all steps in one call (hipGraph replays with the mailbox exchanges captured inside)
    model.reset()
    exchange("ready2", b"x")
    chained = [model.prefill(prompt)]
    exchange("ready3", b"x")
    toks, _ = model.decode(steps)
    chained += [int(t) for t in toks]
    extra = {}
    if spec.get("stochastic"):
        # pass 3: SamplingMethod::Stochastic on the shards -- every rank gathers the whole logit row (tp::gather_logits) and draws with the same seed
        st = spec["stochastic"]
        model.reset()
        model.set_sampling(seed=st["seed"], **st["settings"])
        exchange("ready4", b"x")
        st_tokens = [model.prefill(prompt)]
        st_logits = [model.read_logits()]
        for i in range(steps):
            exchange(f"st{i}", b"x")
            toks, _ = model.decode(1)
            st_tokens.append(int(toks[0]))
            st_logits.append(model.read_logits())
        exchange("ready5", b"x")
        many, _ = model.decode(3)  # several replays of the captured graph in one call
        extra.update(st_tokens=np.array(st_tokens, np.int64), st_logits=np.stack(st_logits), st_many=np.array(many, np.int64))
        if spec.get("tree"):  # a stochastic tree pass on top: per-node seeds
            from uzu_amd.trie import TrieNode
            root = TrieNode(int(many[-1]))
            a, b = TrieNode(5), TrieNode(9)
            root.add(a), root.add(b)
            a.add(TrieNode(11)), b.add(TrieNode(13))
            flat = root.linearize()
            exchange("ready6", b"x")
            sampled = model.verify_tree(flat.token_ids(), flat.nodes())
            extra.update(st_tree_sampled=np.array(sampled, np.int64), st_tree_logits=model.read_tree_logits(), st_tree_ctx=model.context_length,
                         st_tree_heights=flat.nodes()[:, 2].astype(np.int64))
            model.accept([0])
        model.set_sampling(None)
    if spec.get("tree"):
        # pass 4: greedy verify -> accept on the shards: a chain along the chained stream with one wrong branch
        from uzu_amd.trie import TrieNode
        model.reset()
        exchange("ready7", b"x")
        first = model.prefill(prompt)
        root = TrieNode(first)
        n1, n2, wrong = TrieNode(chained[1]), TrieNode(chained[2]), TrieNode((chained[1] + 1) % cfg.vocab_size)
        root.add(n1), root.add(wrong)
        n1.add(n2)
        flat = root.linearize()
        exchange("ready8", b"x")
        sampled = model.verify_tree(flat.token_ids(), flat.nodes())
        tree_logits = model.read_tree_logits()
        accepted = flat.accept(sampled)
        model.accept([index for index, _, _ in accepted])
        exchange("ready9", b"x")
        after, _ = model.decode(2)
        extra.update(tree_sampled=np.array(sampled, np.int64), tree_logits=tree_logits, tree_accepted=np.array([i for i, _, _ in accepted], np.int64),
                     tree_after=np.array(after, np.int64), tree_tokens=flat.token_ids().astype(np.int64))
    err = group.p2p_error()
    np.savez(os.path.join(workdir, f"out_{rank}.npz"), tf_tokens=np.array(tf_tokens, np.int64), tf_logits=np.stack(tf_logits), chained=np.array(chained, np.int64),
             vocab_offset=off, p2p_error=err, launches=model.decode_launch_count, **extra)
    exchange("done", b"x")  # nobody tears its mailbox down while a peer may still write into it
    model.close()
    group.close()
    ctx.close()
    sys.exit(0 if err == 0 else 2)


if __name__ == "__main__":
    main()
