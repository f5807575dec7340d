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
    err = group.p2p_error()
    np.savez(os.path.join(workdir, f"out_{rank}.npz"), tf_tokens=np.array(tf_tokens, np.int64), tf_logits=np.stack(tf_logits), chained=np.array(chained, np.int64),
             vocab_offset=off, p2p_error=err, launches=model.decode_launch_count)
    exchange("done", b"x")  # nobody tears its mailbox down while a peer may still write into it
    model.close()
    group.close()
    ctx.close()
    sys.exit(0 if err == 0 else 2)


if __name__ == "__main__":
    main()
