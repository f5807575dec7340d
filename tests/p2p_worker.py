"""Worker of tests/test_tp.py::test_p2p_all_reduce_two_processes_one_gpu: one of two processes that share GPU 0 and exchange
through each other's mailboxes (hipIpc handles travel over files in `workdir`).  Exit code 0 = every exchange gave the expected
bit pattern."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, size, workdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    from uzu_amd import tp as TP
    from uzu_amd.backend import Context
    ctx = Context.new(0)
    group = TP.TpGroup.local(ctx, rank, size)

    def all_gather(mine: bytes):
        with open(os.path.join(workdir, f"handle_{rank}.tmp"), "wb") as f:
            f.write(mine)
        os.replace(os.path.join(workdir, f"handle_{rank}.tmp"), os.path.join(workdir, f"handle_{rank}"))
        out = []
        for r in range(size):
            path = os.path.join(workdir, f"handle_{r}")
            t0 = time.time()
            while not os.path.exists(path):
                if time.time() - t0 > 60:
                    raise TimeoutError(f"rank {rank}: no handle from rank {r}")
                time.sleep(0.01)
            out.append(open(path, "rb").read())
        return out

    group.enable_p2p(all_gather)
    if len(sys.argv) > 4 and sys.argv[4] == "skew":
        # A rank that is late by more than the bounded wait (UZU_TP_TIMEOUT_MS, set short by the test): the punctual rank gives up, and the
        # failure must reach EVERY rank -- the late one finds each flag it waits for (the punctual rank had pushed its row before it gave
        # up) and would otherwise carry on with a sum it cannot trust.  Both ranks must end with a non-zero error word.
        vec = np.full(1024, float(rank + 1), np.float32)
        for it in range(4):
            if it == 2 and rank == 1:
                time.sleep(1.5)
            buf = ctx.buffer_from(vec)
            group.all_reduce_sum_f32(buf, 1024)
            ctx.synchronize()
            got = buf.download(np.float32, 1024)
            if it < 2 and not np.all(got == np.float32(sum(range(1, size + 1)))):
                print(f"rank {rank}: exchange {it} wrong before the skew")
                sys.exit(1)
        err = group.p2p_error()
        print(f"rank {rank}: error word {err}, last sums {'NaN' if np.isnan(got).all() else got[0]}")
        group.close()
        ctx.close()
        sys.exit(0 if err != 0 and np.isnan(got).all() else 1)
    rng = [np.random.default_rng(100 + r) for r in range(size)]
    ok = True
    for it, count in enumerate([1024, 4096, 5120, 8192, 7, 1024, 1024, 2048]):
        parts = [g.normal(0, 1, count).astype(np.float32) for g in rng]  # every rank draws everybody's vector: same streams
        want = parts[0].copy()
        for r in range(1, size):
            want = want + parts[r]  # rank order, f32: what the kernel computes
        buf = ctx.buffer_from(parts[rank])
        group.all_reduce_sum_f32(buf, count)
        ctx.synchronize()
        got = buf.download(np.float32, count)
        if not np.array_equal(got, want):
            print(f"rank {rank}: exchange {it} (count {count}) differs: max |d| {np.abs(got - want).max()}")
            ok = False
    # the arg-max key exchange: max over ranks of one u64 per rank
    keys = [np.array([(0x8000000000000000 >> r) | (12345 + r)], dtype=np.uint64) for r in range(size)]
    kb = ctx.buffer_from(keys[rank])
    group.all_reduce_max_u64(kb, 1)
    ctx.synchronize()
    if int(kb.download(np.uint64, 1)[0]) != int(max(int(k[0]) for k in keys)):
        print(f"rank {rank}: key exchange wrong")
        ok = False
    if group.p2p_error():
        print(f"rank {rank}: a bounded wait gave up at exchange {group.p2p_error()}")
        ok = False
    # graph replay: the same captured exchange launched 20 times (the sequence number lives in device memory)
    cb_buf = ctx.buffer_from(np.full(1024, float(rank + 1), np.float32))
    import ctypes as C
    from uzu_amd import _ffi
    lib = _ffi.lib()
    # capture through the HIP runtime directly: begin capture on the context stream, encode the exchange, instantiate, replay
    hip = C.CDLL("libamdhip64.so")
    stream = C.c_void_p(ctx.stream)
    graph, gexec = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamBeginCapture(stream, 2) == 0  # hipStreamCaptureModeRelaxed
    group.all_reduce_sum_f32(cb_buf, 1024)
    assert hip.hipStreamEndCapture(stream, C.byref(graph)) == 0
    assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
    for _ in range(5):
        assert hip.hipGraphLaunch(gexec, stream) == 0
    ctx.synchronize()
    total = float(sum(range(1, size + 1)))
    got = cb_buf.download(np.float32, 1024)
    # after replay j every rank holds sum over ranks of its previous value: values grow by a factor `size` per replay once all
    # ranks hold the same number: v1 = total, v2 = size * total, ...
    want_v = total * (size ** 4)
    if not np.all(got == np.float32(want_v)):
        print(f"rank {rank}: graph replay gave {got[0]} instead of {want_v}")
        ok = False
    group.close()
    ctx.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
