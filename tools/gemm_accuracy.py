"""Compare the matrix-core GEMM and the GEMV-tiled path with the CPU reference on a few shapes (run on a GPU box).
UZU_GEMM_MIN_M=1000000 disables the matrix-core path."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_gpu_kernels as T
from uzu_amd import backend as B
from helpers import ulp_diff_bf16

ctx = B.Context.new()
for (n, k, g, m, act) in [(1544, 256, 64, 32, "uniform"), (1544, 256, 64, 32, "shifted"), (256, 512, 64, 32, "uniform"), (8224, 1024, 128, 64, "uniform"),
                          (8224, 1024, 128, 64, "shifted"), (1024, 3584, 128, 64, "normal")]:
    rng = np.random.default_rng(n + k + m)
    q = T.quant_matrix(rng, n, k, 4, g, 0)
    if act == "uniform":
        a = T.bf16(rng.uniform(-1, 1, size=(m, k)))
    elif act == "shifted":
        a = T.bf16(rng.uniform(-1, 1, size=(m, k)) + 0.7)
    else:
        a = T.bf16(rng.normal(size=(m, k)) * 2)
    want, got = T.oracle_matmul(a, q, m), T.hip_matmul(ctx, a, q, m)
    u = ulp_diff_bf16(want, got)
    print(f"n={n} k={k} g={g} m={m} {act:8s} identical={np.mean(want == got):.5f} max_ulp={u.max():.2f} mean_ulp={u.mean():.5f}")
