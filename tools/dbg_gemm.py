import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import test_gpu_kernels as T
from uzu_amd import backend
ctx = backend.Context.new(0)
def f32(b): return (b.astype(np.uint32) << 16).view(np.float32)
bad = 0
for rnd in range(6):
  for m in [16, 37, 128, 200]:
    for group_size in [64, 128]:
      for method in [0, 1, 2]:
        for bits in [4, 8]:
            rng = np.random.default_rng(bits * 1000 + method * 100 + group_size + m + rnd)
            n, k = 200, 512
            q = T.quant_matrix(rng, n, k, bits, group_size, method)
            a = T.activations(rng, m, k)
            want, got = T.oracle_matmul(a, q, m), T.hip_matmul(ctx, a, q, m)
            u = T.ulp_diff_bf16(want, got)
            if u.max() > 1.0:
                bad += 1
                rows = np.where(u.max(1) > 1)[0]; cols = np.where(u.max(0) > 1)[0]
                print("BAD", rnd, (m, group_size, method, bits), "max ulps", u.max(), "n bad", (u > 1).sum(), "rows", rows[:8], "..", rows[-3:], len(rows), "cols", cols[:8], "..", cols[-3:], len(cols))
print("bad", bad)
