#!/bin/bash
# GPU box: the row sums filed by the producers of a prefill GEMM's activation rows (norm, GatedActMul epilogue, DeltaNet norm-gate) against the GEMM's own
# pre-pass launch (lab build: UZU_GEMM_TABLES=0): launches per pass, prompt tok/s, and how far the last-row logits move (f32 sums in another order).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r6f; mkdir -p $O
export UZU_HIP_LIB=$ROOT/uzu_amd/lib_lab/libuzu_hip.so
for model in qwen3.5-0.8b llama-3-8b; do
  for t in 1 0 1 0; do
    UZU_GEMM_TABLES=$t timeout 300 python tools/ab_prefill_bits.py --model $model --prompt 2043 --dump $O/logits_${model}_$t.npy 2>/dev/null | tail -1 | sed "s/^/tables=$t /" | tee -a $O/rowsum_check.txt
  done
  python - <<PY | tee -a $O/rowsum_check.txt
import numpy as np
a = np.load("$O/logits_${model}_1.npy"); b = np.load("$O/logits_${model}_0.npy")
f = lambda x: (x.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
print("$model: logits bit-identical fraction %.4f, max |diff| / sigma %.5f, argmax equal %s" % ((a == b).mean(), np.abs(f(a) - f(b)).max() / f(b).std(), int(f(a).argmax()) == int(f(b).argmax())))
PY
done
rm -f $O/logits_*.npy
