ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r6f; mkdir -p $O
export UZU_HIP_LIB=$ROOT/uzu_amd/lib_lab/libuzu_hip.so
for mask in 0 1 2 3; do
  UZU_LAB_RS_MASK=$mask timeout 300 python tools/ab_prefill_bits.py --model qwen3.5-0.8b --prompt 2043 --dump $O/l_$mask.npy 2>/dev/null | tail -1 | cut -c1-330 | sed "s/^/mask=$mask /"
done
UZU_GEMM_TABLES=0 timeout 300 python tools/ab_prefill_bits.py --model qwen3.5-0.8b --prompt 2043 --dump $O/l_t0.npy 2>/dev/null | tail -1 | cut -c1-330
python - <<PY
import numpy as np
f = lambda x: (x.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
b = np.load("$O/l_t0.npy")
for m in "0123":
    a = np.load("$O/l_%s.npy" % m)
    print("mask", m, "vs tables=0: identical %.4f  max diff/sigma %.5f" % ((a == b).mean(), np.abs(f(a) - f(b)).max() / f(b).std()))
PY
rm -f $O/l_*.npy
