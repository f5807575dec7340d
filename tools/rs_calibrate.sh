ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r6f; mkdir -p $O
export UZU_HIP_LIB=$ROOT/uzu_amd/lib_lab/libuzu_hip.so
run() { env "$@" timeout 300 python tools/ab_prefill_bits.py --model qwen3.5-0.8b --prompt 2043 --dump $O/l_$NAME.npy 2>/dev/null | tail -1 | cut -c1-120; }
NAME=old run UZU_LAB_RS_MASK=0
NAME=new run UZU_LAB_RS_MASK=3
NAME=chain run UZU_LAB_RS_MASK=0 UZU_HIP_TUNE=dn_split=0
NAME=exact run UZU_HIP_EXACT=1
python - <<PY
import numpy as np
f = lambda x: (x.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
L = {n: f(np.load("$O/l_%s.npy" % n)) for n in ("old", "new", "chain", "exact")}
s = L["exact"].std()
for a, b in (("old", "exact"), ("new", "exact"), ("chain", "exact"), ("new", "old"), ("chain", "old")):
    d = np.abs(L[a] - L[b])
    print("%5s vs %5s: max %.4f sigma  rms %.5f sigma  argmax equal %s" % (a, b, d.max() / s, np.sqrt((d ** 2).mean()) / s, L[a].argmax() == L[b].argmax()))
PY
rm -f $O/l_*.npy
