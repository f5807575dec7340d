#!/bin/bash
# round 3, call 14: prefill profile (kernel trace) after a GEMM change; GEMM parity first
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r3p; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "norm" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/trace -o r3p -- python $ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline > $ROOT/$O/prof.log 2>&1
cd $ROOT
tail -1 $O/prof.log | cut -c1-400
f=$(ls $O/trace/*kernel_stats.csv | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e3:10.1f} us  avg {float(r['AverageNs'])/1e3:8.2f}")
PY
