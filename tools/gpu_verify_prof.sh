#!/bin/bash
# GPU box: per-kernel times of a speculative verify pass (VERIFY_NODES, default 16) -- rocprofv3 --kernel-trace --stats over tools/verify_cost.py
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
N=${VERIFY_NODES:-16}
rm -rf /tmp/vprof
VERIFY_NODES=$N timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vprof -- python $R/tools/verify_cost.py > /tmp/vprof.json 2> /tmp/vprof.err
f=$(find /tmp/vprof -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out/verify
cp "$f" $R/gpurun_out/verify/kernel_stats_$N.csv 2>/dev/null
cp /tmp/vprof.json $R/gpurun_out/verify/verify_cost_$N.json
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  calls {int(r["Calls"]):6d}  avg {float(r["AverageNs"])/1e3:8.2f} us  {r["Name"][:110]}')
PY
