#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r6c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_moe.py tests/test_gpu_dflash.py -m gpu -q --tb=short 2>&1 | grep -v "^E    +" | tail -60 > $O/pytest_moe_dflash.log; tail -25 $O/pytest_moe_dflash.log
UZU_HIP_POISON=2 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_tree_verify.py tests/test_gpu_layer_options.py tests/test_gpu_dflash.py tests/test_gpu_moe.py -m gpu -q --tb=line -k "not scale and not census" 2>&1 | grep -v "^E    +" > $O/pytest_poison2_full.log; grep -E "^FAILED|passed|failed|core" $O/pytest_poison2_full.log | head -40
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r6c/prefill_trace -- python $ROOT/tools/prefill_profile.py > $ROOT/gpurun_out/r6c/prefill_trace.log 2>&1
cd $ROOT; tail -2 $O/prefill_trace.log
python - <<'PY'
import csv, glob
f = sorted(glob.glob('gpurun_out/r6c/prefill_trace/**/*kernel_stats.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
out = open('gpurun_out/r6c/prefill_kernel_stats.csv', 'w')
out.write('kernel,calls,total_us,avg_us,pct\n')
for r in rows[:40]:
    line = f"{r['Name'][:90].replace(',', ';')},{r['Calls']},{int(r['TotalDurationNs']) / 1e3:.1f},{float(r['AverageNs']) / 1e3:.3f},{r['Percentage']}"
    out.write(line + '\n')
    print(line)
PY
find gpurun_out/r6c/prefill_trace -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
