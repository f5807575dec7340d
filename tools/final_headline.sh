#!/bin/bash
# GPU box: the headline subset of tools/final_round.sh on the final build -- GPU suite, smoke, the default bench line + the driver's arguments, the rocprofv3 passes, the prefill table
R=${1:-r6}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/final2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^E    +" | tail -15 > $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
timeout 300 python bench.py --config c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
bash tools/refresh_profiles.sh $R > $O/refresh.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_prefill_trace -- python $ROOT/tools/prefill_profile.py > $ROOT/$O/prefill_trace.log 2>&1
cd $ROOT
python - <<PY
import csv, glob
f = sorted(glob.glob('gpurun_out/${R}_prefill_trace/**/*kernel_stats.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
out = open('$O/prefill_kernel_stats.csv', 'w')
out.write('kernel,calls,total_us,avg_us,pct\n')
for r in rows[:40]:
    out.write(f"{r['Name'][:100].replace(',', ';')},{r['Calls']},{int(r['TotalDurationNs']) / 1e3:.1f},{float(r['AverageNs']) / 1e3:.3f},{r['Percentage']}\n")
PY
find gpurun_out/${R}_prefill_trace -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
tail -3 $O/pytest.log; tail -1 $O/smoke.log; head -18 $O/prefill_kernel_stats.csv | cut -c1-130
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/final2/bench_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d.get('roofline') or {}
    print(f.split('/')[-1], d['value'], d.get('ms_per_step'), 'prefill', d.get('prefill_tokens_per_s'), (d.get('prefill_roofline') or {}).get('frac'), 'frac', r.get('frac'), 'parity', (d.get('parity') or {}).get('tokens_equal'), (d.get('parity') or {}).get('of'))
PY
