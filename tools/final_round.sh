#!/bin/bash
# GPU box (gpurun): the end-of-round measurement set -- full GPU test suite, smoke, the bench line of every BASELINE configuration,
# the three rocprofv3 passes behind profiles/<round>_*, the per-workgroup timelines.  Results under gpurun_out/final and
# gpurun_out/<round>_*; afterwards, locally: python tools/summarize_profile.py <round> gpurun_out/<round>_trace gpurun_out/<round>_pmc gpurun_out/<round>_mfma
R=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^E    +" | tail -15 > $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
timeout 300 python bench.py --model llama-3-8b --steps 64 --warmup 4 --no-cpu-baseline > $O/bench_llama_int4.json 2> $O/bench_llama_int4.err
timeout 600 python bench.py --config c3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
timeout 300 python bench.py --config c4 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
timeout 600 python bench.py --config c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
timeout 300 python tools/rht_decode_cost.py > $O/rht_decode.json 2> $O/rht_decode.err
bash tools/refresh_profiles.sh $R > $O/refresh.log 2>&1
cd $ROOT
[ -f uzu_amd/lib_tl/libuzu_hip.so ] && bash tools/timeline_run.sh $O > $O/timeline.log 2>&1
tail -4 $O/pytest.log; tail -2 $O/smoke.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/final/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get('roofline') or {}
        print(f.split('/')[-1], d['value'], d['unit'], 'ms/step', d.get('ms_per_step'), 'prefill', d.get('prefill_tokens_per_s'), 'frac', r.get('frac'), (r.get('rocprofv3') or {}).get('frac'),
              'cpu', (d.get('cpu_baseline') or {}).get('value'))
    except Exception as e:
        print(f, 'ERR', e)
PY
timeout 300 python tools/verify_cost.py > $O/verify_cost.json 2> $O/verify_cost.err
timeout 500 python tools/spec_round_cost.py --out $O/spec_round_cost.json > $O/spec_round_cost.log 2>&1
timeout 600 python bench.py --exact --steps 32 --warmup 5 --no-cpu-baseline > $O/bench_exact.json 2> $O/bench_exact.err
timeout 1500 python tools/parity_census.py --out $O/parity_census_c2.json > $O/parity_census_c2.log 2>&1

# per-kernel table of the prefill passes alone
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
head -16 $O/prefill_kernel_stats.csv | cut -c1-140
