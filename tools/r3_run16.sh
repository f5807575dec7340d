#!/bin/bash
# round 3, call 16: GEMM offset tables from load time / from the normalisation: parity + prefill A/B + profile
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r3r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -4
for t in 0 1; do
  UZU_GEMM_TABLES=$t timeout 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline > $O/qwen_tables$t.json 2> $O/qwen_tables$t.err
  UZU_GEMM_TABLES=$t timeout 400 python bench.py --model llama-3-8b --steps 32 --warmup 4 --no-cpu-baseline > $O/llama_tables$t.json 2> $O/llama_tables$t.err
done
python - "$O" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'prefill', d.get('prefill_tokens_per_s'), d.get('timed_tokens_crc32'))
    except Exception as e:
        print(f, 'ERR', e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/trace -o r3r -- python $ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline > $ROOT/$O/prof.log 2>&1
cd $ROOT
f=$(ls $O/trace/*kernel_stats.csv | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:16]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e3:10.1f} us  avg {float(r['AverageNs'])/1e3:8.2f}")
PY
