#!/bin/bash
# round 3, call 13: trie mask + in-kernel GEMM row sums + XCD-aware chunk scan: parity, prefill A/B
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r3n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm or trie or delta or attention or matmul" 2>&1 | tail -6 > $O/pytest_sel.txt
tail -4 $O/pytest_sel.txt
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "prefill or fixture or full_size" 2>&1 | tail -4 > $O/pytest_model.txt
tail -3 $O/pytest_model.txt
for pre in 1 0; do
  UZU_GEMM_PREPASS=$pre timeout 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline > $O/qwen_prepass$pre.json 2> $O/qwen_prepass$pre.err
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
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/trace -o r3n -- python $ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline > $ROOT/$O/prof.log 2>&1
cd $ROOT
f=$(ls $O/trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls $O/trace/*kernel_stats.csv | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:26]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e3:10.1f} us  avg {float(r['AverageNs'])/1e3:8.2f}")
PY
