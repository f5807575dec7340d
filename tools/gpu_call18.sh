#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "attention" 2>&1 | grep -v "^E    +" | tail -8 > $O/pytest.log
for t in 0 4; do
  UZU_ATTN_TPW=$t timeout 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline > $O/qwen_tpw$t.json 2> $O/qwen_tpw$t.err
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/c18/trace -- python $ROOT/bench.py --steps 16 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/c18/trace.log 2>&1
cd $ROOT
tail -4 $O/pytest.log
python - <<'PY'
import json,glob,csv
for f in sorted(glob.glob('gpurun_out/c18/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], 'prefill', d.get('prefill_tokens_per_s'))
for f in glob.glob('gpurun_out/c18/trace/*/*kernel_stats.csv'):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:14]: print(r['Name'][:60], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
