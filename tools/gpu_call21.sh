#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c21; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "delta_net" 2>&1 | grep -v "^E    +" | tail -25 > $O/pytest.log
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -k "qwen or tiny" 2>&1 | grep -v "^E    +" | tail -12 > $O/pytest_model.log
for v in 1 0; do
  UZU_DN_CHUNK_MFMA=$v timeout 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline > $O/qwen_mfma$v.json 2> $O/qwen_mfma$v.err
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/c21/trace -- python $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/c21/trace.log 2>&1
cd $ROOT
tail -12 $O/pytest.log; tail -4 $O/pytest_model.log
python - <<'PY'
import json,glob,csv
for f in sorted(glob.glob('gpurun_out/c21/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], 'prefill', d.get('prefill_tokens_per_s'))
for f in glob.glob('gpurun_out/c21/trace/*/*kernel_stats.csv'):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:8]: print(r['Name'][:60], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
