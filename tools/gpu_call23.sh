#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c23; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short 2>&1 | grep -v "^E    +" | tail -15 > $O/pytest.log
for v in 1 0; do
  UZU_GEMM_ACT=$v timeout 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline > $O/qwen_act$v.json 2> $O/qwen_act$v.err
  UZU_GEMM_ACT=$v timeout 300 python bench.py --model llama-3-8b --steps 16 --warmup 2 --no-cpu-baseline > $O/llama_act$v.json 2> $O/llama_act$v.err
done
tail -6 $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c23/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], 'prefill', d.get('prefill_tokens_per_s'))
PY
