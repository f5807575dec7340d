#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c24; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -k "gated_act or tiny or fused" 2>&1 | grep -v "^E    +" | tail -8 > $O/pytest.log
for v in 1 0 1 0; do
  UZU_GEMM_ACT=$v timeout 300 python bench.py --steps 16 --warmup 2 --no-cpu-baseline 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('qwen act$v', d['value'], d['prefill_tokens_per_s'])"
  UZU_GEMM_ACT=$v timeout 300 python bench.py --model llama-3-8b --steps 8 --warmup 2 --no-cpu-baseline 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('llama act$v', d['value'], d['prefill_tokens_per_s'])"
done > $O/ab.txt
tail -4 $O/pytest.log; cat $O/ab.txt
