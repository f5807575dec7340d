#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r6f; mkdir -p $O
bash tools/rowsum_check.sh
timeout 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench_c2.json; python -c "
import json; d=json.load(open('$O/bench_c2.json')); print('c2', d['value'], d['prefill_tokens_per_s'], d['prefill_roofline']['frac'], d.get('parity',{}).get('tokens_equal_oracle'))"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | grep -v "^E    +" | tail -30 > $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log
