#!/bin/bash
# prefill GEMM A/B: pair-form int4 conversion (p) x software-pipelined k16 steps (s): default lib = p1 s1, lib_g10 = p1 s0, lib_g00 = p0 s0 (round-1 loop), lib_g01 = p0 s1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c19; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k "gemm or matmul" 2>&1 | grep -v "^E    +" | tail -10 > $O/pytest.log
for v in lib lib_g10 lib_g00 lib_g01; do
  echo "== $v"; LD_LIBRARY_PATH=$ROOT/uzu_amd/$v KB_GEMM=1 timeout 200 tools/kbench 2>&1 | grep -E "gemm_q|TFLOP"
done > $O/kbench.txt 2>&1
for v in lib lib_g00; do
  UZU_HIP_LIB=$ROOT/uzu_amd/$v/libuzu_hip.so timeout 300 python bench.py --model llama-3-8b --steps 16 --warmup 2 --no-cpu-baseline > $O/llama_$v.json 2> $O/llama_$v.err
  UZU_HIP_LIB=$ROOT/uzu_amd/$v/libuzu_hip.so timeout 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline > $O/qwen_$v.json 2> $O/qwen_$v.err
done
tail -5 $O/pytest.log; cat $O/kbench.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c19/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], 'prefill', d.get('prefill_tokens_per_s'))
PY
