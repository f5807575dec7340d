#!/bin/bash
# round 2, call 12: four-item ring in the bandwidth-regime GEMVs (A/B against the double buffer via UZU_DEC_DEEP_MB=0),
# attn_dec with all query heads of a KV head per workgroup + more splits at long context
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c12; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q -x --tb=short 2>&1 | tail -15 > $O/pytest.log
for deep in 16 0; do
  UZU_DEC_DEEP_MB=$deep timeout 300 python bench.py --model llama-3-8b --steps 48 --warmup 4 --no-cpu-baseline > $O/llama_int4_deep$deep.json 2> $O/llama_int4_deep$deep.err
  UZU_DEC_DEEP_MB=$deep timeout 300 python bench.py --config c4 --steps 48 --warmup 4 --no-cpu-baseline > $O/llama_int8_deep$deep.json 2> $O/llama_int8_deep$deep.err
  UZU_DEC_DEEP_MB=$deep timeout 300 python bench.py --steps 192 --warmup 8 --no-cpu-baseline > $O/qwen_deep$deep.json 2> $O/qwen_deep$deep.err
done
UZU_DEC_DEEP_MB=12 timeout 300 python bench.py --model llama-3-8b --steps 48 --warmup 4 --no-cpu-baseline > $O/llama_int4_deep12.json 2> $O/llama_int4_deep12.err
timeout 600 python bench.py --config c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
tail -8 $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c12/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d.get('kernel_us_per_step') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], {n:round(v['us']/v['calls'],1) for n,v in k.items()})
    except Exception as e: print(f, 'ERR', e)
PY
