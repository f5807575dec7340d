#!/bin/bash
# wide workgroups with dynamic (LDS counter) batch hand-out: UZU_DEC_WIDE=0 (off) / 1 (int4 non-readout) / 2 (every bandwidth-regime kernel)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short 2>&1 | grep -v "^E    +" | tail -12 > $O/pytest.log
for w in 1 2 0; do
  UZU_DEC_WIDE=$w timeout 300 python bench.py --model llama-3-8b --steps 48 --warmup 4 --no-cpu-baseline > $O/llama_int4_wide$w.json 2> $O/llama_int4_wide$w.err
  UZU_DEC_WIDE=$w timeout 300 python bench.py --config c4 --steps 48 --warmup 4 --no-cpu-baseline > $O/llama_int8_wide$w.json 2> $O/llama_int8_wide$w.err
  UZU_DEC_WIDE=$w timeout 300 python bench.py --steps 192 --warmup 8 --no-cpu-baseline > $O/qwen_wide$w.json 2> $O/qwen_wide$w.err
done
for w in 1 2; do UZU_DEC_WIDE=$w timeout 600 python bench.py --config c5 --no-cpu-baseline > $O/c5_wide$w.json 2> $O/c5_wide$w.err; done
tail -5 $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c17/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d.get('kernel_us_per_step') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], {n:round(v['us']/v['calls'],1) for n,v in k.items() if 'gemv' in n})
    except Exception as e: print(f, 'ERR', e)
PY
