#!/bin/bash
# round 2, call 14: wide workgroups (12-16 waves) for the bandwidth-regime prologue GEMVs, A/B via UZU_DEC_WIDE; flash-attention
# task mapping for any GQA factor; full GPU test suite
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c14; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^E    +" | tail -30 > $O/pytest.log
for w in 1 0; do
  UZU_DEC_WIDE=$w timeout 300 python bench.py --model llama-3-8b --steps 48 --warmup 4 --no-cpu-baseline > $O/llama_int4_wide$w.json 2> $O/llama_int4_wide$w.err
  UZU_DEC_WIDE=$w timeout 300 python bench.py --config c4 --steps 48 --warmup 4 --no-cpu-baseline > $O/llama_int8_wide$w.json 2> $O/llama_int8_wide$w.err
  UZU_DEC_WIDE=$w timeout 300 python bench.py --steps 192 --warmup 8 --no-cpu-baseline > $O/qwen_wide$w.json 2> $O/qwen_wide$w.err
  UZU_DEC_WIDE=$w timeout 600 python bench.py --config c5 --no-cpu-baseline > $O/c5_wide$w.json 2> $O/c5_wide$w.err
done
tail -12 $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c14/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d.get('kernel_us_per_step') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'prefill', d.get('prefill_tokens_per_s'), {n:round(v['us']/v['calls'],1) for n,v in k.items()})
    except Exception as e: print(f, 'ERR', e)
PY
