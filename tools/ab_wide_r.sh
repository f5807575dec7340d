#!/bin/bash
# GPU box (gpurun): same-box A/B of the decode GEMV plan switches on the bandwidth-regime models (Llama-3-8B int4, Qwen3-14B-class):
#   UZU_DEC_WIDE_R=0|1|2   rows per lane group of the wide workgroups: the round-count rule (shipping) | one | two everywhere
#   UZU_DEC_SPREAD=1|0     a matrix smaller than one round of the resident waves runs on every CU (shipping) | on as few as fill 16 waves
# usage: [AB_VAR=UZU_DEC_SPREAD AB_VALUES="1 0 1 0"] tools/ab_wide_r.sh     (prints tokens/s and the per-kernel us of the HIP-event profile step)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
VAR=${AB_VAR:-UZU_DEC_WIDE_R}
O=gpurun_out/ab_wr; mkdir -p $O
for r in ${AB_VALUES:-0 2 0 2}; do
  env $VAR=$r timeout 300 python bench.py --model llama-3-8b --steps 48 --warmup 4 --no-cpu-baseline > $O/llama_r${r}_$RANDOM.json 2>/dev/null
  env $VAR=$r timeout 600 python bench.py --config c5 --no-cpu-baseline > $O/c5_r${r}_$RANDOM.json 2>/dev/null
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob('gpurun_out/ab_wr/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get('kernel_us_per_step') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], {n.replace('gemv_dec', 'g'): round(v['us'] / v['calls'], 2) for n, v in k.items() if 'gemv' in n or 'attn' in n})
    except Exception as e:
        print(f, 'ERR', e)
PY
