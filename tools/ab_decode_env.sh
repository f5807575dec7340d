#!/bin/bash
# GPU box (gpurun): same-box A/B of the bandwidth-regime decode GEMV switches through environment variables of ONE library build
# (box-to-box variation is +-5-8 %, so every comparison runs inside one call).
#   UZU_DEC_WIDE=0|1|2   4-wave workgroups everywhere | 12-16-wave workgroups for int4 kernels with K >= 4096 (shipping) | for every
#                        bandwidth-regime kernel; the wide workgroups hand their batches out through an LDS counter
# usage: tools/ab_decode_env.sh [out-dir]     (prints tokens/s and per-kernel us of the HIP-event profile step)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
O=${1:-gpurun_out/ab_decode}; mkdir -p $O
for w in 1 2 0; do
  UZU_DEC_WIDE=$w timeout 300 python bench.py --model llama-3-8b --steps 48 --warmup 4 --no-cpu-baseline > $O/llama_int4_wide$w.json 2> $O/llama_int4_wide$w.err
  UZU_DEC_WIDE=$w timeout 300 python bench.py --config c4 --steps 48 --warmup 4 --no-cpu-baseline > $O/llama_int8_wide$w.json 2> $O/llama_int8_wide$w.err
  UZU_DEC_WIDE=$w timeout 300 python bench.py --steps 192 --warmup 8 --no-cpu-baseline > $O/qwen_wide$w.json 2> $O/qwen_wide$w.err
  UZU_DEC_WIDE=$w timeout 600 python bench.py --config c5 --no-cpu-baseline > $O/c5_wide$w.json 2> $O/c5_wide$w.err
done
python - "$O" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get('kernel_us_per_step') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], {n: round(v['us'] / v['calls'], 1) for n, v in k.items() if 'gemv' in n})
    except Exception as e:
        print(f, 'ERR', e)
PY
