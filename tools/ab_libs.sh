#!/bin/bash
# GPU box (gpurun): same-box A/B of in-tree library builds (uzu_amd/lib_<name>/libuzu_hip.so, selected through UZU_HIP_LIB;
# `lib` = the shipping build).  Box-to-box variation is +-5-8 %, so every comparison runs inside one call; each variant is run
# twice, interleaved, to see the run-to-run noise on the box.
# usage: AB_LIBS="lib_base lib" tools/ab_libs.sh [out-dir]     (AB_TESTS=1: the decode-related GPU tests on the shipping build first)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
O=${1:-gpurun_out/ab_libs}; mkdir -p $O
if [ "${AB_TESTS:-0}" = 1 ]; then
  timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short 2>&1 | grep -v "^E    +" | tail -12 > $O/pytest.log
  timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -k gemv 2>&1 | tail -5 >> $O/pytest.log
  grep -E 'passed|failed|error' $O/pytest.log
fi
for rep in 1 2; do
  for v in ${AB_LIBS:-lib_base lib}; do
    L=$ROOT/uzu_amd/$v/libuzu_hip.so
    [ -f $L ] || continue
    UZU_HIP_LIB=$L timeout 300 python bench.py --steps 256 --warmup 8 --no-cpu-baseline > $O/qwen_${v}_$rep.json 2> $O/qwen_${v}_$rep.err
    [ "${AB_LLAMA:-1}" = 1 ] && UZU_HIP_LIB=$L timeout 300 python bench.py --model llama-3-8b --steps 48 --warmup 4 --no-cpu-baseline > $O/llama_${v}_$rep.json 2> $O/llama_${v}_$rep.err
  done
done
python - "$O" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get('kernel_us_per_step') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], {n.replace('gemv_dec', 'g'): round(v['us'] / v['calls'], 2) for n, v in k.items() if 'gemv' in n or 'attn' in n})
    except Exception as e:
        print(f, 'ERR', e)
PY
