#!/bin/bash
# round 3, call 2: the LDS-staged weight stream (csrc/k_stream.hip) -- parity first, then same-box A/B against the register GEMVs
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r3b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "stream" 2>&1 | tail -25 > $O/pytest_stream.txt
echo "pytest rc=$?"; tail -25 $O/pytest_stream.txt
for m in 0 1; do
  UZU_DEC_STREAM=$m KB_LLAMA=1 timeout 120 tools/kbench > $O/kbench_llama_stream$m.txt 2>&1
  echo "--- kbench stream=$m"; cat $O/kbench_llama_stream$m.txt
done
for m in 0 1; do
  UZU_DEC_STREAM=$m timeout 400 python bench.py --model llama-3-8b --steps 48 --warmup 4 --no-cpu-baseline > $O/llama_int4_stream$m.json 2> $O/llama_int4_stream$m.err
  UZU_DEC_STREAM=$m timeout 300 python bench.py --steps 192 --warmup 8 --no-cpu-baseline > $O/qwen_stream$m.json 2> $O/qwen_stream$m.err
done
python - "$O" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get('kernel_us_per_step') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('timed_tokens_crc32'), {n: round(v['us'] / v['calls'], 1) for n, v in k.items() if 'gemv' in n})
    except Exception as e:
        print(f, 'ERR', e)
PY
L=$ROOT/uzu_amd/lib_tl/libuzu_hip.so
UZU_HIP_LIB=$L timeout 400 python tools/timeline.py --model llama-3-8b --detail 9 > $O/timeline_llama_stream.txt 2> $O/timeline_llama_stream.err
head -40 $O/timeline_llama_stream.txt; tail -3 $O/timeline_llama_stream.err
