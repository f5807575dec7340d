#!/bin/bash
# round 3, call 12: the weight-stream A/B behind profiles/r3_kbench_stream_ab.txt and r3_timeline_llama_stream_*.txt, default policy check
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r3m; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "stream" 2>&1 | tail -5 > $O/pytest_stream.txt
tail -3 $O/pytest_stream.txt
{
for cfg in "register GEMV|UZU_DEC_STREAM=0" "default policy|UZU_DEC_STREAM=1" "LDS stream, dot2 consumers, every supported shape|UZU_DEC_STREAM=2" "LDS stream, MFMA consumers where supported|UZU_DEC_STREAM=2 UZU_STREAM_MFMA=1"; do
  name=${cfg%%|*}; envs=${cfg##*|}
  echo "--- $name ($envs)"
  env $envs KB_LLAMA=1 timeout 120 tools/kbench 2>&1
done
} > $O/kbench_stream_ab.txt
cat $O/kbench_stream_ab.txt
for m in 0 1; do
  UZU_DEC_STREAM=$m timeout 300 python bench.py --steps 192 --warmup 8 --no-cpu-baseline > $O/qwen_stream$m.json 2> $O/qwen_stream$m.err
done
UZU_DEC_STREAM=2 timeout 400 python bench.py --model llama-3-8b --steps 48 --warmup 4 --no-cpu-baseline > $O/llama_int4_stream2.json 2> $O/llama_int4_stream2.err
UZU_DEC_STREAM=0 timeout 400 python bench.py --model llama-3-8b --steps 48 --warmup 4 --no-cpu-baseline > $O/llama_int4_stream0.json 2> $O/llama_int4_stream0.err
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
UZU_DEC_STREAM=2 UZU_HIP_LIB=$L timeout 400 python tools/timeline.py --model llama-3-8b > $O/timeline_llama_stream_dot2.txt 2> $O/tl1.err
UZU_DEC_STREAM=2 UZU_STREAM_MFMA=1 UZU_HIP_LIB=$L timeout 400 python tools/timeline.py --model llama-3-8b > $O/timeline_llama_stream_mfma.txt 2> $O/tl2.err
UZU_DEC_STREAM=0 UZU_HIP_LIB=$L timeout 400 python tools/timeline.py --model llama-3-8b > $O/timeline_llama_register.txt 2> $O/tl3.err
sed -n 8,13p $O/timeline_llama_stream_dot2.txt | cut -c1-200
