#!/bin/bash
# round 3, call 9: MFMA-consumer stream kernel v2 (every consumer on every slot, scalar-run loader): parity, kbench A/B, llama bench A/B, timeline
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r3k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "stream" 2>&1 | tail -25 > $O/pytest_stream.txt
echo "pytest rc=$?"; tail -8 $O/pytest_stream.txt
run_kb() { local name=$1; shift; env "$@" KB_LLAMA=1 timeout 120 tools/kbench > $O/kbench_$name.txt 2>&1; echo "--- kbench $name ($*)"; cat $O/kbench_$name.txt; }
run_kb reg UZU_DEC_STREAM=0
run_kb mfma UZU_DEC_STREAM=1
run_kb mfma_d0 UZU_DEC_STREAM=1 UZU_STREAM_PRO_DELAY=0
run_kb mfma_d24 UZU_DEC_STREAM=1 UZU_STREAM_PRO_DELAY=24
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
UZU_HIP_LIB=$L timeout 400 python tools/timeline.py --model llama-3-8b > $O/timeline_llama_stream.txt 2> $O/timeline_llama_stream.err
sed -n 8,30p $O/timeline_llama_stream.txt | cut -c1-200; tail -3 $O/timeline_llama_stream.err
