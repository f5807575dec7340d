#!/bin/bash
# round 3, call 5: exact mode, batched activation epilogue, stream kernel: full GPU suite + A/B benches
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r3f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -40 > $O/pytest.txt
echo "pytest rc=$?"; tail -30 $O/pytest.txt
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
