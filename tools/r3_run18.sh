#!/bin/bash
# round 3, call 18: rows-normalisation kernel: bit identity, model parity, token identity of the bench stream
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r3s; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "normalization" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3
for t in 0 1; do
  UZU_NORM_ROWS=$t timeout 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline > $O/qwen_rows$t.json 2> $O/qwen_rows$t.err
done
python - "$O" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + '/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], 'prefill', d.get('prefill_tokens_per_s'), d.get('timed_tokens_crc32'))
PY
