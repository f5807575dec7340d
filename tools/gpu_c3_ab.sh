#!/bin/bash
# GPU box: BASELINE configs[2] (bench.py --config c3) with the 256-thread GEMM forced / the plan's choice (ping-pong form on the long-K linears), alternating runs
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/gemm_ab; mkdir -p $O
for r in 1 2 3; do
  UZU_HIP_TUNE=gemm_form=0 timeout 600 python bench.py --config c3 --steps 4 --warmup 1 --no-cpu-baseline > $O/c3_pp0_$r.json 2> $O/c3_pp0.err
  timeout 600 python bench.py --config c3 --steps 4 --warmup 1 --no-cpu-baseline > $O/c3_auto_$r.json 2> $O/c3_auto.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/gemm_ab/c3_*_?.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], "ms/step", d["ms_per_step"], "single", d.get("single_sequence_prefill_tokens_per_s"), "crc", d.get("first_tokens"))
    except Exception as e:
        print(f, "ERR", e)
PY
