#!/bin/bash
# GPU box: the headline line (bench.py, BASELINE configs[1]) with the 256-thread GEMM forced / the plan's choice, alternating runs: prefill tokens/s
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/gemm_ab; mkdir -p $O
for r in 1 2 3; do
  UZU_HIP_TUNE=gemm_form=0 timeout 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline > $O/c2_f0_$r.json 2> $O/c2_f0.err
  timeout 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline > $O/c2_plan_$r.json 2> $O/c2_plan.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/gemm_ab/c2_*_?.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "decode", d["value"], "prefill", d.get("prefill_tokens_per_s"), "crc", d.get("timed_tokens_crc32"), (d.get("parity") or {}).get("tokens_equal"))
    except Exception as e:
        print(f, "ERR", e)
PY
