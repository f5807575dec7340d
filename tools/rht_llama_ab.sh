#!/bin/bash
# GPU box: Llama-3-8B with RHT linears through the fused decode step, stripe epilogues gated at 10 MB per matrix (shipped) vs lifted (lab build, UZU_DEC_STRIPE_MAX_MB)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r6e; mkdir -p $O
export UZU_HIP_LIB=$ROOT/uzu_amd/lib_lab/libuzu_hip.so
for mb in 10 100000; do
  UZU_DEC_STRIPE_MAX_MB=$mb timeout 600 python tools/rht_decode_cost.py --models llama-3-8b > $O/rht_llama_stripe_max_$mb.json 2> $O/rht_llama_stripe_max_$mb.err
  python - <<PY
import json
try:
    r = json.load(open("$O/rht_llama_stripe_max_$mb.json"))["models"][0]
    print("max_mb=$mb", {k: (v["tokens_per_s"], v["launches_per_token"]) for k, v in r.items() if isinstance(v, dict)}, "equal:", r["rht_fused_tokens_equal_unfused"], "rht/plain:", r["rht_over_plain"])
    print({k: v for k, v in list(r["rht_fused"]["kernel_us_per_step"].items())[:8]})
except Exception as e:
    print("max_mb=$mb failed", e); print(open("$O/rht_llama_stripe_max_$mb.err").read()[-1500:])
PY
done
