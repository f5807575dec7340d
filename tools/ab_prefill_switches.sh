#!/bin/bash
# GPU box (gpurun): the round-4 prefill switches of one library build, side by side (tools/ab_prefill_bits.py prints one JSON line per run).
#   equal `logits_sha256` expected for UZU_HIP_TUNE=conv_apply4=0 and norm_partials=0 against the default; dn_split=0 sums in another order.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
O=${1:-gpurun_out/ab_prefill}; mkdir -p $O
run() { env "$@" timeout 300 python tools/ab_prefill_bits.py $MODEL_ARGS 2>> $O/err.log | tail -1; }
MODEL_ARGS="--model tiny --prompt 700"
{
  echo "# tiny-qwen at model_dim 1024, 700-token prompt"
  run X=1; run UZU_HIP_TUNE=conv_apply4=0; run UZU_HIP_TUNE=norm_partials=0; run UZU_HIP_TUNE=dn_split=0
  MODEL_ARGS="--model qwen3.5-0.8b --prompt 2043"
  echo "# qwen3.5-0.8b, 2043-token prompt"
  run X=1; run UZU_HIP_TUNE=conv_apply4=0,norm_partials=0,dn_split=0
  [ -n "$AB_EACH" ] && { run UZU_HIP_TUNE=conv_apply4=0; run UZU_HIP_TUNE=norm_partials=0; run UZU_HIP_TUNE=dn_split=0; }
} | tee $O/switches.txt
