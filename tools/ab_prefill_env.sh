#!/bin/bash
# GPU box (gpurun): same-box A/B of the prefill switches of one library build.
#   UZU_GEMM_ACT=0|1       GatedActMul as its own kernel | in the up projection's GEMM epilogue (shipping)
#   UZU_ATTN_TPW=4|2|1|0   wave tasks per workgroup of the flash-attention prefill kernel (0 = chosen by grid size, shipping)
# The GEMM loop variants (pair-form conversion, pipelined k16 steps) are compile-time: build k_gemm128.hip with
# -DUZU_GEMM_PAIR_DEQUANT=0 / -DUZU_GEMM_PIPE=0 into a second library directory and compare with
#   LD_LIBRARY_PATH=uzu_amd/<dir> KB_GEMM=1 tools/kbench        (profiles/r2_gemm128_pmc.txt holds the round-2 run)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
run() { "$@" 2> /dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'tok/s decode,', d.get('prefill_tokens_per_s'), 'tok/s prefill')"; }
for v in 1 0; do
  echo "UZU_GEMM_ACT=$v qwen3.5-0.8b:"; UZU_GEMM_ACT=$v run python bench.py --steps 16 --warmup 2 --no-cpu-baseline
  echo "UZU_GEMM_ACT=$v llama-3-8b:"; UZU_GEMM_ACT=$v run python bench.py --model llama-3-8b --steps 8 --warmup 2 --no-cpu-baseline
done
for v in 0 4; do echo "UZU_ATTN_TPW=$v qwen3.5-0.8b:"; UZU_ATTN_TPW=$v run python bench.py --steps 16 --warmup 2 --no-cpu-baseline; done
