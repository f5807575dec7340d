#!/bin/bash
# GPU box: the add-norm split's upper bound.  Lab build (uzu_amd/lib_lab: make LAB=1 OUT=../lib_lab) with UZU_LAB_NORM_NOADD=0 / 1, alternating on one box:
# =1 removes from every normed decode GEMV the shortcut row load, the add and the residual store -- all a producer-side residual + sum-of-squares split could take
# out of the consumer (numerics WRONG on purpose: timing only).  Qwen3.5-0.8B (headline) and Llama-3-8B int4.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r6e; mkdir -p $O
export UZU_HIP_LIB=$ROOT/uzu_amd/lib_lab/libuzu_hip.so
for rep in 1 2 3; do
  for v in 0 1; do
    UZU_LAB_NORM_NOADD=$v timeout 300 python bench.py --steps 256 --warmup 16 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('qwen3.5-0.8b noadd=$v rep=$rep', d['value'], d['ms_per_step'], d.get('gpu_ms_per_step_events'))" | tee -a $O/addnorm_bound.txt
  done
done
for rep in 1 2; do
  for v in 0 1; do
    UZU_LAB_NORM_NOADD=$v timeout 400 python bench.py --model llama-3-8b --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('llama-3-8b-int4 noadd=$v rep=$rep', d['value'], d['ms_per_step'], d.get('gpu_ms_per_step_events'))" | tee -a $O/addnorm_bound.txt
  done
done
