#!/bin/bash
# GPU box (gpurun): A/B of the decode GEMV across in-tree library builds (uzu_amd/lib = shipping, lib_r1 = round-1 HEAD,
# lib_nont = default-policy weight loads) with tools/kbench (graph-captured launches, weights rotated through 96
# buffers so they stream from HBM), then the rows-per-wave / resident-workgroups sweep on the shipping build.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for v in ${AB_LIBS:-lib_r1 lib lib_nont}; do
  [ -f uzu_amd/$v/libuzu_hip.so ] || continue
  echo "=== $v: Qwen3.5-0.8B decode shapes"
  KB_NBUF=96 LD_LIBRARY_PATH=$ROOT/uzu_amd/$v timeout 120 tools/kbench
  echo "=== $v: Llama-3-8B decode shapes + read-outs"
  KB_LLAMA=1 LD_LIBRARY_PATH=$ROOT/uzu_amd/$v timeout 120 tools/kbench
done
echo "=== sweep on lib: UZU_DEC_R x UZU_DEC_CAP (big matrices)"
for R in 1 2 4; do for CAP in 2 4 6 8; do
  echo "--- R=$R CAP=$CAP"
  UZU_DEC_R=$R UZU_DEC_CAP=$CAP KB_LLAMA=1 LD_LIBRARY_PATH=$ROOT/uzu_amd/lib timeout 120 tools/kbench
done; done
echo "=== sweep on lib: UZU_DEC_R x UZU_DEC_TW (small matrices)"
for R in 1 2 4; do for TW in 4 8 16; do
  echo "--- R=$R TW=$TW"
  UZU_DEC_R=$R UZU_DEC_TW=$TW KB_NBUF=96 LD_LIBRARY_PATH=$ROOT/uzu_amd/lib timeout 120 tools/kbench 2>&1 | grep "gemv_dec"
done; done
