#!/bin/bash
# GPU box: ping-pong GEMM variants (uzu_amd/lib_v/<name>/libuzu_hip.so) against the 256-thread form, tools/kbench KB_GEMM_AB; names ending in t are -DUZU_GEMM_PP_TIMING builds
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/gemm_ab; mkdir -p $O; rm -f $O/kbench_variants.txt
if [ -n "$PYTEST" ]; then UZU_HIP_TUNE=gemm_form=${FORM:-1} timeout 420 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm or matmul or large_tile" --tb=short 2>&1 | tail -8 > $O/pytest_pp.log; tail -4 $O/pytest_pp.log; fi
for v in ${VARIANTS:-$(ls uzu_amd/lib_v)}; do
  echo "=== variant $v" >> $O/kbench_variants.txt
  T=""; case $v in t*|*t) T=1;; esac
  env LD_LIBRARY_PATH=$PWD/uzu_amd/lib_v/$v KB_GEMM_AB=1 ${SHORT:+KB_GEMM_AB_SHORT=1} ${T:+KB_GEMM_PP_TIMING=1} timeout 120 tools/kbench >> $O/kbench_variants.txt 2>&1
done
grep -v "us/launch" $O/kbench_variants.txt
