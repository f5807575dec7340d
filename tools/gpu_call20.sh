#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/c20
bash tools/pmc_gemm.sh; bash tools/pmc_gemm2.sh
cd $ROOT
for i in 1 2 11 12; do cp gpurun_out/pmc_gemm_$i.txt gpurun_out/c20/ 2>/dev/null; echo "## pass $i"; grep -E "917504|114688" gpurun_out/pmc_gemm_$i.txt | cut -c1-700; done
