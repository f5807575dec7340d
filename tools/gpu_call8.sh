#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rf --tb=short -k "activation_transform or int8_symmetric or output_rht or sparse or argmax_exact or capture" 2>&1 | grep -v "^  File \"<frozen" > $O/pytest_full.log
grep -n "passed\|failed" $O/pytest_full.log | tail -3; grep -n "^FAILED\|^E  " $O/pytest_full.log | head -60
