#!/bin/bash
# round 3, call 6: engine-level sliding window / sinks; the in-launch edge against the kernel boundary (tools/edge_lab.hip)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r3g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q -x -k "sliding or ring or mask or attention_prepare or kv_cache" 2>&1 | tail -15 > $O/pytest_sw.txt
echo "pytest rc=$?"; tail -12 $O/pytest_sw.txt
timeout 120 tools/edge_lab | tee $O/edge_lab.txt
