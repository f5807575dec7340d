#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c13; mkdir -p $O
UZU_DEC_DEEP_MB=0 timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -k "full_size" 2>&1 | tail -5 > $O/pytest_deep0.log
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q --tb=short 2>&1 | grep -v "^E    +" | tail -40 > $O/pytest.log
echo "--- deep off"; cat $O/pytest_deep0.log; echo "--- default"; cat $O/pytest.log
