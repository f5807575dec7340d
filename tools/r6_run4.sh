#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r6d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_weaver.py -m gpu -q --tb=short -x 2>&1 | grep -v "^E    +" | tail -60 > $O/pytest_weaver.log; tail -30 $O/pytest_weaver.log
timeout 900 python -m pytest tests/test_gpu_dflash.py tests/test_gpu_layer_options.py tests/test_gpu_prefill_switches.py -m gpu -q --tb=short 2>&1 | grep -v "^E    +" | tail -40 > $O/pytest_dflash_layeropts.log; tail -8 $O/pytest_dflash_layeropts.log
UZU_HIP_POISON=2 timeout 900 python -m pytest tests/test_gpu_weaver.py tests/test_gpu_dflash.py tests/test_gpu_layer_options.py tests/test_gpu_tree_verify.py -m gpu -q --tb=line 2>&1 | grep -v "^E    +" > $O/pytest_poison2.log; grep -E "^FAILED|passed|failed|core" $O/pytest_poison2.log | head -40
