#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c22; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^E    +" | tail -30 > $O/pytest.log
tail -25 $O/pytest.log
