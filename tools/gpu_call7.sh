#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c7; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -rf --tb=short 2>&1 | grep -v "^  File \"<frozen" > $O/pytest_full.log
grep -n "passed\|failed" $O/pytest_full.log | tail -3; grep -n "^FAILED\|^E  " $O/pytest_full.log | head -60
