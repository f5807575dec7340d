#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c9; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -rf --tb=short -s -k "fixture or rope_variants or output_rht or sequence_states" 2>&1 | grep -v "^  File \"<frozen" > $O/pytest_full.log
grep -n "passed\|failed" $O/pytest_full.log | tail -3; grep -n "^FAILED\|^E  \|^fixture" $O/pytest_full.log | head -40
