#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c6; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -60) > $O/pytest.log
tail -30 $O/pytest.log
