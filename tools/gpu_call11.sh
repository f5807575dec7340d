#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c11; mkdir -p $O
timeout 400 python -m pytest tests/test_tp.py -m gpu -q -rf --tb=short -k "p2p" 2>&1 | tail -30 > $O/pytest_p2p.log
timeout 900 python bench.py --config c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
cat $O/pytest_p2p.log | tail -25; tail -1 $O/bench_c5.json | cut -c1-600; tail -2 $O/bench_c5.err
