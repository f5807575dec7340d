#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c10; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -rf --tb=short -k "qwen3_14b" 2>&1 | tail -15 > $O/pytest.log
timeout 900 python bench.py --config c3 > $O/bench_c3.json 2> $O/bench_c3.err
timeout 600 python bench.py --config c4 --steps 32 --warmup 4 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
timeout 900 python bench.py --config c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
tail -3 $O/pytest.log; for f in $O/bench_c*.json; do tail -1 $f | cut -c1-700; done; tail -3 $O/bench_c3.err $O/bench_c5.err
