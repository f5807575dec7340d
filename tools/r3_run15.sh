#!/bin/bash
# round 3, call 15: the whole GPU suite + smoke + default bench (what the driver runs at round end)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r3q; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v "^E    +" | tail -25 > $O/pytest.log
tail -6 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -1 $O/bench_n1.json | cut -c1-900
