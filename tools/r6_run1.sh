#!/bin/bash
# GPU box, round 6, first call: baseline of the round's build + the records the round-5 review asked for (reference-order price tag, the
# reference's GEMV microbenchmark grid, the GPU suite under the NaN-poison CI mode, censuses of the side configurations).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r6a; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 600 $O/bench_n1.json | head -c 300; echo
timeout 400 python bench.py --exact > $O/bench_exact.json 2> $O/bench_exact.err; head -c 400 $O/bench_exact.json; echo; tail -3 $O/bench_exact.err
timeout 600 python tools/gemv_grid.py --out $O/gemv_grid.json > $O/gemv_grid.log 2>&1; tail -3 $O/gemv_grid.log
# the GPU suite with every fresh allocation / recycled workspace block full of NaNs (level 1), then the engine-level files with scratch + KV poisoned too (level 2)
UZU_HIP_POISON=1 timeout 1200 python -m pytest tests -m gpu -q --tb=short -x -k "not scale and not census" 2>&1 | grep -v "^E    +" | tail -25 > $O/pytest_poison1.log; tail -5 $O/pytest_poison1.log
UZU_HIP_POISON=2 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_tree_verify.py tests/test_gpu_layer_options.py -m gpu -q --tb=short -k "not scale and not census" 2>&1 | grep -v "^E    +" | tail -40 > $O/pytest_poison2.log; tail -8 $O/pytest_poison2.log
timeout 500 python tools/parity_census.py --config c4 --variants 32 --steps 8 --budget-s 400 --out $O/parity_census_c4.json > $O/census_c4.log 2>&1; tail -2 $O/census_c4.log
timeout 600 python tools/parity_census.py --config c5 --variants 32 --steps 4 --budget-s 500 --out $O/parity_census_c5.json > $O/census_c5.log 2>&1; tail -2 $O/census_c5.log
