#!/bin/bash
# GPU box, round 6, second call: the DFlash drafter against the oracle, the tests the first call left open, the cost of a speculation round, c3 census,
# the level-2 poison failures in full
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r6b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dflash.py tests/test_gpu_prefill_switches.py tests/test_gpu_layer_options.py -m gpu -q --tb=short -x 2>&1 | grep -v "^E    +" | tail -40 > $O/pytest_dflash.log; tail -12 $O/pytest_dflash.log
timeout 600 python tools/spec_round_cost.py --out $O/spec_round_cost.json > $O/spec_round_cost.log 2>&1; tail -30 $O/spec_round_cost.log
timeout 700 python tools/parity_census.py --config c3 --variants 32 --steps 2 --budget-s 600 --out $O/parity_census_c3.json > $O/census_c3.log 2>&1; tail -2 $O/census_c3.log
UZU_HIP_POISON=2 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_tree_verify.py tests/test_gpu_layer_options.py -m gpu -q --tb=line -k "not scale and not census" 2>&1 | grep -v "^E    +" > $O/pytest_poison2_full.log; grep -E "^FAILED|passed|failed|core" $O/pytest_poison2_full.log | head -60
