#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c2; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $O/pytest.log
timeout 120 tools/lat_lab > $O/lat_lab.txt 2>&1
timeout 120 tools/lat_lab_preload > $O/lat_lab_preload.txt 2>&1
timeout 300 python bench.py --steps 128 --warmup 8 --no-cpu-baseline > $O/bench_lib.json 2> $O/bench_lib.err
UZU_HIP_LIB=$ROOT/uzu_amd/lib_tl/libuzu_hip.so timeout 300 python tools/timeline.py > $O/timeline.txt 2> $O/timeline.err
UZU_HIP_LIB=$ROOT/uzu_amd/lib_tl/libuzu_hip.so timeout 400 python tools/timeline.py --model llama-3-8b > $O/timeline_llama.txt 2> $O/timeline_llama.err
PMC_TAG=c2/pmc_gemv timeout 900 tools/pmc_gemv.sh > $O/pmc.log 2>&1
timeout 400 python bench.py --model llama-3-8b --steps 32 --warmup 4 --no-cpu-baseline > $O/bench_llama.json 2> $O/bench_llama.err
tail -5 $O/pytest.log; cat $O/lat_lab.txt; grep -h '"value"' $O/bench_*.json | cut -c1-200
