#!/bin/bash
# round-2 GPU call 1: parity of the packed-dot GEMV, A/B against the round-1 build, decode timeline, seed search
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c1; mkdir -p $O
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $O/pytest.log
timeout 60 tools/dot2_probe > $O/dot2.txt 2>&1
timeout 900 tools/ab_gemv.sh > $O/ab_gemv.txt 2>&1
for v in lib_r1 lib lib_nont; do
  UZU_HIP_LIB=$ROOT/uzu_amd/$v/libuzu_hip.so timeout 300 python bench.py --steps 128 --warmup 8 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
done
UZU_HIP_LIB=$ROOT/uzu_amd/lib_tl/libuzu_hip.so timeout 300 python tools/timeline.py > $O/timeline.txt 2> $O/timeline.err
timeout 400 python bench.py --model llama-3-8b --steps 32 --warmup 4 --no-cpu-baseline > $O/bench_llama.json 2> $O/bench_llama.err
timeout 500 python tools/seed_search.py --model qwen3.5-0.8b --prompt 2040 --steps 24 --sigma 0.6 0.4 --seeds 42 43 44 45 > $O/seeds.txt 2> $O/seeds.err
tail -5 $O/pytest.log; cat $O/dot2.txt; grep -h '"value"' $O/bench_*.json | cut -c1-200
