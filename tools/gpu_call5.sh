#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c5; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -q -k "not llama3_8b_full_depth" 2>&1 | tail -40) > $O/pytest.log
for v in lib_r1 lib lib_r1 lib; do
  UZU_HIP_LIB=$ROOT/uzu_amd/$v/libuzu_hip.so timeout 300 python bench.py --steps 128 --warmup 8 --no-cpu-baseline >> $O/bench_$v.json 2>> $O/bench_$v.err
done
for v in lib_r1 lib; do
  UZU_HIP_LIB=$ROOT/uzu_amd/$v/libuzu_hip.so timeout 400 python bench.py --model llama-3-8b --steps 32 --warmup 4 --no-cpu-baseline > $O/bench_llama_$v.json 2> $O/bench_llama_$v.err
done
timeout 900 python tools/seed_search.py --model qwen3.5-0.8b --prompt 2040 --steps 24 --sigma 0.6 --seeds $(seq 200 700) --quiet --min-distinct 12 --min-gap 0.4 > $O/seeds.txt 2> $O/seeds.err
tail -8 $O/pytest.log; grep -h '"value"' $O/bench_*.json | cut -c1-150; cat $O/seeds.txt | grep distinct
