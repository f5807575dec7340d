#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c3; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $O/pytest.log
for v in lib lib_nopre; do
  UZU_HIP_LIB=$ROOT/uzu_amd/$v/libuzu_hip.so timeout 300 python bench.py --steps 128 --warmup 8 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
done
UZU_HIP_LIB=$ROOT/uzu_amd/lib_tl/libuzu_hip.so timeout 300 python tools/timeline.py > $O/timeline.txt 2> $O/timeline.err
UZU_HIP_LIB=$ROOT/uzu_amd/lib_tl/libuzu_hip.so timeout 400 python tools/timeline.py --model llama-3-8b > $O/timeline_llama.txt 2> $O/timeline_llama.err
timeout 400 python bench.py --model llama-3-8b --steps 32 --warmup 4 --no-cpu-baseline > $O/bench_llama.json 2> $O/bench_llama.err
KB_LLAMA=1 timeout 120 tools/kbench > $O/kbench_llama.txt 2>&1
tail -8 $O/pytest.log; grep -h '"value"' $O/bench_*.json | cut -c1-200
