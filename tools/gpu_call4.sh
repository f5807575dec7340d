#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c4; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q -k "not llama3_8b_full_depth" 2>&1 | tail -40) > $O/pytest.log
timeout 300 python bench.py --steps 128 --warmup 8 --no-cpu-baseline > $O/bench_lib.json 2> $O/bench_lib.err
UZU_HIP_LIB=$ROOT/uzu_amd/lib_tl/libuzu_hip.so timeout 300 python tools/timeline.py > $O/timeline.txt 2> $O/timeline.err
UZU_HIP_LIB=$ROOT/uzu_amd/lib_tl/libuzu_hip.so timeout 400 python tools/timeline.py --model llama-3-8b > $O/timeline_llama.txt 2> $O/timeline_llama.err
timeout 400 python bench.py --model llama-3-8b --steps 32 --warmup 4 --no-cpu-baseline > $O/bench_llama.json 2> $O/bench_llama.err
for d in 0 1; do echo "== UZU_DEC_DEFER=$d"; UZU_DEC_DEFER=$d KB_LLAMA=1 timeout 120 tools/kbench; done > $O/kbench_llama.txt 2>&1
timeout 600 python tools/seed_search.py --model qwen3.5-0.8b --prompt 2040 --steps 24 --sigma 0.6 --seeds $(seq 100 160) > $O/seeds.txt 2> $O/seeds.err
tail -8 $O/pytest.log; grep -h '"value"' $O/bench_*.json | cut -c1-200
