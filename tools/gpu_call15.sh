#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c15; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -k "full_size or model_directory" 2>&1 | grep -v "^E    +" | tail -12 > $O/pytest.log
UZU_HIP_LIB=$ROOT/uzu_amd/lib_tl/libuzu_hip.so timeout 400 python tools/timeline.py --model llama-3-8b > $O/timeline_llama.txt 2> $O/timeline_llama.err
UZU_DEC_WIDE=0 UZU_HIP_LIB=$ROOT/uzu_amd/lib_tl/libuzu_hip.so timeout 400 python tools/timeline.py --model llama-3-8b > $O/timeline_llama_narrow.txt 2> $O/timeline_llama_narrow.err
bash tools/refresh_profiles.sh r2 > $O/refresh.log 2>&1
cd $ROOT
tail -6 $O/pytest.log; sed -n 6,24p $O/timeline_llama.txt; echo; sed -n 6,14p $O/timeline_llama_narrow.txt; tail -3 $O/refresh.log
