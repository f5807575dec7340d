#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c16; mkdir -p $O
UZU_DEC_WIDE=0 UZU_HIP_LIB=$ROOT/uzu_amd/lib_tl/libuzu_hip.so timeout 400 python tools/timeline.py --model llama-3-8b --detail 9 > $O/timeline_llama_narrow.txt 2> $O/timeline_llama_narrow.err
UZU_HIP_LIB=$ROOT/uzu_amd/lib_tl/libuzu_hip.so timeout 400 python tools/timeline.py --detail 15 > $O/timeline_qwen.txt 2> $O/timeline_qwen.err
grep -A5 "^ *\(3\|4\|5\|160\) " $O/timeline_llama_narrow.txt | head -60; tail -8 $O/timeline_qwen.txt
