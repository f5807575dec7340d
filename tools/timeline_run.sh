#!/bin/bash
# GPU box (gpurun): per-workgroup timelines of one replayed decode step on the -DUZU_TIMELINE build (make OUT=../lib_tl
# HIPFLAGS="... -DUZU_TIMELINE" in uzu_amd/csrc): where a launch's time goes, and who finishes late (--detail).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
O=${1:-gpurun_out/timeline}; mkdir -p $O
L=$ROOT/uzu_amd/lib_tl/libuzu_hip.so
UZU_HIP_LIB=$L timeout 400 python tools/timeline.py --detail 15 > $O/timeline_qwen.txt 2> $O/timeline_qwen.err
UZU_HIP_LIB=$L timeout 400 python tools/timeline.py --model llama-3-8b --detail 9 > $O/timeline_llama.txt 2> $O/timeline_llama.err
UZU_DEC_WIDE=0 UZU_HIP_LIB=$L timeout 400 python tools/timeline.py --model llama-3-8b --detail 9 > $O/timeline_llama_narrow.txt 2> $O/timeline_llama_narrow.err
tail -3 $O/timeline_qwen.txt
