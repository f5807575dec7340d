#!/bin/bash
# Runs on the GPU box (gpurun): the three rocprofv3 passes behind profiles/<round>_*.  Raw output under gpurun_out/<round>_*;
# afterwards, locally:  python tools/summarize_profile.py <round> gpurun_out/<round>_trace gpurun_out/<round>_pmc gpurun_out/<round>_mfma
# (--pmc passes carry --kernel-trace only, never a sys/hip/hsa trace)
R=${1:-r1}
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 64 --warmup 8 --no-cpu-baseline"
rm -rf $ROOT/gpurun_out/${R}_trace $ROOT/gpurun_out/${R}_pmc $ROOT/gpurun_out/${R}_mfma
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_trace -- $CMD > $ROOT/gpurun_out/${R}_trace.log 2>&1
UZU_BENCH_NO_FLOOR_PROBE=1 timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/${R}_pmc -- $CMD > $ROOT/gpurun_out/${R}_pmc.log 2>&1
UZU_BENCH_NO_FLOOR_PROBE=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $ROOT/gpurun_out/${R}_mfma -- $CMD > $ROOT/gpurun_out/${R}_mfma.log 2>&1
# keep what is merged back small: the per-dispatch kernel trace of the PMC passes is not needed
find $ROOT/gpurun_out/${R}_pmc $ROOT/gpurun_out/${R}_mfma -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
du -sh $ROOT/gpurun_out/${R}_trace $ROOT/gpurun_out/${R}_pmc $ROOT/gpurun_out/${R}_mfma
