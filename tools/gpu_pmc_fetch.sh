#!/bin/bash
# GPU box: the FETCH_SIZE pass behind profiles/<round>_pmc_fetch.json on its own (bench.py, short run; counters serialise the kernels)
R=${1:-r5}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/${R}_pmc
UZU_BENCH_NO_FLOOR_PROBE=1 timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/${R}_pmc -- python $ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline > $ROOT/gpurun_out/${R}_pmc.log 2>&1
echo "rc $?"
find $ROOT/gpurun_out/${R}_pmc -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
du -sh $ROOT/gpurun_out/${R}_pmc; tail -5 $ROOT/gpurun_out/${R}_pmc.log | cut -c1-300
