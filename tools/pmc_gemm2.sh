#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TCP|TCC|TD)_[A-Z0-9_]+" | sort -u | tr "\n" " " > $R/gpurun_out/tc_counters.txt
i=10
for set in "TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES GRBM_GUI_ACTIVE" \
           "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_REQ TCP_TCR_TCP_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  KB_GEMM=1 timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc$i -- $R/tools/kbench > /tmp/pmc$i.log 2>&1
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then tail -5 /tmp/pmc$i.log > $R/gpurun_out/pmc_gemm_$i.txt; continue; fi
  python3 - "$f" > $R/gpurun_out/pmc_gemm_$i.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:40], r["Grid_Size"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    if "mfma128" not in k[0]: continue
    print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
done
