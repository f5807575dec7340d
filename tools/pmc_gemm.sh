#!/bin/bash
# SQ counter passes over the prefill GEMM micro-benchmark (tools/kbench KB_GEMM=1); summaries land in gpurun_out/pmc_gemm_*.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  KB_GEMM=1 KB_REPS=2 timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc$i -- $R/tools/kbench > /dev/null 2>&1
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" > $R/gpurun_out/pmc_gemm_$i.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:40], r["Grid_Size"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    if "gemm" not in k[0]: continue
    print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
done
