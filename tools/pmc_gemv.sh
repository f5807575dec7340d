#!/bin/bash
# GPU box: counter passes over the decode GEMV at the Llama-3-8B shapes + the two read-outs (tools/kbench KB_LLAMA=1),
# one rocprofv3 run per counter set (kernel-trace only beside --pmc); per-kernel means land in gpurun_out/pmc_gemv_<i>.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" \
           "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_REQ" \
           "FETCH_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmcv$i
  KB_LLAMA=1 ${KB_ENV} timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmcv$i -- $R/tools/kbench > /tmp/pmcv$i.log 2>&1
  f=$(find /tmp/pmcv$i -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then tail -5 /tmp/pmcv$i.log > $R/gpurun_out/${PMC_TAG:-pmc_gemv}_$i.txt; continue; fi
  python3 - "$f" > $R/gpurun_out/${PMC_TAG:-pmc_gemv}_$i.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:64], r["Grid_Size"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    if "gemv" not in k[0]: continue
    print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()}, "launches", max(cnt[(k, c)] for c in d))
PY
done
