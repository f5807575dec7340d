#!/bin/bash
# GPU box: SQ / LDS / TA counter passes over tools/kbench KB_GEMM_AB (one 4096 x 14336 x 4096 GEMM, the 256-thread form and the ping-pong form);
# per-kernel means land in gpurun_out/gemm_ab/pmc_<i>.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/gemm_ab
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INST_LEVEL_VMEM" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  KB_GEMM_AB=1 KB_GEMM_AB_ONE=1 timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc$i -- $R/tools/kbench > /tmp/pmc$i.log 2>&1
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then tail -5 /tmp/pmc$i.log > $R/gpurun_out/gemm_ab/pmc_$i.txt; continue; fi
  python3 - "$f" > $R/gpurun_out/gemm_ab/pmc_$i.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:70], r["Grid_Size"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    if "mfma128" not in k[0]: continue
    print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
done
cat $R/gpurun_out/gemm_ab/pmc_*.txt
