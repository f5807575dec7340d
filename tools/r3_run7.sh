#!/bin/bash
# round 3, call 7: HybridSpec completeness (QLoRA, RHT embeddings, Hadamard in norm / lookup), text front-end, TP regression
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r3h; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x -k "qlora or hadamard or hybrid or generate or rht or sliding or tp_sharded or abi" 2>&1 | tail -15 > $O/pytest.txt
echo "pytest rc=$?"; tail -12 $O/pytest.txt
