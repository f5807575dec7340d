#!/bin/bash
# round 3, call 1: the whole GPU suite after the hygiene fixes, smoke, and the 2-rank --share-gpu dry run of the TP decode path
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
echo "pytest rc=$?"; tail -5 $O/pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --share-gpu --steps 32 --warmup 4 --no-cpu-baseline > $O/share2.json 2> $O/share2.err
echo "share2 rc=$?"; tail -c 1200 $O/share2.json; tail -5 $O/share2.err
