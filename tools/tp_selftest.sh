#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
O=gpurun_out/c25; mkdir -p $O
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dist --steps 32 --warmup 4 --no-cpu-baseline > $O/tp1.json 2> $O/tp1.err
echo "rc=$?"; tail -c 1500 $O/tp1.json; tail -5 $O/tp1.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --force-dist --no-p2p --steps 32 --warmup 4 --no-cpu-baseline > $O/tp1_rccl.json 2> $O/tp1_rccl.err
echo "rc=$?"; python -c "import json;d=json.loads(open('$O/tp1_rccl.json').read().strip().splitlines()[-1]);print(d['value'], d.get('tp'))"; tail -3 $O/tp1_rccl.err
timeout 300 python -m pytest tests/test_tp.py -m gpu -q --tb=short 2>&1 | tail -5
