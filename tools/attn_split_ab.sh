#!/bin/bash
# GPU box, lab build: key splits / tasks per workgroup of the prefill attention on the 0.8B's 6 attention layers (8 q heads x 256, 2 kv heads), one 2043-token pass
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export UZU_HIP_LIB=$ROOT/uzu_amd/lib_lab/libuzu_hip.so
run() { env "$@" timeout 300 python tools/ab_prefill_bits.py --model qwen3.5-0.8b --prompt 2043 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['prefill_ms'], d['prefill_launches'], d['logits_sha256'])"; }
for rep in 1 2; do
  run X=1
  run UZU_ATTN_KSPLIT=3
  run UZU_ATTN_KSPLIT=4
  run UZU_ATTN_KSPLIT=6
  run UZU_ATTN_KSPLIT=8
  run UZU_ATTN_KSPLIT=4 UZU_ATTN_TPW=2
  run UZU_ATTN_KSPLIT=1
done
