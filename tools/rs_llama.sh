ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export UZU_HIP_LIB=$ROOT/uzu_amd/lib_lab/libuzu_hip.so
run() { env "$@" timeout 300 python tools/ab_prefill_bits.py --model $MODEL --prompt $PROMPT 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$MODEL $PROMPT $*', d['prefill_ms'], d['prefill_launches'])"; }
for rep in 1 2; do
for MODEL in llama-3-8b; do for PROMPT in 2043 4096; do
  run UZU_GEMM_TABLES=0
  run UZU_LAB_RS_MASK=0
  run UZU_LAB_RS_MASK=1
done; done; done
