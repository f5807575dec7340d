ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for rep in 1 2 3; do for v in x=1 prep_fused=0; do
  UZU_HIP_TUNE=$v timeout 300 python tools/ab_prefill_bits.py --model qwen3.5-0.8b --prompt 2043 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['prefill_ms'], d['prefill_launches'], d['logits_sha256'], d['tokens'][:4])"
done; done
timeout 600 python -m pytest tests/test_gpu_prefill_switches.py tests/test_gpu_tree_verify.py tests/test_gpu_layer_options.py -m gpu -q --tb=short 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -k "chained_stream or deltanet or delta or batched or prefill or gemma or sliding or ring or sink" 2>&1 | tail -5
timeout 200 python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('c2', d['value'], d['prefill_tokens_per_s'], d['prefill_roofline']['frac'], d.get('parity'))" | cut -c1-600
