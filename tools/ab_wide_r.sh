O=gpurun_out/ab_wr; mkdir -p $O
for r in 0 2 0 2; do
  UZU_DEC_WIDE_R=$r timeout 300 python bench.py --model llama-3-8b --steps 48 --warmup 4 --no-cpu-baseline > $O/llama_r${r}_$RANDOM.json 2>/dev/null
  UZU_DEC_WIDE_R=$r timeout 600 python bench.py --config c5 --no-cpu-baseline > $O/c5_r${r}_$RANDOM.json 2>/dev/null
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob('gpurun_out/ab_wr/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d.get('kernel_us_per_step') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], {n.replace('gemv_dec', 'g'): round(v['us'] / v['calls'], 2) for n, v in k.items() if 'gemv' in n or 'attn' in n})
    except Exception as e:
        print(f, 'ERR', e)
PY
