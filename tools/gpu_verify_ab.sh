#!/bin/bash
# GPU box: few-rows / tree-verify tests, then the verify cost table (tools/verify_cost.py)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/verify; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tree_verify.py tests/test_gpu_model.py -m gpu -q -x -k "few_rows or tree or verify or speculat or rows" --tb=short 2>&1 | tail -6 > $O/pytest.log; tail -3 $O/pytest.log
timeout 300 python tools/verify_cost.py > $O/verify_cost.json 2> $O/verify_cost.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/verify/verify_cost.json"))
print("decode", d["decode_us_per_token"])
for v in d["verify"]: print(v["nodes"], v["verify_us"], v["verify_gpu_us"], v["launches"], v["break_even_accepted_tokens"])
PY
