#!/bin/bash
cd /root/repo
python -m pytest tests/test_headline_parity_gpu.py -q --timeout=900 -k "dalle_example_shape_step" -s 2>&1 | grep -i "forced\|passed\|failed" | tail -5
python - <<'PY'
import json
s=json.load(open("/root/repo/gpurun_out/parity_dalle_example.json"))["steps"][0]
print({k:(round(v[0],4),v[1]) for k,v in s.items() if k.startswith("worst_grad")})
t=sorted(s["grad_rel_l2_vs_forced_fp32w_fa_oracle"].items(), key=lambda t:-t[1])[:6]
print("forced+fa:", [(k, round(v,4)) for k,v in t])
PY
