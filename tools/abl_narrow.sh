timeout 900 python -m pytest tests/test_gpu_networks.py -q -m gpu -x -k "policy_variant or benched_configuration" 2>&1 | tail -8
python - <<PY
import json
d = json.load(open('gpurun_out/parity_errors.json'))
for k, v in d.items():
    if k.startswith('policy') or k.startswith('fullsize'): print(k, v)
PY
