timeout 900 python -m pytest tests/test_gpu_networks.py -q -m gpu -k "teacher_forced" 2>&1 | tail -3
python - <<PY
import json
d = json.load(open('gpurun_out/parity_errors.json'))
for k, v in d.items():
    if k.startswith('teacher'): print(k, '%.2e' % v)
PY
