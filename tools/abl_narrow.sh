# scratch driver for the probe of the moment (rewritten per experiment)
timeout 300 python -m pytest tests/test_gpu_networks.py -q -m gpu -k "teacher_forced" 2>&1 | tail -4
python - <<PY
import json
d = json.load(open('gpurun_out/parity_errors.json'))
for k, v in d.items():
    if k.startswith('teacher_forced/n_layers') and 'fp32' in k: print(k, '%.2e' % v)
PY
