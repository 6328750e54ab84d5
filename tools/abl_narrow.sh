# scratch driver for the probe of the moment (rewritten per experiment)
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout=300 2>&1 | tail -6
cp gpurun_out/parity_errors.json gpurun_out/parity_errors_r02k.json
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r02k.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/bench_r02k.json').read().strip().splitlines()[-1]); print('train', d['value'], d['ms_per_step'], d['strict_parity']['value'], d['roofline']['frac'])"
timeout 200 python bench.py --workload infer --steps 10 --warmup 2 > gpurun_out/bench_infer_r02k.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/bench_infer_r02k.json').read().strip().splitlines()[-1]); print('infer', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --workload wsi --steps 254 --warmup 2 > gpurun_out/bench_wsi_r02k.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/bench_wsi_r02k.json').read().strip().splitlines()[-1]); print('wsi', d['value'], d['ms_per_step'])"
