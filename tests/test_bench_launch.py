"""bench.py's launch path without GPUs: `python bench.py --gpus 2` (no torchrun around it) must re-execute itself as one process per
rank, rendezvous on 127.0.0.1 and print exactly ONE JSON line from rank 0 carrying the contract keys.  DL_BENCH_DRYRUN=1 routes the model
through the test emulation backend on CPU tensors and DL_BENCH_BACKEND=gloo replaces RCCL; the numbers are meaningless by construction."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_bench_self_launches_two_ranks_and_prints_one_json_line():
    env = dict(os.environ, DL_BENCH_DRYRUN='1', DL_BENCH_BACKEND='gloo', OMP_NUM_THREADS='2')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--size', '64', '--batch', '1',
                        '--ngf', '8', '--no-cpu-baseline', '--strict'], capture_output=True, text=True, timeout=800, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    # the gloo transport itself writes '[Gloo] Rank ...' fragments to stdout from C++, from both ranks at once (RCCL does not): look for the contract
    # line by its first key instead of trying to recognise everything that is NOT it
    lines = [l[l.index('{"metric"'):] for l in r.stdout.splitlines() if '{"metric"' in l]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
              'roofline', 'cpu_baseline', 'strict_parity'):
        assert k in d, k
    assert d['n_gpus'] == 2 and d['config']['rccl_ranks'] == 2 and d['config']['global_batch'] == 2 and d['scaling'] == 'weak'
    sp = d['strict_parity']
    assert sp['value'] > 0 and sp['ms_per_step'] > 0 and 0 < sp['headline_vs_strict']['generated_images_max_abs_over_max'] < 0.2
    # the 2-rank line explains itself (VERDICT r3 #8a): ranks counted by a real all-reduce, and the gradient exchange of the last step
    ex = d['exchange']
    assert set(ex['bytes']) == {'G', 'D'} and all(v > 0 for v in ex['bytes'].values()) and all(v >= 1 for v in ex['buckets'].values())
    assert ex['buckets']['G'] >= 6           # 5 generators, the first one (the pass ends with it) in two halves

