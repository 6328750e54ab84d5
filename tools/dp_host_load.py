"""Host issue time of the 5 G + 5 D step when N trainer processes share a few CPU cores (multi-GPU readiness without an 8-GPU node).

On an 8-GPU node the step is issued by 8 Python processes; round 5 measured 35 ms of host time per 76 ms step for ONE process with the whole host to itself.
This driver starts N independent processes of the REAL engine on the ONE GPU of a gpurun box, each pinned to `cores` CPU cores (sched_setaffinity), on tiles
small enough that the GPU work per step is a few ms (ngf 64, 64 x 64, batch 1: the same ~2 400 launches per step as the benched shape, the GPU is not the
limit), and reports per process the wall time per step -- which at this size IS the host issue time under contention (caches, memory bandwidth, the KFD
submission path shared by all processes).  No gradient exchange: the collectives are RCCL's threads, not Python's.

  python tools/dp_host_load.py [--procs 8] [--cores 2] [--steps 10]     ->  one JSON line per configuration, gpurun_out/dp_host_load.json
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, cores, steps, size):
    cpus = sorted(os.sched_getaffinity(0))
    mine = {cpus[(rank * cores + i) % len(cpus)] for i in range(cores)}
    os.sched_setaffinity(0, mine)
    import torch
    torch.set_num_threads(1)
    sys.path.insert(0, ROOT)
    import bench
    from deepliif_amd import models as M
    args = argparse.Namespace(ngf=64, norm='instance', precision='bf16', batch=1, size=size)
    torch.manual_seed(0)
    opt = bench.make_opt(args, 0)
    sys.stdout, saved = open(os.devnull, 'w'), sys.stdout
    model = M.create_model(opt)
    model.setup(opt)
    sys.stdout = saved
    g = torch.Generator().manual_seed(1)
    batch = {'A': (torch.rand(1, 3, size, size, generator=g) * 2 - 1).cuda(), 'B': [(torch.rand(1, 3, size, size, generator=g) * 2 - 1).cuda() for _ in range(5)],
             'A_paths': ['x']}
    for _ in range(3):
        model.set_input(batch)
        model.optimize_parameters()
    torch.cuda.synchronize()
    # rendezvous through the file system: every process starts its timed steps together
    open(f'/tmp/dp_host_load_ready_{rank}', 'w').close()
    while not os.path.exists('/tmp/dp_host_load_go'):
        time.sleep(0.01)
    t0 = time.perf_counter()
    for _ in range(steps):
        model.set_input(batch)
        model.optimize_parameters()
    host = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    print(json.dumps({'rank': rank, 'cpus': sorted(mine), 'host_issue_ms_per_step': round(host * 1e3, 2), 'wall_ms_per_step': round(wall * 1e3, 2)}), flush=True)


def run(procs, cores, steps, size):
    for f in os.listdir('/tmp'):
        if f.startswith('dp_host_load_'):
            os.remove(os.path.join('/tmp', f))
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--worker', str(r), '--cores', str(cores), '--steps', str(steps), '--size', str(size)],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for r in range(procs)]
    t0 = time.time()
    while sum(os.path.exists(f'/tmp/dp_host_load_ready_{r}') for r in range(procs)) < procs and time.time() - t0 < 600:
        time.sleep(0.05)
    open('/tmp/dp_host_load_go', 'w').close()
    rows = []
    for p in ps:
        out, _ = p.communicate(timeout=900)
        rows += [json.loads(l) for l in out.splitlines() if l.startswith('{')]
    host = [r['host_issue_ms_per_step'] for r in rows]
    wall = [r['wall_ms_per_step'] for r in rows]
    return {'procs': procs, 'cores_per_proc': cores, 'tile': size, 'steps': steps, 'finished': len(rows),
            'host_issue_ms_per_step': {'min': min(host), 'max': max(host), 'mean': round(sum(host) / len(host), 2)} if host else None,
            'wall_ms_per_step': {'min': min(wall), 'max': max(wall), 'mean': round(sum(wall) / len(wall), 2)} if wall else None}


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--worker', type=int, default=-1)
    ap.add_argument('--procs', type=int, default=8)
    ap.add_argument('--cores', type=int, default=2)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--size', type=int, default=64)
    a = ap.parse_args()
    if a.worker >= 0:
        worker(a.worker, a.cores, a.steps, a.size)
    else:
        res = [run(1, a.cores, a.steps, a.size), run(a.procs, a.cores, a.steps, a.size), run(a.procs, 1, a.steps, a.size)]
        os.makedirs('gpurun_out', exist_ok=True)
        json.dump(res, open('gpurun_out/dp_host_load.json', 'w'), indent=1)
        for r in res:
            print(json.dumps(r))
