"""Per-step wall times of the benched training step (sync after every step) + Python GC and caching-allocator activity per step:
looks for periodic slow steps.   python tools/step_jitter.py [precision=bf16] [steps=40] [gc=on|off|freeze]"""
import gc, os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deepliif_amd import models as M

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
mode = sys.argv[3] if len(sys.argv) > 3 else 'on'
a = argparse.Namespace(batch=8, size=512, ngf=64, norm='instance', precision=prec)
torch.manual_seed(0)
opt = bench.make_opt(a, 0)
so = sys.stdout
sys.stdout = open(os.devnull, 'w')
model = M.create_model(opt)
model.setup(opt)
sys.stdout = so
dev = torch.device('cuda', 0)
def synth(seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(8, 3, 512, 512, generator=g) * 2 - 1).to(dev)
batch = {'A': synth(1234), 'B': [synth(1235 + i) for i in range(5)], 'A_paths': ['synthetic']}
for _ in range(3):
    model.set_input(batch); model.optimize_parameters()
torch.cuda.synchronize()
if mode == 'off':
    gc.disable()
elif mode == 'freeze':
    gc.collect(); gc.freeze()
rows = []
for i in range(steps):
    g0 = [s['collections'] for s in gc.get_stats()]
    m0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    model.set_input(batch); model.optimize_parameters()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    g1 = [s['collections'] for s in gc.get_stats()]
    m1 = torch.cuda.memory_stats()
    rows.append(((t2 - t0) * 1e3, (t1 - t0) * 1e3, [b - a_ for a_, b in zip(g0, g1)], m1['num_device_alloc'] - m0['num_device_alloc'], m1['num_device_free'] - m0['num_device_free'],
                 m1['reserved_bytes.all.current'] / 2**30))
ts = sorted(r[0] for r in rows)
print(f'{prec} gc={mode}: median {ts[len(ts)//2]:.2f} ms, mean {sum(ts)/len(ts):.2f}, min {ts[0]:.2f}, max {ts[-1]:.2f}')
for i, r in enumerate(rows):
    print(f'step {i:2d}: {r[0]:7.2f} ms (host issue {r[1]:7.2f})  gc {r[2]}  device alloc/free {r[3]}/{r[4]}  reserved {r[5]:.1f} GiB')
