"""Where does the HOST time of one training step go?  cProfile over a few eager steps of the benched model (5 G + 5 D, batch 8, 512 x 512), sorted by own time.
The GPU runs behind; the profile is of the issuing Python thread (ctypes calls included as built-in calls).
  python tools/host_profile.py [steps=5]  ->  gpurun_out/host_profile.txt"""
import argparse, cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from deepliif_amd import models as M
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
args = argparse.Namespace(ngf=64, norm='instance', precision='bf16', batch=8, size=512)
torch.manual_seed(0)
opt = bench.make_opt(args, 0)
sys.stdout, saved = open(os.devnull, 'w'), sys.stdout
model = M.create_model(opt)
model.setup(opt)
sys.stdout = saved
g = torch.Generator().manual_seed(1)
batch = {'A': (torch.rand(8, 3, 512, 512, generator=g) * 2 - 1).cuda(), 'B': [(torch.rand(8, 3, 512, 512, generator=g) * 2 - 1).cuda() for _ in range(5)], 'A_paths': ['x']}
for _ in range(3):
    model.set_input(batch); model.optimize_parameters()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    model.set_input(batch); model.optimize_parameters()
host = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    model.set_input(batch); model.optimize_parameters()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45)
os.makedirs('gpurun_out', exist_ok=True)
with open('gpurun_out/host_profile.txt', 'w') as f:
    f.write(f'host issue time per step (no profiler): {host * 1e3:.2f} ms; wall per step incl. the final synchronize: {wall * 1e3:.2f} ms; profile over {steps} steps follows\n')
    f.write(s.getvalue())
print(open('gpurun_out/host_profile.txt').read()[:6000])
