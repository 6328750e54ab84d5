"""Sample sclk / power (rocm-smi) while one kernel of the training step runs back to back for a few seconds.
usage: clock_probe.py [fwd|wgrad|norm] [seconds]"""
import os, sys, time, threading, subprocess, re, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepliif_amd import _lib as L, ops
from deepliif_amd.engine import Precision
from deepliif_amd.geometry import ConvSpec
be = ops.impl(); prec = Precision.get('bf16'); DEV = 'cuda'
which = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
spec = ConvSpec('conv', 256, 256, 3, 1, 1)
w = torch.randn(256, 256, 3, 3, device=DEV) * 0.02
x = torch.randn(8, 128, 128, 256, device=DEV).to(prec.dtype)
out = torch.empty_like(x)
pf = ops.PackedWeights(spec.forward_plan(), DEV, False); be.pack_weights(pf, w)
grad = torch.zeros(256, 256, 3, 3, device=DEV)
samples, stop = [], False

def sampler():
    while not stop:
        try:
            o = subprocess.run(['rocm-smi', '-c', '-P', '--showmemuse'], capture_output=True, text=True, timeout=5).stdout
            sclk = re.search(r'sclk clock level.*?\((\d+)Mhz\)', o)
            pw = re.search(r'Power \(W\):\s*([\d.]+)', o) or re.search(r'Socket Power \(W\):\s*([\d.]+)', o)
            samples.append((time.time(), sclk.group(1) if sclk else '?', pw.group(1) if pw else '?'))
        except Exception as e:
            samples.append((time.time(), 'err', str(e)[:40]))
        time.sleep(0.2)

def run(n):
    for _ in range(n):
        if which == 'fwd':
            be.conv_forward(pf, x, out, 128, 128, None, 0, 0, prec.prec)
        elif which == 'wgrad':
            be.conv_wgrad(out, x, grad, 3, 1, 1, 0, 0, 0, prec.prec, False)
        else:
            be.norm_forward(x, out, 256, L.NORM_INSTANCE, L.ACT_RELU, None, None, None, None, -1.0, None)

run(20); torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); iters = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    run(200); iters += 200
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; th.join()
print(which, 'avg_us', e0.elapsed_time(e1) * 1e3 / iters, 'env', {k: v for k, v in os.environ.items() if k.startswith('DL_')})
print('samples (sclk MHz, W):', [(s[1], s[2]) for s in samples])
