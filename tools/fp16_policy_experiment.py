"""VERDICT r4 #7, the "one cheap experiment": would a single-pass FP16 storage + product policy (same MFMA rate as bf16, 3 more mantissa bits) land nearer the
1e-3 bar than the bf16 policy?  CPU only: the oracle's resnet_9blocks at full size (ngf 64, 1 x 3 x 512 x 512, seeded N(0, 0.02) weights) with every convolution's
operands and result rounded to the 16-bit format (products of two 16-bit values are exact in fp32 and the sums run in fp32, like the MFMA), against the fp32 oracle.
The bf16 row calibrates the emulation against the GPU measurement (profiles/parity_errors_r04.json: 8.1e-2 on this network).
  python tools/fp16_policy_experiment.py  ->  profiles/r05/fp16_policy_experiment.json"""
import json, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from oracle import deepliif_oracle as O
from golden_util import seeded_uniform


class sixteen_bit:
    """every F.conv2d / F.conv_transpose2d: operands and result through `dtype` (storage + product rounding of a 16-bit policy; accumulation stays fp32)"""
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        self.c, self.ct = F.conv2d, F.conv_transpose2d
        r = lambda t: t.to(self.dtype).float() if torch.is_tensor(t) and t.is_floating_point() else t
        F.conv2d = lambda x, w, b=None, *a, **k: r(self.c(r(x), r(w), b, *a, **k))
        F.conv_transpose2d = lambda x, w, b=None, *a, **k: r(self.ct(r(x), r(w), b, *a, **k))

    def __exit__(self, *exc):
        F.conv2d, F.conv_transpose2d = self.c, self.ct


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


torch.set_num_threads(min(32, os.cpu_count() or 1))
out = {}
for arch, norm in (('resnet_9blocks', 'instance'), ('resnet_9blocks', 'batch'), ('unet_512', 'batch')):
    sd = O.random_state_dict(arch, 3, 3, 64, norm, 'zero', 4, generator=torch.Generator().manual_seed(21))
    x = seeded_uniform((1, 3, 512, 512), 22)
    with torch.no_grad():
        ref = O.run_generator(arch, {k: v.clone() for k, v in sd.items()}, x.clone(), norm, 'zero')
        for name, dt in (('bf16', torch.bfloat16), ('fp16', torch.float16)):
            with sixteen_bit(dt):
                y = O.run_generator(arch, {k: v.clone() for k, v in sd.items()}, x.clone(), norm, 'zero')
            out[f'forward/{arch}-{norm}/{name}'] = rel(y, ref)
            print(arch, norm, name, out[f'forward/{arch}-{norm}/{name}'], flush=True)
# backward: the magnitudes an fp16 policy would have to STORE -- the gradient arriving at each conv output of the generator for a unit-scale loss
sd = O.random_state_dict('resnet_9blocks', 3, 3, 64, 'instance', 'zero', 4, generator=torch.Generator().manual_seed(21))
x = seeded_uniform((1, 3, 512, 512), 22)
grads = []
orig = F.conv2d
def hook_conv(xx, w, b=None, *a, **k):
    y = orig(xx, w, b, *a, **k)
    if y.requires_grad:
        y.register_hook(lambda g: grads.append((tuple(g.shape), float(g.abs().max()), float((g.abs() < 6.1e-5).float().mean()), float((g.abs() < 6e-8).float().mean()))))
    return y
F.conv2d = hook_conv
sdo = {k: v.clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd.items()}
y = O.run_generator('resnet_9blocks', sdo, x.clone(), 'instance', 'zero')
# the step's own generator loss scale: 100 * SmoothL1(mean over 3 x 512 x 512) / 6 networks
loss = (100.0 / 6.0) * F.smooth_l1_loss(y, seeded_uniform((1, 3, 512, 512), 23))
loss.backward()
F.conv2d = orig
out['backward/dy_of_conv_outputs'] = {'n_tensors': len(grads), 'max_abs_range': [min(g[1] for g in grads), max(g[1] for g in grads)],
                                      'fraction_below_fp16_min_normal_6.1e-5': [min(g[2] for g in grads), max(g[2] for g in grads)],
                                      'fraction_below_fp16_min_subnormal_6e-8': [min(g[3] for g in grads), max(g[3] for g in grads)]}
print(out['backward/dy_of_conv_outputs'])
os.makedirs(os.path.join(ROOT, 'profiles', 'r05'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'profiles', 'r05', 'fp16_policy_experiment.json'), 'w'), indent=1, sort_keys=True)
