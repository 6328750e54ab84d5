"""Kernel micro-benchmarks on the GPU box (HIP events on torch's current stream). Writes gpurun_out/microbench.json."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepliif_amd import _lib as L
from deepliif_amd import ops
from deepliif_amd.engine import Precision
from deepliif_amd.geometry import ConvSpec, cpad

DEV = 'cuda'
be = ops.impl()


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def conv_case(name, kind, cin, cout, k, s, p, N, H, W, precname, op=0):
    prec = Precision.get(precname)
    spec = ConvSpec(kind, cin, cout, k, s, p, L.PAD_ZERO, op)
    ho, wo = spec.out_hw(H, W)
    wshape = (cout, cin, k, k) if kind == 'conv' else (cin, cout, k, k)
    w = (torch.randn(wshape, device=DEV) * 0.02)
    x = torch.randn(N, H, W, cpad(cin), device=DEV).to(prec.dtype)
    dy = torch.randn(N, ho, wo, cpad(cout), device=DEV).to(prec.dtype)
    pf = ops.PackedWeights(spec.forward_plan(), DEV, prec.prec == 3); be.pack_weights(pf, w)
    pd = ops.PackedWeights(spec.dgrad_plan(), DEV, prec.prec == 3); be.pack_weights(pd, w)
    out = torch.empty(N, ho, wo, cpad(cout), device=DEV, dtype=prec.dtype)
    dx = torch.empty_like(x)
    grad = torch.zeros(wshape, device=DEV)
    hq, wq = (ho, wo) if kind == 'conv' else (H, W)
    dq = (ho, wo) if (kind == 'conv' and s == 2) else (H, W)
    flops = 2.0 * N * ho * wo * cout * cin * k * k if kind == 'conv' else 2.0 * N * H * W * cout * cin * k * k
    res = {}
    t = timeit(lambda: be.conv_forward(pf, x, out, hq, wq, None, 0, 0, prec.prec)); res['fwd_ms'] = t * 1e3; res['fwd_tflops'] = flops / t / 1e12
    t = timeit(lambda: be.conv_forward(pd, dy, dx, dq[0], dq[1], None, 0, 0, prec.prec)); res['dgrad_ms'] = t * 1e3; res['dgrad_tflops'] = flops / t / 1e12
    if kind == 'conv':
        f = lambda: be.conv_wgrad(dy, x, grad, k, s, p, 0, 0, 0, prec.prec, False)
    else:
        f = lambda: be.conv_wgrad(x, dy, grad, k, s, p, 0, 0, 0, prec.prec, False)
    t = timeit(f); res['wgrad_ms'] = t * 1e3; res['wgrad_tflops'] = flops / t / 1e12
    t = timeit(lambda: be.pack_weights(pf, w)); res['pack_ms'] = t * 1e3
    print(name, precname, json.dumps(res), flush=True)
    return res


def norm_case(N, H, W, C, precname):
    prec = Precision.get(precname)
    y = torch.randn(N, H, W, C, device=DEV).to(prec.dtype)
    z = torch.empty_like(y)
    dz = torch.randn_like(y)
    dy = torch.empty_like(y)
    st = [None]
    def f():
        st[0] = be.norm_forward(y, z, C, L.NORM_INSTANCE, L.ACT_RELU, None, None, None, None, -1.0, None)
    t = timeit(f)
    nbytes = y.numel() * y.element_size()
    res = {'fwd_ms': t * 1e3, 'fwd_GBs_alg': 3 * nbytes / t / 1e9}
    t = timeit(lambda: be.norm_backward(dz, y, dy, st[0], C, L.NORM_INSTANCE, L.ACT_RELU, None, None, None))
    res.update({'bwd_ms': t * 1e3, 'bwd_GBs_alg': 5 * nbytes / t / 1e9})
    print('norm', (N, H, W, C), precname, json.dumps(res), flush=True)
    return res


if __name__ == '__main__':
    out = {}
    for pn in ('bf16', 'fp32'):
        out[f'res3x3_256_{pn}'] = conv_case('res3x3_256@128', 'conv', 256, 256, 3, 1, 1, 8, 128, 128, pn)
        out[f'down3x3_64_128_{pn}'] = conv_case('down3x3 64->128@512', 'conv', 64, 128, 3, 2, 1, 8, 512, 512, pn)
        out[f'stem7x7_{pn}'] = conv_case('stem7x7 3->64@512', 'conv', 3, 64, 7, 1, 3, 8, 512, 512, pn)
        out[f'head7x7_{pn}'] = conv_case('head7x7 64->3@512', 'conv', 64, 3, 7, 1, 3, 8, 512, 512, pn)
        out[f'up3x3_256_128_{pn}'] = conv_case('convT 256->128@128', 'convT', 256, 128, 3, 2, 1, 8, 128, 128, pn, op=1)
        out[f'd4x4_512_{pn}'] = conv_case('D 512->512 k4s1@32', 'conv', 512, 512, 4, 1, 1, 8, 32, 32, pn)
        out[f'unet_inner_{pn}'] = conv_case('unet 512->512 k4s2@4', 'conv', 512, 512, 4, 2, 1, 8, 4, 4, pn)
        out[f'norm256_{pn}'] = norm_case(8, 128, 128, 256, pn)
        out[f'norm64_{pn}'] = norm_case(8, 512, 512, 64, pn)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/microbench.json', 'w'), indent=1)
