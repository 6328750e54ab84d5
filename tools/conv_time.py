"""Time the ResnetBlock conv shape (3x3 256->256 @ 8x128x128) in isolation: forward, data gradient, weight gradient.

  python tools/conv_time.py [precision=fp32] [which=fwd,dgrad,wgrad]
Environment switches of the library (DL_X3_VAR, DL_CONV_ABLATE, DL_NO_X3_GLDS, ...) are read by the C side at first use, so every variant is a
separate process (tools/gpu_r03_var.sh loops over them)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepliif_amd import _lib as L, ops
from deepliif_amd.engine import Precision
from deepliif_amd.geometry import ConvSpec
be = ops.impl()
prec = Precision.get(sys.argv[1] if len(sys.argv) > 1 else 'fp32')
which = (sys.argv[2] if len(sys.argv) > 2 else 'fwd,dgrad,wgrad').split(',')
DEV = 'cuda'
spec = ConvSpec('conv', 256, 256, 3, 1, 1)
data = os.environ.get('TIME_DATA', 'randn')       # randn | zero | bf16 (fp32 values that are exactly representable in bf16): DVFS / data-dependence probe
w = torch.randn(256, 256, 3, 3, device=DEV) * 0.02
x = torch.randn(8, 128, 128, 256, device=DEV).to(prec.dtype)
dy = torch.randn(8, 128, 128, 256, device=DEV).to(prec.dtype)
if data == 'halfzero':          # what a data gradient really reads behind a ReLU: about half of dL/dy is exactly zero (r05: the fwd / dgrad asymmetry)
    x = x * (torch.rand_like(x.float()) < 0.5).to(x.dtype)
    dy = dy * (torch.rand_like(dy.float()) < 0.5).to(dy.dtype)
if data == 'zero':
    w, x, dy = w * 0, x * 0, dy * 0
elif data == 'bf16':
    w, x, dy = w.bfloat16().float(), x.bfloat16().to(prec.dtype), dy.bfloat16().to(prec.dtype)
out = torch.empty_like(x)
SPLIT = os.environ.get('TIME_SPLIT') == '1' and prec.prec == L.PREC_BF16X3      # strict policy: feed the producer-written split copies (conv_gemm_w4x3_kernel's input)
if SPLIT:
    def _split_copy(t):
        hi = t.to(torch.bfloat16)
        lo = (t - hi.float()).to(torch.bfloat16)
        g = torch.stack([hi.reshape(*t.shape[:3], -1, 8), lo.reshape(*t.shape[:3], -1, 8)], dim=4)
        return g.contiguous().view(torch.int16).reshape(*t.shape[:3], -1).view(torch.float32).reshape(t.shape)
    x, dy = _split_copy(x), _split_copy(dy)
KW = {'in_split': True} if SPLIT else {}
pf = ops.PackedWeights(spec.forward_plan(), DEV, prec.prec == L.PREC_BF16X3); be.pack_weights(pf, w)
pd = ops.PackedWeights(spec.dgrad_plan(), DEV, prec.prec == L.PREC_BF16X3); be.pack_weights(pd, w)
grad = torch.zeros(256, 256, 3, 3, device=DEV)


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
    return s.elapsed_time(e) / iters * 1e3


res = {}
if 'fwd' in which:
    res['fwd_us'] = timeit(lambda: be.conv_forward(pf, x, out, 128, 128, None, 0, 0, prec.prec, **KW)); res['fwd_kernel'] = be.last_conv_kernel
if 'fwdstats' in which:          # the forward as the step launches it: bias + fused norm statistics in the store epilogue
    bias = torch.randn(256, device=DEV)
    if 'biasprobe' in which:         # is the cost of a bias its fetch or its VALUES?  (r05: 9-12 us per forward launch whatever the fetch)
        for tag, b in (('zero', torch.zeros(256, device=DEV)), ('small', bias * 1e-3), ('randn', bias), ('const', torch.full((256,), 0.5, device=DEV))):
            res[f'fwd_bias_{tag}_us'] = timeit(lambda: be.conv_forward(pf, x, out, 128, 128, b, 0, 0, prec.prec, **KW))
        res['fwd_again_us'] = timeit(lambda: be.conv_forward(pf, x, out, 128, 128, None, 0, 0, prec.prec, **KW))
    res['fwd_bias_us'] = timeit(lambda: be.conv_forward(pf, x, out, 128, 128, bias, 0, 0, prec.prec, **KW))
    res['fwd_bias_stats_us'] = timeit(lambda: be.conv_forward(pf, x, out, 128, 128, bias, 0, 0, prec.prec, want_stats=True, **KW))
    res['fwd_stats_us'] = timeit(lambda: be.conv_forward(pf, x, out, 128, 128, None, 0, 0, prec.prec, want_stats=True, **KW))
if 'dgrad' in which:
    res['dgrad_us'] = timeit(lambda: be.conv_forward(pd, dy, out, 128, 128, None, 0, 0, prec.prec, **KW))
if 'wgrad' in which:
    res['wgrad_us'] = timeit(lambda: be.conv_wgrad(dy, x, grad, 3, 1, 1, L.PAD_ZERO, 0, 0, prec.prec, False))
gf = 2 * 8 * 128 * 128 * 256 * 2304 / 1e9
print(prec.name, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in res.items()}, {k.replace('_us', '_tf'): round(gf / v * 1e3, 1) for k, v in res.items() if k.endswith('_us')},
      {k: v for k, v in os.environ.items() if k.startswith('DL_') or k == 'TIME_DATA'})
