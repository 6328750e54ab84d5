"""Weight gradient of the ResnetBlock conv shape (3x3 256->256 @ 8x128x128) in isolation: one launch per layer (kernel + slab reduction) against the batched
launch of 18 layers (ops.HipBackend: dl_conv_wgrad_multi + one batched reduction).

  python tools/wgrad_time.py [precision=bf16] [layers=18]
The library's switches (DL_NO_WGRAD_W4, DL_WGRAD_TR_ASM, ...) are read once by the C side: every variant is its own process (tools/gpu_r05_wgrad.sh)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepliif_amd import _lib as L, ops
from deepliif_amd.engine import Precision
be = ops.impl()
prec = Precision.get(sys.argv[1] if len(sys.argv) > 1 else 'bf16')
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 18
DEV = 'cuda'
SPLIT = os.environ.get('TIME_SPLIT') == '1' and prec.prec == L.PREC_BF16X3


def _split_copy(t):
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    g = torch.stack([hi.reshape(*t.shape[:3], -1, 8), lo.reshape(*t.shape[:3], -1, 8)], dim=4)
    return g.contiguous().view(torch.int16).reshape(*t.shape[:3], -1).view(torch.float32).reshape(t.shape)


xs = [torch.randn(8, 128, 128, 256, device=DEV).to(prec.dtype) for _ in range(NL)]
dys = [torch.randn(8, 128, 128, 256, device=DEV).to(prec.dtype) for _ in range(NL)]
if SPLIT:
    xs, dys = [_split_copy(t) for t in xs], [_split_copy(t) for t in dys]
KW = {'p_split': True, 'q_split': True} if SPLIT else {}
grads = [torch.zeros(256, 256, 3, 3, device=DEV) for _ in range(NL)]


def one_pass(batch):
    ops._WGRAD_BATCH = batch
    be.wgrad_defer_begin()
    for dy, x, g in zip(dys, xs, grads):
        be.conv_wgrad(dy, x, g, 3, 1, 1, L.PAD_ZERO, 0, 0, prec.prec, False, **KW)
    be.wgrad_defer_end()


def immediate():
    for dy, x, g in zip(dys, xs, grads):
        be.conv_wgrad(dy, x, g, 3, 1, 1, L.PAD_ZERO, 0, 0, prec.prec, False, **KW)


def timeit(fn, iters=6, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3 / NL


gf = 2 * 8 * 128 * 128 * 256 * 2304 / 1e9
res = {'precision': prec.name, 'layers': NL, 'split_copies': SPLIT,
       'env': {k: v for k, v in os.environ.items() if k.startswith('DL_')}}
ONLY = os.environ.get('TIME_ONLY')          # e.g. batched: a PMC pass over the batched launch alone
for name, fn in [v for v in (('immediate_us_per_layer', immediate), ('deferred_reduce_us_per_layer', lambda: one_pass(False)), ('batched_us_per_layer', lambda: one_pass(True))) if not ONLY or v[0].startswith(ONLY)]:
    us = timeit(fn)
    res[name] = round(us, 1)
    res[name.replace('_us_per_layer', '_tf')] = round(gf / us * 1e3, 1)
if not ONLY:
    ref = [g.clone() for g in grads]
    one_pass(False)
    torch.cuda.synchronize()
    res['batched_vs_single_max_rel'] = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(ref, grads))
print(json.dumps(res))
