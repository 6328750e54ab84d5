"""probe: is there a fixed per-launch cost in the dominant conv kernel?  3x3 256->256 at 128x128, batch 4 / 8 / 16 (event-timed)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepliif_amd import _lib as L, ops
from deepliif_amd.geometry import ConvSpec
be = ops.impl()
spec = ConvSpec('conv', 256, 256, 3, 1, 1, L.PAD_ZERO)
w = torch.randn(256, 256, 3, 3, device='cuda') * 0.02
pf = ops.PackedWeights(spec.forward_plan(), 'cuda', False); be.pack_weights(pf, w)
for n in (4, 8, 16, 32):
    x = torch.randn(n, 128, 128, 256, device='cuda').to(torch.bfloat16)
    out = torch.empty_like(x)
    f = lambda: be.conv_forward(pf, x, out, 128, 128, None, L.ACT_NONE, L.ACT_NONE, L.PREC_BF16, want_stats=True)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) * 50
    print('batch', n, be.last_conv_kernel, 'us/launch', round(t, 1), 'TF/s', round(2.0 * n * 128 * 128 * 256 * 2304 / t / 1e6, 1))
