"""probe: ResnetGenerator stem forward (3 -> 64, 7x7) at 8x512x512 through dl_conv_forward, event-timed; for rocprofv3 kernel traces"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepliif_amd import _lib as L, ops
from deepliif_amd.engine import Precision
from deepliif_amd.geometry import ConvSpec
be = ops.impl()
spec = ConvSpec('conv', 3, 64, 7, 1, 3, L.PAD_ZERO)
w = torch.randn(64, 3, 7, 7, device='cuda') * 0.02
pf = ops.PackedWeights(spec.forward_plan(), 'cuda', False); be.pack_weights(pf, w)
b = torch.zeros(64, device='cuda')
x = torch.zeros(8, 512, 512, 8, device='cuda', dtype=torch.bfloat16); x[..., :3] = torch.randn(8, 512, 512, 3, device='cuda').to(torch.bfloat16)
out = torch.empty(8, 512, 512, 64, device='cuda', dtype=torch.bfloat16)
for stats in (False, True):
    f = lambda: be.conv_forward(pf, x, out, 512, 512, b, L.ACT_NONE, L.ACT_NONE, L.PREC_BF16, want_stats=stats)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); s.record()
    for _ in range(20): f()
    e.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
    print('stats', stats, be.last_conv_kernel, 'gpu us/launch', s.elapsed_time(e) * 50, 'host us/call', (t1 - t0) / 20 * 1e6)
