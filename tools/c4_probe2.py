import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepliif_amd import _lib as L, ops
from deepliif_amd.geometry import ConvSpec
be = ops.impl()
spec = ConvSpec('conv', 3, 64, 7, 1, 3, L.PAD_ZERO)
w = torch.randn(64, 3, 7, 7, device='cuda') * 0.02
pf = ops.PackedWeights(spec.forward_plan(), 'cuda', False); be.pack_weights(pf, w)
b = torch.zeros(64, device='cuda')
for (n, h, wd) in ((8, 512, 512), (1, 512, 512), (1, 64, 64), (8, 128, 128), (2, 512, 512)):
    x = torch.zeros(n, h, wd, 8, device='cuda', dtype=torch.bfloat16)
    out = torch.empty(n, h, wd, 64, device='cuda', dtype=torch.bfloat16)
    f = lambda: be.conv_forward(pf, x, out, h, wd, b, L.ACT_NONE, L.ACT_NONE, L.PREC_BF16, want_stats=False)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    print((n, h, wd), be.last_conv_kernel, 'gpu us/launch', round(s.elapsed_time(e) * 50, 1))
