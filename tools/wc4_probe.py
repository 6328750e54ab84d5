import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepliif_amd import _lib as L, ops
be = ops.impl()
for (n, h, w) in ((8, 512, 512), (1, 512, 512)):
    dy = torch.randn(n, h, w, 64, device='cuda').to(torch.bfloat16)
    x = torch.zeros(n, h, w, 8, device='cuda', dtype=torch.bfloat16); x[..., :3] = 1
    grad = torch.zeros(64, 3, 7, 7, device='cuda')
    f = lambda: be.conv_wgrad(dy, x, grad, 7, 1, 3, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, L.PREC_BF16, True)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    print((n, h, w), 'us/launch', round(s.elapsed_time(e) * 50, 1))
