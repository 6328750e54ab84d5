"""probe: dl_conv_narrow_forward launch time vs image height (HIP events), for the rocprofv3 kernel trace"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepliif_amd import _lib as L, ops
from deepliif_amd.engine import Precision
from deepliif_amd.geometry import ConvSpec
be = ops.impl()
prec = Precision.get('bf16')
spec = ConvSpec('conv', 64, 3, 7, 1, 3, L.PAD_ZERO)
w = torch.randn(3, 64, 7, 7, device='cuda') * 0.02
pf = ops.PackedWeights(spec.narrow_forward_plan(), 'cuda', False); be.pack_weights(pf, w)
b = torch.zeros(3, device='cuda')
for (n, h, wd) in ((8, 512, 512), (8, 128, 512), (8, 512, 128), (1, 512, 512), (8, 64, 64)):
    x = torch.randn(n, h, wd, 64, device='cuda').to(torch.bfloat16)
    out = torch.empty(n, h, wd, 8, device='cuda', dtype=torch.bfloat16)
    for _ in range(2):
        be.conv_narrow_forward(pf, x, out, 3, 7, 3, b, L.ACT_TANH)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        be.conv_narrow_forward(pf, x, out, 3, 7, 3, b, L.ACT_TANH)
    e.record(); torch.cuda.synchronize()
    print((n, h, wd), 'us per launch', s.elapsed_time(e) * 100)
