"""conv_s2d_kernel against torch's conv2d on the GPU (debug aid): which output rows / pixels / channels differ, per shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepliif_amd import _lib as L, ops
from deepliif_amd.geometry import ConvSpec
be = ops.impl()
for (N, H, W, co) in [(2, 64, 256, 128), (4, 512, 256, 128), (8, 512, 512, 128), (8, 128, 512, 128), (16, 64, 512, 128)]:
    torch.manual_seed(0)
    spec = ConvSpec('conv', 64, co, 3, 2, 1, L.PAD_ZERO, 0)
    w = (torch.randn(co, 64, 3, 3, device='cuda') * 0.05).bfloat16().float()
    x = torch.randn(N, H, W, 64, device='cuda').bfloat16()
    plan = spec.forward_plan()
    packed = ops.PackedWeights(plan, 'cuda', False)
    be.pack_weights(packed, w)
    out = torch.empty(N, H // 2, W // 2, co, device='cuda', dtype=torch.bfloat16)
    be.conv_forward(packed, x, out, H // 2, W // 2, None, L.ACT_NONE, L.ACT_NONE, L.PREC_BF16, 1)
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w, stride=2, padding=1).permute(0, 2, 3, 1)
    err = (out.float() - ref).abs()
    bad = err > 0.05
    print((N, H, W, co), be.last_conv_kernel, 'max err', float(err.max()), 'bad frac', float(bad.float().mean()))
    if bad.any():
        print('  bad rows (ho) per image 0:', sorted(set(bad[0].any(dim=2).any(dim=1).nonzero().flatten().tolist()))[:40])
        print('  bad images:', bad.any(dim=3).any(dim=2).any(dim=1).nonzero().flatten().tolist())
        r = bad[0].any(dim=2).any(dim=1).nonzero().flatten()[0].item() if bad[0].any() else None
        if r is not None:
            print('  row', r, 'bad wo:', bad[0, r].any(dim=1).nonzero().flatten().tolist()[:20], '... count', int(bad[0, r].any(dim=1).sum()),
                  'bad co count', int(bad[0, r].any(dim=0).sum()))
