"""Run only the dominant conv kernel (ResnetBlock 3x3 256->256 @ 8x128x128, bf16) a few times: PMC target."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepliif_amd import _lib as L, ops
from deepliif_amd.engine import Precision
from deepliif_amd.geometry import ConvSpec
be = ops.impl(); prec = Precision.get('bf16'); DEV = 'cuda'
which = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
# other shapes of the step for the PMC passes over the conv_gemm_glds family: name -> (kind, cin, cout, k, stride, pad, out_pad, H)
SHAPES = {'down1': ('conv', 64, 128, 3, 2, 1, 0, 512), 'up2': ('convT', 128, 64, 3, 2, 1, 1, 256), 'dc2': ('conv', 64, 128, 4, 2, 1, 0, 256),
          'dc1': ('conv', 6, 64, 4, 2, 1, 0, 512), 'up1': ('convT', 256, 128, 3, 2, 1, 1, 128)}
if which in SHAPES:
    from deepliif_amd.geometry import cpad
    kind, cin, cout, k, st, pd, op, H = SHAPES[which]
    spec = ConvSpec(kind, cin, cout, k, st, pd, L.PAD_ZERO, op)
    ho, wo = spec.out_hw(H, H)
    w = torch.randn((cout, cin, k, k) if kind == 'conv' else (cin, cout, k, k), device=DEV) * 0.02
    x = torch.randn(8, H, H, cpad(cin), device=DEV).to(prec.dtype)
    out = torch.empty(8, ho, wo, cpad(cout), device=DEV, dtype=prec.dtype)
    pf = ops.PackedWeights(spec.forward_plan(), DEV, False); be.pack_weights(pf, w)
    hq, wq = (ho, wo) if kind == 'conv' else (H, H)
    for _ in range(5):
        be.conv_forward(pf, x, out, hq, wq, None, 0, 0, prec.prec)
    torch.cuda.synchronize()
    print(which, be.last_conv_kernel)
    sys.exit(0)
spec = ConvSpec('conv', 256, 256, 3, 1, 1)
w = torch.randn(256, 256, 3, 3, device=DEV) * 0.02
x = torch.randn(8, 128, 128, 256, device=DEV).to(prec.dtype)
out = torch.empty_like(x)
pf = ops.PackedWeights(spec.forward_plan(), DEV, False); be.pack_weights(pf, w)
grad = torch.zeros(256, 256, 3, 3, device=DEV)
for _ in range(5):
    if which == 'fwd':
        be.conv_forward(pf, x, out, 128, 128, None, 0, 0, prec.prec)
    else:
        be.conv_wgrad(out, x, grad, 3, 1, 1, 0, 0, 0, prec.prec, False)
torch.cuda.synchronize()
