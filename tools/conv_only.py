"""Run only the dominant conv kernel (ResnetBlock 3x3 256->256 @ 8x128x128, bf16) a few times: PMC target."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepliif_amd import _lib as L, ops
from deepliif_amd.engine import Precision
from deepliif_amd.geometry import ConvSpec
be = ops.impl(); prec = Precision.get('bf16'); DEV = 'cuda'
which = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
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
