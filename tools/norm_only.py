"""PMC target: the norm backward kernels and the axpby kernel on the SAME ResnetBlock tensors (8x128x128x256 bf16), a few launches each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepliif_amd import _lib as L, ops
from deepliif_amd.engine import Precision
be = ops.impl(); prec = Precision.get('bf16'); DEV = 'cuda'
shape = (8, 128, 128, 256)
y = torch.randn(shape, device=DEV).to(prec.dtype)
dz = torch.randn(shape, device=DEV).to(prec.dtype)
z = torch.empty_like(y); dy = torch.empty_like(y)
cs = torch.zeros(256, device=DEV)
for _ in range(5):
    st = be.norm_forward(y, z, 256, L.NORM_BATCH, L.ACT_RELU, torch.ones(256, device=DEV), torch.zeros(256, device=DEV), None, None, -1.0, None)
    be.norm_backward(dz, y, dy, st, 256, L.NORM_BATCH, L.ACT_RELU, torch.ones(256, device=DEV), torch.zeros(256, device=DEV), torch.zeros(256, device=DEV), cs)
    be.axpby(1.0, dy, 1.0, dz, dy)
torch.cuda.synchronize()
