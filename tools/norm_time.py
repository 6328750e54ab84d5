"""Time the norm streaming kernels (+ axpby as the in-situ reference) on one ResnetBlock tensor with events; prints us per call.
The tensors are cycled over a 2 GB pool so that nothing is served from the 256 MB Infinity Cache (as in the training step)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepliif_amd import _lib as L, ops
from deepliif_amd.engine import Precision
be = ops.impl(); prec = Precision.get('bf16'); DEV = 'cuda'
shape = (8, 128, 128, 256); NP = 10
ys = [torch.randn(shape, device=DEV).to(prec.dtype) for _ in range(NP)]
dzs = [torch.randn(shape, device=DEV).to(prec.dtype) for _ in range(NP)]
outs = [torch.empty(shape, dtype=prec.dtype, device=DEV) for _ in range(NP)]
g, b, cs = torch.ones(256, device=DEV), torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)
st = be.norm_forward(ys[0], outs[0], 256, L.NORM_BATCH, L.ACT_RELU, g, b, None, None, -1.0, None)
def timeit(fn, n=40):
    for i in range(5): fn(i % NP)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i % NP)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
fwd = timeit(lambda i: be.norm_forward(ys[i], outs[i], 256, L.NORM_BATCH, L.ACT_RELU, g, b, None, None, -1.0, None))
bwd = timeit(lambda i: be.norm_backward(dzs[i], ys[i], outs[i], st, 256, L.NORM_BATCH, L.ACT_RELU, g, torch.zeros(256, device=DEV), torch.zeros(256, device=DEV), cs))
axp = timeit(lambda i: be.axpby(1.0, outs[i], 1.0, dzs[i], outs[i]))
print(f"grid_mul={os.environ.get('DL_NORM_GRID_MUL','1')}: norm_forward(all kernels) {fwd:.1f} us  norm_backward(all kernels) {bwd:.1f} us  axpby {axp:.1f} us")
