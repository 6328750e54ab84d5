"""Split-K sweep of the small-M, large-K PatchGAN convs (D c5: 4x4 s1 512->512 @ 8x32x32 -> 31x31) on the 256x256-tile kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from layer_budget import L, ops, be, timeit, ConvSpec, cpad, Precision, DEV
prec = Precision.get('bf16')
for (name, cin, cout, k, s, p, H) in (('D c5', 512, 512, 4, 1, 1, 32), ('D c4', 256, 512, 4, 2, 1, 64), ('D c3', 128, 256, 4, 2, 1, 128)):
    spec = ConvSpec('conv', cin, cout, k, s, p, L.PAD_ZERO, 0)
    ho, wo = spec.out_hw(H, H)
    w = torch.randn(cout, cin, k, k, device=DEV) * 0.02
    b = torch.zeros(cout, device=DEV)
    x = torch.randn(8, H, H, cpad(cin), device=DEV).to(prec.dtype)
    pf = ops.PackedWeights(spec.forward_plan(), DEV, False); be.pack_weights(pf, w)
    out = torch.empty(8, ho, wo, cpad(cout), device=DEV, dtype=prec.dtype)
    ref = None
    for sk in (None, 1, 2, 4, 8):
        f = lambda: be.conv_forward(pf, x, out, ho, wo, b, L.ACT_NONE, L.ACT_NONE, prec.prec, splitk=sk)
        t = timeit(f)
        if ref is None: ref = out.float().clone()
        err = float((out.float() - ref).abs().max() / ref.abs().max())
        print(name, 'fwd splitk', sk, '%.1f us' % (t * 1e6), be.last_conv_kernel, 'rel diff vs auto %.2e' % err, flush=True)
