"""dl_conv_forward_add (conv_gemm_w4_kernel's store pass with the ADD option): out = conv(in) + addend, the way a ResnetBlock's first conv meets the
gradient that came down the skip connection (networks.py:509-513), against the CPU emulation's conv followed by an fp32 add; the engine-level switch
(DL_CONV_ADD=0 -> separate axpby) must leave a whole Resnet-9 backward within bf16 rounding of the fused form."""
import pytest
import torch

import fake_backend
from deepliif_amd import _lib as L
from deepliif_amd import ops
from deepliif_amd.engine import Precision
from deepliif_amd.geometry import ConvSpec, cpad

from test_gpu_kernels import DEV, _run_conv, hip, rel, rnd, sync, tol

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('direction', ['fwd', 'dgrad'])
@pytest.mark.parametrize('shape', [(8, 128, 128, 256, 256), (4, 128, 128, 128, 256)], ids=lambda s: 'n%d-%dx%d-ci%d-co%d' % s)
def test_conv_plus_addend_in_the_store_pass(shape, direction):
    N, H, W_, cin, cout = shape
    if direction == 'dgrad':
        if cin != cout:
            pytest.skip('the data gradient of a 128 -> 256 layer has 128 output channels: not a conv_gemm_w4_kernel shape (Co % 256)')
        cin, cout = cout, cin            # the data gradient of a cout -> cin layer: contracted channels = cout
    prec = Precision.get('bf16')
    spec = ConvSpec('conv', cin, cout, 3, 1, 1, L.PAD_ZERO, 0) if direction == 'fwd' else ConvSpec('conv', cout, cin, 3, 1, 1, L.PAD_ZERO, 0)
    w = rnd((spec.cout, spec.cin, 3, 3), 1, prec, 0.05)
    fake, real = fake_backend.FakeBackend(), hip()
    plan = spec.forward_plan() if direction == 'fwd' else spec.dgrad_plan()
    n_in = spec.cin if direction == 'fwd' else spec.cout
    n_out = spec.cout if direction == 'fwd' else spec.cin
    x = rnd((N, H, W_, n_in), 3, prec).to(prec.dtype)
    addend = rnd((N, H, W_, n_out), 5, prec).to(prec.dtype)
    exp_conv = _run_conv(fake, direction, spec, prec, x, w, None, L.ACT_NONE, L.ACT_NONE, H, W_)
    exp = (exp_conv.float() + addend.float()).to(prec.dtype)
    packed = ops.PackedWeights(plan, DEV, False)
    real.pack_weights(packed, w.to(DEV))
    out = torch.empty((N, H, W_, cpad(n_out)), dtype=prec.dtype, device=DEV)
    assert real.conv_forward_add(packed, x.to(DEV), addend.to(DEV), out, H, W_, prec.prec), 'the ResnetBlock shape is expected to have the fused form'
    sync()
    assert real.last_conv_kernel == 'conv_gemm_w4_kernel'
    assert rel(out, exp) < tol(prec)
    # in place: the addend IS the output buffer
    buf = addend.to(DEV).clone()
    assert real.conv_forward_add(packed, x.to(DEV), buf, buf, H, W_, prec.prec)
    sync()
    assert torch.equal(buf, out)
    # a layer the w4 kernel does not serve: no fused form, nothing launched, the buffer untouched
    spec2 = ConvSpec('conv', 64, 64, 3, 1, 1, L.PAD_ZERO, 0)
    p2 = ops.PackedWeights(spec2.forward_plan(), DEV, False)
    real.pack_weights(p2, rnd((64, 64, 3, 3), 7, prec, 0.05).to(DEV))
    x2 = rnd((1, 32, 32, 64), 8, prec).to(prec.dtype).to(DEV)
    o2 = torch.full((1, 32, 32, 64), 3.0, dtype=prec.dtype, device=DEV)
    assert not real.conv_forward_add(p2, x2, o2, o2, 32, 32, prec.prec)
    sync()
    assert float((o2.float() - 3.0).abs().max()) == 0.0


def test_resnet_backward_with_and_without_the_fused_add(monkeypatch):
    """a full-width Resnet-9 backward at 128-pixel rows, batch 4 (256 tiles: where the blocks take conv_gemm_w4_kernel): the fused add against the separate axpby -- one
    bf16 rounding instead of two per block, so the gradients agree to bf16 accuracy, not bit for bit"""
    from deepliif_amd import engine as E
    from deepliif_amd import networks as N
    from golden_util import seeded_uniform
    torch.manual_seed(3)
    net = N.define_G(3, 3, 64, 'resnet_9blocks', 'instance', False, 'normal', 0.02, [0], 'zero')
    net.train()
    x = seeded_uniform((4, 3, 512, 512), 52)
    r = torch.randn(4, 3, 512, 512, generator=torch.Generator().manual_seed(53))
    grads = {}
    for fused in (True, False):
        monkeypatch.setattr(ops, '_CONV_ADD', fused)
        ops._impl = None
        prec = E.Precision.get('bf16')
        tape = E.Tape()
        ctx = E.Ctx(prec, tape, training=True)
        xa = E.to_engine(x.to(DEV), prec)
        xa.needs_grad = True
        for p in net.parameters():
            p.grad = torch.zeros_like(p)
        ya = net.run(ctx, xa)
        ya.grad = E.to_engine(r.to(DEV), prec).t
        calls = {'add': 0}
        be = ops.impl()
        orig = be.conv_forward_add

        def counted(*a, **k):
            ok = orig(*a, **k)
            calls['add'] += bool(ok)
            return ok
        be.conv_forward_add = counted
        tape.backward()
        torch.cuda.synchronize()
        assert calls['add'] == (9 if fused else 0), calls       # one per ResnetBlock: its first conv's data gradient meets the skip gradient
        grads[fused] = (E.from_engine(E.Act(xa.grad, xa.C)).clone(), torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone())
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    assert l2(grads[True][0], grads[False][0]) < 2e-2 and l2(grads[True][1], grads[False][1]) < 2e-2
