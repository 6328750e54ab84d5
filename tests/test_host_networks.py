"""Host logic of the engine-backed networks (deepliif_amd/networks.py + engine.py) on CPU: the ops backend is replaced by
the formula emulation (tests/fake_backend.py), everything above it -- layer programs, tape autograd, UNet skip aliasing,
in-place concat routing, parameter binding / state_dict keys -- is the product code.  Compared with the oracle in fp32
(tolerance 2e-4 relative; both sides are fp32 CPU arithmetic in different summation orders)."""
import os

import numpy as np
import pytest
import torch

import fake_backend
from deepliif_amd import engine as E
from deepliif_amd import networks as N
from golden_util import seeded_uniform
from oracle import deepliif_oracle as O


@pytest.fixture(autouse=True)
def _fake():
    fake_backend.install()
    yield
    fake_backend.uninstall()


def rel(a, b, floor=1e-30):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(floor))


CASES = [
    ('resnet_9blocks', 3, 8, 'batch', 'zero', (2, 3, 32, 32)),
    ('resnet_2blocks', 3, 8, 'instance', 'zero', (2, 3, 32, 32)),
    ('unet_32', 3, 8, 'batch', 'zero', (2, 3, 32, 32)),
    ('unet_32', 9, 8, 'instance', 'zero', (1, 9, 32, 32)),
    ('unet_64', 3, 8, 'batch', 'zero', (1, 3, 64, 64)),
    ('n_layers', 6, 8, 'batch', 'zero', (2, 6, 64, 64)),
    ('n_layers', 12, 8, 'instance', 'zero', (1, 12, 64, 64)),
    ('n_layers', 6, 8, 'instance', 'zero', (3, 6, 100, 76)),       # 100 -> 50 -> 25 -> 12: an odd map in front of a stride-2 conv
    ('resnet_9blocks', 3, 8, 'instance', 'zero', (1, 3, 36, 52)),   # batch 1, H != W
    ('resnet_2blocks', 3, 8, 'instance', 'reflect', (2, 3, 20, 28)),  # ReflectionPad2d(3) stem / head on 20x28 -> 5x7 block maps
    ('resnet_9blocks', 3, 8, 'batch', 'reflect', (1, 3, 32, 32)),
    # CLI-exposed non-default options (cli.py:103, 176-179): --upsample resize_conv, --net-d pixel
    ('resnet_9blocks:resize_conv', 3, 8, 'batch', 'zero', (2, 3, 32, 32)),
    ('resnet_2blocks:resize_conv', 3, 8, 'instance', 'reflect', (1, 3, 20, 28)),
    ('pixel', 6, 8, 'batch', 'zero', (2, 6, 24, 40)),
    ('pixel', 6, 8, 'instance', 'zero', (1, 6, 16, 16)),
]


def build(arch, cin, nf, norm, pad):
    if arch in ('n_layers', 'pixel'):
        return N.define_D(cin, nf, arch, 4, norm, 'normal', 0.02, [])
    arch, _, ups = arch.partition(':')
    return N.define_G(cin, 3, nf, arch, norm, False, 'normal', 0.02, [], pad, ups or 'convtranspose')


@pytest.mark.parametrize('arch,cin,nf,norm,pad,shape', CASES)
def test_forward_backward_matches_oracle(arch, cin, nf, norm, pad, shape):
    sd = O.random_state_dict(arch, cin, 3, nf, norm, pad, 4, generator=torch.Generator().manual_seed(5))
    net = build(arch, cin, nf, norm, pad)
    net.load_state_dict(sd, strict=True)            # same keys / shapes as the reference
    net.train()
    x = seeded_uniform(shape, 6)
    prec = E.Precision.get('fp32')
    tape = E.Tape()
    ctx = E.Ctx(prec, tape, training=True)
    xa = E.to_engine(x, prec)
    xa.needs_grad = True
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    ya = net.run(ctx, xa)
    y = E.from_engine(ya)

    # oracle
    sdo = {k: v.clone() for k, v in sd.items()}
    params = {k: v.requires_grad_(True) for k, v in sdo.items() if v.is_floating_point() and 'running' not in k}
    xo = x.clone().requires_grad_(True)
    if arch in ('n_layers', 'pixel'):
        yo = O.run_discriminator(arch, sdo, xo, norm, 4, update_running=(norm == 'batch'))
    else:
        yo = O.run_generator(arch, sdo, xo, norm, pad, update_running=(norm == 'batch'))
    assert rel(y, yo.detach()) < 2e-4

    r = torch.randn(yo.shape, generator=torch.Generator().manual_seed(7))
    grads = torch.autograd.grad((yo * r).sum(), [xo] + list(params.values()))
    ya.grad = E.to_engine(r, prec).t
    tape.backward()
    dx = E.from_engine(E.Act(xa.grad, xa.C))
    assert rel(dx, grads[0]) < 2e-4
    gscale = max(float(g.abs().max()) for g in grads[1:])
    named = dict(net.named_parameters())
    for (k, _), g in zip(params.items(), grads[1:]):
        assert rel(named[k].grad, g, floor=0.05 * gscale) < 5e-4, k
    # BatchNorm running statistics follow nn.BatchNorm2d
    if norm == 'batch':
        for k, v in net.state_dict().items():
            if 'running' in k:
                assert rel(v, sdo[k].detach(), floor=1e-3) < 1e-4, k


def test_inference_seam_per_sample_norm():
    """net(x) with N>1 reproduces N single-tile forwards (the reference only ever runs one tile per forward)."""
    sd = O.random_state_dict('resnet_2blocks', 3, 3, 8, 'batch', 'zero', generator=torch.Generator().manual_seed(1))
    net = build('resnet_2blocks', 3, 8, 'batch', 'zero').set_precision('fp32')
    net.load_state_dict(sd)
    net.eval()
    x = seeded_uniform((3, 3, 16, 16), 2)
    y = net(x)
    for i in range(3):
        yo = O.resnet_generator(sd, x[i:i + 1], 'batch', 'zero', 2)
        assert rel(y[i:i + 1], yo) < 2e-4


def test_seeded_init_matches_reference_rng_order():
    """define_G/define_D under torch.manual_seed(0) must give the reference's weights (fixture: per-key sums)."""
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'seeded_init.npz'))
    cases = {'resnet_9blocks_batch': lambda: N.define_G(3, 3, 64, 'resnet_9blocks', 'batch', False, 'normal', 0.02, [], 'zero'),
             'resnet_9blocks_instance': lambda: N.define_G(3, 3, 64, 'resnet_9blocks', 'instance', False, 'normal', 0.02, [], 'zero'),
             'unet_512_batch': lambda: N.define_G(3, 3, 64, 'unet_512', 'batch', False, 'normal', 0.02, []),
             'n_layers_batch': lambda: N.define_D(6, 64, 'n_layers', 4, 'batch', 'normal', 0.02, [])}
    for tag, fn in cases.items():
        torch.manual_seed(0)
        sd = fn().state_dict()
        assert list(sd.keys()) == [str(k) for k in z[f'{tag}/keys']]
        assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in z[f'{tag}/shapes']]
        sums = np.array([v.double().sum().item() for v in sd.values()])
        asums = np.array([v.double().abs().sum().item() for v in sd.values()])
        assert np.allclose(sums, z[f'{tag}/sums'], rtol=1e-9, atol=1e-9), tag
        assert np.allclose(asums, z[f'{tag}/abs_sums'], rtol=1e-9, atol=1e-9), tag


def test_dropout_backward_reuses_the_forward_mask():
    """E.dropout: the gradient is masked/scaled with exactly the mask of the forward pass."""
    torch.manual_seed(3)
    prec = E.Precision.get('fp32')
    tape = E.Tape()
    ctx = E.Ctx(prec, tape, training=True)
    x = E.Act(torch.rand(1, 8, 8, 16) + 0.5, 16, True)
    y = E.dropout(ctx, x, 0.5)
    y.grad = torch.ones_like(y.t)
    tape.backward()
    mask = (y.t > 0).float()
    assert 0.3 < float(mask.mean()) < 0.7
    assert torch.allclose(y.t, x.t * mask * 2)
    assert torch.allclose(x.grad, mask * 2)


def test_networks_with_dropout_train_and_eval():
    """use_dropout=True keeps the reference's Sequential indices (state_dict keys) and is the identity in eval mode."""
    sd_ref_keys = list(O.random_state_dict('resnet_2blocks', 3, 3, 8, 'batch', 'zero').keys())
    net = N.define_G(3, 3, 8, 'resnet_2blocks', 'batch', True, 'normal', 0.02, [], 'zero').set_precision('fp32')
    keys = [k for k in net.state_dict().keys()]
    # with dropout the second conv of a block moves from index 3 to 4 (networks.py:493-506)
    assert 'model.10.conv_block.4.weight' in keys and 'model.10.conv_block.3.weight' not in keys
    assert len(keys) == len(sd_ref_keys)
    x = seeded_uniform((1, 3, 16, 16), 1)
    net.eval()
    a, b = net(x), net(x)
    assert torch.equal(a, b)
    net.train()
    c, d = net(x), net(x)
    assert not torch.equal(c, d)          # two different dropout draws


def test_narrow_transposed_conv_route_on_the_emulated_backend():
    """The inference-only route of UnetGenerator's outermost up-convolution (engine.conv: one 1x1 GEMM with a row per (ky, kx, co) +
    dl_convt4_gather) through the host code on the emulated backend, against torch's conv_transpose2d -- pins the derived weight image
    W'[(ky*4+kx)*Cout+co][ci] = W[ci][co][ky][kx] and the 2x2 gather formula on CPU; the GPU kernels are checked in test_gpu_kernels.py."""
    import fake_backend
    from deepliif_amd import _lib as L
    from deepliif_amd import ops
    from deepliif_amd.geometry import ConvSpec
    ops._impl = fake_backend.FakeBackend()
    try:
        g = torch.Generator().manual_seed(9)
        n, h, w, cin, cout = 2, 5, 7, 16, 3
        wt = (torch.randn(cin, cout, 4, 4, generator=g) * 0.1).to(torch.bfloat16).float()
        bias = torch.randn(cout, generator=g) * 0.1
        x = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16)
        layer = E.ConvLayer(ConvSpec('convT', cin, cout, 4, 2, 1), torch.nn.Parameter(wt.clone()), torch.nn.Parameter(bias.clone()))
        y = E.conv(E.Ctx(E.Precision.get('bf16'), None, training=False), E.Act(x, cin, False), layer, act=L.ACT_TANH, in_act=L.ACT_RELU)
        assert ops._impl.calls.get('convt4_gather', 0) == 1
        ref = torch.tanh(torch.nn.functional.conv_transpose2d(torch.relu(x.float()).permute(0, 3, 1, 2), wt, bias, stride=2, padding=1)).permute(0, 2, 3, 1)
        assert tuple(y.t.shape) == (n, 2 * h, 2 * w, 8)
        assert float((y.t[..., :cout].float() - ref).abs().max()) < 1e-2 and float(y.t[..., cout:].float().abs().max()) == 0.0
    finally:
        ops._impl = None


@pytest.mark.parametrize('tag', ['in9_n2'])
def test_attention_unet_on_the_emulated_backend(tag):
    """`unet_512_attention` (att_unet.py:117-199): module tree / state_dict keys of the reference, the layer program, the zero-copy
    [gated skip | up] concatenation, the gate and sigmoid backward, BatchNorm running-statistic updates -- against the reference-generated fixture
    (tests/golden/att_unet.npz).  Gradient tolerance: see att_util.GRAD_TOL."""
    import att_util
    cin, sd, x, r = att_util.case(tag)
    net = N.define_G(cin, 3, 64, 'unet_512_attention', 'batch', False, 'normal', 0.02, [])
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd, strict=True)
    net.set_precision('fp32').train()
    prec = E.Precision.get('fp32')
    tape = E.Tape()
    ctx = E.Ctx(prec, tape, training=True)
    xa = E.to_engine(x, prec)
    xa.needs_grad = True
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    ya = net.run(ctx, xa)
    y = E.from_engine(ya)
    ya.grad = E.to_engine(r, prec).t
    tape.backward()
    dx = E.from_engine(E.Act(xa.grad, xa.C))
    running = {k: v.clone() for k, v in net.state_dict().items() if 'running_' in k}
    net.eval()
    for m in net.modules():                      # the reference's inference toggle (util/__init__.py:743-755)
        if isinstance(m, torch.nn.BatchNorm2d):
            m.track_running_stats = False
    net.batched_per_sample_norm = False          # the fixture's eval forward is ONE reference forward over the batch of 2 (statistics over the batch)
    with torch.no_grad():
        y_eval = net(x)
    att_util.check_against_fixture(tag, y, dx, {k: p.grad for k, p in net.named_parameters()}, running, y_eval, 2e-4)
