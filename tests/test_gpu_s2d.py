"""GPU parity of conv_s2d_kernel (csrc/conv_s2d.hip): the stride-2 3x3 convolution over a 64-channel tensor with the weights held in
registers -- ResnetGenerator down1 forward (networks.py:400-404) and the data gradient of up2 (networks.py:425-436) -- called through
the C ABI (dl_conv_forward) and checked against the CPU formula emulation (tests/fake_backend.py), incl. the fused statistics.
Every case asserts that the dispatch really took the new kernel (dl_conv_kernel_name)."""
import ctypes as C

import pytest
import torch

import fake_backend
from deepliif_amd import _lib as L
from deepliif_amd import ops
from deepliif_amd.engine import Precision
from deepliif_amd.geometry import ConvSpec, cpad, fill_conv_desc

from test_gpu_kernels import DEV, DRY, _run_conv, hip, rel, rnd, sync, tol

pytestmark = pytest.mark.gpu


def _kernel_name(be, plan, x, out, hq, wq, prec):
    n, hi, wi, _ = x.shape
    _, ho, wo, cop = out.shape
    d = fill_conv_desc(plan, n, hi, wi, x.shape[3], ho, wo, cop, cop, hq, wq, L.DL_BF16, prec.prec, L.ACT_NONE, L.ACT_NONE, 0, 1)
    return L.load().dl_conv_kernel_name(C.byref(d)).decode()


S2D_CASES = [
    # kind, cin, cout, N, H, W of the layer INPUT, direction
    ('conv', 64, 128, 2, 64, 256, 'fwd'),          # one 128-pixel segment, 32 output rows
    ('conv', 64, 128, 1, 8, 512, 'fwd'),           # two segments, 4 output rows: strips of 2 rows (the halo row of every strip but the first is real data)
    ('conv', 64, 256, 3, 32, 256, 'fwd'),          # two channel tiles
    ('conv', 64, 128, 8, 512, 512, 'fwd'),         # down1 at the benched size: 256 workgroups x 16 rows
    ('convT', 128, 64, 2, 32, 128, 'dgrad'),       # up2 data gradient: dy 64 x 256 -> dx 32 x 128
    ('convT', 128, 64, 8, 256, 256, 'dgrad'),      # ... at the benched size
]


@pytest.mark.parametrize('case', S2D_CASES, ids=lambda c: f'{c[0]}{c[1]}-{c[2]}n{c[3]}h{c[4]}w{c[5]}{c[6]}')
def test_s2d_kernel_against_the_emulation(case):
    kind, cin, cout, N, H, W_, direction = case
    prec = Precision.get('bf16')
    spec = ConvSpec(kind, cin, cout, 3, 2, 1, L.PAD_ZERO, 1 if kind == 'convT' else 0)
    wshape = (cout, cin, 3, 3) if kind == 'conv' else (cin, cout, 3, 3)
    w = rnd(wshape, 1, prec, 0.05)
    fake, real = fake_backend.FakeBackend(), hip()
    if direction == 'fwd':
        bias = rnd((cout,), 2, Precision.get('fp32'), 0.1)
        x = rnd((N, H, W_, cin), 3, prec).to(prec.dtype)
        plan = spec.forward_plan()
        ho, wo = spec.out_hw(H, W_)
        probe = torch.empty((N, ho, wo, cpad(cout)), dtype=prec.dtype, device=DEV)
        assert _kernel_name(real, plan, x, probe, ho, wo, prec) == 'conv_s2d_kernel'
        for act in (L.ACT_NONE, L.ACT_RELU):
            exp = _run_conv(fake, 'fwd', spec, prec, x, w, bias, act, L.ACT_NONE, H, W_)
            first = None
            for rep in range(3):         # repeated: a staging race would show up as run-to-run differences
                got = _run_conv(real, 'fwd', spec, prec, x.to(DEV), w.to(DEV), bias.to(DEV), act, L.ACT_NONE, H, W_, splitk=1)
                assert DRY or real.last_conv_kernel.startswith('conv_s2d'), real.last_conv_kernel      # (split-K chosen by size would route small cases to the gather GEMM)
                sync()
                assert rel(got, exp) < tol(prec), ('fwd', act, rep)
                if first is None:
                    first = got.clone()
                else:
                    assert torch.equal(got, first), 'run-to-run difference'
        # without a bias (the data-gradient form of the same descriptor)
        exp = _run_conv(fake, 'fwd', spec, prec, x, w, None, L.ACT_NONE, L.ACT_NONE, H, W_)
        got = _run_conv(real, 'fwd', spec, prec, x.to(DEV), w.to(DEV), None, L.ACT_NONE, L.ACT_NONE, H, W_, splitk=1)
        assert DRY or real.last_conv_kernel.startswith('conv_s2d'), real.last_conv_kernel      # (split-K chosen by size would route small cases to the gather GEMM)
        sync()
        assert rel(got, exp) < tol(prec), 'no bias'
    else:
        ho, wo = spec.out_hw(H, W_)
        dy = rnd((N, ho, wo, cout), 4, prec).to(prec.dtype)
        plan = spec.dgrad_plan()
        probe = torch.empty((N, H, W_, cpad(cin)), dtype=prec.dtype, device=DEV)
        assert _kernel_name(real, plan, dy, probe, H, W_, prec) == 'conv_s2d_kernel'
        exp = _run_conv(fake, 'dgrad', spec, prec, dy, w, None, L.ACT_NONE, L.ACT_NONE, H, W_)
        first = None
        for rep in range(3):
            got = _run_conv(real, 'dgrad', spec, prec, dy.to(DEV), w.to(DEV), None, L.ACT_NONE, L.ACT_NONE, H, W_, splitk=1)
            assert DRY or real.last_conv_kernel.startswith('conv_s2d'), real.last_conv_kernel      # (split-K chosen by size would route small cases to the gather GEMM)
            sync()
            assert rel(got, exp) < tol(prec), ('dgrad', rep)
            if first is None:
                first = got.clone()
            else:
                assert torch.equal(got, first), 'run-to-run difference (dgrad)'


def test_s2d_kernel_writes_into_a_channel_slice_and_reads_from_one():
    """in_pstride / out_pstride wider than the channel count (UNet-style concat buffers)"""
    prec = Precision.get('bf16')
    spec = ConvSpec('conv', 64, 128, 3, 2, 1, L.PAD_ZERO, 0)
    N, H, W_ = 2, 16, 256
    w = rnd((128, 64, 3, 3), 1, prec, 0.05)
    bias = rnd((128,), 2, Precision.get('fp32'), 0.1)
    x = rnd((N, H, W_, 64), 3, prec).to(prec.dtype)
    fake, real = fake_backend.FakeBackend(), hip()
    exp = _run_conv(fake, 'fwd', spec, prec, x, w, bias, L.ACT_NONE, L.ACT_NONE, H, W_)
    wide_in = torch.full((N, H, W_, 128), 7.0, dtype=prec.dtype, device=DEV)
    wide_in[..., 64:] = x.to(DEV)
    wide_out = torch.full((N, H // 2, W_ // 2, 256), -3.0, dtype=prec.dtype, device=DEV)
    plan = spec.forward_plan()
    packed = ops.PackedWeights(plan, DEV, False)
    real.pack_weights(packed, w.to(DEV))
    real.conv_forward(packed, wide_in[..., 64:], wide_out[..., 128:], H // 2, W_ // 2, bias.to(DEV), L.ACT_NONE, L.ACT_NONE, prec.prec, 1)
    sync()
    assert DRY or real.last_conv_kernel.startswith('conv_s2d')
    assert rel(wide_out[..., 128:], exp) < tol(prec)
    assert float((wide_out[..., :128].float() + 3.0).abs().max()) == 0.0, 'the other half of the buffer must be untouched'


@pytest.mark.parametrize('scope', [L.NORM_INSTANCE, L.NORM_BATCH])
@pytest.mark.parametrize('shape', [(2, 64, 256), (8, 512, 512), (1, 8, 512)])
def test_s2d_fused_norm_statistics(shape, scope):
    """the statistics chunks of conv_s2d_kernel (one per workgroup: row segment x strip) feed dl_norm_forward exactly like the stand-alone pass"""
    N, H, W_ = shape
    prec = Precision.get('bf16')
    cin, cout = 64, 128
    spec = ConvSpec('conv', cin, cout, 3, 2, 1, L.PAD_ZERO, 0)
    w = rnd((cout, cin, 3, 3), 1, prec, 0.05).to(DEV)
    bias = rnd((cout,), 2, Precision.get('fp32'), 0.1).to(DEV)
    x = rnd((N, H, W_, cin), 3, prec).to(prec.dtype).to(DEV)
    be = hip()
    plan = spec.forward_plan()
    packed = ops.PackedWeights(plan, DEV, False)
    be.pack_weights(packed, w)
    ho, wo = spec.out_hw(H, W_)
    y = torch.empty((N, ho, wo, cout), dtype=prec.dtype, device=DEV)
    nch = be.conv_forward(packed, x, y, ho, wo, bias, L.ACT_NONE, L.ACT_NONE, prec.prec, splitk=1, want_stats=True)
    assert DRY or (be.last_conv_kernel.startswith('conv_s2d') and nch > 0)
    affine = scope == L.NORM_BATCH
    g = (1 + 0.1 * torch.randn(cout, generator=torch.Generator().manual_seed(4))).to(DEV) if affine else None
    b = (0.1 * torch.randn(cout, generator=torch.Generator().manual_seed(5))).to(DEV) if affine else None
    z1 = torch.empty_like(y)
    st1 = be.norm_forward(y, z1, cout, scope, L.ACT_RELU, g, b, None, None, -1.0, None, ext_nchunks=nch)
    st1 = [t.clone() for t in st1[:2]]
    z2 = torch.empty_like(y)
    st2 = be.norm_forward(y, z2, cout, scope, L.ACT_RELU, g, b, None, None, -1.0, None)
    sync()
    assert rel(st1[0], st2[0]) < 1e-5 and rel(st1[1], st2[1]) < 1e-5
    assert rel(z1, z2) <= 2.0 ** -7
    assert float((z1 != z2).float().mean()) < 1e-3


# ------------------------------------------------------------------------------------------------------------------------------------------
# conv_s2u_kernel (csrc/conv_s2u.hip): the mirror image -- ResnetGenerator up2 forward (ConvTranspose2d(128, 64, k3, s2, p1, op1)) and the data gradient of
# down1 (Conv2d(64, 128, k3, s2, p1)): four sub-pixel phases, weights in registers, input rows streamed once
# ------------------------------------------------------------------------------------------------------------------------------------------
S2U_CASES = [
    # kind, cin, cout, N, H, W of the layer INPUT, direction
    ('convT', 128, 64, 2, 16, 64, 'fwd'),          # one 64-pixel segment, 16 input rows
    ('convT', 128, 64, 1, 6, 128, 'fwd'),          # two segments (pixel 64 of segment 0 is real data, of segment 1 the padding), strips of 1-2 rows
    ('convT', 128, 128, 3, 8, 64, 'fwd'),          # two channel tiles
    ('convT', 128, 64, 8, 256, 256, 'fwd'),        # up2 at the benched size
    ('conv', 64, 128, 2, 32, 128, 'dgrad'),        # down1 data gradient: dy 16 x 64 -> dx 32 x 128
    ('conv', 64, 128, 8, 512, 512, 'dgrad'),       # ... at the benched size
]


@pytest.mark.parametrize('case', S2U_CASES, ids=lambda c: f'{c[0]}{c[1]}-{c[2]}n{c[3]}h{c[4]}w{c[5]}{c[6]}')
def test_s2u_kernel_against_the_emulation(case):
    kind, cin, cout, N, H, W_, direction = case
    prec = Precision.get('bf16')
    spec = ConvSpec(kind, cin, cout, 3, 2, 1, L.PAD_ZERO, 1 if kind == 'convT' else 0)
    wshape = (cout, cin, 3, 3) if kind == 'conv' else (cin, cout, 3, 3)
    w = rnd(wshape, 1, prec, 0.05)
    fake, real = fake_backend.FakeBackend(), hip()
    if direction == 'fwd':
        bias = rnd((cout,), 2, Precision.get('fp32'), 0.1)
        x = rnd((N, H, W_, cin), 3, prec).to(prec.dtype)
        ho, wo = spec.out_hw(H, W_)
        probe = torch.empty((N, ho, wo, cpad(cout)), dtype=prec.dtype, device=DEV)
        assert _kernel_name(real, spec.forward_plan(), x, probe, H, W_, prec) == 'conv_s2u_kernel'
        for act, b in ((L.ACT_NONE, bias), (L.ACT_RELU, bias), (L.ACT_NONE, None)):
            exp = _run_conv(fake, 'fwd', spec, prec, x, w, b, act, L.ACT_NONE, H, W_)
            first = None
            for rep in range(2):
                got = _run_conv(real, 'fwd', spec, prec, x.to(DEV), w.to(DEV), None if b is None else b.to(DEV), act, L.ACT_NONE, H, W_, splitk=1)
                assert DRY or real.last_conv_kernel.startswith('conv_s2u'), real.last_conv_kernel      # (split-K chosen by size would route small cases to the gather GEMM)
                sync()
                assert rel(got, exp) < tol(prec), ('fwd', act, rep)
                if first is None:
                    first = got.clone()
                else:
                    assert torch.equal(got, first), 'run-to-run difference'
    else:
        ho, wo = spec.out_hw(H, W_)
        dy = rnd((N, ho, wo, cout), 4, prec).to(prec.dtype)
        probe = torch.empty((N, H, W_, cpad(cin)), dtype=prec.dtype, device=DEV)
        assert _kernel_name(real, spec.dgrad_plan(), dy, probe, ho, wo, prec) == 'conv_s2u_kernel'
        exp = _run_conv(fake, 'dgrad', spec, prec, dy, w, None, L.ACT_NONE, L.ACT_NONE, H, W_)
        first = None
        for rep in range(2):
            got = _run_conv(real, 'dgrad', spec, prec, dy.to(DEV), w.to(DEV), None, L.ACT_NONE, L.ACT_NONE, H, W_, splitk=1)
            assert DRY or real.last_conv_kernel.startswith('conv_s2u'), real.last_conv_kernel      # (split-K chosen by size would route small cases to the gather GEMM)
            sync()
            assert rel(got, exp) < tol(prec), ('dgrad', rep)
            if first is None:
                first = got.clone()
            else:
                assert torch.equal(got, first), 'run-to-run difference (dgrad)'


@pytest.mark.parametrize('scope', [L.NORM_INSTANCE, L.NORM_BATCH])
@pytest.mark.parametrize('shape', [(2, 16, 64), (8, 256, 256), (1, 6, 128)])
def test_s2u_fused_norm_statistics(shape, scope):
    N, H, W_ = shape
    prec = Precision.get('bf16')
    cin, cout = 128, 64
    spec = ConvSpec('convT', cin, cout, 3, 2, 1, L.PAD_ZERO, 1)
    w = rnd((cin, cout, 3, 3), 1, prec, 0.05).to(DEV)
    bias = rnd((cout,), 2, Precision.get('fp32'), 0.1).to(DEV)
    x = rnd((N, H, W_, cin), 3, prec).to(prec.dtype).to(DEV)
    be = hip()
    packed = ops.PackedWeights(spec.forward_plan(), DEV, False)
    be.pack_weights(packed, w)
    ho, wo = spec.out_hw(H, W_)
    y = torch.empty((N, ho, wo, cout), dtype=prec.dtype, device=DEV)
    nch = be.conv_forward(packed, x, y, H, W_, bias, L.ACT_NONE, L.ACT_NONE, prec.prec, splitk=1, want_stats=True)
    assert DRY or (be.last_conv_kernel.startswith('conv_s2u') and nch > 0)
    affine = scope == L.NORM_BATCH
    g = (1 + 0.1 * torch.randn(cout, generator=torch.Generator().manual_seed(4))).to(DEV) if affine else None
    b = (0.1 * torch.randn(cout, generator=torch.Generator().manual_seed(5))).to(DEV) if affine else None
    z1 = torch.empty_like(y)
    st1 = be.norm_forward(y, z1, cout, scope, L.ACT_RELU, g, b, None, None, -1.0, None, ext_nchunks=nch)
    st1 = [t.clone() for t in st1[:2]]
    z2 = torch.empty_like(y)
    st2 = be.norm_forward(y, z2, cout, scope, L.ACT_RELU, g, b, None, None, -1.0, None)
    sync()
    assert rel(st1[0], st2[0]) < 1e-5 and rel(st1[1], st2[1]) < 1e-5
    assert rel(z1, z2) <= 2.0 ** -7
    assert float((z1 != z2).float().mean()) < 1e-3
