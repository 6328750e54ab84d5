"""conv_dot_fwd_kernel / conv_dot_dgrad_kernel (csrc/conv_dot.hip): the PatchGAN's one-channel prediction layer Conv2d(512, 1, k4, s1, p1)
(NLayerDiscriminator, networks.py:655-660), forward and data gradient through dl_conv_forward against the CPU emulation; the dispatch must take
the new kernels (dl_conv_kernel_name) whatever split-K the host asked for."""
import ctypes as C

import pytest
import torch

import fake_backend
from deepliif_amd import _lib as L
from deepliif_amd.engine import Precision
from deepliif_amd.geometry import ConvSpec, cpad, fill_conv_desc

from test_gpu_kernels import DEV, DRY, _run_conv, hip, rel, rnd, sync, tol

pytestmark = pytest.mark.gpu


def _name(plan, n, hi, wi, cin_p, ho, wo, cop, bias_n):
    d = fill_conv_desc(plan, n, hi, wi, cin_p, ho, wo, cop, cop, ho, wo, L.DL_BF16, L.PREC_BF16, L.ACT_NONE, L.ACT_NONE, bias_n, 1)
    return L.load().dl_conv_kernel_name(C.byref(d)).decode()


@pytest.mark.parametrize('shape', [(8, 31, 31), (2, 8, 8), (3, 17, 9), (1, 4, 4)], ids=lambda s: 'n%d-%dx%d' % s)
@pytest.mark.parametrize('splitk', [None, 1, 3])
def test_patchgan_prediction_layer_forward_and_data_gradient(shape, splitk):
    N, H, W_ = shape
    prec = Precision.get('bf16')
    spec = ConvSpec('conv', 512, 1, 4, 1, 1, L.PAD_ZERO, 0)
    w = rnd((1, 512, 4, 4), 1, prec, 0.05)
    bias = rnd((1,), 2, Precision.get('fp32'), 0.1)
    x = rnd((N, H, W_, 512), 3, prec).to(prec.dtype)
    ho, wo = spec.out_hw(H, W_)
    assert _name(spec.forward_plan(), N, H, W_, 512, ho, wo, 8, 1) == 'conv_dot_fwd_kernel'
    assert _name(spec.dgrad_plan(), N, ho, wo, 8, H, W_, 512, 0) == 'conv_dot_dgrad_kernel'
    fake, real = fake_backend.FakeBackend(), hip()
    for act in (L.ACT_NONE, L.ACT_LRELU):
        exp = _run_conv(fake, 'fwd', spec, prec, x, w, bias, act, L.ACT_NONE, H, W_)
        got = _run_conv(real, 'fwd', spec, prec, x.to(DEV), w.to(DEV), bias.to(DEV), act, L.ACT_NONE, H, W_, splitk=splitk)
        sync()
        assert DRY or real.last_conv_kernel == 'conv_dot_fwd_kernel'
        assert rel(got, exp) < tol(prec), ('fwd', act)
        assert float(got[..., 1:].float().abs().max()) == 0.0, 'the padding channels of the one-channel prediction stay zero'
    dy = torch.zeros(N, ho, wo, cpad(1))
    dy[..., :1] = rnd((N, ho, wo, 1), 4, prec)
    exp = _run_conv(fake, 'dgrad', spec, prec, dy.to(prec.dtype), w, None, L.ACT_NONE, L.ACT_NONE, H, W_)
    got = _run_conv(real, 'dgrad', spec, prec, dy.to(prec.dtype).to(DEV), w.to(DEV), None, L.ACT_NONE, L.ACT_NONE, H, W_, splitk=splitk)
    sync()
    assert DRY or real.last_conv_kernel == 'conv_dot_dgrad_kernel'
    assert rel(got, exp) < tol(prec), 'dgrad'


@pytest.mark.parametrize('shape', [(8, 512, 512), (2, 8, 512), (3, 20, 512), (1, 4, 512)], ids=lambda s: 'n%d-%dx%d' % s)
def test_patchgan_first_layer_forward(shape):
    """conv_d1_kernel (csrc/conv_d1.hip): Conv2d(6, 64, k4, s2, p1) + LeakyReLU(0.2) (networks.py:638-641) on 512-pixel-wide inputs"""
    N, H, W_ = shape
    prec = Precision.get('bf16')
    spec = ConvSpec('conv', 6, 64, 4, 2, 1, L.PAD_ZERO, 0)
    w = rnd((64, 6, 4, 4), 1, prec, 0.05)
    bias = rnd((64,), 2, Precision.get('fp32'), 0.1)
    x = torch.zeros(N, H, W_, cpad(6))
    x[..., :6] = rnd((N, H, W_, 6), 3, prec)
    x = x.to(prec.dtype)
    ho, wo = spec.out_hw(H, W_)
    assert _name(spec.forward_plan(), N, H, W_, 8, ho, wo, 64, 64) == 'conv_d1_kernel'
    fake, real = fake_backend.FakeBackend(), hip()
    for act, b in ((L.ACT_LRELU, bias), (L.ACT_NONE, bias), (L.ACT_LRELU, None)):
        exp = _run_conv(fake, 'fwd', spec, prec, x, w, b, act, L.ACT_NONE, H, W_)
        first = None
        for rep in range(2):
            got = _run_conv(real, 'fwd', spec, prec, x.to(DEV), w.to(DEV), None if b is None else b.to(DEV), act, L.ACT_NONE, H, W_, splitk=1)
            sync()
            assert DRY or real.last_conv_kernel == 'conv_d1_kernel'
            assert rel(got, exp) < tol(prec), ('fwd', act, rep)
            if first is None:
                first = got.clone()
            else:
                assert torch.equal(got, first), 'run-to-run difference'


@pytest.mark.parametrize('shape', [(8, 512, 512), (2, 8, 256), (1, 6, 512), (3, 20, 256)], ids=lambda s: 'n%d-%dx%d' % s)
def test_patchgan_first_layer_data_gradient(shape):
    """conv_d1g_kernel (csrc/conv_d1g.hip): the data gradient of Conv2d(6, 64, k4, s2, p1) -- the four sub-pixel phases as the M dimension of one MFMA tile"""
    N, H, W_ = shape
    prec = Precision.get('bf16')
    spec = ConvSpec('conv', 6, 64, 4, 2, 1, L.PAD_ZERO, 0)
    w = rnd((64, 6, 4, 4), 1, prec, 0.05)
    ho, wo = spec.out_hw(H, W_)
    dy = rnd((N, ho, wo, 64), 4, prec).to(prec.dtype)
    d = fill_conv_desc(spec.dgrad_plan(), N, ho, wo, 64, H, W_, 8, 8, ho, wo, L.DL_BF16, L.PREC_BF16, L.ACT_NONE, L.ACT_NONE, 0, 1)
    assert L.load().dl_conv_kernel_name(C.byref(d)).decode() == 'conv_d1g_kernel'
    fake, real = fake_backend.FakeBackend(), hip()
    exp = _run_conv(fake, 'dgrad', spec, prec, dy, w, None, L.ACT_NONE, L.ACT_NONE, H, W_)
    first = None
    for rep in range(3):
        got = _run_conv(real, 'dgrad', spec, prec, dy.to(DEV), w.to(DEV), None, L.ACT_NONE, L.ACT_NONE, H, W_, splitk=1)
        sync()
        assert DRY or real.last_conv_kernel == 'conv_d1g_kernel'
        assert rel(got, exp) < tol(prec), ('dgrad', rep)
        assert float(got[..., 6:].float().abs().max()) == 0.0, 'the padding channels of the gradient stay zero'
        if first is None:
            first = got.clone()
        else:
            assert torch.equal(got, first), 'run-to-run difference'
