"""GPU parity of every C-ABI kernel family, called through deepliif_amd.ops.HipBackend (ctypes -> libdeepliif_hip.so) and
checked against the CPU formula emulation (tests/fake_backend.py, itself pinned to torch's convolutions by
tests/test_geometry.py and to the oracle by tests/test_host_networks.py).

Tolerances (relative to the max |expected| of the tensor):
  fp32 policy (split-bf16 x3 MFMA, fp32 storage) : 1e-4   -- north-star bar is 1e-3
  bf16 policy (bf16 storage, one MFMA pass)      : inputs/weights are pre-rounded to bf16 so both sides see identical
                                                   operands; remaining error = bf16 output rounding (2^-9) + fp32
                                                   summation order -> 6e-3
"""
import numpy as np
import pytest
import torch

import fake_backend
from deepliif_amd import _lib as L
from deepliif_amd import ops
from deepliif_amd.engine import Precision
from deepliif_amd.geometry import ConvSpec, cpad

import os

pytestmark = pytest.mark.gpu
# DL_TEST_DRYRUN=1 runs this file on CPU with the emulation on both sides: a self-check of the test code only
DRY = os.environ.get('DL_TEST_DRYRUN') == '1'
DEV = 'cpu' if DRY else 'cuda'


def hip():
    if DRY:
        return fake_backend.FakeBackend()
    ops._impl = None
    return ops.impl()


def sync():
    if not DRY:
        torch.cuda.synchronize()


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


def tol(prec):
    return 1e-4 if prec.name == 'fp32' else 6e-3


def rnd(shape, seed, prec, scale=1.0):
    t = torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale
    return t.to(prec.dtype).float() if prec.dtype == torch.bfloat16 else t


# ------------------------------------------------------------------------------------------------ hardware probes
def test_probe_mfma_fragment_layout():
    if DRY:
        pytest.skip('hardware probe')
    lib = L.load()
    g = torch.Generator().manual_seed(0)
    a = torch.randn(16, 32, generator=g).bfloat16()
    b = torch.randn(32, 16, generator=g).bfloat16()       # asymmetric operands: catches row/col swaps
    d = torch.zeros(16, 16, device=DEV)
    ad, bd = a.to(DEV), b.to(DEV)          # keep the device copies alive across the launch
    L.check(lib.dl_probe_mfma16(ad.data_ptr(), bd.data_ptr(), d.data_ptr(), torch.cuda.current_stream().cuda_stream), 'probe')
    sync()
    assert rel(d, a.float() @ b.float()) < 1e-5


def test_probe_ds_read_tr16_b64():
    """lane (m, g) reads 8 B at tile[4g + (m>>2)][(m&3)*4]; the transposing read must return tile[4g + j][m], j = 0..3"""
    if DRY:
        pytest.skip('hardware probe')
    lib = L.load()
    src = torch.arange(64 * 16, dtype=torch.int16).view(64, 16)
    dst = torch.zeros(64, 4, dtype=torch.int16, device=DEV)
    sd = src.to(DEV)
    L.check(lib.dl_probe_trread(sd.data_ptr(), dst.data_ptr(), torch.cuda.current_stream().cuda_stream), 'probe')
    sync()
    dst = dst.cpu()
    exp = torch.zeros(64, 4, dtype=torch.int16)
    for lane in range(64):
        m, g = lane & 15, lane >> 4
        for j in range(4):
            exp[lane, j] = src[4 * g + j, m]
    if not torch.equal(dst, exp):
        np.save('gpurun_out/trread_actual.npy', dst.numpy())
    assert torch.equal(dst, exp), f'ds_read_b64_tr_b16 mapping differs; lane0={dst[0].tolist()} lane1={dst[1].tolist()} lane17={dst[17].tolist()}'


# ------------------------------------------------------------------------------------------------ conv forward / dgrad
CONV_CASES = [
    # kind, cin, cout, k, s, p, pad_mode, out_pad, N, H, W
    ('conv', 3, 64, 7, 1, 3, L.PAD_ZERO, 0, 2, 40, 36),
    ('conv', 3, 64, 7, 1, 3, L.PAD_REFLECT, 0, 1, 24, 20),
    ('conv', 64, 128, 3, 2, 1, L.PAD_ZERO, 0, 2, 32, 24),
    ('conv', 256, 256, 3, 1, 1, L.PAD_ZERO, 0, 2, 24, 20),
    ('conv', 128, 128, 3, 1, 1, L.PAD_REFLECT, 0, 1, 12, 16),
    ('convT', 256, 128, 3, 2, 1, L.PAD_ZERO, 1, 2, 12, 10),
    ('convT', 128, 64, 3, 2, 1, L.PAD_ZERO, 1, 1, 16, 16),
    ('conv', 64, 3, 7, 1, 3, L.PAD_ZERO, 0, 2, 24, 28),
    ('conv', 6, 64, 4, 2, 1, L.PAD_ZERO, 0, 2, 32, 32),
    ('conv', 12, 64, 4, 2, 1, L.PAD_ZERO, 0, 1, 16, 16),
    ('conv', 256, 512, 4, 2, 1, L.PAD_ZERO, 0, 2, 16, 16),
    ('conv', 512, 512, 4, 1, 1, L.PAD_ZERO, 0, 2, 9, 9),
    ('conv', 512, 1, 4, 1, 1, L.PAD_ZERO, 0, 2, 8, 8),
    ('convT', 1024, 512, 4, 2, 1, L.PAD_ZERO, 0, 2, 4, 4),
    ('convT', 128, 3, 4, 2, 1, L.PAD_ZERO, 0, 1, 16, 16),
    ('conv', 512, 512, 4, 2, 1, L.PAD_ZERO, 0, 8, 2, 2),      # UNet innermost down: split-K path
    ('conv', 64, 128, 3, 2, 1, L.PAD_ZERO, 0, 2, 25, 19),     # odd sizes: the stride-2 data gradient has ragged sub-pixel phases
    ('conv', 6, 64, 4, 2, 1, L.PAD_ZERO, 0, 3, 25, 38),       # PatchGAN first conv on an odd-height tile
    ('conv', 128, 256, 4, 2, 1, L.PAD_ZERO, 0, 1, 5, 5),      # tiny odd map, split-K
    ('convT', 512, 512, 4, 2, 1, L.PAD_ZERO, 0, 8, 1, 1),     # UNet innermost up
]


def _run_conv(be, plan_kind, spec, prec, x, w, bias, act, in_act, H, W_, splitk=None):
    plan = spec.forward_plan() if plan_kind == 'fwd' else spec.dgrad_plan()
    dev = x.device
    packed = ops.PackedWeights(plan, dev, prec.prec == L.PREC_BF16X3)
    be.pack_weights(packed, w)
    n = x.shape[0]
    if plan_kind == 'fwd':
        ho, wo = spec.out_hw(H, W_)
        hq, wq = (ho, wo) if spec.kind == 'conv' else (H, W_)
        cop = cpad(spec.cout)
    else:
        ho, wo = H, W_                                  # dx has the layer-input size
        oh, ow = spec.out_hw(H, W_)
        hq, wq = ((H + 1) // 2, (W_ + 1) // 2) if (spec.kind == 'conv' and spec.stride == 2) else (H, W_)     # ceil: odd sizes
        cop = cpad(spec.cin)
    out = torch.empty((n, ho, wo, cop), dtype=prec.dtype, device=dev)
    be.conv_forward(packed, x, out, hq, wq, bias, act, in_act, prec.prec, splitk)
    return out


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: f'{c[0]}{c[1]}-{c[2]}k{c[3]}s{c[4]}')
def test_conv_forward_and_dgrad(case, precname):
    kind, cin, cout, k, s, p, pm, op, N, H, W_ = case
    prec = Precision.get(precname)
    spec = ConvSpec(kind, cin, cout, k, s, p, pm, op)
    wshape = (cout, cin, k, k) if kind == 'conv' else (cin, cout, k, k)
    w = rnd(wshape, 1, prec, 0.05)
    bias = rnd((cout,), 2, Precision.get('fp32'), 0.1)
    x = torch.zeros(N, H, W_, cpad(cin))
    x[..., :cin] = rnd((N, H, W_, cin), 3, prec)
    fake, real = fake_backend.FakeBackend(), hip()
    for act, in_act in ((L.ACT_NONE, L.ACT_NONE), (L.ACT_TANH, L.ACT_LRELU)):
        exp = _run_conv(fake, 'fwd', spec, prec, x.to(prec.dtype), w, bias, act, in_act, H, W_)
        got = _run_conv(real, 'fwd', spec, prec, x.to(prec.dtype).to(DEV), w.to(DEV), bias.to(DEV), act, in_act, H, W_)
        sync()
        assert rel(got, exp) < tol(prec), ('fwd', act, in_act)
    # forced split-K must agree with the single-pass result
    got2 = _run_conv(real, 'fwd', spec, prec, x.to(prec.dtype).to(DEV), w.to(DEV), bias.to(DEV), L.ACT_NONE, L.ACT_NONE, H, W_, splitk=3)
    exp = _run_conv(fake, 'fwd', spec, prec, x.to(prec.dtype), w, bias, L.ACT_NONE, L.ACT_NONE, H, W_)
    assert rel(got2, exp) < tol(prec), 'splitk'
    if pm == L.PAD_ZERO:
        ho, wo = spec.out_hw(H, W_)
        dy = torch.zeros(N, ho, wo, cpad(cout))
        dy[..., :cout] = rnd((N, ho, wo, cout), 4, prec)
        exp = _run_conv(fake, 'dgrad', spec, prec, dy.to(prec.dtype), w, None, L.ACT_NONE, L.ACT_NONE, H, W_)
        got = _run_conv(real, 'dgrad', spec, prec, dy.to(prec.dtype).to(DEV), w.to(DEV), None, L.ACT_NONE, L.ACT_NONE, H, W_)
        sync()
        assert rel(got, exp) < tol(prec), 'dgrad'


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: f'{c[0]}{c[1]}-{c[2]}k{c[3]}s{c[4]}')
def test_conv_wgrad(case, precname):
    kind, cin, cout, k, s, p, pm, op, N, H, W_ = case
    prec = Precision.get(precname)
    spec = ConvSpec(kind, cin, cout, k, s, p, pm, op)
    ho, wo = spec.out_hw(H, W_)
    x = torch.zeros(N, H, W_, cpad(cin))
    x[..., :cin] = rnd((N, H, W_, cin), 5, prec)
    dy = torch.zeros(N, ho, wo, cpad(cout))
    dy[..., :cout] = rnd((N, ho, wo, cout), 6, prec)
    fake, real = fake_backend.FakeBackend(), hip()
    if kind == 'conv':
        P, Q, gshape = dy, x, (cout, cin, k, k)
        args = (k, s, p, pm, L.ACT_NONE, L.ACT_LRELU)
    else:
        P, Q, gshape = x, dy, (cin, cout, k, k)
        args = (k, s, p, L.PAD_ZERO, L.ACT_RELU, L.ACT_NONE)
    g_exp = torch.zeros(gshape)
    fake.conv_wgrad(P.to(prec.dtype), Q.to(prec.dtype), g_exp, *args, prec.prec, False)
    g0 = torch.randn(gshape, generator=torch.Generator().manual_seed(9))
    g_got = g0.clone().to(DEV)
    real.conv_wgrad(P.to(prec.dtype).to(DEV), Q.to(prec.dtype).to(DEV), g_got, *args, prec.prec, True)
    sync()
    # accumulate=True adds onto the existing gradient; bf16 policy: operands identical on both sides, fp32 accumulation
    assert rel(g_got.cpu() - g0, g_exp) < (1e-4 if precname == 'fp32' else 1e-3)
    g_got2 = torch.empty(gshape, device=DEV)
    real.conv_wgrad(P.to(prec.dtype).to(DEV), Q.to(prec.dtype).to(DEV), g_got2, *args, prec.prec, False, splitk=1)
    sync()
    assert rel(g_got2, g_exp) < (1e-4 if precname == 'fp32' else 1e-3)
    # no staged activation: the bf16 policy takes the direct-to-LDS (global_load_lds) kernel where the shape allows it
    args0 = args[:4] + (L.ACT_NONE, L.ACT_NONE)
    g_exp0 = torch.zeros(gshape)
    fake.conv_wgrad(P.to(prec.dtype), Q.to(prec.dtype), g_exp0, *args0, prec.prec, False)
    for sk in (None, 2):
        g_got3 = torch.empty(gshape, device=DEV)
        real.conv_wgrad(P.to(prec.dtype).to(DEV), Q.to(prec.dtype).to(DEV), g_got3, *args0, prec.prec, False, splitk=sk)
        sync()
        assert rel(g_got3, g_exp0) < (1e-4 if precname == 'fp32' else 1e-3), ('plain', sk)


# ------------------------------------------------------------------------------------------------ norm / elementwise / loss / adam
@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
@pytest.mark.parametrize('scope', [L.NORM_INSTANCE, L.NORM_BATCH])
@pytest.mark.parametrize('shape,C', [((2, 20, 24, 64), 64), ((3, 7, 5, 256), 256), ((2, 33, 31, 8), 3), ((8, 1, 1, 512), 512), ((1, 64, 64, 16), 12)])
def test_norm_forward_backward(shape, C, scope, precname):
    prec = Precision.get(precname)
    N, H, W_, Cp = shape
    if scope == L.NORM_INSTANCE and H * W_ == 1:
        pytest.skip('instance norm over a single pixel is degenerate (var = 0)')
    y = torch.zeros(shape)
    y[..., :C] = rnd((N, H, W_, C), 1, prec) * 1.7 + 0.3
    res = torch.zeros(shape)
    res[..., :C] = rnd((N, H, W_, C), 2, prec)
    dz = torch.zeros(shape)
    dz[..., :C] = rnd((N, H, W_, C), 3, prec)
    affine = scope == L.NORM_BATCH
    gamma = (1 + 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(4))) if affine else None
    beta = (0.1 * torch.randn(C, generator=torch.Generator().manual_seed(5))) if affine else None
    fake, real = fake_backend.FakeBackend(), hip()
    # every activation the C ABI accepts is its own kernel instantiation (tanh never follows a norm in the reference networks,
    # but dl_norm_desc.act allows it)
    for act, use_res in ((L.ACT_RELU, False), (L.ACT_NONE, True), (L.ACT_LRELU, False), (L.ACT_TANH, True)):
        rm_f, rv_f = (torch.zeros(C), torch.ones(C)) if affine else (None, None)
        rm_r, rv_r = (torch.zeros(C, device=DEV), torch.ones(C, device=DEV)) if affine else (None, None)
        z_f = torch.empty(shape, dtype=prec.dtype)
        st_f = fake.norm_forward(y.to(prec.dtype), z_f, C, scope, act, gamma, beta, rm_f, rv_f, 0.1 if affine else -1.0, res.to(prec.dtype) if use_res else None)
        z_r = torch.empty(shape, dtype=prec.dtype, device=DEV)
        g_r, b_r = (gamma.to(DEV), beta.to(DEV)) if affine else (None, None)
        st_r = real.norm_forward(y.to(prec.dtype).to(DEV), z_r, C, scope, act, g_r, b_r, rm_r, rv_r, 0.1 if affine else -1.0,
                                 res.to(prec.dtype).to(DEV) if use_res else None)
        sync()
        t = 2e-5 if precname == 'fp32' else 1e-2
        assert rel(z_r, z_f) < t, ('z', act)
        assert rel(st_r[0][:, :C], st_f[0][:, :C]) < 1e-4 and rel(st_r[1][:, :C], st_f[1][:, :C]) < 1e-4
        if affine:
            assert rel(rm_r, rm_f) < 1e-4 and rel(rv_r, rv_f) < 1e-4
        dy_f = torch.empty(shape, dtype=prec.dtype)
        dg_f, db_f = (torch.zeros(C), torch.zeros(C)) if affine else (None, None)
        cs_f, cs_r = torch.ones(C), torch.ones(C, device=DEV)
        fake.norm_backward(dz.to(prec.dtype), y.to(prec.dtype), dy_f, st_f, C, scope, act, gamma, dg_f, db_f, cs_f)
        dy_r = torch.empty(shape, dtype=prec.dtype, device=DEV)
        dg_r, db_r = (torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)) if affine else (None, None)
        real.norm_backward(dz.to(prec.dtype).to(DEV), y.to(prec.dtype).to(DEV), dy_r, st_r, C, scope, act, g_r, dg_r, db_r, cs_r)
        sync()
        assert rel(dy_r, dy_f) < (1e-4 if precname == 'fp32' else 2e-2), ('dy', act)
        # fused conv-bias gradient: 1 + sum over pixels of dy (analytically 0: compare absolutely, scaled by sum |dy|)
        assert float((cs_r.cpu() - cs_f).abs().max()) <= 1e-5 * float(dy_f.float().abs().sum()) / C + 1e-5, 'dy channel sums'
        if affine:
            assert rel(dg_r, dg_f) < 1e-3 and rel(db_r, db_f) < 1e-3


BIG_CASES = [
    # shapes large enough for the 256x256 tiles (>= 256 tiles): the 8-phase kernel (and DL_CONV_8PH=0: the one-barrier kernel)
    ('conv', 256, 256, 3, 1, 1, 8, 128, 128),        # ResnetBlock conv at the 512x512 training shape: forward + data gradient
    ('conv', 128, 256, 3, 2, 1, 8, 256, 256),        # second down conv: forward (9 taps x 128 ch) + 4-phase stride-2 data gradient
    ('convT', 256, 128, 3, 2, 1, 8, 128, 128),       # first up conv: 4-phase forward (Co = 128: plain tiles), data gradient 128 -> 256
    ('conv', 64, 256, 1, 1, 0, 4, 128, 128),         # odd number of K steps (1 tap x 64 channels = 1 step)
    ('conv', 192, 256, 3, 1, 1, 4, 128, 128),        # Cin padded to 256 (zero channels), 36 steps
    ('conv', 64, 256, 3, 1, 1, 16, 64, 128),         # kernel-column reuse with ONE channel chunk per tap (9 steps), 64-row images
    ('conv', 128, 256, 3, 1, 1, 4, 128, 256),        # 256-pixel rows: no kernel-column reuse
    ('conv', 256, 256, 3, 1, 1, 5, 120, 136),        # ragged: 81 600 pixels = 318.75 tiles (masked last tile), 16 320-pixel images
                                                     # are not a multiple of the 256-pixel tile, so tiles straddle two images
]


@pytest.mark.parametrize('precname', ['bf16', 'fp32'])
@pytest.mark.parametrize('case', BIG_CASES, ids=lambda c: f'{c[0]}{c[1]}-{c[2]}k{c[3]}s{c[4]}n{c[6]}')
def test_conv_big_tiles(case, precname):
    """bf16: conv_gemm_8ph_kernel; fp32 (strict policy): conv_gemm_8ph_x3_kernel / conv_gemm_glds_x3_kernel (csrc/conv_x3.h)"""
    kind, cin, cout, k, s_, p, N, H, W_ = case
    prec = Precision.get(precname)
    spec = ConvSpec(kind, cin, cout, k, s_, p, L.PAD_ZERO, 1 if kind == 'convT' else 0)
    wshape = (cout, cin, k, k) if kind == 'conv' else (cin, cout, k, k)
    w = rnd(wshape, 1, prec, 0.05)
    bias = rnd((cout,), 2, Precision.get('fp32'), 0.1)
    x = torch.zeros(N, H, W_, cpad(cin))
    x[..., :cin] = rnd((N, H, W_, cin), 3, prec)
    fake, real = fake_backend.FakeBackend(), hip()
    exp = _run_conv(fake, 'fwd', spec, prec, x.to(prec.dtype), w, bias, L.ACT_NONE, L.ACT_NONE, H, W_)
    for rep in range(3):            # repeated: a staging race would show up as run-to-run differences
        got = _run_conv(real, 'fwd', spec, prec, x.to(prec.dtype).to(DEV), w.to(DEV), bias.to(DEV), L.ACT_NONE, L.ACT_NONE, H, W_)
        sync()
        assert rel(got, exp) < tol(prec), ('fwd', rep)
        if rep == 0:
            first = got.clone()
        else:
            assert torch.equal(got, first), 'run-to-run difference'
    ho, wo = spec.out_hw(H, W_)
    dy = torch.zeros(N, ho, wo, cpad(cout))
    dy[..., :cout] = rnd((N, ho, wo, cout), 4, prec)
    exp = _run_conv(fake, 'dgrad', spec, prec, dy.to(prec.dtype), w, None, L.ACT_NONE, L.ACT_NONE, H, W_)
    for rep in range(3):
        got = _run_conv(real, 'dgrad', spec, prec, dy.to(prec.dtype).to(DEV), w.to(DEV), None, L.ACT_NONE, L.ACT_NONE, H, W_)
        sync()
        assert rel(got, exp) < tol(prec), ('dgrad', rep)
        if rep == 0:
            first = got.clone()
        else:
            assert torch.equal(got, first), 'run-to-run difference (dgrad)'


def test_pack_weights_batch_matches_single_packs():
    """dl_pack_weights_batch over a table of job records must write exactly what dl_pack_weights writes image by image
    (forward / 4-phase data-gradient / transposed / kernel-column-stacked layouts, with and without the lo image)."""
    be = hip()
    specs = [ConvSpec('conv', 3, 64, 7, 1, 3), ConvSpec('conv', 64, 128, 3, 2, 1), ConvSpec('conv', 256, 256, 3, 1, 1),
             ConvSpec('convT', 256, 128, 3, 2, 1, L.PAD_ZERO, 1), ConvSpec('conv', 6, 64, 4, 2, 1), ConvSpec('conv', 512, 1, 4, 1, 1),
             ConvSpec('conv', 64, 3, 7, 1, 3),
             # r05: the shapes that take the tiled form of the batch kernel (csrc/pack_tile.h; host twin: test_pack_tile_host.py) -- UNet-512 / PatchGAN 4x4 layers
             # (one-phase and 4-phase images, master rows = image rows and = contracted channels), a 1x1 layer, one real row out of 128
             ConvSpec('conv', 128, 256, 4, 2, 1), ConvSpec('conv', 512, 512, 4, 2, 1), ConvSpec('convT', 1024, 512, 4, 2, 1),
             ConvSpec('convT', 512, 256, 4, 2, 1), ConvSpec('conv', 256, 128, 1, 1, 0), ConvSpec('convT', 128, 64, 3, 2, 1, L.PAD_ZERO, 1)]
    jobs, refs = [], []
    for i, spec in enumerate(specs):
        wshape = (spec.cout, spec.cin, spec.k, spec.k) if spec.kind == 'conv' else (spec.cin, spec.cout, spec.k, spec.k)
        w = rnd(wshape, 10 + i, Precision.get('fp32'), 0.05).to(DEV)
        plans = [spec.forward_plan(), spec.dgrad_plan()]
        if spec.is_narrow():
            plans.append(spec.narrow_forward_plan())
        for plan in plans:
            for with_lo in (False, True):
                single = ops.PackedWeights(plan, DEV, with_lo)
                be.pack_weights(single, w)
                batched = ops.PackedWeights(plan, DEV, with_lo)
                batched.hi.fill_(0x7fc0)               # (a bf16 NaN pattern: anything the pack does not overwrite shows up)
                if with_lo:
                    batched.lo.fill_(0x7fc0)
                jobs.append((batched, w))
                refs.append(single)
    table = be.pack_batch_build(jobs)
    assert table[0].numel() == len(jobs) * int(be.lib.dl_pack_job_bytes()) and table[1].numel() == 2 * table[2] and table[2] >= len(jobs)
    tiled = (table[1][0::2] & (1 << 30)) != 0
    assert tiled.any() and not tiled.all()                     # both forms of the kernel are in this launch
    be.pack_batch_run(table, len(jobs))
    sync()
    for (batched, _), single in zip(jobs, refs):
        assert torch.equal(batched.hi.view(torch.int16), single.hi.view(torch.int16))
        if single.lo is not None:
            assert torch.equal(batched.lo.view(torch.int16), single.lo.view(torch.int16))
    os.environ['DL_PACK_TILED'] = '0'                           # the A/B switch: every job in the chunk-per-thread form
    L.load().dl_switches_reload()                               # (the library reads its switches once, at load)
    try:
        chunk_table = be.pack_batch_build(jobs)
    finally:
        del os.environ['DL_PACK_TILED']
        L.load().dl_switches_reload()
    assert not ((chunk_table[1][0::2] & (1 << 30)) != 0).any()
    for batched, _ in jobs:
        batched.hi.fill_(0x7fc0)               # (a bf16 NaN pattern: anything the pack does not overwrite shows up)
    be.pack_batch_run(chunk_table, len(jobs))
    sync()
    for (batched, _), single in zip(jobs, refs):
        assert torch.equal(batched.hi.view(torch.int16), single.hi.view(torch.int16))


@pytest.mark.parametrize('precname,ca,q_act', [('bf16', 256, L.ACT_NONE), ('fp32', 256, L.ACT_NONE), ('fp32', 128, L.ACT_LRELU), ('fp32', 256, L.ACT_RELU)])
def test_wgrad_fast_path_ragged_pixels(precname, ca, q_act):
    """direct-to-LDS weight gradient (wgrad_glds_kernel for the bf16 policy, wgrad_glds_x3_kernel for the strict one; ca -> 256, 3x3) over a
    pixel count that is neither a multiple of the K step nor of the split-K chunk, with tiles straddling images: 5 x 120 x 136 = 81 600
    pixels.  The strict kernel also applies the staged operand activation while it splits the tile in LDS."""
    prec = Precision.get(precname)
    N, H, W_, C = 5, 120, 136, 256
    x = rnd((N, H, W_, C), 5, prec).to(prec.dtype)
    dy = rnd((N, H, W_, ca), 6, prec).to(prec.dtype)
    fake, real = fake_backend.FakeBackend(), hip()
    g_exp = torch.zeros(ca, C, 3, 3)
    fake.conv_wgrad(dy, x, g_exp, 3, 1, 1, L.PAD_ZERO, L.ACT_NONE, q_act, prec.prec, False)
    t = 1e-3 if precname == 'bf16' else 1e-4
    for sk in (None, 1, 7):
        g = torch.empty(ca, C, 3, 3, device=DEV)
        real.conv_wgrad(dy.to(DEV), x.to(DEV), g, 3, 1, 1, L.PAD_ZERO, L.ACT_NONE, q_act, prec.prec, False, splitk=sk)
        sync()
        assert rel(g, g_exp) < t, ('splitk', sk)
        if sk is None:
            g2 = torch.empty_like(g)
            real.conv_wgrad(dy.to(DEV), x.to(DEV), g2, 3, 1, 1, L.PAD_ZERO, L.ACT_NONE, q_act, prec.prec, False, splitk=sk)
            sync()
            assert torch.equal(g, g2), 'run-to-run difference (fixed-order reduction expected)'


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
@pytest.mark.parametrize('shape,pad', [((2, 9, 10, 16), 3), ((1, 5, 7, 8), 1), ((3, 32, 24, 64), 3), ((2, 4, 4, 8), 3), ((1, 128, 128, 256), 1)])
def test_reflect_fold(shape, pad, precname):
    """dl_reflect_fold (backward of nn.ReflectionPad2d) against the formula backend, including maps barely larger than the pad
    (4 x 4 with pad 3: the mirrored ranges of both edges overlap)."""
    prec = Precision.get(precname)
    n, h, w, c = shape
    src = rnd((n, h + 2 * pad, w + 2 * pad, c), 7, prec).to(prec.dtype)
    exp = torch.empty(shape, dtype=prec.dtype)
    fake_backend.FakeBackend().reflect_fold(src, exp, pad)
    got = torch.empty(shape, dtype=prec.dtype, device=DEV)
    hip().reflect_fold(src.to(DEV), got, pad)
    sync()
    assert rel(got, exp) < (1e-6 if precname == 'fp32' else 2.0 ** -7)     # bf16: same fp32 sums, different order -> one-ulp flips


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
@pytest.mark.parametrize('case', [(3, 64, 7, 3, 1, 24, 20), (128, 128, 3, 1, 1, 12, 16), (32, 32, 3, 1, 2, 10, 6), (64, 3, 7, 3, 2, 40, 24),
                                  (256, 256, 3, 1, 2, 64, 64)], ids=lambda c: f'c{c[0]}-{c[1]}k{c[2]}p{c[3]}n{c[4]}_{c[5]}x{c[6]}')
def test_reflect_conv_data_gradient(case, precname):
    """Data gradient of nn.ReflectionPad2d(p) + Conv2d(padding=0): dl_conv_forward with the pad-0 plan over the PADDED extent (the
    output, (H+2p) x (W+2p), is larger than the input dy and every tap offset is <= 0 -- a geometry no other layer produces),
    then dl_reflect_fold.  Against the formula backend, which test_geometry pins to torch.autograd."""
    cin, cout, k, p, N, H, W_ = case
    prec = Precision.get(precname)
    spec = ConvSpec('conv', cin, cout, k, 1, p, L.PAD_REFLECT, 0)
    w = rnd((cout, cin, k, k), 1, prec, 0.05)
    dy = torch.zeros(N, H, W_, cpad(cout))
    dy[..., :cout] = rnd((N, H, W_, cout), 4, prec)
    plan = spec.dgrad_plan()
    outs = []
    for be, dev in ((fake_backend.FakeBackend(), 'cpu'), (hip(), DEV)):
        packed = ops.PackedWeights(plan, dev, prec.prec == L.PREC_BF16X3)
        be.pack_weights(packed, w.to(dev))
        dxp = torch.empty((N, H + 2 * p, W_ + 2 * p, cpad(cin)), dtype=prec.dtype, device=dev)
        be.conv_forward(packed, dy.to(prec.dtype).to(dev), dxp, H + 2 * p, W_ + 2 * p, None, L.ACT_NONE, L.ACT_NONE, prec.prec)
        dx = torch.empty((N, H, W_, cpad(cin)), dtype=prec.dtype, device=dev)
        be.reflect_fold(dxp, dx, p)
        outs.append((dxp, dx))
    sync()
    (dxp_f, dx_f), (dxp_r, dx_r) = outs
    assert rel(dxp_r, dxp_f) < tol(prec), 'gradient w.r.t. the padded input'
    assert rel(dx_r, dx_f) < tol(prec), 'folded gradient'


STATS_CASES = [
    # kind, cin, cout, k, s, p, N, H, W      (bf16 direct-to-LDS dispatch: 256x16 / 128x64 / 128x128 / 256x256 tiles, 4-phase convT)
    ('conv', 3, 64, 7, 1, 3, 2, 32, 32),
    ('conv', 64, 128, 3, 2, 1, 2, 32, 32),
    ('conv', 256, 256, 3, 1, 1, 2, 32, 32),
    ('conv', 256, 256, 3, 1, 1, 8, 128, 128),
    ('convT', 256, 128, 3, 2, 1, 2, 16, 16),
    ('conv', 6, 64, 4, 2, 1, 2, 64, 64),
    ('conv', 64, 8, 3, 1, 1, 1, 32, 32),
]


@pytest.mark.parametrize('precname', ['bf16', 'fp32'])
@pytest.mark.parametrize('scope', [L.NORM_INSTANCE, L.NORM_BATCH])
@pytest.mark.parametrize('case', STATS_CASES, ids=lambda c: f'{c[0]}{c[1]}-{c[2]}k{c[3]}s{c[4]}n{c[6]}')
def test_conv_fused_norm_statistics(case, scope, precname):
    """dl_conv_forward(stats_part) + dl_norm_forward(ext_nchunks) must give the statistics / output of the stand-alone pass over
    the same stored y (the sums are taken over the stored -- bf16-rounded or fp32 -- values in both).  fp32 = the strict policy's
    direct-to-LDS kernels (csrc/conv_x3.h); its 7x7 / 3-channel stem has no fused statistics (the c4 kernel is bf16-only) unless the
    generic strict tile takes it."""
    kind, cin, cout, k, s, p, N, H, W_ = case
    prec = Precision.get(precname)
    spec = ConvSpec(kind, cin, cout, k, s, p, L.PAD_ZERO, 1 if kind == 'convT' else 0)
    wshape = (cout, cin, k, k) if kind == 'conv' else (cin, cout, k, k)
    w = rnd(wshape, 1, prec, 0.05).to(DEV)
    bias = rnd((cout,), 2, Precision.get('fp32'), 0.1).to(DEV)
    x = torch.zeros(N, H, W_, cpad(cin))
    x[..., :cin] = rnd((N, H, W_, cin), 3, prec)
    x = x.to(prec.dtype).to(DEV)
    be = hip()
    plan = spec.forward_plan()
    packed = ops.PackedWeights(plan, DEV, prec.prec == L.PREC_BF16X3)
    be.pack_weights(packed, w)
    ho, wo = spec.out_hw(H, W_)
    hq, wq = (ho, wo) if kind == 'conv' else (H, W_)
    y = torch.empty((N, ho, wo, cpad(cout)), dtype=prec.dtype, device=DEV)
    nch = be.conv_forward(packed, x, y, hq, wq, bias, L.ACT_NONE, L.ACT_NONE, prec.prec, splitk=1, want_stats=True)
    assert nch > 0, 'this case is expected to take the fused-statistics path'
    affine = scope == L.NORM_BATCH
    g = (1 + 0.1 * torch.randn(cout, generator=torch.Generator().manual_seed(4))).to(DEV) if affine else None
    b = (0.1 * torch.randn(cout, generator=torch.Generator().manual_seed(5))).to(DEV) if affine else None
    z1 = torch.empty_like(y)
    st1 = be.norm_forward(y, z1, cout, scope, L.ACT_RELU, g, b, None, None, -1.0, None, ext_nchunks=nch)
    y2 = torch.empty_like(y)
    assert be.conv_forward(packed, x, y2, hq, wq, bias, L.ACT_NONE, L.ACT_NONE, prec.prec, splitk=1) == 0
    z2 = torch.empty_like(y)
    st2 = be.norm_forward(y2, z2, cout, scope, L.ACT_RELU, g, b, None, None, -1.0, None)
    sync()
    assert torch.equal(y, y2)
    assert rel(st1[0][:, :cout], st2[0][:, :cout]) < 1e-5 and rel(st1[1][:, :cout], st2[1][:, :cout]) < 1e-5
    # same scale/shift up to fp32 summation order -> identical up to rare one-ulp bf16 rounding flips (fp32 storage: up to ~1e-6)
    if precname == 'bf16':
        assert rel(z1, z2) <= 2.0 ** -7
        assert float((z1 != z2).float().mean()) < 1e-3
    else:
        assert rel(z1, z2) <= 1e-5
    assert float(z1[..., cout:].float().abs().max()) == 0.0 if cpad(cout) > cout else True


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
def test_elementwise_family(precname):
    prec = Precision.get(precname)
    fake, real = fake_backend.FakeBackend(), hip()
    shape = (2, 9, 11, 32)
    a, b = rnd(shape, 1, prec), rnd(shape, 2, prec)
    buf = rnd((2, 9, 11, 64), 3, prec)
    t = 1e-6 if precname == 'fp32' else 8e-3
    for act in (L.ACT_RELU, L.ACT_LRELU, L.ACT_TANH):
        yf = torch.empty(shape, dtype=prec.dtype); fake.act_forward(act, a.to(prec.dtype), yf)
        yr = torch.empty(shape, dtype=prec.dtype, device=DEV); real.act_forward(act, a.to(prec.dtype).to(DEV), yr)
        assert rel(yr, yf) < t
        df = torch.empty(shape, dtype=prec.dtype); fake.act_backward(act, b.to(prec.dtype), yf, df)
        dr = torch.empty(shape, dtype=prec.dtype, device=DEV); real.act_backward(act, b.to(prec.dtype).to(DEV), yr, dr)
        assert rel(dr, df) < 2 * t
    # axpby into a channel slice of a wider buffer
    bf_, br_ = buf.clone().to(prec.dtype), buf.clone().to(prec.dtype).to(DEV)
    fake.axpby(0.25, a.to(prec.dtype), -1.5, b.to(prec.dtype), bf_[..., 32:])
    real.axpby(0.25, a.to(prec.dtype).to(DEV), -1.5, b.to(prec.dtype).to(DEV), br_[..., 32:])
    assert rel(br_, bf_) < t
    # channel copies (cat / slice), 3 channels at offset 3
    cf, cr = torch.zeros((2, 9, 11, 8), dtype=prec.dtype), torch.zeros((2, 9, 11, 8), dtype=prec.dtype, device=DEV)
    fake.copy_channels(a.to(prec.dtype), 5, cf, 3, 3); real.copy_channels(a.to(prec.dtype).to(DEV), 5, cr, 3, 3)
    fake.copy_channels(b.to(prec.dtype), 0, cf, 3, 3, True); real.copy_channels(b.to(prec.dtype).to(DEV), 0, cr, 3, 3, True)
    assert rel(cr, cf) < t
    # channel sums
    sf, sr = torch.ones(20), torch.ones(20, device=DEV)
    fake.channel_sum(a.to(prec.dtype), 20, sf, True); real.channel_sum(a.to(prec.dtype).to(DEV), 20, sr, True)
    assert rel(sr, sf) < 1e-4
    # layout converters
    x = torch.randn(2, 3, 9, 11)
    nf, nr = torch.ones((2, 9, 11, 8), dtype=prec.dtype), torch.ones((2, 9, 11, 8), dtype=prec.dtype, device=DEV)
    fake.nchw_to_nhwc(x, nf, 0, 8); real.nchw_to_nhwc(x.to(DEV), nr, 0, 8)
    assert torch.equal(nr.cpu(), nf)
    of, orr = torch.empty(2, 3, 9, 11), torch.empty(2, 3, 9, 11, device=DEV)
    fake.nhwc_to_nchw(nf, 0, of); real.nhwc_to_nchw(nr, 0, orr)
    assert torch.equal(orr.cpu(), of)


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
def test_losses(precname):
    prec = Precision.get(precname)
    fake, real = fake_backend.FakeBackend(), hip()
    x = torch.zeros(2, 30, 30, 8); x[..., :1] = rnd((2, 30, 30, 1), 1, prec) * 3
    a = torch.zeros(2, 16, 16, 8); a[..., :3] = rnd((2, 16, 16, 3), 2, prec)
    b = torch.zeros(2, 16, 16, 8); b[..., :3] = rnd((2, 16, 16, 3), 3, prec) * 2
    for kind, t, tgt, C, const in ((L.LOSS_BCE_LOGITS, x, None, 1, 1.0), (L.LOSS_BCE_LOGITS, x, None, 1, 0.0), (L.LOSS_MSE, x, None, 1, 1.0),
                                   (L.LOSS_SMOOTH_L1, a, b, 3, 0.0), (L.LOSS_L1, a, b, 3, 0.0)):
        lf, lr = torch.zeros(1), torch.zeros(1, device=DEV)
        gf = torch.empty(t.shape, dtype=prec.dtype)
        gr = torch.full(t.shape, 7.0, dtype=prec.dtype, device=DEV)
        fake.loss(kind, t.to(prec.dtype), tgt.to(prec.dtype) if tgt is not None else None, const, C, lf, gf, 0.37)
        real.loss(kind, t.to(prec.dtype).to(DEV), tgt.to(prec.dtype).to(DEV) if tgt is not None else None, const, C, lr, gr, 0.37)
        sync()
        assert abs(float(lr) - float(lf)) < 1e-5 * max(1.0, abs(float(lf))), kind
        assert rel(gr, gf) < (1e-5 if precname == 'fp32' else 8e-3), kind


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
def test_loss_accumulates_weighted_terms(precname):
    """dl_loss_acc: loss_out[0] (+)= out_scale * mean -- the five weighted L1 terms of VGGLoss land in one slot (networks.py:738-743)"""
    prec = Precision.get(precname)
    real = hip()
    a = torch.zeros(2, 16, 16, 8); a[..., :5] = rnd((2, 16, 16, 5), 2, prec)
    b = torch.zeros(2, 16, 16, 8); b[..., :5] = rnd((2, 16, 16, 5), 3, prec)
    out = torch.full((1,), 3.0, device=DEV)
    real.loss(L.LOSS_L1, a.to(prec.dtype).to(DEV), b.to(prec.dtype).to(DEV), 0.0, 5, out, None, 1.0, out_scale=0.25, accumulate=False)
    real.loss(L.LOSS_L1, a.to(prec.dtype).to(DEV), b.to(prec.dtype).to(DEV), 0.0, 5, out, None, 1.0, out_scale=2.0, accumulate=True)
    sync()
    exp = 2.25 * float((a.to(prec.dtype).float()[..., :5] - b.to(prec.dtype).float()[..., :5]).abs().mean())
    assert abs(float(out) - exp) < 1e-5 * exp


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 16, 16, 8), (1, 37, 51, 64), (3, 8, 10, 128)])
def test_maxpool2_forward_backward_against_torch(precname, shape):
    """nn.MaxPool2d(2, 2) of the VGG19 features: values, floor mode on odd sizes, and ATen's first-maximum tie rule in backward (ties are the
    NORM after a ReLU: whole windows of zeros)"""
    prec = Precision.get(precname)
    real = hip()
    n, h, w, c = shape
    x = torch.relu(rnd(shape, 5, prec)).to(prec.dtype)                  # ~half the entries are exactly 0
    he, we = h // 2 * 2, w // 2 * 2
    x[:, 0:he:4, 0:we:4] = x[:, 1:he:4, 1:we:4]                       # positive ties inside windows (top-left == bottom-right)
    xt = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    yt = torch.nn.functional.max_pool2d(xt, 2, 2)
    dy = rnd((n, h // 2, w // 2, c), 6, prec).to(prec.dtype)
    yt.backward(dy.float().permute(0, 3, 1, 2))
    y = torch.empty((n, h // 2, w // 2, c), dtype=prec.dtype, device=DEV)
    dx = torch.full(shape, 9.0, dtype=prec.dtype, device=DEV)
    real.maxpool2_forward(x.to(DEV), y)
    real.maxpool2_backward(x.to(DEV), dy.to(DEV), dx)
    sync()
    assert torch.equal(y.cpu().float(), yt.detach().permute(0, 2, 3, 1))
    assert torch.equal(dx.cpu().float(), xt.grad.permute(0, 2, 3, 1))


def test_adam_matches_torch():
    n = 100003
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(n, generator=g) * 0.02
    real = hip()
    p_t = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p_t], lr=2e-4, betas=(0.5, 0.999))
    p_r = p0.clone().to(DEV); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * 1e-3
        p_t.grad = grad.clone()
        opt.step()
        real.adam_step(p_r, grad.to(DEV), m, v, 2e-4, 0.5, 0.999, 1e-8, step, 1.0)
    sync()
    assert float((p_r.cpu() - p_t.detach()).abs().max()) < 2e-8


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
@pytest.mark.parametrize('pm', [L.PAD_ZERO, L.PAD_REFLECT])
def test_narrow_cout_head_conv(precname, pm):
    """7x7, 64 -> 3 head (networks.py:438-443) through the kernel-column-stacked path: dl_pack_weights(stack_kw) +
    dl_conv_forward(raw_out) + dl_shift_sum, and dl_shift_stack + dl_conv_wgrad(stack_kw); engine-level, both backends."""
    from deepliif_amd import engine as E
    prec = Precision.get(precname)
    spec = ConvSpec('conv', 64, 3, 7, 1, 3, pm)
    assert spec.is_narrow()
    w0 = rnd((3, 64, 7, 7), 1, prec, 0.05)
    b0 = rnd((3,), 2, Precision.get('fp32'), 0.1)
    x0 = rnd((2, 20, 24, 64), 3, prec)
    g0 = torch.zeros(2, 20, 24, 8)
    g0[..., :3] = rnd((2, 20, 24, 3), 4, prec)
    res = {}
    for name, dev in (('fake', 'cpu'), ('real', DEV)):
        ops._impl = fake_backend.FakeBackend() if name == 'fake' else hip()
        w = torch.nn.Parameter(w0.clone().to(dev)); w.grad = torch.zeros_like(w)
        b = torch.nn.Parameter(b0.clone().to(dev)); b.grad = torch.zeros_like(b)
        layer = E.ConvLayer(spec, w, b)
        tape = E.Tape() if pm == L.PAD_ZERO else None
        ctx = E.Ctx(prec, tape, training=True)
        xa = E.Act(x0.to(prec.dtype).to(dev), 64, False)
        y = E.conv(ctx, xa, layer, act=L.ACT_TANH)
        out = {'y': y.t.float().cpu()}
        if tape is not None:
            y.grad = g0.to(prec.dtype).to(dev)
            tape.backward()
            out['dw'], out['db'] = w.grad.cpu(), b.grad.cpu()
        res[name] = out
    sync()
    ops._impl = None
    t = 1e-4 if precname == 'fp32' else 8e-3
    assert rel(res['real']['y'], res['fake']['y']) < t
    if pm == L.PAD_ZERO:
        assert rel(res['real']['dw'], res['fake']['dw']) < (1e-4 if precname == 'fp32' else 2e-3)
        assert rel(res['real']['db'], res['fake']['db']) < 1e-3


@pytest.mark.parametrize('precname', ['bf16', 'fp32'])
@pytest.mark.parametrize('shape', [(2, 70, 130), (1, 16, 58), (1, 150, 64), (3, 33, 59), (1, 512, 512)])
@pytest.mark.parametrize('cout,act', [(3, L.ACT_TANH), (1, L.ACT_NONE)])
def test_narrow_roll_kernel_against_torch(shape, cout, act, precname):
    """dl_conv_narrow_forward (rolling input rows, every kernel row at once, kernel-column sum from LDS) vs torch's conv2d on the same
    operands: several strips with a ragged last one, row bands with recomputed halo rows, images narrower than a strip.
    fp32 = dl_conv_narrow_forward_x3 (strict policy: fp32 rows split into bf16 hi / lo while the fragments are read, lo weights in LDS, eight
    rolling blocks) against the fp64 convolution of the fp32 operands."""
    from deepliif_amd import engine as E
    n, h, w = shape
    prec = Precision.get(precname)
    real = hip()
    spec = ConvSpec('conv', 64, cout, 7, 1, 3, L.PAD_ZERO)
    w0 = rnd((cout, 64, 7, 7), 11, prec, 0.05)
    b0 = rnd((cout,), 12, Precision.get('fp32'), 0.1)
    x0 = rnd((n, h, w, 64), 13, prec)
    ref = torch.nn.functional.conv2d(x0.double().permute(0, 3, 1, 2), w0.double(), b0.double(), padding=3)
    if act == L.ACT_TANH:
        ref = torch.tanh(ref)
    ref = ref.float()
    wp = torch.nn.Parameter(w0.clone().to(DEV))
    layer = E.ConvLayer(spec, wp, torch.nn.Parameter(b0.clone().to(DEV)))
    layer.ensure_packed(prec, need_dgrad=False)
    xd = x0.to(prec.dtype).to(DEV)
    assert real.conv_narrow_supported(xd, 64, cout, 7, 3, L.PAD_ZERO)
    out = torch.full((n, h, w, 8), 5.0, dtype=prec.dtype, device=DEV)
    real.conv_narrow_forward(layer.packed_fwd, xd, out, cout, 7, 3, layer.bias.detach(), act)
    sync()
    got = out.float().cpu()
    assert (got[..., cout:] == 0).all(), 'padded channels must be written as zeros'
    assert rel(got[..., :cout].permute(0, 3, 1, 2), ref) < (6e-3 if precname == 'bf16' else 1e-4)
    if precname == 'fp32':
        out2 = torch.empty_like(out)
        real.conv_narrow_forward(layer.packed_fwd, xd, out2, cout, 7, 3, layer.bias.detach(), act)
        sync()
        assert torch.equal(out, out2)              # run to run


@pytest.mark.parametrize('precname', ['bf16', 'fp32'])
@pytest.mark.parametrize('shape', [(2, 64, 128), (1, 8, 64), (3, 36, 192), (1, 512, 512)])
@pytest.mark.parametrize('pm', [L.PAD_ZERO, L.PAD_REFLECT])
def test_c4_patch_kernel_stem_forward_and_head_dgrad(shape, pm, precname):
    """conv_c4_patch_kernel (<= 4 real input channels, 7x7, input patch staged once in two 8-byte-shifted copies): the ResnetGenerator
    stem forward incl. the fused norm statistics, and the head's data gradient (a 3 -> 64 conv with the flipped kernel), vs torch.
    fp32 = the strict policy's conv_c4_patch_x3_kernel (csrc/conv_x3.h: fp32 patch split once per staged pixel, three products)."""
    from deepliif_amd import engine as E
    n, h, w = shape
    prec = Precision.get(precname)
    kname = 'conv_c4_patch_kernel' if precname == 'bf16' else 'conv_c4_patch_x3_kernel'
    tol = 6e-3 if precname == 'bf16' else 1e-4
    real = hip()
    # ---- stem: 3 -> 64, bias, statistics for the following norm
    spec = ConvSpec('conv', 3, 64, 7, 1, 3, pm)
    w0 = rnd((64, 3, 7, 7), 21, prec, 0.05)
    b0 = rnd((64,), 22, Precision.get('fp32'), 0.1)
    x0 = torch.zeros(n, h, w, 8)
    x0[..., :3] = rnd((n, h, w, 3), 23, prec)
    layer = E.ConvLayer(spec, torch.nn.Parameter(w0.clone().to(DEV)), torch.nn.Parameter(b0.clone().to(DEV)))
    layer.ensure_packed(prec, need_dgrad=False)
    out = torch.empty((n, h, w, 64), dtype=prec.dtype, device=DEV)
    nch = real.conv_forward(layer.packed_fwd, x0.to(prec.dtype).to(DEV), out, h, w, layer.bias.detach(), L.ACT_NONE, L.ACT_NONE, prec.prec, want_stats=True)
    sync()
    assert real.last_conv_kernel == kname and nch == (h // 4) * (w // 64)
    xp = x0[..., :3].permute(0, 3, 1, 2)
    xp = torch.nn.functional.pad(xp, (3, 3, 3, 3), mode='reflect') if pm == L.PAD_REFLECT else torch.nn.functional.pad(xp, (3, 3, 3, 3))
    ref = torch.nn.functional.conv2d(xp, w0, b0)
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert rel(got, ref) < tol
    part = ops.WS.get('norm_ws', 1, out.device)[:n * nch * 2 * 64].view(n, nch, 2, 64).cpu()
    assert torch.allclose(part[:, :, 0].sum(1), got.sum(dim=(2, 3)), rtol=1e-4, atol=1e-2)
    assert torch.allclose(part[:, :, 1].sum(1), (got * got).sum(dim=(2, 3)), rtol=1e-4, atol=1e-2)
    # ---- head data gradient: dx[64] from dy[3] (zero padding only: the reflect gradient runs over the padded extent, another plan)
    if pm != L.PAD_ZERO:
        return
    hspec = ConvSpec('conv', 64, 3, 7, 1, 3, L.PAD_ZERO)
    hw = rnd((3, 64, 7, 7), 24, prec, 0.05)
    hl = E.ConvLayer(hspec, torch.nn.Parameter(hw.clone().to(DEV)), None)
    hl.ensure_packed(prec, need_dgrad=True)
    dy0 = torch.zeros(n, h, w, 8)
    dy0[..., :3] = rnd((n, h, w, 3), 25, prec)
    dx = torch.empty((n, h, w, 64), dtype=prec.dtype, device=DEV)
    real.conv_forward(hl.packed_dgrad, dy0.to(prec.dtype).to(DEV), dx, h, w, None, L.ACT_NONE, L.ACT_NONE, prec.prec)
    sync()
    assert real.last_conv_kernel == kname
    xt = torch.zeros(n, 64, h, w, requires_grad=True)
    torch.nn.functional.conv2d(xt, hw, padding=3).backward(dy0[..., :3].permute(0, 3, 1, 2).contiguous())
    assert rel(dx.float().cpu().permute(0, 3, 1, 2), xt.grad) < tol


@pytest.mark.parametrize('precname', ['bf16', 'fp32'])
@pytest.mark.parametrize('shape', [(2, 64, 128), (1, 4, 64), (3, 36, 192), (2, 256, 256)])
def test_c4_weight_gradient_of_the_7x7_layers(shape, precname):
    """wgrad_c4_kernel (csrc/wgrad_c4.h): dW of the Resnet stem (3 -> 64: wide = dL/dy, small = x) and of its head (64 -> 3: wide = x,
    small = dL/dy, mirrored kernel indices), persistent workgroups + fixed-order combine, vs torch autograd; accumulate semantics.
    fp32 = the strict policy's wgrad_c4_x3_kernel (csrc/wgrad_x3.h)."""
    n, h, w = shape
    prec = Precision.get(precname)
    tol = 2e-3 if precname == 'bf16' else 1e-4
    real = hip()
    for cin, cout in ((3, 64), (64, 3)):
        x0 = torch.zeros(n, h, w, max(8, cin)); x0[..., :cin] = rnd((n, h, w, cin), 31, prec)
        dy0 = torch.zeros(n, h, w, max(8, cout)); dy0[..., :cout] = rnd((n, h, w, cout), 32, prec)
        wt = torch.zeros(cout, cin, 7, 7, requires_grad=True)
        torch.nn.functional.conv2d(x0[..., :cin].permute(0, 3, 1, 2), wt, padding=3).backward(dy0[..., :cout].permute(0, 3, 1, 2).contiguous())
        g0 = rnd((cout, cin, 7, 7), 33, Precision.get('fp32'))
        grad = g0.clone().to(DEV)
        P, Q = dy0.to(prec.dtype).to(DEV), x0.to(prec.dtype).to(DEV)
        assert real.wgrad_c4_applies(P, Q, grad, 7, 1, 3, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, prec.prec)
        real.conv_wgrad(P, Q, grad, 7, 1, 3, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, prec.prec, True)
        sync()
        assert rel(grad.cpu() - g0, wt.grad) < tol, (cin, cout)
        real.conv_wgrad(P, Q, grad, 7, 1, 3, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, prec.prec, False)
        sync()
        assert rel(grad.cpu(), wt.grad) < tol, (cin, cout, 'overwrite')


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
def test_dropout_mask_properties(precname):
    """nn.Dropout(0.5) (networks.py:493-494, 604-605): the RNG stream cannot match torch's, so the test pins the properties the
    training graph relies on: keep probability, 1/(1-p) scaling, same seed -> same mask (backward), different seed -> different."""
    prec = Precision.get(precname)
    real = hip()
    if DRY:
        pytest.skip('statistical test of the device hash RNG')
    x = torch.ones(2, 64, 64, 64, dtype=prec.dtype, device=DEV)
    y1, y2, y3 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    real.dropout(x, y1, 0.5, 1234)
    real.dropout(x, y2, 0.5, 1234)
    real.dropout(x, y3, 0.5, 1235)
    sync()
    y1f = y1.float()
    assert set(torch.unique(y1f).tolist()) == {0.0, 2.0}
    keep = float((y1f > 0).float().mean())
    assert abs(keep - 0.5) < 0.01, keep
    assert torch.equal(y1, y2)
    assert float((y1f != y3.float()).float().mean()) > 0.4
    # no visible structure along channels / pixels
    per_c = (y1f > 0).float().mean(dim=(0, 1, 2))
    assert float((per_c - 0.5).abs().max()) < 0.03
    # p = 0.25 on a strided (channel-slice) view, in place
    buf = torch.ones(2, 16, 16, 64, dtype=prec.dtype, device=DEV)
    real.dropout(buf[..., 32:], buf[..., 32:], 0.25, 7)
    sync()
    assert torch.all(buf[..., :32] == 1)
    k = float((buf[..., 32:].float() > 0).float().mean())
    assert abs(k - 0.75) < 0.02 and abs(float(buf[..., 32:].float().max()) - 1 / 0.75) < 1e-2


@pytest.mark.parametrize('norm_kind,nact', [('instance', L.ACT_RELU), ('batch', L.ACT_LRELU), ('instance', L.ACT_NONE)])
@pytest.mark.parametrize('c1,c2,k,s,p,n,hw', [
    (64, 64, 3, 1, 1, 2, 128),       # 128 x 64 tiles, one phase (enough tiles that the dispatch does not split K)
    (128, 128, 4, 2, 1, 4, 128),     # PatchGAN-like consumer: stride 2 -> the data gradient runs in 4 sub-pixel phases
    (256, 256, 3, 1, 1, 4, 128),     # the ResnetBlock shape: 256 x 256 tiles (8-phase kernel), 256 tiles
])
def test_norm_backward_reductions_fused_into_the_data_gradient(norm_kind, nact, c1, c2, k, s, p, n, hw):
    """conv_a -> norm -> act -> conv_b: the data gradient of conv_b (dl_conv_forward_bnstats) leaves sum(dn), sum(dn * xhat) of the norm's
    backward in its store epilogue and dl_norm_backward(ext_nchunks) skips its own pass over y and dz.  Same graph with the fusion
    switched off (the default route; DL_BNSTATS=1 enables the fusion -- measured break-even to slightly slower on the training step, ops.py): gradients agree to summation order; the fused route must actually have been taken."""
    if DRY:
        pytest.skip('needs the HIP library')
    from deepliif_amd import engine as E
    prec = Precision.get('bf16')
    be = hip()
    spec_a = ConvSpec('conv', c1, c1, 3, 1, 1, L.PAD_ZERO)
    spec_b = ConvSpec('conv', c1, c2, k, s, p, L.PAD_ZERO)
    wa0, wb0 = rnd((c1, c1, 3, 3), 1, prec, 0.05), rnd((c2, c1, k, k), 2, prec, 0.05)
    x0 = rnd((n, hw, hw, c1), 3, prec)
    ho = spec_b.out_hw(hw, hw)[0]
    g0 = rnd((n, ho, ho, c2), 4, prec)
    bn = torch.nn.BatchNorm2d(c1).to(DEV) if norm_kind == 'batch' else None
    if bn is not None:
        with torch.no_grad():
            bn.weight.copy_(torch.rand(c1, generator=torch.Generator().manual_seed(5)) + 0.5)
            bn.bias.copy_(torch.randn(c1, generator=torch.Generator().manual_seed(6)) * 0.2)
    res, taken = {}, {}
    for fused in (True, False):
        ops._BNSTATS = fused
        wa = torch.nn.Parameter(wa0.clone().to(DEV)); wa.grad = torch.zeros_like(wa)
        wb = torch.nn.Parameter(wb0.clone().to(DEV)); wb.grad = torch.zeros_like(wb)
        la, lb = E.ConvLayer(spec_a, wa, None), E.ConvLayer(spec_b, wb, None)
        if bn is not None:
            bn.weight.grad, bn.bias.grad = torch.zeros_like(bn.weight), torch.zeros_like(bn.bias)
        tape = E.Tape()
        ctx = E.Ctx(prec, tape, training=True)
        xa = E.Act(x0.to(torch.bfloat16).to(DEV), c1, True)
        y = E.conv(ctx, xa, la, stats=True)
        z = E.norm_act(ctx, y, E.NormLayer(norm_kind, c1, bn), nact)
        o = E.conv(ctx, z, lb)
        seen = []
        orig = be.norm_backward

        def spy(*a, **kw):
            seen.append(kw.get('ext_nchunks', 0))
            return orig(*a, **kw)
        be.norm_backward = spy
        try:
            o.grad = g0.to(torch.bfloat16).to(DEV)
            tape.backward()
        finally:
            be.norm_backward = orig
        sync()
        taken[fused] = seen
        res[fused] = {'dx': xa.grad.float().cpu(), 'dwa': wa.grad.cpu(), 'dwb': wb.grad.cpu()}
        if bn is not None:
            res[fused]['dgamma'], res[fused]['dbeta'] = bn.weight.grad.cpu().clone(), bn.bias.grad.cpu().clone()
    ops._BNSTATS = os.environ.get('DL_BNSTATS', '0') == '1'
    ops._impl = None
    assert taken[True] and taken[True][0] > 0, taken          # the conv epilogue produced the reductions
    assert taken[False] == [0], taken
    for key in res[True]:
        # identical inputs; the two routes differ in fp32 summation order of the reductions only (then one bf16 rounding of dy)
        assert rel(res[True][key], res[False][key]) < 4e-3, (key, rel(res[True][key], res[False][key]))


@pytest.mark.parametrize('n,h,w,cin,cout', [(2, 24, 40, 128, 3), (1, 7, 5, 16, 3), (3, 9, 33, 64, 1), (8, 256, 256, 128, 3)])
def test_narrow_transposed_conv_as_one_gemm_plus_gather(n, h, w, cin, cout):
    """UnetGenerator's outermost ConvTranspose2d(2*ngf, 3, 4, 2, 1) + Tanh behind an in-place ReLU (networks.py:573-576), inference route:
    dl_conv_forward(raw_out) with one GEMM row per (ky, kx, co) + dl_convt4_gather, against torch's conv_transpose2d on the same bf16-rounded
    operands, and against the 4-phase gather-GEMM route it replaces."""
    if DRY:
        pytest.skip('needs the HIP library')
    from deepliif_amd import engine as E
    prec = Precision.get('bf16')
    be = hip()
    spec = ConvSpec('convT', cin, cout, 4, 2, 1)
    w0 = rnd((cin, cout, 4, 4), 21, prec, 0.05)
    b0 = rnd((cout,), 22, Precision.get('fp32'), 0.1)
    x0 = rnd((n, h, w, cin), 23, prec)
    ref = torch.tanh(torch.nn.functional.conv_transpose2d(torch.relu(x0).permute(0, 3, 1, 2), w0, b0, stride=2, padding=1)).permute(0, 2, 3, 1)
    outs = {}
    for route in (True, False):
        E._CONVT4 = route
        layer = E.ConvLayer(spec, torch.nn.Parameter(w0.clone().to(DEV)), torch.nn.Parameter(b0.clone().to(DEV)))
        calls = []
        orig = be.convt4_gather
        be.convt4_gather = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            y = E.conv(E.Ctx(prec, None, training=False), E.Act(x0.to(torch.bfloat16).to(DEV), cin, False), layer, act=L.ACT_TANH, in_act=L.ACT_RELU)
        finally:
            be.convt4_gather = orig
        sync()
        assert bool(calls) == route
        assert tuple(y.t.shape) == (n, 2 * h, 2 * w, 8) and float(y.t[..., cout:].float().abs().max()) == 0.0
        outs[route] = y.t[..., :cout].float().cpu()
        assert rel(outs[route], ref) < 8e-3, (route, rel(outs[route], ref))
    E._CONVT4 = os.environ.get('DL_CONVT4', '1') != '0'
    ops._impl = None
    assert rel(outs[True], outs[False]) < 8e-3


@pytest.mark.parametrize('precname', ['fp32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 9, 7, 64), (1, 16, 16, 512), (3, 5, 4, 8)])
def test_attention_gate_and_sigmoid(shape, precname):
    """dl_gate_forward / dl_gate_backward (x * psi, one-channel psi broadcast over the channels; att_unet.py:113-115) and the sigmoid entry of
    dl_act_forward / dl_act_backward against the formula emulation"""
    prec = Precision.get(precname)
    x = rnd(shape, 1, prec).to(prec.dtype)
    g = rnd(shape, 2, prec).to(prec.dtype)
    psi = torch.zeros(shape[:3] + (8,))
    psi[..., 0] = torch.sigmoid(rnd(shape[:3], 3, prec))
    psi = (psi.to(prec.dtype).float() if prec.dtype == torch.bfloat16 else psi).to(prec.dtype)
    fake, real = fake_backend.FakeBackend(), hip()
    out_f, out_r = torch.empty(shape, dtype=prec.dtype), torch.empty(shape, dtype=prec.dtype, device=DEV)
    fake.gate_forward(x, psi, out_f)
    real.gate_forward(x.to(DEV), psi.to(DEV), out_r)
    sync()
    assert rel(out_r, out_f) < (1e-6 if precname == 'fp32' else 1e-2)
    dx_f, dpsi_f = torch.empty(shape, dtype=prec.dtype), torch.empty(psi.shape, dtype=prec.dtype)
    dx_r, dpsi_r = torch.empty(shape, dtype=prec.dtype, device=DEV), torch.full(psi.shape, 7.0, dtype=prec.dtype, device=DEV)
    fake.gate_backward(g, x, psi, dx_f, dpsi_f)
    real.gate_backward(g.to(DEV), x.to(DEV), psi.to(DEV), dx_r, dpsi_r)
    sync()
    assert rel(dx_r, dx_f) < (1e-6 if precname == 'fp32' else 1e-2)
    assert rel(dpsi_r, dpsi_f) < (1e-5 if precname == 'fp32' else 1e-2) and float(dpsi_r[..., 1:].abs().max()) == 0.0
    real.gate_backward(g.to(DEV), x.to(DEV), psi.to(DEV), None, dpsi_r)           # dx is optional
    sync()
    assert rel(dpsi_r, dpsi_f) < (1e-5 if precname == 'fp32' else 1e-2)
    y_f, y_r = torch.empty(shape, dtype=prec.dtype), torch.empty(shape, dtype=prec.dtype, device=DEV)
    fake.act_forward(L.ACT_SIGMOID, x, y_f)
    real.act_forward(L.ACT_SIGMOID, x.to(DEV), y_r)
    sync()
    assert rel(y_r, y_f) < (2e-6 if precname == 'fp32' else 1e-2)
    fake.act_backward(L.ACT_SIGMOID, g, y_f, dx_f)
    real.act_backward(L.ACT_SIGMOID, g.to(DEV), y_f.to(DEV), dx_r)
    sync()
    assert rel(dx_r, dx_f) < (2e-6 if precname == 'fp32' else 1e-2)


def _split_copy(t):
    """host-side model of store_split8 (csrc/common.h): per group of 8 channels [8 bf16 hi | 8 bf16 lo] in the bytes of the fp32 group"""
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    g = torch.stack([hi.reshape(*t.shape[:3], -1, 8), lo.reshape(*t.shape[:3], -1, 8)], dim=4)       # [N,H,W,groups,2,8] bf16
    return g.contiguous().view(torch.int16).reshape(*t.shape[:3], -1).view(torch.float32).reshape(t.shape)


@pytest.mark.parametrize('case', [('conv', 256, 256, 3, 1, 1, 8, 64, 128), ('conv', 128, 256, 3, 2, 1, 4, 64, 64), ('convT', 256, 128, 3, 2, 1, 2, 32, 32),
                                  ('conv', 64, 128, 4, 2, 1, 2, 48, 40), ('conv', 512, 512, 4, 1, 1, 2, 9, 9)], ids=lambda c: f'{c[0]}{c[1]}-{c[2]}k{c[3]}s{c[4]}')
def test_split_copy_inputs_are_bit_identical_to_the_in_kernel_split(case):
    """strict policy: the norm kernels can write a SPLIT COPY of their fp32 output (hi = bf16(v), lo = bf16(v - hi) per group of 8 channels) and the
    direct-to-LDS strict kernels read it instead of splitting the fp32 values while staging (dl_conv_desc.in_split, dl_wgrad_desc.p_split /
    q_split).  Same split formula on both routes -> forward, data gradient and weight gradient must be BIT-IDENTICAL; and the copy the norm
    kernels write must be exactly the host model of the layout."""
    kind, cin, cout, k, s_, p, N, H, W_ = case
    prec = Precision.get('fp32')
    spec = ConvSpec(kind, cin, cout, k, s_, p, L.PAD_ZERO, 1 if (kind == 'convT' and k == 3) else 0)
    wshape = (cout, cin, k, k) if kind == 'conv' else (cin, cout, k, k)
    w = rnd(wshape, 1, prec, 0.05).to(DEV)
    x = rnd((N, H, W_, cin), 3, prec).to(DEV)
    ho, wo = spec.out_hw(H, W_)
    dy = rnd((N, ho, wo, cout), 4, prec).to(DEV)
    real = hip()
    # the norm kernels' split outputs
    zs = torch.empty_like(x)
    z = torch.empty_like(x)
    st = real.norm_forward(x, z, cin, L.NORM_INSTANCE, L.ACT_RELU, None, None, None, None, -1.0, None, z_split=zs)
    sync()
    assert torch.equal(zs.view(torch.int32), _split_copy(z).view(torch.int32)), 'dl_norm_forward(z_split)'
    dys = torch.empty_like(x)
    dyo = torch.empty_like(x)
    real.norm_backward(rnd((N, H, W_, cin), 5, prec).to(DEV), x, dyo, st, cin, L.NORM_INSTANCE, L.ACT_RELU, None, None, None, None, dy_split=dys)
    sync()
    assert torch.equal(dys.view(torch.int32), _split_copy(dyo).view(torch.int32)), 'dl_norm_backward(dy_split)'
    xs, dys = _split_copy(x), _split_copy(dy)
    for plan_kind, src, srcs in (('fwd', x, xs), ('dgrad', dy, dys)):
        plan = spec.forward_plan() if plan_kind == 'fwd' else spec.dgrad_plan()
        packed = ops.PackedWeights(plan, DEV, True)
        real.pack_weights(packed, w)
        if plan_kind == 'fwd':
            oshape, (hq, wq) = (N, ho, wo, cpad(cout)), ((ho, wo) if kind == 'conv' else (H, W_))
        else:
            oshape = (N, H, W_, cpad(cin))
            hq, wq = ((H + 1) // 2, (W_ + 1) // 2) if (kind == 'conv' and s_ == 2) else (H, W_)
        a, b = torch.empty(oshape, device=DEV), torch.empty(oshape, device=DEV)
        real.conv_forward(packed, src, a, hq, wq, None, L.ACT_NONE, L.ACT_NONE, prec.prec)
        k_plain = real.last_conv_kernel
        real.conv_forward(packed, srcs, b, hq, wq, None, L.ACT_NONE, L.ACT_NONE, prec.prec, in_split=True)
        sync()
        assert 'x3' in real.last_conv_kernel, real.last_conv_kernel
        if real.last_conv_kernel == k_plain:
            assert torch.equal(a, b), plan_kind
        else:
            # the ResnetBlock shape: split copies go to conv_gemm_w4x3_kernel (r04), fp32 inputs stay on the 8-phase strict kernel -- same split
            # formula and products, another summation order (32x32x16 fragments, (chunk, kh, kw) K order)
            # likewise the fused four-phase tile: split copies go to conv_s2f_x3_kernel (r04), fp32 inputs stay on the 4-phase strict kernel
            assert real.last_conv_kernel in ('conv_gemm_w4x3_kernel', 'conv_s2f_x3_kernel') and rel(b, a) < 2e-6, (plan_kind, real.last_conv_kernel, rel(b, a))
    gshape = wshape
    P, Ps, Q, Qs = (dy, dys, x, xs) if kind == 'conv' else (x, xs, dy, dys)
    g0 = torch.empty(gshape, device=DEV)
    real.conv_wgrad(P, Q, g0, k, s_, p, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, prec.prec, False)
    for ps, qs in ((True, True), (True, False), (False, True)):
        g1 = torch.empty(gshape, device=DEV)
        real.conv_wgrad(Ps if ps else P, Qs if qs else Q, g1, k, s_, p, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, prec.prec, False, p_split=ps, q_split=qs)
        sync()
        assert torch.equal(g0, g1), ('wgrad', ps, qs)


@pytest.mark.parametrize('case', [(256, 256, 8, 128), (64, 256, 8, 128), (256, 512, 4, 128)], ids=lambda c: f'ci{c[0]}-co{c[1]}-n{c[2]}-h{c[3]}')
def test_strict_w4_kernel_on_split_copies(case):
    """The ResnetBlock conv under the strict policy with a SPLIT-COPY input -- forward (fused statistics, bias) and data gradient (reversed kw order)
    against the fp32 reference of the emulation backend at fp32-class tolerance, run-to-run identical, image borders included by construction of the
    shape.  Default dispatch: conv_gemm_8ph_x3_kernel; with DL_CONV_W4X3=1 (tests/test_gpu_switches.py) conv_gemm_w4x3_kernel (csrc/conv_w4x3.hip)."""
    cin, cout, N, H = case
    W_ = 128
    prec = Precision.get('fp32')
    spec = ConvSpec('conv', cin, cout, 3, 1, 1, L.PAD_ZERO, 0)
    w = rnd((cout, cin, 3, 3), 1, prec, 0.05)
    bias = rnd((cout,), 2, prec, 0.1)
    x = rnd((N, H, W_, cin), 3, prec)
    dy = rnd((N, H, W_, cout), 4, prec)
    fake, real = fake_backend.FakeBackend(), hip()
    exp_f = _run_conv(fake, 'fwd', spec, prec, x, w, bias, L.ACT_NONE, L.ACT_NONE, H, W_)
    exp_d = _run_conv(fake, 'dgrad', spec, prec, dy, w, None, L.ACT_NONE, L.ACT_NONE, H, W_)
    for plan_kind, src, exp, b in (('fwd', x, exp_f, bias), ('dgrad', dy, exp_d, None)):
        plan = spec.forward_plan() if plan_kind == 'fwd' else spec.dgrad_plan()
        packed = ops.PackedWeights(plan, DEV, True)
        real.pack_weights(packed, w.to(DEV))
        srcs = _split_copy(src.to(DEV))
        outs = []
        for rep in range(2):
            o = torch.empty(exp.shape, device=DEV)
            nch = real.conv_forward(packed, srcs, o, H, W_, None if b is None else b.to(DEV), L.ACT_NONE, L.ACT_NONE, prec.prec, in_split=True,
                                    want_stats=(plan_kind == 'fwd'))
            sync()
            outs.append(o)
        if os.environ.get('DL_CONV_W4X3') == '1' and plan_kind == 'fwd':          # opt-in kernel (tests/test_gpu_switches.py runs this test under the switch)
            assert real.last_conv_kernel == 'conv_gemm_w4x3_kernel', real.last_conv_kernel
        assert torch.equal(outs[0], outs[1]), 'run-to-run difference'
        assert rel(outs[0], exp) < 2e-5, (plan_kind, rel(outs[0], exp))
        if plan_kind == 'fwd':
            assert nch == H * W_ // 256
            z1, z2 = torch.empty_like(outs[0]), torch.empty_like(outs[0])
            st1 = real.norm_forward(outs[0], z1, cout, L.NORM_INSTANCE, L.ACT_RELU, None, None, None, None, -1.0, None, ext_nchunks=nch)
            st2 = real.norm_forward(outs[0], z2, cout, L.NORM_INSTANCE, L.ACT_RELU, None, None, None, None, -1.0, None)
            sync()
            assert rel(st1[0], st2[0]) < 1e-5 and rel(st1[1], st2[1]) < 1e-5 and rel(z1, z2) < 1e-5



@pytest.mark.parametrize('case', [('convT', 128, 64, 3, 1, 256, 256, 'fwd'), ('convT', 256, 128, 3, 2, 128, 256, 'fwd'), ('conv', 64, 128, 3, 1, 512, 512, 'dgrad'),
                                  ('conv', 64, 128, 4, 1, 512, 512, 'dgrad'), ('conv', 128, 256, 4, 4, 200, 328, 'dgrad')],
                         ids=lambda c: f'{c[0]}{c[1]}-{c[2]}k{c[3]}n{c[4]}-{c[5]}x{c[6]}-{c[7]}')
def test_strict_fused_stride2_tile_on_split_copies(case):
    """conv_s2f_x3_kernel (csrc/conv_s2f_x3.hip): the stride-2 layers under the strict policy with a SPLIT-COPY input -- ConvTranspose2d(3, 2, 1, 1) forward
    (bias, fused statistics over pixels and phases) and the data gradients of Conv2d(3 | 4, 2, 1) (4 / 9 distinct input offsets), at sizes that pass the
    kernel's size rule (>= 256 tiles of 256 phase-grid pixels) incl. a ragged last tile; against the fp32 reference of the emulation backend at fp32-class
    tolerance, run-to-run identical, and within 2e-6 of the 4-phase strict kernel that fp32 inputs keep."""
    kind, cin, cout, k, N, H, W_, plan_kind = case
    prec = Precision.get('fp32')
    spec = ConvSpec(kind, cin, cout, k, 2, 1, L.PAD_ZERO, 1 if kind == 'convT' else 0)
    wshape = (cout, cin, k, k) if kind == 'conv' else (cin, cout, k, k)
    w = rnd(wshape, 1, prec, 0.05)
    fake, real = fake_backend.FakeBackend(), hip()
    if plan_kind == 'fwd':
        src, bias = rnd((N, H, W_, cin), 3, prec), rnd((cout,), 2, prec, 0.1)
        hq, wq = H, W_
    else:
        ho, wo = spec.out_hw(H, W_)
        src, bias = rnd((N, ho, wo, cout), 4, prec), None
        hq, wq = (H + 1) // 2, (W_ + 1) // 2
    exp = _run_conv(fake, plan_kind, spec, prec, src, w, bias, L.ACT_NONE, L.ACT_NONE, H, W_)
    plan = spec.forward_plan() if plan_kind == 'fwd' else spec.dgrad_plan()
    packed = ops.PackedWeights(plan, DEV, True)
    real.pack_weights(packed, w.to(DEV))
    srcd = src.to(DEV)
    srcs = _split_copy(srcd)
    want = plan_kind == 'fwd'
    outs = []
    for rep in range(2):
        o = torch.full(exp.shape, 7.0, device=DEV)
        nch = real.conv_forward(packed, srcs, o, hq, wq, None if bias is None else bias.to(DEV), L.ACT_NONE, L.ACT_NONE, prec.prec, in_split=True, want_stats=want)
        sync()
        outs.append(o)
    assert (real.last_conv_kernel == 'conv_s2f_x3_kernel') == (os.environ.get('DL_CONV_S2F') != '0' and os.environ.get('DL_CONV_S2FX3') != '0'), real.last_conv_kernel
    assert torch.equal(outs[0], outs[1]), 'run-to-run difference'
    assert rel(outs[0], exp) < 2e-5, (plan_kind, rel(outs[0], exp))
    plain = torch.empty(exp.shape, device=DEV)
    real.conv_forward(packed, srcd, plain, hq, wq, None if bias is None else bias.to(DEV), L.ACT_NONE, L.ACT_NONE, prec.prec)
    sync()
    assert 'x3' in real.last_conv_kernel and rel(outs[0], plain) < 2e-6, (real.last_conv_kernel, rel(outs[0], plain))
    if want:
        assert nch == (hq * wq) // 256 if (hq * wq) % 256 == 0 else nch == 0
        if nch:
            z1, z2 = torch.empty_like(outs[0]), torch.empty_like(outs[0])
            st1 = real.norm_forward(outs[0], z1, cout, L.NORM_INSTANCE, L.ACT_RELU, None, None, None, None, -1.0, None, ext_nchunks=nch)
            st2 = real.norm_forward(outs[0], z2, cout, L.NORM_INSTANCE, L.ACT_RELU, None, None, None, None, -1.0, None)
            sync()
            assert rel(st1[0], st2[0]) < 1e-5 and rel(st1[1], st2[1]) < 1e-5 and rel(z1, z2) < 1e-5
