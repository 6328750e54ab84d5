"""Host-side geometry (deepliif_amd/geometry.py): the gather-GEMM / pack / wgrad formulas of include/deepliif_hip.h, emulated
literally on CPU (tests/emu.py), must reproduce torch's Conv2d / ConvTranspose2d forward, data-gradient and weight-gradient
for every layer shape on the path (networks.py:386-444, 576-609, 638-660).  fp64, tolerance 1e-10."""
import pytest

import fake_backend
import torch
import torch.nn.functional as F

from deepliif_amd import _lib as L
from deepliif_amd.geometry import ConvSpec, cpad
from emu import emu_gather_gemm, emu_pack, emu_wgrad, from_nhwc, to_nhwc

CASES = [
    # kind, cin, cout, k, stride, pad, pad_mode, out_pad, H
    ('conv', 3, 8, 7, 1, 3, L.PAD_ZERO, 0, 9),        # resnet stem, zero pad
    ('conv', 3, 8, 7, 1, 3, L.PAD_REFLECT, 0, 9),     # resnet stem, reflect pad
    ('conv', 8, 16, 3, 2, 1, L.PAD_ZERO, 0, 8),       # resnet down
    ('conv', 16, 16, 3, 1, 1, L.PAD_ZERO, 0, 6),      # resnet block
    ('conv', 16, 16, 3, 1, 1, L.PAD_REFLECT, 0, 6),   # resnet block, reflect
    ('convT', 16, 8, 3, 2, 1, L.PAD_ZERO, 1, 5),      # resnet up
    ('conv', 8, 3, 7, 1, 3, L.PAD_ZERO, 0, 8),        # resnet head
    ('conv', 6, 8, 4, 2, 1, L.PAD_ZERO, 0, 8),        # D first / unet down
    ('conv', 8, 8, 4, 1, 1, L.PAD_ZERO, 0, 6),        # D stride-1 tail (H -> H-1)
    ('conv', 8, 1, 4, 1, 1, L.PAD_ZERO, 0, 5),        # D head
    ('convT', 16, 8, 4, 2, 1, L.PAD_ZERO, 0, 3),      # unet up
    ('convT', 16, 3, 4, 2, 1, L.PAD_ZERO, 0, 4),      # unet outermost up
    ('conv', 8, 8, 4, 2, 1, L.PAD_ZERO, 0, 2),        # unet innermost down (2x2 -> 1x1)
    ('convT', 8, 8, 4, 2, 1, L.PAD_ZERO, 0, 1),       # unet innermost up (1x1 -> 2x2)
]


@pytest.mark.parametrize('kind,cin,cout,k,s,p,pm,op,H', CASES)
def test_layer_formulas(kind, cin, cout, k, s, p, pm, op, H):
    torch.manual_seed(0)
    dt = torch.float64
    N, W_ = 2, H + 1
    spec = ConvSpec(kind, cin, cout, k, s, p, pm, op)
    x = torch.randn(N, cin, H, W_, dtype=dt, requires_grad=True)
    if kind == 'conv':
        w = torch.randn(cout, cin, k, k, dtype=dt, requires_grad=True)
        xp = F.pad(x, (p, p, p, p), mode='reflect') if pm == L.PAD_REFLECT else x
        y = F.conv2d(xp, w, stride=s, padding=0 if pm == L.PAD_REFLECT else p)
    else:
        w = torch.randn(cin, cout, k, k, dtype=dt, requires_grad=True)
        y = F.conv_transpose2d(x, w, stride=s, padding=p, output_padding=op)
    Ho, Wo = y.shape[2], y.shape[3]
    assert (Ho, Wo) == spec.out_hw(H, W_)
    r = torch.randn_like(y)
    dx_ref, dw_ref = torch.autograd.grad((y * r).sum(), [x, w])

    # ---- forward
    fp = spec.forward_plan()
    Wp = emu_pack(fp, w.detach())
    hq, wq = (Ho, Wo) if kind == 'conv' else (H, W_)
    out = emu_gather_gemm(fp, to_nhwc(x.detach(), cpad(cin)), Wp, Ho, Wo, hq, wq, cpad(cout))
    assert torch.allclose(from_nhwc(out, cout), y.detach(), atol=1e-10)
    assert out[..., cout:].abs().max() == 0 if cpad(cout) > cout else True

    # ---- data gradient
    if pm == L.PAD_ZERO:
        dp = spec.dgrad_plan()
        Wd = emu_pack(dp, w.detach())
        if kind == 'conv':
            # stride 2: four sub-pixel phases over a ceil(H/2) x ceil(W/2) grid; with an odd size the odd phases have one row /
            # column less and the kernels drop the outputs that fall outside dx (W_ = H + 1 makes one of the two sizes odd)
            hq, wq = (H, W_) if s == 1 else ((H + 1) // 2, (W_ + 1) // 2)
        else:
            hq, wq = H, W_
        dx = emu_gather_gemm(dp, to_nhwc(r, cpad(cout)), Wd, H, W_, hq, wq, cpad(cin))
        assert torch.allclose(from_nhwc(dx, cin), dx_ref, atol=1e-10)

    if pm == L.PAD_REFLECT:
        # gradient w.r.t. the explicitly padded input (pad-0 plan over the (H+2p) x (W+2p) extent), then the reflection fold
        dp = spec.dgrad_plan()
        Wd = emu_pack(dp, w.detach())
        Hp, Wpd = H + 2 * p, W_ + 2 * p
        dxp = emu_gather_gemm(dp, to_nhwc(r, cpad(cout)), Wd, Hp, Wpd, Hp, Wpd, cpad(cin))
        dx = torch.zeros(N, H, W_, cpad(cin), dtype=dt)
        fake_backend.FakeBackend().reflect_fold(dxp, dx, p)
        assert torch.allclose(from_nhwc(dx.to(dt), cin), dx_ref, atol=1e-5)       # the formula backend folds in fp32

    # ---- weight gradient
    if kind == 'conv':
        g = emu_wgrad(to_nhwc(r, cpad(cout)), to_nhwc(x.detach(), cpad(cin)), k, k, s, p, pm, cout, cin)
    else:
        g = emu_wgrad(to_nhwc(x.detach(), cpad(cin)), to_nhwc(r, cpad(cout)), k, k, s, p, L.PAD_ZERO, cin, cout)
    assert torch.allclose(g, dw_ref, atol=1e-10)


@pytest.mark.parametrize('hw', [(8, 6), (9, 6), (8, 7), (25, 19), (5, 5), (3, 2)])
def test_stride2_dgrad_even_and_odd_sizes(hw):
    """4-phase data-gradient of the stride-2 convs for both kernel sizes, on even grids (the 512-pixel path) and on odd ones
    (arbitrary tile sizes: torch handles them, so must we)"""
    dt = torch.float64
    H, W_ = hw
    for k in (3, 4):
        spec = ConvSpec('conv', 8, 16, k, 2, 1)
        x = torch.randn(1, 8, H, W_, dtype=dt, requires_grad=True)
        w = torch.randn(16, 8, k, k, dtype=dt)
        y = F.conv2d(x, w, stride=2, padding=1)
        r = torch.randn_like(y)
        dx_ref, = torch.autograd.grad((y * r).sum(), [x])
        dp = spec.dgrad_plan()
        dx = emu_gather_gemm(dp, to_nhwc(r, 16), emu_pack(dp, w), H, W_, (H + 1) // 2, (W_ + 1) // 2, 8)
        assert torch.allclose(from_nhwc(dx, 8), dx_ref, atol=1e-10)


# ---------------------------------------------------------------------------------------------------------------
# LDS layout of the strict-policy direct-to-LDS kernels (csrc/conv_x3.h): bank-conflict model of ds_read_b128
# ---------------------------------------------------------------------------------------------------------------
# ds_read_b128 is serviced in four 16-lane groups; bank of byte address a = (a / 4) % 64 (MI355X_MICROARCH.md, LDS table)
_B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
_B128_GROUPS += [[l + 32 for l in g] for g in _B128_GROUPS]


def _b128_worst_way(addr_of_lane):
    worst = 0
    for g in _B128_GROUPS:
        slots = {}
        for lane in g:
            a = addr_of_lane(lane)
            slots.setdefault((a // 16) % 16, set()).add(a)
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst


def _x3_swz(h):          # conv_x3.h: x3_swz
    return (((h >> 2) & 1) << 2) | ((h & 1) << 1) | (((h >> 1) ^ (h >> 2)) & 1)


def test_x3_swizzle_is_conflict_free():
    """fp32 activation rows (128 B = 32 channels): lane (fr, fg) reads the 16-byte chunks 2*fg and 2*fg + 1 of row fr.  With the bf16 kernels'
    swizzle c ^ ((row >> 1) & 7) those reads are 2-way conflicting; with c ^ x3_swz((row >> 1) & 7) they are conflict-free.  The weight rows
    ([32 hi | 32 lo] bf16) read chunk fg and 4 + fg under the old swizzle, as the bf16 kernel does for kk = 0 / 1."""
    assert sorted(_x3_swz(h) for h in range(8)) == list(range(8)), 'the swizzle must be a permutation (source-side / read-side involution)'
    for e in range(2):
        old = _b128_worst_way(lambda l: (l & 15) * 128 + (((2 * (l >> 4) + e) ^ (((l & 15) >> 1) & 7)) << 4))
        new = _b128_worst_way(lambda l: (l & 15) * 128 + (((2 * (l >> 4) + e) ^ _x3_swz(((l & 15) >> 1) & 7)) << 4))
        assert old == 2 and new == 1, (e, old, new)
        w = _b128_worst_way(lambda l: (l & 15) * 128 + (((e * 4 + (l >> 4)) ^ (((l & 15) >> 1) & 7)) << 4))
        assert w == 1, (e, w)


def test_conv_descriptor_cache_hands_out_independent_copies():
    """fill_conv_desc keeps the filled descriptor on the plan (r05: ~10-24 us of ctypes field stores per launch) -- callers set in_split / splitk on what they get,
    so every call must return its OWN copy, equal to a freshly filled one"""
    from deepliif_amd.geometry import ConvSpec, fill_conv_desc, _fill_conv_desc
    from deepliif_amd import _lib as L
    plan = ConvSpec('conv', 64, 128, 3, 2, 1).forward_plan()
    args = (2, 64, 64, 64, 32, 32, 128, 128, 32, 32, L.DL_BF16, L.PREC_BF16, L.ACT_RELU, L.ACT_NONE, 128, 1)
    a = fill_conv_desc(plan, *args)
    a.in_split, a.splitk = 1, 7
    b = fill_conv_desc(plan, *args)
    assert (b.in_split, b.splitk) == (0, 1) and b is not a
    assert bytes(b) == bytes(_fill_conv_desc(plan, *args))
    c = fill_conv_desc(plan, *(args[:-1] + (4,)))                 # another split-K: another entry
    assert c.splitk == 4 and bytes(fill_conv_desc(plan, *args)) == bytes(b)
