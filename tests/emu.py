"""Literal CPU emulation of the C-ABI kernel *formulas* (include/deepliif_hip.h), used by the CPU test-suite to check the
host-side geometry / packing logic against torch's own convolutions, and as the fake backend for host-logic tests.
Test infrastructure only."""
import numpy as np
import torch

from deepliif_amd import _lib as L
from deepliif_amd.geometry import GatherPlan


def emu_pack(plan: GatherPlan, src: torch.Tensor) -> torch.Tensor:
    """dl_pack_weights: src[A][B][KH][KW] -> W[rows_pad][kstride] (fp32 here, no bf16 rounding)."""
    A, B = src.shape[0], src.shape[1]
    W = torch.zeros(plan.rows_pad, plan.kstride, dtype=src.dtype)
    for ph, taps in enumerate(plan.phase_taps):
        for tl, (_, _, kh, kw) in enumerate(taps):
            k0 = plan.kbase[ph] + tl * plan.cc_pad
            blk = src[:, :, kh, kw]                      # [A][B]
            blk = blk if plan.row_is_a else blk.t()      # [row][contracted]
            W[:plan.rows_real, k0:k0 + plan.cc_real] = blk
    return W


def _reflect(i, n):
    i = -i if i < 0 else i
    return 2 * n - 2 - i if i >= n else i


def emu_gather_gemm(plan: GatherPlan, x: torch.Tensor, W: torch.Tensor, ho: int, wo: int, hq: int, wq: int, co_pad: int) -> torch.Tensor:
    """dl_conv_forward without bias/activation: x [N,Hi,Wi,Cp] -> out [N,Ho,Wo,co_pad]."""
    N, Hi, Wi, Cp = x.shape
    assert Cp == plan.cc_pad
    out = torch.zeros(N, ho, wo, co_pad, dtype=x.dtype)
    for ph, taps in enumerate(plan.phase_taps):
        oh, ow = plan.phase_off[ph]
        for tl, (dh, dw, _, _) in enumerate(taps):
            k0 = plan.kbase[ph] + tl * plan.cc_pad
            Wt = W[:co_pad, k0:k0 + plan.cc_pad]                     # [co][ci]
            for a in range(hq):
                if a * plan.out_step + oh >= ho:         # odd output size: this phase has one row less than the phase grid
                    continue
                hi = a * plan.in_step + dh
                if plan.pad_mode == L.PAD_REFLECT:
                    hi = _reflect(hi, Hi)
                elif not (0 <= hi < Hi):
                    continue
                for b in range(wq):
                    if b * plan.out_step + ow >= wo:
                        continue
                    wi = b * plan.in_step + dw
                    if plan.pad_mode == L.PAD_REFLECT:
                        wi = _reflect(wi, Wi)
                    elif not (0 <= wi < Wi):
                        continue
                    out[:, a * plan.out_step + oh, b * plan.out_step + ow, :] += x[:, hi, wi, :] @ Wt.t()
    return out


def emu_wgrad(P: torch.Tensor, Q: torch.Tensor, KH, KW, step, pad, pad_mode, CA, CB) -> torch.Tensor:
    """dl_conv_wgrad: grad[a][b][kh][kw] = sum_{n,hp,wp} P[n,hp,wp,a] * Q[n, hp*step-pad+kh, wp*step-pad+kw, b]."""
    N, Hp, Wp, _ = P.shape
    _, Hq, Wq, _ = Q.shape
    g = torch.zeros(CA, CB, KH, KW, dtype=P.dtype)
    for kh in range(KH):
        for kw in range(KW):
            for hp in range(Hp):
                h = hp * step - pad + kh
                if pad_mode == L.PAD_REFLECT:
                    h = _reflect(h, Hq)
                elif not (0 <= h < Hq):
                    continue
                for wp in range(Wp):
                    w = wp * step - pad + kw
                    if pad_mode == L.PAD_REFLECT:
                        w = _reflect(w, Wq)
                    elif not (0 <= w < Wq):
                        continue
                    g[:, :, kh, kw] += P[:, hp, wp, :CA].t() @ Q[:, h, w, :CB]
    return g


def to_nhwc(x: torch.Tensor, cp: int) -> torch.Tensor:
    N, C, H, W = x.shape
    out = torch.zeros(N, H, W, cp, dtype=x.dtype)
    out[..., :C] = x.permute(0, 2, 3, 1)
    return out


def from_nhwc(x: torch.Tensor, c: int) -> torch.Tensor:
    return x[..., :c].permute(0, 3, 1, 2).contiguous()
