"""CPU emulation of the ops backend (deepliif_amd/ops.py HipBackend) -- TEST INFRASTRUCTURE ONLY.

It implements the documented formula of every C-ABI entry point (include/deepliif_hip.h) with plain torch CPU ops, so the
CPU test-suite can run the *host* logic (geometry, tape autograd, network programs, model step, optimizer, data-parallel
exchange) without a GPU and compare it with the oracle.  It is never importable from the product package.
"""
import torch

from deepliif_amd import _lib as L
from deepliif_amd import ops


def _act(act, v):
    if act == L.ACT_RELU:
        return torch.relu(v)
    if act == L.ACT_LRELU:
        return torch.where(v > 0, v, 0.2 * v)
    if act == L.ACT_TANH:
        return torch.tanh(v)
    if act == L.ACT_SIGMOID:
        return torch.sigmoid(v)
    return v


def _act_grad_from_output(act, y):
    if act == L.ACT_RELU:
        return (y > 0).to(y.dtype)
    if act == L.ACT_LRELU:
        return torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.2))
    if act == L.ACT_TANH:
        return 1 - y * y
    if act == L.ACT_SIGMOID:
        return y * (1 - y)
    return torch.ones_like(y)


def _reflect(idx, n):
    idx = idx.abs()
    return torch.where(idx >= n, 2 * n - 2 - idx, idx)


def _gather(x, hq, wq, step, dh, dw, reflect):
    """x [N,H,W,C] -> g[n,a,b,:] = x[n, a*step+dh, b*step+dw, :] (zero / reflect outside)."""
    N, H, W, C = x.shape
    hi = torch.arange(hq) * step + dh
    wi = torch.arange(wq) * step + dw
    if reflect:
        hi, wi = _reflect(hi, H), _reflect(wi, W)
        return x[:, hi][:, :, wi]
    mh, mw = (hi >= 0) & (hi < H), (wi >= 0) & (wi < W)
    g = x[:, hi.clamp(0, H - 1)][:, :, wi.clamp(0, W - 1)]
    return g * (mh[:, None] & mw[None, :]).to(x.dtype)[None, :, :, None]


class FakeBackend:
    emulates_any_half = True        # storage rounding follows the tensors' dtype (bf16 or fp16): ops.impl() returns this object in either half mode

    def __init__(self):
        self.calls = {}

    def _count(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1

    def pack_batch_build(self, jobs):
        self._count('pack_batch_build')
        return list(jobs)

    def pack_batch_run(self, table, count):
        self._count('pack_batch')
        assert len(table) == count
        for packed, src in table:
            self.pack_weights(packed, src)

    def pack_weights(self, packed, src):
        self._count('pack')
        plan = packed.plan
        W = torch.zeros(plan.rows_pad, plan.kstride, dtype=torch.float32)
        for ph, taps in enumerate(plan.phase_taps):
            for tl, (_, _, kh, kw) in enumerate(taps):
                k0 = plan.kbase[ph] + tl * plan.cc_pad
                if plan.stack_kw:            # row = a*KW + kw, taps give kh only
                    blk = src[:, :, kh, :].float().permute(0, 2, 1).reshape(-1, src.shape[1])      # [(a,kw)][b]
                    W[:plan.rows_real, k0:k0 + plan.cc_real] = blk
                    continue
                blk = src[:, :, kh, kw].float()
                W[:plan.rows_real, k0:k0 + plan.cc_real] = blk if plan.row_is_a else blk.t()
        packed.fake_w = W

    def norm_ws_token(self):
        return 0

    supports_split = False          # the emulation works on fp32 tensors: split copies (engine.norm_act) are a GPU-side storage detail

    def conv_forward(self, packed, x, out, hq, wq, bias, act, in_act, prec, splitk=None, raw_out=False, want_stats=False, bn=None, in_split=False):
        self._count('conv')
        plan = packed.plan
        xv = _act(in_act, x.float())
        cop = out.shape[3]
        acc = torch.zeros(out.shape, dtype=torch.float32)
        for ph, taps in enumerate(plan.phase_taps):
            oh, ow = plan.phase_off[ph]
            for tl, (dh, dw, _, _) in enumerate(taps):
                k0 = plan.kbase[ph] + tl * plan.cc_pad
                Wt = packed.fake_w[:cop, k0:k0 + plan.cc_pad]
                g = _gather(xv, hq, wq, plan.in_step, dh, dw, plan.pad_mode == L.PAD_REFLECT)
                dst = acc[:, oh::plan.out_step, ow::plan.out_step, :][:, :hq, :wq]          # odd sizes: an odd phase is one row / column shorter
                dst += (g @ Wt.t())[:, :dst.shape[1], :dst.shape[2]]
        if bias is not None:
            acc[..., :bias.numel()] += bias.float()
        out.copy_(_act(act, acc).to(out.dtype))

    def conv_wgrad(self, P, Q, grad, k, step, pad, pad_mode, p_act, q_act, prec, accumulate, splitk=None, stack_kw=0, p_split=False, q_split=False):
        self._count('wgrad')
        Pv, Qv = _act(p_act, P.float()), _act(q_act, Q.float())
        CA, CB = grad.shape[0], grad.shape[1]
        N, Hp, Wp, _ = Pv.shape
        g = torch.zeros_like(grad)
        if stack_kw:          # P channels = (a, kw); vertical taps only
            for kh in range(k):
                qg = _gather(Qv, Hp, Wp, step, kh - pad, 0, pad_mode == L.PAD_REFLECT)
                r = torch.einsum('nhwa,nhwb->ab', Pv[..., :CA * stack_kw], qg[..., :CB])       # [(a,kw)][b]
                g[:, :, kh, :] = r.view(CA, stack_kw, CB).permute(0, 2, 1)
            if accumulate:
                grad.add_(g)
            else:
                grad.copy_(g)
            return
        for kh in range(k):
            for kw in range(k):
                qg = _gather(Qv, Hp, Wp, step, kh - pad, kw - pad, pad_mode == L.PAD_REFLECT)
                g[:, :, kh, kw] = torch.einsum('nhwa,nhwb->ab', Pv[..., :CA], qg[..., :CB])
        if accumulate:
            grad.add_(g)
        else:
            grad.copy_(g)

    def norm_forward(self, y, z, C_real, scope, act, gamma, beta, running_mean, running_var, momentum, residual, ext_nchunks=0, z_split=None):
        assert ext_nchunks == 0
        self._count('norm_fwd')
        yv = y.float()
        dims = (0, 1, 2) if scope == L.NORM_BATCH else (1, 2)
        mean = yv.mean(dim=dims, keepdim=True)
        var = yv.var(dim=dims, unbiased=False, keepdim=True)
        rstd = 1.0 / torch.sqrt(var + 1e-5)
        Cp = y.shape[3]
        ga = torch.zeros(Cp)
        be = torch.zeros(Cp)
        ga[:C_real] = gamma.float() if gamma is not None else 1.0
        if beta is not None:
            be[:C_real] = beta.float()
        scale = ga * rstd
        shift = be - mean * scale
        if running_mean is not None and momentum >= 0:
            cnt = yv.numel() // Cp
            running_mean.mul_(1 - momentum).add_(momentum * mean.reshape(-1)[:C_real])
            running_var.mul_(1 - momentum).add_(momentum * var.reshape(-1)[:C_real] * cnt / max(cnt - 1, 1))
        v = _act(act, yv * scale + shift)
        if residual is not None:
            v = v + residual.float()
        z.copy_(v.to(z.dtype))
        N = y.shape[0]
        stats = torch.empty(4, N, Cp)
        for i, t in enumerate((mean, rstd, scale, shift)):
            stats[i] = t.reshape(-1, Cp).expand(N, Cp)
        return stats

    def norm_backward(self, dz, y, dy, stats, C_real, scope, act, gamma, dgamma, dbeta, dy_chansum=None, ext_nchunks=0, dy_split=None):
        assert ext_nchunks == 0, 'the emulation never reports fused norm-backward reductions'
        self._count('norm_bwd')
        yv, g = y.float(), dz.float()
        N, H, W, Cp = yv.shape
        mean, rstd, scale, shift = (stats[i].view(N, 1, 1, Cp) for i in range(4))
        nv = yv * scale + shift
        if act == L.ACT_RELU:
            dn = g * (nv > 0)
        elif act == L.ACT_LRELU:
            dn = torch.where(nv > 0, g, 0.2 * g)
        elif act == L.ACT_TANH:
            dn = g * (1 - torch.tanh(nv) ** 2)
        else:
            dn = g
        xh = (yv - mean) * rstd
        dims = (0, 1, 2) if scope == L.NORM_BATCH else (1, 2)
        c1 = dn.mean(dim=dims, keepdim=True)
        c2 = (dn * xh).mean(dim=dims, keepdim=True)
        ga = torch.zeros(Cp)
        ga[:C_real] = gamma.float() if gamma is not None else 1.0
        dyv = ga * rstd * (dn - c1 - xh * c2)
        dy.copy_(dyv.to(dy.dtype))
        if dy_chansum is not None:
            dy_chansum.add_(dyv.sum(dim=(0, 1, 2))[:C_real])
        if dgamma is not None:
            dgamma.add_((dn * xh).sum(dim=(0, 1, 2))[:C_real])
            dbeta.add_(dn.sum(dim=(0, 1, 2))[:C_real])

    def act_forward(self, act, x, y):
        self._count('act')
        y.copy_(_act(act, x.float()).to(y.dtype))

    def act_backward(self, act, dy, y, dx):
        self._count('act_bwd')
        dx.copy_((dy.float() * _act_grad_from_output(act, y.float())).to(dx.dtype))

    def gate_forward(self, x, psi, out):
        self._count('gate')
        out.copy_((x.float() * psi.float()[..., :1]).to(out.dtype))

    def gate_backward(self, g, x, psi, dx, dpsi):
        self._count('gate_bwd')
        if dx is not None:
            dx.copy_((g.float() * psi.float()[..., :1]).to(dx.dtype))
        d = torch.zeros(dpsi.shape, dtype=torch.float32)
        d[..., 0] = (g.float() * x.float()).sum(-1)
        dpsi.copy_(d.to(dpsi.dtype))

    def dropout(self, x, y, p, seed):
        self._count('dropout')
        g = torch.Generator().manual_seed(int(seed) % (2 ** 63))
        keep = (torch.rand(x.shape, generator=g) >= p).float()
        y.copy_((x.float() * keep / (1.0 - p)).to(y.dtype))

    def axpby(self, alpha, a, beta, b, out):
        self._count('axpby')
        v = alpha * a.float()
        if b is not None:
            v = v + beta * b.float()
        out.copy_(v.to(out.dtype))

    def copy_channels(self, src, s_c0, dst, d_c0, nch, accumulate=False):
        self._count('copy')
        v = src[..., s_c0:s_c0 + nch].float()
        if accumulate:
            v = v + dst[..., d_c0:d_c0 + nch].float()
        dst[..., d_c0:d_c0 + nch] = v.to(dst.dtype)

    def channel_sum(self, x, C_real, out, accumulate):
        self._count('csum')
        s = x.float().sum(dim=(0, 1, 2))[:C_real]
        if accumulate:
            out.add_(s)
        else:
            out.copy_(s)

    def nchw_to_nhwc(self, src, dst, c0, zero_pad_to):
        self._count('to_nhwc')
        C = src.shape[1]
        dst[..., c0:c0 + C] = src.permute(0, 2, 3, 1).to(dst.dtype)
        if zero_pad_to > c0 + C:
            dst[..., c0 + C:zero_pad_to] = 0

    def nhwc_to_nchw(self, src, c0, dst):
        self._count('to_nchw')
        C = dst.shape[1]
        dst.copy_(src[..., c0:c0 + C].permute(0, 3, 1, 2).float())

    def conv_narrow_supported(self, x, cin_p, cout, k, pad, pad_mode, act=L.ACT_NONE):
        return act in (L.ACT_NONE, L.ACT_TANH) and cout in (1, 3) and x.dtype == torch.bfloat16 and cin_p == 64 and k == 7 and cout <= 4 and pad == 3 and pad_mode == L.PAD_ZERO

    def conv_narrow_forward(self, packed, x, out, cout, k, pad, bias, act):
        # same mathematics as the two-call form: raw (co, kw) planes from the vertical taps, then the shifted sum over kw
        self._count('conv_narrow')
        n, h, w, _ = x.shape
        T = torch.empty((n, h, w, packed.plan.rows_pad if False else ((cout * k + 7) // 8 * 8 if cout * k > 8 else 8)), dtype=torch.float32)
        tc = 8
        while tc < cout * k:
            tc *= 2
        T = torch.empty((n, h, w, tc), dtype=torch.float32)
        self.conv_forward(packed, x, T, h, w, None, L.ACT_NONE, L.ACT_NONE, L.PREC_BF16, raw_out=True)
        self.shift_sum(T, cout, k, pad, L.PAD_ZERO, bias, act, out)

    def shift_sum(self, T, cout, kw, pad, pad_mode, bias, act, out):
        self._count('shift_sum')
        n, h, w, _ = T.shape
        acc = torch.zeros(n, h, w, cout)
        idx = torch.arange(w)
        for k in range(kw):
            ws = idx + k - pad
            if pad_mode == L.PAD_REFLECT:
                ws = _reflect(ws, w)
                acc += T[:, :, ws][..., k:cout * kw:kw]
            else:
                m = ((ws >= 0) & (ws < w)).float()
                acc += T[:, :, ws.clamp(0, w - 1)][..., k:cout * kw:kw] * m[None, None, :, None]
        if bias is not None:
            acc += bias.float()
        out.zero_()
        out[..., :cout] = _act(act, acc).to(out.dtype)

    def convt4_gather(self, T, cout, bias, act, out):
        """dl_convt4_gather: out[n,oy,ox,co] = act(bias + sum over the 2x2 (ky,kx) with (oy+1-ky), (ox+1-kx) even of T[n,(oy+1-ky)/2,(ox+1-kx)/2,(ky*4+kx)*cout+co])"""
        self._count('convt4_gather')
        n, h, w, _ = T.shape
        acc = torch.zeros(n, 2 * h, 2 * w, cout)
        for ky in range(4):
            for kx in range(4):
                oy = 2 * torch.arange(h) - 1 + ky
                ox = 2 * torch.arange(w) - 1 + kx
                my, mx = (oy >= 0) & (oy < 2 * h), (ox >= 0) & (ox < 2 * w)
                src = T[:, my][:, :, mx][..., (ky * 4 + kx) * cout:(ky * 4 + kx + 1) * cout].float()
                acc[:, oy[my][:, None], ox[mx][None, :]] += src
        if bias is not None:
            acc += bias.float()
        out.zero_()
        out[..., :cout] = _act(act, acc).to(out.dtype)

    def reflect_fold(self, src, dst, pad):
        self._count('reflect_fold')
        n, h, w, cp = dst.shape

        def sources(i, size):           # padded indices that nn.ReflectionPad2d(pad) fills from interior index i
            out = [i + pad]
            if 1 <= i <= pad:
                out.append(pad - i)
            if size - 1 - pad <= i <= size - 2:
                out.append(pad + 2 * size - 2 - i)
            return out
        acc = torch.zeros(dst.shape, dtype=torch.float32)
        sv = src.float()
        for i in range(h):
            for hp in sources(i, h):
                row = sv[:, hp]                                  # [N, W+2p, Cp]
                for j in range(w):
                    for wp in sources(j, w):
                        acc[:, i, j] += row[:, wp]
        dst.copy_(acc.to(dst.dtype))

    def shift_stack(self, dy, cout, kw, pad, D):
        self._count('shift_stack')
        n, h, w, _ = dy.shape
        D.zero_()
        idx = torch.arange(w)
        for k in range(kw):
            ws = idx - (k - pad)
            m = ((ws >= 0) & (ws < w)).to(dy.dtype)
            D[..., k:cout * kw:kw] = dy[:, :, ws.clamp(0, w - 1)][..., :cout] * m[None, None, :, None]

    def loss(self, kind, x, target, target_const, C_real, loss_out, grad, grad_scale, out_scale=1.0, accumulate=False):
        self._count('loss')
        v = x[..., :C_real].float()
        t = target[..., :C_real].float() if target is not None else torch.full_like(v, target_const)
        if kind == L.LOSS_BCE_LOGITS:
            l = v.clamp(min=0) - v * t + torch.log1p(torch.exp(-v.abs()))
            g = torch.sigmoid(v) - t
        elif kind == L.LOSS_MSE:
            l = (v - t) ** 2
            g = 2 * (v - t)
        elif kind == L.LOSS_L1:
            l = (v - t).abs()
            g = torch.sign(v - t)
        elif kind == L.LOSS_LINEAR:
            l = t * v
            g = t
        else:
            d = v - t
            l = torch.where(d.abs() < 1, 0.5 * d * d, d.abs() - 0.5)
            g = d.clamp(-1, 1)
        loss_out[0] = (loss_out[0] if accumulate else 0.0) + out_scale * l.mean()
        if grad is not None:
            grad.zero_()
            grad[..., :C_real] = (g * grad_scale / v.numel()).to(grad.dtype)

    def upsample2(self, src, dst, backward=False):
        self._count('upsample2')
        if backward:
            n, h2, w2, c = src.shape
            dst.copy_(src.float().reshape(n, h2 // 2, 2, w2 // 2, 2, c).sum(dim=(2, 4)).to(dst.dtype))
        else:
            dst.copy_(src.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2))

    def kldiv(self, x, t, C_real, loss_out, grad, grad_scale, out_scale=1.0, accumulate=False):
        self._count('kldiv')
        v, tv = x[..., :C_real].double().reshape(-1), t[..., :C_real].double().reshape(-1)
        q, p = torch.softmax(v, 0), torch.softmax(tv, 0)
        kl = (p * (torch.log_softmax(tv, 0) - torch.log_softmax(v, 0))).sum()
        loss_out[0] = (loss_out[0] if accumulate else 0.0) + out_scale * kl.float()
        if grad is not None:
            grad.zero_()
            grad[..., :C_real] = (grad_scale * (q - p)).reshape(x[..., :C_real].shape).to(grad.dtype)

    def maxpool2_forward(self, x, y):
        self._count('maxpool_fwd')
        n, h, w, c = x.shape
        v = x[:, :h // 2 * 2, :w // 2 * 2].float().reshape(n, h // 2, 2, w // 2, 2, c)
        y.copy_(v.amax(dim=(2, 4)).to(y.dtype))

    def maxpool2_backward(self, x, dy, dx):
        self._count('maxpool_bwd')
        n, h, w, c = x.shape
        ho, wo = h // 2, w // 2
        v = x[:, :ho * 2, :wo * 2].float().reshape(n, ho, 2, wo, 2, c).permute(0, 1, 3, 5, 2, 4).reshape(n, ho, wo, c, 4)
        arg = v.argmax(dim=-1)                        # first maximum in row-major window order
        g = torch.zeros_like(v).scatter_(-1, arg.unsqueeze(-1), dy.float().unsqueeze(-1))
        dx.zero_()
        dx[:, :ho * 2, :wo * 2] = g.reshape(n, ho, wo, c, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(n, ho * 2, wo * 2, c).to(dx.dtype)

    def adam_step(self, p, g, m, v, lr, b1, b2, eps, step, gscale):
        self._count('adam')
        gi = g * gscale
        m.mul_(b1).add_(gi, alpha=1 - b1)
        v.mul_(b2).addcmul_(gi, gi, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        p.addcdiv_(m, (v.sqrt() / bc2 ** 0.5).add_(eps), value=-lr / bc1)


    # ---- tiles (include/deepliif_hip.h dl_tile_*): written from the header's formulas, not from the oracle
    @staticmethod
    def _tile_pixels(img, H0, W0, ox, oy, tile, pad, pad_rgb):
        import numpy as np
        a = img.numpy()
        patch = tile - 2 * pad
        out = np.empty((tile, tile, 3), dtype=np.uint8)
        out[...] = [pad_rgb & 255, (pad_rgb >> 8) & 255, (pad_rgb >> 16) & 255]

        def mirror(c, n):
            m = c % (2 * n)
            return np.where(c < n, c, np.where(m < n, m, 2 * n - 1 - m))
        sy = mirror(oy + np.arange(patch), H0)
        sx = mirror(ox + np.arange(patch), W0)
        out[pad:pad + patch, pad:pad + patch] = a[sy][:, sx]
        return out

    def tile_gather(self, images, H0, W0, origins, tile, pad, pad_rgb, lut, out):
        self._count('tile_gather')
        out.zero_()
        for t, (ox, oy) in enumerate(origins.tolist()):
            for s, im in enumerate(images):
                px = self._tile_pixels(im, H0, W0, ox, oy, tile, pad, pad_rgb)
                out[t, :, :, 3 * s:3 * s + 3] = lut[torch.from_numpy(px.astype('int64'))].to(out.dtype)

    def tile_gray_stats(self, image, H0, W0, origins, tile, pad, pad_rgb, stats):
        self._count('tile_gray_stats')
        for t, (ox, oy) in enumerate(origins.tolist()):
            px = torch.from_numpy(self._tile_pixels(image, H0, W0, ox, oy, tile, pad, pad_rgb).astype('int64'))
            g = (19595 * px[..., 0] + 38470 * px[..., 1] + 7471 * px[..., 2] + 0x8000) >> 16
            v = g[(g != 0) & (g != 255)]
            stats[t, 0], stats[t, 1], stats[t, 2] = v.numel(), int(v.sum()), int((v * v).sum())

    def tile_paste(self, tiles, tile, rects, dst):
        self._count('tile_paste')
        for slot, l, t, w, h, px, py, rgb in rects.tolist():
            if slot < 0:
                dst[py:py + h, px:px + w] = torch.tensor([rgb & 255, (rgb >> 8) & 255, (rgb >> 16) & 255], dtype=torch.uint8)
                continue
            v = tiles[slot, t:t + h, l:l + w, :3].float()
            dst[py:py + h, px:px + w] = (((v + 1.0) * 0.5) * 255.0).to(torch.int32).to(torch.uint8)


def install():
    """Route deepliif_amd.ops through the CPU emulation (tests only); returns the backend for call counting."""
    fb = FakeBackend()
    ops._impl = fb
    return fb


def uninstall():
    ops._impl = None
