"""Per-layer-shape time budget of the 5G+5D training step (BASELINE configs[2]) on the GPU box.

Every conv shape of Resnet-9 and NLayerD(n=4) at batch 8, 512x512 is timed in isolation (forward, data gradient, weight gradient;
HIP events on torch's current stream) and multiplied by the number of launches per step, so the table says where the step's conv
time goes and how far each shape is from the MFMA / HBM roofline.  Writes gpurun_out/layer_budget_<tag>.json.

  python tools/layer_budget.py [tag] [precision]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepliif_amd import _lib as L
from deepliif_amd import ops
from deepliif_amd.engine import Precision
from deepliif_amd.geometry import ConvSpec, cpad

DEV = 'cuda'
be = ops.impl()
NG, ND = 5, 5          # generators / discriminators in the step


def timeit(fn, iters=12, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def split_input(t, prec):
    """LB_SPLIT=1 (strict policy): hand the convolutions the SPLIT COPY of their input ([8 bf16 hi | 8 bf16 lo] per group of 8 channels), as the norm
    kernels do inside the training step -- the route conv_gemm_8ph_x3 / conv_s2f_x3 / the split-reading glds kernels take there"""
    if os.environ.get('LB_SPLIT') != '1' or prec.prec != L.PREC_BF16X3 or t.dtype != torch.float32 or t.shape[3] < 32 or not be.conv_takes_split(t, prec.prec, L.ACT_NONE, L.PAD_ZERO):
        return t, {}
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    g = torch.stack([hi.reshape(*t.shape[:3], -1, 8), lo.reshape(*t.shape[:3], -1, 8)], dim=4)
    return g.contiguous().view(torch.int16).reshape(*t.shape[:3], -1).view(torch.float32).reshape(t.shape), {'in_split': True}


def conv_case(name, kind, cin, cout, k, s, p, N, H, W, prec, n_fwd, n_dgrad, n_wgrad, op=0, act=L.ACT_NONE, bias=True):
    spec = ConvSpec(kind, cin, cout, k, s, p, L.PAD_ZERO, op)
    ho, wo = spec.out_hw(H, W)
    wshape = (cout, cin, k, k) if kind == 'conv' else (cin, cout, k, k)
    w = torch.randn(wshape, device=DEV) * 0.02
    b = torch.zeros(cout, device=DEV) if bias else None
    x = torch.randn(N, H, W, cpad(cin), device=DEV).to(prec.dtype)
    dy = torch.randn(N, ho, wo, cpad(cout), device=DEV).to(prec.dtype)
    hq, wq = (ho, wo) if kind == 'conv' else (H, W)
    dq = ((H + 1) // 2, (W + 1) // 2) if (kind == 'conv' and s == 2) else (H, W)
    flops = 2.0 * N * (ho * wo if kind == 'conv' else H * W) * cout * cin * k * k
    esz = x.element_size()
    io_bytes = (x.numel() + dy.numel()) * esz + w.numel() * 2
    res = {'name': name, 'gflop': flops / 1e9, 'io_mb': io_bytes / 1e6, 'n_fwd': n_fwd, 'n_dgrad': n_dgrad, 'n_wgrad': n_wgrad}
    narrow = spec.is_narrow()
    if n_fwd:
        if narrow:
            pf = ops.PackedWeights(spec.narrow_forward_plan(), DEV, prec.prec == 3); be.pack_weights(pf, w)
            T = torch.empty((N, ho, wo, cpad(cout * k)), dtype=torch.float32, device=DEV)
            out = torch.empty(N, ho, wo, cpad(cout), device=DEV, dtype=prec.dtype)

            roll = (prec.prec == L.PREC_BF16 or (prec.prec == L.PREC_BF16X3 and x.dtype == torch.float32)) and be.conv_narrow_supported(x, cpad(cin), cout, k, p, L.PAD_ZERO)
            res['fwd_path'] = 'dl_conv_narrow_forward (rolling rows)' if roll else 'dl_conv_forward(raw) + dl_shift_sum'

            def f():
                if roll:
                    be.conv_narrow_forward(pf, x, out, cout, k, p, b, act)
                    return
                be.conv_forward(pf, x, T, ho, wo, None, L.ACT_NONE, L.ACT_NONE, prec.prec, raw_out=True)
                be.shift_sum(T, cout, k, p, L.PAD_ZERO, b, act, out)
        else:
            pf = ops.PackedWeights(spec.forward_plan(), DEV, prec.prec == 3); be.pack_weights(pf, w)
            out = torch.empty(N, ho, wo, cpad(cout), device=DEV, dtype=prec.dtype)

            xin, kw = split_input(x, prec)

            def f():
                be.conv_forward(pf, xin, out, hq, wq, b, act, L.ACT_NONE, prec.prec, want_stats=(act == L.ACT_NONE), **kw)
        t = timeit(f)
        res['fwd_us'], res['fwd_tf'], res['fwd_kernel'] = t * 1e6, flops / t / 1e12, be.last_conv_kernel
    if n_dgrad:
        pd = ops.PackedWeights(spec.dgrad_plan(), DEV, prec.prec == 3); be.pack_weights(pd, w)
        dx = torch.empty_like(x)
        dyin, kw = split_input(dy, prec)
        t = timeit(lambda: be.conv_forward(pd, dyin, dx, dq[0], dq[1], None, L.ACT_NONE, L.ACT_NONE, prec.prec, **kw))
        res['dgrad_us'], res['dgrad_tf'], res['dgrad_kernel'] = t * 1e6, flops / t / 1e12, be.last_conv_kernel
    if n_wgrad:
        grad = torch.zeros(wshape, device=DEV)
        if narrow:
            D = torch.empty((N, ho, wo, cpad(cout * k)), dtype=dy.dtype, device=DEV)

            c4 = getattr(be, 'wgrad_c4_applies', None) is not None and be.wgrad_c4_applies(dy, x, grad, k, 1, p, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, prec.prec)
            res['wgrad_path'] = 'wgrad_c4_kernel' if c4 else 'dl_shift_stack + stacked dl_conv_wgrad'

            def f():
                if c4:
                    be.conv_wgrad(dy, x, grad, k, 1, p, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, prec.prec, True)
                    return
                be.shift_stack(dy, cout, k, p, D)
                be.conv_wgrad(D, x, grad, k, 1, p, L.PAD_ZERO, L.ACT_NONE, L.ACT_NONE, prec.prec, True, stack_kw=k)
        elif kind == 'conv':
            f = lambda: be.conv_wgrad(dy, x, grad, k, s, p, 0, 0, 0, prec.prec, True)
        else:
            f = lambda: be.conv_wgrad(x, dy, grad, k, s, p, 0, 0, 0, prec.prec, True)
        t = timeit(f)
        res['wgrad_us'], res['wgrad_tf'] = t * 1e6, flops / t / 1e12
    res['step_ms'] = (n_fwd * res.get('fwd_us', 0) + n_dgrad * res.get('dgrad_us', 0) + n_wgrad * res.get('wgrad_us', 0)) / 1e3
    print(json.dumps(res), flush=True)
    return res


def norm_case(name, N, H, W, C, prec, act, n_fwd, n_bwd, residual=False):
    y = torch.randn(N, H, W, C, device=DEV).to(prec.dtype)
    z = torch.empty_like(y)
    r = torch.randn_like(y) if residual else None
    dz = torch.randn_like(y)
    dy = torch.empty_like(y)
    st = [None]

    def f():
        st[0] = be.norm_forward(y, z, C, L.NORM_INSTANCE, act, None, None, None, None, -1.0, r)
    t = timeit(f)
    nbytes = y.numel() * y.element_size()
    res = {'name': name, 'fwd_us': t * 1e6, 'fwd_GBs': (3 + (1 if residual else 0)) * nbytes / t / 1e9, 'n_fwd': n_fwd, 'n_bwd': n_bwd}
    t = timeit(lambda: be.norm_backward(dz, y, dy, st[0], C, L.NORM_INSTANCE, act, None, None, None))
    res.update({'bwd_us': t * 1e6, 'bwd_GBs': 5 * nbytes / t / 1e9})
    res['step_ms'] = (n_fwd * res['fwd_us'] + n_bwd * res['bwd_us']) / 1e3
    print(json.dumps(res), flush=True)
    return res


if __name__ == '__main__':
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
    prec = Precision.get(sys.argv[2] if len(sys.argv) > 2 else 'bf16')
    N, S = 8, 512
    out = []
    R, LR, TH = L.ACT_RELU, L.ACT_LRELU, L.ACT_TANH
    # ---- Resnet-9 generator (x NG).  The stem's input is real_A: no data gradient.
    out.append(conv_case('G stem 7x7 3->64 @512', 'conv', 3, 64, 7, 1, 3, N, S, S, prec, NG, 0, NG))
    out.append(conv_case('G down1 3x3s2 64->128 @512->256', 'conv', 64, 128, 3, 2, 1, N, S, S, prec, NG, NG, NG))
    out.append(conv_case('G down2 3x3s2 128->256 @256->128', 'conv', 128, 256, 3, 2, 1, N, S // 2, S // 2, prec, NG, NG, NG))
    out.append(conv_case('G block 3x3 256->256 @128', 'conv', 256, 256, 3, 1, 1, N, S // 4, S // 4, prec, 18 * NG, 18 * NG, 18 * NG))
    out.append(conv_case('G up1 convT3x3s2 256->128 @128->256', 'convT', 256, 128, 3, 2, 1, N, S // 4, S // 4, prec, NG, NG, NG, op=1))
    out.append(conv_case('G up2 convT3x3s2 128->64 @256->512', 'convT', 128, 64, 3, 2, 1, N, S // 2, S // 2, prec, NG, NG, NG, op=1))
    out.append(conv_case('G head 7x7 64->3 @512', 'conv', 64, 3, 7, 1, 3, N, S, S, prec, NG, NG, NG, act=TH))
    # ---- NLayerD(n=4) (x ND).  r06: backward_D runs every discriminator ONCE on cat(fake pairs, real pairs) -- batch 2N: 1 fwd + 1 bwd (no data gradient into the
    # detached input pair) -- and backward_G once on batch N (fwd + data gradient only).  LB_D_PAIR=0: the two calls per discriminator of rounds 1-5.
    pair = os.environ.get('LB_D_PAIR', '1') != '0'
    dcases = [('D c1 4x4s2 6->64 @512->256', 6, 64, 2, S, LR), ('D c2 4x4s2 64->128 @256->128', 64, 128, 2, S // 2, L.ACT_NONE), ('D c3 4x4s2 128->256 @128->64', 128, 256, 2, S // 4, L.ACT_NONE),
              ('D c4 4x4s2 256->512 @64->32', 256, 512, 2, S // 8, L.ACT_NONE), ('D c5 4x4s1 512->512 @32->31', 512, 512, 1, S // 16, L.ACT_NONE),
              ('D c6 4x4s1 512->1 @31->30', 512, 1, 1, S // 16 - 1, L.ACT_NONE)]
    for name, ci, co, st, hw, act in dcases:
        first = ci == 6
        if pair:
            out.append(conv_case(name + ' x2N (D step)', 'conv', ci, co, 4, st, 1, 2 * N, hw, hw, prec, ND, 0 if first else ND, ND, act=act))
            out.append(conv_case(name + ' (G step)', 'conv', ci, co, 4, st, 1, N, hw, hw, prec, ND, ND, 0, act=act))
        else:
            out.append(conv_case(name, 'conv', ci, co, 4, st, 1, N, hw, hw, prec, 3 * ND, ND if first else 3 * ND, 2 * ND, act=act))
    # ---- norms (instance): G 23 per net (fwd + bwd), D 4 per net (3 fwd, 3 bwd)
    out.append(norm_case('norm 64 @512 relu', N, S, S, 64, prec, R, 2 * NG, 2 * NG))
    out.append(norm_case('norm 128 @256 relu', N, S // 2, S // 2, 128, prec, R, 2 * NG, 2 * NG))
    out.append(norm_case('norm 256 @128 relu', N, S // 4, S // 4, 256, prec, R, 10 * NG, 10 * NG))
    out.append(norm_case('norm 256 @128 +res', N, S // 4, S // 4, 256, prec, L.ACT_NONE, 9 * NG, 9 * NG, residual=True))
    for name, hw, c in (('norm 128 @128 lrelu (D)', S // 4, 128), ('norm 256 @64 lrelu (D)', S // 8, 256), ('norm 512 @32 lrelu (D)', S // 16, 512)):
        if pair:
            out.append(norm_case(name + ' x2N', 2 * N, hw, hw, c, prec, LR, ND, ND))
            out.append(norm_case(name, N, hw, hw, c, prec, LR, ND, ND))
        else:
            out.append(norm_case(name, N, hw, hw, c, prec, LR, 3 * ND, 3 * ND))
    tot = sum(r['step_ms'] for r in out)
    print('sum of isolated launches per step: %.1f ms' % tot)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open(f'gpurun_out/layer_budget_{tag}.json', 'w'), indent=1)
    print('%-40s %8s %8s %8s %8s' % ('layer', 'fwd us', 'dgrad us', 'wgrad us', 'step ms'))
    for r in out:
        print('%-40s %8.1f %8.1f %8.1f %8.2f   %s' % (r['name'], r.get('fwd_us', 0), r.get('dgrad_us', r.get('bwd_us', 0)), r.get('wgrad_us', 0), r['step_ms'],
                                                      ('fwd %.0f TF dgrad %.0f TF wgrad %.0f TF' % (r.get('fwd_tf', 0), r.get('dgrad_tf', 0), r.get('wgrad_tf', 0))) if 'gflop' in r else ('%.0f / %.0f GB/s' % (r['fwd_GBs'], r['bwd_GBs']))))
