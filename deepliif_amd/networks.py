"""Drop-in counterparts of deepliif/models/networks.py for the hot path: same constructor signatures, same module tree
(so state_dict keys, init_weights RNG order and the BatchNorm toggles of deepliif/util/__init__.py:743-770 keep working),
but `forward` runs the MI355X engine (deepliif_amd.engine -> HIP kernels) instead of ATen.

The torch.nn leaf modules (Conv2d, ConvTranspose2d, BatchNorm2d, ...) are used as *parameter containers* only -- their own
forward is never called.  Reference citations: networks.py:25-44 get_norm_layer, :84-139 init_weights/init_net,
:142-238 define_G/define_D, :244-317 GANLoss, :357-513 ResnetGenerator, :516-615 UnetGenerator, :618-664 NLayerDiscriminator.
"""
from __future__ import annotations

import functools
import os
from typing import List, Optional

import torch
import torch.nn as nn
from torch.nn import init
from torch.optim import lr_scheduler

from . import _lib as L
from . import engine as E
from .geometry import ConvSpec

DEFAULT_PRECISION = os.environ.get('DEEPLIIF_AMD_PRECISION', 'bf16')


class Identity(nn.Module):
    def forward(self, x):
        return x


def get_norm_layer(norm_type='instance'):
    """networks.py:25-44.  'batch': BatchNorm2d(affine, tracked);  'instance': InstanceNorm2d(no affine, untracked)."""
    if norm_type == 'batch':
        return functools.partial(nn.BatchNorm2d, affine=True, track_running_stats=True)
    if norm_type == 'instance':
        return functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=False)
    if norm_type == 'none':
        return lambda c: Identity()
    raise NotImplementedError('normalization layer [%s] is not supported by the MI355X engine' % norm_type)


def _norm_kind(norm_layer) -> str:
    f = norm_layer.func if isinstance(norm_layer, functools.partial) else norm_layer
    if f is nn.BatchNorm2d:
        return 'batch'
    if f is nn.InstanceNorm2d:
        return 'instance'
    return 'none'


def _uses_bias(norm_layer) -> bool:
    # conv bias only when the following norm has no affine shift (networks.py:381-384,570-573,631-634)
    return _norm_kind(norm_layer) == 'instance'


# -------------------------------------------------------------------------------------------------------------
# engine-backed module base
# -------------------------------------------------------------------------------------------------------------
class EngineNet(nn.Module):
    """Common machinery: precision policy, lazily-built layer bindings, NCHW<->engine conversion at the seam."""

    def __init__(self):
        super().__init__()
        self.precision = DEFAULT_PRECISION
        self.batched_per_sample_norm = True     # N>1 inference reproduces N one-tile forwards
        self._bound = None

    def set_precision(self, name: str):
        self.precision = name
        return self

    def _layers(self):
        if self._bound is None:
            self._bound = self._bind()
        return self._bound

    def _bind(self):
        raise NotImplementedError

    def run(self, ctx: E.Ctx, x: E.Act) -> E.Act:
        raise NotImplementedError

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Inference seam: NCHW fp32 in, NCHW fp32 out (deepliif/models/__init__.py:285-291 calls net(tensor)).
        eval() (init_nets): BatchNorm on the statistics of each tile, which is what the reference's one-tile forwards with nulled running
        statistics compute.  train(): what nn.Module.train() means for the reference's module tree -- TorchServe's handler
        serves the nets in that mode (model-server/net_handler.py:10-12): statistics over the whole batch, running statistics updated
        (momentum 0.1) where the module still tracks them, Dropout(0.5) active."""
        prec = E.Precision.get(self.precision)
        train = self.training
        with E.ops.half_mode(prec.half):
            ctx = E.Ctx(prec, None, training=train, per_sample_norm=self.batched_per_sample_norm and not train)
            return E.from_engine(self.run(ctx, E.to_engine(x, prec)))


def _norm_binding(kind: str, C: int, module: nn.Module) -> Optional[E.NormLayer]:
    if kind == 'none':
        return None
    return E.NormLayer(kind, C, module if kind == 'batch' else None)


# -------------------------------------------------------------------------------------------------------------
# ResnetGenerator
# -------------------------------------------------------------------------------------------------------------
class ResnetBlock(nn.Module):
    """Parameter container with the reference's conv_block indices (networks.py:467-508)."""

    def __init__(self, dim, padding_type, norm_layer, use_dropout, use_bias, use_spectral_norm=False):
        super().__init__()
        if use_spectral_norm:
            raise NotImplementedError('spectral norm is not on the MI355X hot path')
        if padding_type not in ('zero', 'reflect'):
            raise NotImplementedError('padding [%s] is not supported by the MI355X engine' % padding_type)
        seq: List[nn.Module] = []
        self.idx = {}
        for half in (0, 1):
            if padding_type == 'reflect':
                seq.append(nn.ReflectionPad2d(1))
            self.idx[f'conv{half}'] = len(seq)
            seq.append(nn.Conv2d(dim, dim, kernel_size=3, padding=1 if padding_type == 'zero' else 0, bias=use_bias))
            self.idx[f'norm{half}'] = len(seq)
            seq.append(norm_layer(dim))
            if half == 0:
                seq.append(nn.ReLU(True))
                if use_dropout:
                    seq.append(nn.Dropout(0.5))
        self.conv_block = nn.Sequential(*seq)
        self.use_dropout = use_dropout


class ResnetGenerator(EngineNet):
    def __init__(self, input_nc, output_nc, ngf=64, norm_layer=nn.BatchNorm2d, use_dropout=False, n_blocks=6, padding_type='zero',
                 upsample='convtranspose', use_spectral_norm=False):
        assert n_blocks >= 0
        super().__init__()
        if upsample not in ('convtranspose', 'resize_conv') or use_spectral_norm:
            # 'pixel_shuffle' cannot be constructed in the reference either (networks.py:416-420 passes kernel_size to SpectralNorm: TypeError)
            raise NotImplementedError('upsample=convtranspose | resize_conv without spectral norm are on the MI355X hot path')
        self.upsample = upsample
        self.norm_kind = _norm_kind(norm_layer)
        self.padding_type = padding_type
        self.n_blocks = n_blocks
        self.ngf, self.input_nc, self.output_nc = ngf, input_nc, output_nc
        use_bias = _uses_bias(norm_layer)
        pad3 = nn.ReflectionPad2d(3) if padding_type == 'reflect' else nn.ZeroPad2d(3)
        seq: List[nn.Module] = [pad3, nn.Conv2d(input_nc, ngf, kernel_size=7, padding=0, bias=use_bias), norm_layer(ngf), nn.ReLU(True)]
        for i in range(2):
            m = 2 ** i
            seq += [nn.Conv2d(ngf * m, ngf * m * 2, kernel_size=3, stride=2, padding=1, bias=use_bias), norm_layer(ngf * m * 2), nn.ReLU(True)]
        for _ in range(n_blocks):
            seq.append(ResnetBlock(ngf * 4, padding_type, norm_layer, use_dropout, use_bias))
        for i in range(2):
            m = 2 ** (2 - i)
            if upsample == 'resize_conv':       # networks.py:409-415: nearest x2, ReflectionPad2d(1), Conv2d(k3, default bias)
                seq += [nn.Upsample(scale_factor=2, mode='nearest'), nn.ReflectionPad2d(1), nn.Conv2d(ngf * m, ngf * m // 2, kernel_size=3, stride=1, padding=0),
                        norm_layer(ngf * m // 2), nn.ReLU(True)]
                continue
            seq += [nn.ConvTranspose2d(ngf * m, ngf * m // 2, kernel_size=3, stride=2, padding=1, output_padding=1, bias=use_bias),
                    norm_layer(ngf * m // 2), nn.ReLU(True)]
        seq.append(nn.ReflectionPad2d(3) if padding_type == 'reflect' else nn.ZeroPad2d(3))
        seq.append(nn.Conv2d(ngf, output_nc, kernel_size=7, padding=0))
        seq.append(nn.Tanh())
        self.model = nn.Sequential(*seq)

    def _bind(self):
        ngf, k = self.ngf, self.norm_kind
        pm = L.PAD_REFLECT if self.padding_type == 'reflect' else L.PAD_ZERO
        m = self.model
        b = {}
        b['stem'] = (E.ConvLayer(ConvSpec('conv', self.input_nc, ngf, 7, 1, 3, pm), m[1].weight, m[1].bias), _norm_binding(k, ngf, m[2]))
        b['down'] = []
        idx = 4
        for i in range(2):
            c = ngf * 2 ** i
            b['down'].append((E.ConvLayer(ConvSpec('conv', c, 2 * c, 3, 2, 1), m[idx].weight, m[idx].bias), _norm_binding(k, 2 * c, m[idx + 1])))
            idx += 3
        b['blocks'] = []
        for _ in range(self.n_blocks):
            blk = m[idx]
            cb = blk.conv_block
            ent = []
            for half in (0, 1):
                cm, nm = cb[blk.idx[f'conv{half}']], cb[blk.idx[f'norm{half}']]
                ent.append((E.ConvLayer(ConvSpec('conv', ngf * 4, ngf * 4, 3, 1, 1, pm), cm.weight, cm.bias), _norm_binding(k, ngf * 4, nm)))
            b['blocks'].append((ent, blk))
            idx += 1
        b['up'] = []
        for i in range(2):
            c = ngf * 2 ** (2 - i)
            if self.upsample == 'resize_conv':
                b['up'].append((E.ConvLayer(ConvSpec('conv', c, c // 2, 3, 1, 1, L.PAD_REFLECT), m[idx + 2].weight, m[idx + 2].bias),
                                _norm_binding(k, c // 2, m[idx + 3])))
                idx += 5
                continue
            b['up'].append((E.ConvLayer(ConvSpec('convT', c, c // 2, 3, 2, 1, out_pad=1), m[idx].weight, m[idx].bias),
                            _norm_binding(k, c // 2, m[idx + 1])))
            idx += 3
        idx += 1
        b['head'] = E.ConvLayer(ConvSpec('conv', ngf, self.output_nc, 7, 1, 3, pm), m[idx].weight, m[idx].bias)
        return b

    def run(self, ctx: E.Ctx, x: E.Act) -> E.Act:
        b = self._layers()
        # sole_reader: the next convolution is the only reader of these activations (engine.norm_act); the last down stage's output is also the first
        # block's residual, so it is not promised to anyone
        c, n = b['stem']
        h = E.norm_act(ctx, E.conv(ctx, x, c, stats=n is not None), n, L.ACT_RELU, sole_reader=b['down'][0][0])
        for i, (c, n) in enumerate(b['down']):
            h = E.norm_act(ctx, E.conv(ctx, h, c, stats=n is not None), n, L.ACT_RELU, sole_reader=b['down'][i + 1][0] if i + 1 < len(b['down']) else None)
        nblk = len(b['blocks'])
        for ib, (ent, blk) in enumerate(b['blocks']):
            (c1, n1), (c2, n2) = ent
            drop = blk.use_dropout and blk.training
            # without dropout the block's inner activation is read by its second conv only: the strict policy keeps just its split copy
            r = E.norm_act(ctx, E.conv(ctx, h, c1, stats=n1 is not None), n1, L.ACT_RELU, sole_reader=None if drop else c2)
            if drop:                                   # nn.Dropout(0.5) after the first norm+ReLU (networks.py:493-494)
                r = E.dropout(ctx, r, 0.5)
            # the last block's output is read by the first up-convolution only (every other one is the next block's residual as well)
            last_out = b['up'][0][0] if (ib == nblk - 1 and self.upsample == 'convtranspose') else None
            h = E.norm_act(ctx, E.conv(ctx, r, c2, stats=n2 is not None), n2, L.ACT_NONE, residual=h, sole_reader=last_out)
        for i, (c, n) in enumerate(b['up']):
            if self.upsample == 'resize_conv':
                h = E.upsample2(ctx, h)
            nxt = b['up'][i + 1][0] if (i + 1 < len(b['up']) and self.upsample == 'convtranspose') else None
            h = E.norm_act(ctx, E.conv(ctx, h, c, stats=n is not None), n, L.ACT_RELU, sole_reader=nxt)
        return E.conv(ctx, h, b['head'], act=L.ACT_TANH)


# -------------------------------------------------------------------------------------------------------------
# UnetGenerator
# -------------------------------------------------------------------------------------------------------------
class UnetSkipConnectionBlock(nn.Module):
    """Parameter container reproducing the reference's nested Sequential layout (networks.py:583-609)."""

    def __init__(self, outer_nc, inner_nc, input_nc=None, submodule=None, outermost=False, innermost=False, norm_layer=nn.BatchNorm2d,
                 use_dropout=False):
        super().__init__()
        self.outermost, self.innermost = outermost, innermost
        use_bias = _uses_bias(norm_layer)
        if input_nc is None:
            input_nc = outer_nc
        self.outer_nc, self.inner_nc, self.input_nc = outer_nc, inner_nc, input_nc
        downconv = nn.Conv2d(input_nc, inner_nc, kernel_size=4, stride=2, padding=1, bias=use_bias)
        downrelu = nn.LeakyReLU(0.2, True)
        downnorm = norm_layer(inner_nc)
        uprelu = nn.ReLU(True)
        upnorm = norm_layer(outer_nc)
        self.use_dropout = False
        if outermost:
            upconv = nn.ConvTranspose2d(inner_nc * 2, outer_nc, kernel_size=4, stride=2, padding=1)
            seq = [downconv, submodule, uprelu, upconv, nn.Tanh()]
            self.pos = dict(down=0, sub=1, up=3)
        elif innermost:
            upconv = nn.ConvTranspose2d(inner_nc, outer_nc, kernel_size=4, stride=2, padding=1, bias=use_bias)
            seq = [downrelu, downconv, uprelu, upconv, upnorm]
            self.pos = dict(down=1, up=3, upnorm=4)
        else:
            upconv = nn.ConvTranspose2d(inner_nc * 2, outer_nc, kernel_size=4, stride=2, padding=1, bias=use_bias)
            seq = [downrelu, downconv, downnorm, submodule, uprelu, upconv, upnorm]
            self.pos = dict(down=1, downnorm=2, sub=3, up=5, upnorm=6)
            if use_dropout:
                seq.append(nn.Dropout(0.5))
                self.use_dropout = True
        self.model = nn.Sequential(*seq)


class UnetGenerator(EngineNet):
    def __init__(self, input_nc, output_nc, num_downs, ngf=64, norm_layer=nn.BatchNorm2d, use_dropout=False):
        super().__init__()
        self.norm_kind = _norm_kind(norm_layer)
        self.num_downs = num_downs
        blk = UnetSkipConnectionBlock(ngf * 8, ngf * 8, submodule=None, norm_layer=norm_layer, innermost=True)
        for _ in range(num_downs - 5):
            blk = UnetSkipConnectionBlock(ngf * 8, ngf * 8, submodule=blk, norm_layer=norm_layer, use_dropout=use_dropout)
        blk = UnetSkipConnectionBlock(ngf * 4, ngf * 8, submodule=blk, norm_layer=norm_layer)
        blk = UnetSkipConnectionBlock(ngf * 2, ngf * 4, submodule=blk, norm_layer=norm_layer)
        blk = UnetSkipConnectionBlock(ngf, ngf * 2, submodule=blk, norm_layer=norm_layer)
        self.model = UnetSkipConnectionBlock(output_nc, ngf, input_nc=input_nc, submodule=blk, outermost=True, norm_layer=norm_layer)

    def _bind(self):
        k = self.norm_kind
        chain = []
        blk = self.model
        while blk is not None:
            chain.append(blk)
            blk = None if blk.innermost else blk.model[blk.pos['sub']]
        levels = []
        for blk in chain:
            dm, um = blk.model[blk.pos['down']], blk.model[blk.pos['up']]
            down = E.ConvLayer(ConvSpec('conv', blk.input_nc, blk.inner_nc, 4, 2, 1), dm.weight, dm.bias)
            up_in = blk.inner_nc if blk.innermost else blk.inner_nc * 2
            up = E.ConvLayer(ConvSpec('convT', up_in, blk.outer_nc, 4, 2, 1), um.weight, um.bias)
            dn = _norm_binding(k, blk.inner_nc, blk.model[blk.pos['downnorm']]) if 'downnorm' in blk.pos else None
            un = _norm_binding(k, blk.outer_nc, blk.model[blk.pos['upnorm']]) if 'upnorm' in blk.pos else None
            levels.append(dict(block=blk, down=down, up=up, downnorm=dn, upnorm=un, has_downnorm='downnorm' in blk.pos,
                               has_upnorm='upnorm' in blk.pos))
        return levels

    def run(self, ctx: E.Ctx, x: E.Act) -> E.Act:
        """networks.py:611-615.  The in-place LeakyReLU makes every skip carry lrelu(h); the up-path ReLU then
        sees relu(lrelu(h)) = relu(h), so the concat buffer stores the raw pre-activations [h_d | u_d] and the consumers
        apply the activation while staging (in_act), with no standalone activation pass."""
        lv = self._layers()
        D = len(lv)
        dev, dt = x.t.device, ctx.prec.dtype
        n = x.t.shape[0]
        # ---- down path.  cat[d] (d = 1..D-1) = [h_d | u_d] at the resolution of block d's input
        cats: List[Optional[torch.Tensor]] = [None] * D
        firsts: List[Optional[E.Act]] = [None] * D
        h = x
        for d in range(D):
            l = lv[d]
            hi, wi = h.t.shape[1], h.t.shape[2]
            ho, wo = hi // 2, wi // 2
            cin_act = L.ACT_NONE if d == 0 else L.ACT_LRELU
            cin = l['down'].spec.cout
            if d < D - 1:
                # output of this down step is the input (pre-activation) of block d+1 -> first half of cat[d+1]
                cpad_c = E.cpad(cin)
                assert cpad_c == cin, 'UNet feature widths must be powers of two >= 8 for zero-copy concat'
                buf = torch.empty((n, ho, wo, 2 * cin), dtype=dt, device=dev)
                first = buf[..., :cin]
                if l['has_downnorm'] and l['downnorm'] is not None:
                    y = E.conv(ctx, h, l['down'], in_act=cin_act, stats=True)
                    hn = E.norm_act(ctx, y, l['downnorm'], L.ACT_NONE, out=first)
                else:
                    hn = E.conv(ctx, h, l['down'], in_act=cin_act, out=first)
                cats[d + 1], firsts[d + 1] = buf, hn
                h = hn
            else:
                h = E.conv(ctx, h, l['down'], in_act=cin_act)      # innermost: no down-norm
        # ---- up path
        u_in = h                                   # innermost down-conv output; consumers apply ReLU while staging
        for d in range(D - 1, 0, -1):
            l = lv[d]
            cat = cats[d]
            cout = l['up'].spec.cout
            second = cat[..., cout:]
            y = E.conv(ctx, u_in, l['up'], in_act=L.ACT_RELU, stats=True)
            drop = l['block'].use_dropout and l['block'].training      # nn.Dropout(0.5) on the block output (networks.py:604-605)
            u = E.norm_act(ctx, y, l['upnorm'], L.ACT_NONE, out=None if drop else second)
            if drop:
                ud = E.dropout(ctx, u, 0.5)
                ops_impl = E.ops.impl()
                ops_impl.axpby(1.0, ud.t, 0.0, None, second)           # place the dropped tensor into the concat buffer
                placed = E.Act(second, ud.C, ud.needs_grad)
                if ctx.tape is not None and ud.needs_grad:
                    def _route(ud=ud, placed=placed):
                        g = placed.grad
                        placed.grad = None
                        if g is not None:
                            ud.add_grad(g)
                    ctx.tape.record(_route)
                u = placed
            u_in = _JoinedAct(cat, 2 * cout, firsts[d], u, ctx)
        return E.conv(ctx, u_in, lv[0]['up'], act=L.ACT_TANH, in_act=L.ACT_RELU)


class _JoinedAct(E.Act):
    """A concat buffer whose halves were written in place by their producers; routes the gradient back to both."""
    __slots__ = ('first', 'second')

    def __init__(self, t, C, first: E.Act, second: E.Act, ctx: E.Ctx):
        super().__init__(t, C, ctx.tape is not None and (first.needs_grad or second.needs_grad))
        self.first, self.second = first, second

    def add_grad(self, g: torch.Tensor):
        c = self.first.t.shape[3]
        if self.first.needs_grad:
            self.first.add_grad(g[..., :c])
        if self.second.needs_grad:
            self.second.add_grad(g[..., c:])


# -------------------------------------------------------------------------------------------------------------
# Attention U-Net (`--net-gs unet_512_attention`, networks.py:189-190 -> att_unet.py:117-199)
# -------------------------------------------------------------------------------------------------------------
class conv_block(nn.Module):
    """Parameter container, att_unet.py:32-55: Conv2d(k4 s2 p1, bias) [+ BatchNorm2d] + LeakyReLU(0.2) / ReLU (innermost)."""

    def __init__(self, ch_in, ch_out, innermost=False, outermost=False):
        super().__init__()
        conv = nn.Conv2d(ch_in, ch_out, kernel_size=4, stride=2, padding=1, bias=True)
        if outermost:
            self.conv = nn.Sequential(conv, nn.LeakyReLU(0.2, True))
        elif innermost:
            self.conv = nn.Sequential(conv, nn.ReLU(inplace=True))
        else:
            self.conv = nn.Sequential(conv, nn.BatchNorm2d(ch_out), nn.LeakyReLU(0.2, True))
        self.ch_in, self.ch_out, self.innermost, self.outermost = ch_in, ch_out, innermost, outermost


class up_conv(nn.Module):
    """att_unet.py:57-85: ConvTranspose2d(k4 s2 p1) + BatchNorm2d + ReLU; the outermost one has a bias and ends in Tanh; every one but the
    innermost reads the concatenation [gated skip | up] (ch_in * 2 channels)."""

    def __init__(self, ch_in, ch_out, innermost=False, outermost=False):
        super().__init__()
        if outermost:
            self.up = nn.Sequential(nn.ConvTranspose2d(ch_in * 2, ch_out, kernel_size=4, stride=2, padding=1), nn.Tanh())
        else:
            self.up = nn.Sequential(nn.ConvTranspose2d(ch_in if innermost else ch_in * 2, ch_out, kernel_size=4, stride=2, padding=1, bias=False),
                                    nn.BatchNorm2d(ch_out), nn.ReLU(True))
        self.ch_in, self.ch_out, self.innermost, self.outermost = ch_in, ch_out, innermost, outermost


class Attention_block(nn.Module):
    """att_unet.py:88-115: psi = Sigmoid(BN(Conv1x1(relu(BN(Conv1x1(g)) + BN(Conv1x1(x)))))) with ONE channel; returns x * psi."""

    def __init__(self, F_g, F_l, F_int):
        super().__init__()
        self.W_g = nn.Sequential(nn.Conv2d(F_g, F_int, kernel_size=1, stride=1, padding=0, bias=True), nn.BatchNorm2d(F_int))
        self.W_x = nn.Sequential(nn.Conv2d(F_l, F_int, kernel_size=1, stride=1, padding=0, bias=True), nn.BatchNorm2d(F_int))
        self.psi = nn.Sequential(nn.Conv2d(F_int, 1, kernel_size=1, stride=1, padding=0, bias=True), nn.BatchNorm2d(1), nn.Sigmoid())
        self.relu = nn.ReLU(inplace=True)
        self.F_g, self.F_l, self.F_int = F_g, F_l, F_int


class AttU_Net(EngineNet):
    """att_unet.py:117-199, same module tree and state_dict keys (Conv1..Conv8, Up8, Att8, ..., Up2, Att2, Up1; widths are fixed at
    64 ... 512, BatchNorm2d is hard-wired: define_G passes neither ngf nor the norm layer).  Eight stride-2 levels: the input side must be a
    multiple of 256.  On the engine: the 4x4 stride-2 convs and transposed convs are the UNet's gather GEMMs (bias / LeakyReLU / ReLU / Tanh in
    the epilogue or behind the fused norm kernel), the attention blocks' 1x1 convs are plain GEMMs over the pixels, relu(g1 + x1) is the
    residual form of the norm kernel + an input activation of the psi conv, x * psi is dl_gate_forward / _backward, and
    torch.cat((x_gated, d), 1) is zero-copy: the gate and the up path's norm write the two halves of one buffer."""

    def __init__(self, img_ch=3, output_ch=1):
        super().__init__()
        self.Conv1 = conv_block(img_ch, 64, outermost=True)
        self.Conv2 = conv_block(64, 128)
        self.Conv3 = conv_block(128, 256)
        self.Conv4 = conv_block(256, 512)
        self.Conv5 = conv_block(512, 512)
        self.Conv6 = conv_block(512, 512)
        self.Conv7 = conv_block(512, 512)
        self.Conv8 = conv_block(512, 512, innermost=True)
        self.Up8 = up_conv(512, 512, innermost=True)
        self.Att8 = Attention_block(512, 512, 512)
        self.Up7 = up_conv(512, 512)
        self.Att7 = Attention_block(512, 512, 512)
        self.Up6 = up_conv(512, 512)
        self.Att6 = Attention_block(512, 512, 512)
        self.Up5 = up_conv(512, 512)
        self.Att5 = Attention_block(512, 512, 512)
        self.Up4 = up_conv(512, 256)
        self.Att4 = Attention_block(256, 256, 128)
        self.Up3 = up_conv(256, 128)
        self.Att3 = Attention_block(128, 128, 64)
        self.Up2 = up_conv(128, 64)
        self.Att2 = Attention_block(64, 64, 32)
        self.Up1 = up_conv(64, output_ch, outermost=True)
        self.norm_kind = 'batch'

    def _bind(self):
        def conv_of(seq, spec):
            return E.ConvLayer(spec, seq[0].weight, seq[0].bias)

        def bn_of(seq, c, idx=1):
            return E.NormLayer('batch', c, seq[idx]) if isinstance(seq[idx], nn.BatchNorm2d) else None
        downs = []
        for k in range(1, 9):
            blk = getattr(self, f'Conv{k}')
            downs.append((conv_of(blk.conv, ConvSpec('conv', blk.ch_in, blk.ch_out, 4, 2, 1)), bn_of(blk.conv, blk.ch_out), blk))
        ups, atts = {}, {}
        for k in range(8, 0, -1):
            blk = getattr(self, f'Up{k}')
            cin = blk.up[0].weight.shape[0]
            ups[k] = (conv_of(blk.up, ConvSpec('convT', cin, blk.ch_out, 4, 2, 1)), bn_of(blk.up, blk.ch_out), blk)
            if k >= 2:
                a = getattr(self, f'Att{k}')
                atts[k] = dict(wg=(conv_of(a.W_g, ConvSpec('conv', a.F_g, a.F_int, 1, 1, 0)), bn_of(a.W_g, a.F_int)),
                               wx=(conv_of(a.W_x, ConvSpec('conv', a.F_l, a.F_int, 1, 1, 0)), bn_of(a.W_x, a.F_int)),
                               psi=(conv_of(a.psi, ConvSpec('conv', a.F_int, 1, 1, 1, 0)), bn_of(a.psi, 1)))
        return dict(downs=downs, ups=ups, atts=atts)

    def run(self, ctx: E.Ctx, x: E.Act) -> E.Act:
        """att_unet.py:153-199."""
        b = self._layers()
        assert x.t.shape[1] % 256 == 0 and x.t.shape[2] % 256 == 0, 'AttU_Net halves the image eight times: H and W must be multiples of 256'
        xs = []
        h = x
        for conv, bn, blk in b['downs']:
            if bn is None:
                h = E.conv(ctx, h, conv, act=L.ACT_RELU if blk.innermost else L.ACT_LRELU)
            else:
                h = E.norm_act(ctx, E.conv(ctx, h, conv, stats=True), bn, L.ACT_LRELU)
            xs.append(h)                                    # xs[k - 1] = x_k
        n = x.t.shape[0]
        src = xs[7]                                         # x8
        for k in range(8, 1, -1):
            conv, bn, _ = b['ups'][k]
            skip = xs[k - 2]                                # x_{k-1}: what Att_k gates
            c = conv.spec.cout
            assert skip.C == c and E.cpad(c) == c
            hh, ww = skip.t.shape[1], skip.t.shape[2]
            cat = torch.empty((n, hh, ww, 2 * c), dtype=ctx.prec.dtype, device=x.t.device)
            d = E.norm_act(ctx, E.conv(ctx, src, conv, stats=True), bn, L.ACT_RELU, out=cat[..., c:])
            a = b['atts'][k]
            g1 = E.norm_act(ctx, E.conv(ctx, d, a['wg'][0], stats=True), a['wg'][1], L.ACT_NONE)
            s = E.norm_act(ctx, E.conv(ctx, skip, a['wx'][0], stats=True), a['wx'][1], L.ACT_NONE, residual=g1)          # g1 + x1
            p = E.norm_act(ctx, E.conv(ctx, s, a['psi'][0], in_act=L.ACT_RELU, stats=True), a['psi'][1], L.ACT_NONE)     # psi conv over relu(g1 + x1)
            p = E.act_op(ctx, p, L.ACT_SIGMOID)
            xg = E.gate(ctx, skip, p, out=cat[..., :c])
            src = _JoinedAct(cat, 2 * c, xg, d, ctx)
        conv, _, _ = b['ups'][1]
        return E.conv(ctx, src, conv, act=L.ACT_TANH)


# -------------------------------------------------------------------------------------------------------------
# NLayerDiscriminator
# -------------------------------------------------------------------------------------------------------------
class NLayerDiscriminator(EngineNet):
    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, use_spectral_norm=False):
        super().__init__()
        if use_spectral_norm:
            raise NotImplementedError('spectral norm is not on the MI355X hot path')
        self.norm_kind = _norm_kind(norm_layer)
        use_bias = _uses_bias(norm_layer)
        self.input_nc, self.ndf, self.n_layers = input_nc, ndf, n_layers
        seq: List[nn.Module] = [nn.Conv2d(input_nc, ndf, kernel_size=4, stride=2, padding=1), nn.LeakyReLU(0.2, True)]
        prev = 1
        self._chan = []
        for n in range(1, n_layers + 1):
            mult = min(2 ** n, 8)
            stride = 2 if n < n_layers else 1
            seq += [nn.Conv2d(ndf * prev, ndf * mult, kernel_size=4, stride=stride, padding=1, bias=use_bias), norm_layer(ndf * mult),
                    nn.LeakyReLU(0.2, True)]
            self._chan.append((ndf * prev, ndf * mult, stride))
            prev = mult
        seq.append(nn.Conv2d(ndf * prev, 1, kernel_size=4, stride=1, padding=1))
        self._last_in = ndf * prev
        self.model = nn.Sequential(*seq)

    def _bind(self):
        m, k = self.model, self.norm_kind
        b = {'first': E.ConvLayer(ConvSpec('conv', self.input_nc, self.ndf, 4, 2, 1), m[0].weight, m[0].bias), 'mid': []}
        idx = 2
        for cin, cout, stride in self._chan:
            b['mid'].append((E.ConvLayer(ConvSpec('conv', cin, cout, 4, stride, 1), m[idx].weight, m[idx].bias), _norm_binding(k, cout, m[idx + 1])))
            idx += 3
        b['last'] = E.ConvLayer(ConvSpec('conv', self._last_in, 1, 4, 1, 1), m[idx].weight, m[idx].bias)
        return b

    def run(self, ctx: E.Ctx, x: E.Act) -> E.Act:
        b = self._layers()
        h = E.conv(ctx, x, b['first'], act=L.ACT_LRELU)
        for i, (c, n) in enumerate(b['mid']):
            nxt = b['mid'][i + 1][0] if i + 1 < len(b['mid']) else b['last']          # the only reader of this activation
            h = E.norm_act(ctx, E.conv(ctx, h, c, stats=n is not None), n, L.ACT_LRELU, sole_reader=nxt)
        return E.conv(ctx, h, b['last'])

    def forward(self, x):
        prec = E.Precision.get(self.precision)
        with E.ops.half_mode(prec.half):
            ctx = E.Ctx(prec, None, training=False, per_sample_norm=False)
            return E.from_engine(self.run(ctx, E.to_engine(x, prec)))


class PixelDiscriminator(EngineNet):
    """networks.py:667-696 (--net-d pixel): a 1x1 PatchGAN -- conv1x1 (bias) + LeakyReLU, conv1x1 + norm + LeakyReLU, conv1x1 -> 1 channel."""

    def __init__(self, input_nc, ndf=64, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.norm_kind = _norm_kind(norm_layer)
        use_bias = _uses_bias(norm_layer)
        self.input_nc, self.ndf = input_nc, ndf
        self.net = nn.Sequential(nn.Conv2d(input_nc, ndf, kernel_size=1, stride=1, padding=0), nn.LeakyReLU(0.2, True),
                                 nn.Conv2d(ndf, ndf * 2, kernel_size=1, stride=1, padding=0, bias=use_bias), norm_layer(ndf * 2), nn.LeakyReLU(0.2, True),
                                 nn.Conv2d(ndf * 2, 1, kernel_size=1, stride=1, padding=0, bias=use_bias))

    def _bind(self):
        m = self.net
        return {'c0': E.ConvLayer(ConvSpec('conv', self.input_nc, self.ndf, 1, 1, 0), m[0].weight, m[0].bias),
                'c1': (E.ConvLayer(ConvSpec('conv', self.ndf, self.ndf * 2, 1, 1, 0), m[2].weight, m[2].bias), _norm_binding(self.norm_kind, self.ndf * 2, m[3])),
                'c2': E.ConvLayer(ConvSpec('conv', self.ndf * 2, 1, 1, 1, 0), m[5].weight, m[5].bias)}

    def run(self, ctx: E.Ctx, x: E.Act) -> E.Act:
        b = self._layers()
        h = E.conv(ctx, x, b['c0'], act=L.ACT_LRELU)
        c, n = b['c1']
        h = E.norm_act(ctx, E.conv(ctx, h, c, stats=n is not None), n, L.ACT_LRELU)
        return E.conv(ctx, h, b['c2'])

    def forward(self, x):
        prec = E.Precision.get(self.precision)
        with E.ops.half_mode(prec.half):
            ctx = E.Ctx(prec, None, training=False, per_sample_norm=False)
            return E.from_engine(self.run(ctx, E.to_engine(x, prec)))


# -------------------------------------------------------------------------------------------------------------
# factories / init  (networks.py:84-238)
# -------------------------------------------------------------------------------------------------------------
def init_weights(net, init_type='normal', init_gain=0.02):
    def init_func(m):
        classname = m.__class__.__name__
        if hasattr(m, 'weight') and (classname.find('Conv') != -1 or classname.find('Linear') != -1):
            if init_type == 'normal':
                init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == 'xavier':
                init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == 'kaiming':
                init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
            elif init_type == 'orthogonal':
                init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
            if hasattr(m, 'bias') and m.bias is not None:
                init.constant_(m.bias.data, 0.0)
        elif classname.find('BatchNorm2d') != -1:
            init.normal_(m.weight.data, 1.0, init_gain)
            init.constant_(m.bias.data, 0.0)
    net.apply(init_func)


def init_net(net, init_type='normal', init_gain=0.02, gpu_ids=[]):
    """networks.py:118-139.  One process per GPU: the net goes to gpu_ids[0]; there is no DataParallel / DDP wrapper
    (gradient exchange is done on flat buffers by deepliif_amd.distributed, not by per-module reducers)."""
    if len(gpu_ids) > 0:
        assert torch.cuda.is_available()
        net.to(torch.device('cuda', gpu_ids[0]))
    init_weights(net, init_type, init_gain=init_gain)
    return net


def define_G(input_nc, output_nc, ngf, netG, norm='batch', use_dropout=False, init_type='normal', init_gain=0.02, gpu_ids=[],
             padding_type='reflect', upsample='convtranspose'):
    norm_layer = get_norm_layer(norm_type=norm)
    if netG.startswith('resnet_'):
        n_blocks = int(netG.split('_')[1].replace('blocks', ''))
        net = ResnetGenerator(input_nc, output_nc, ngf, norm_layer=norm_layer, use_dropout=use_dropout, n_blocks=n_blocks,
                              padding_type=padding_type, upsample=upsample)
    elif netG in ('unet_32', 'unet_64', 'unet_128', 'unet_256', 'unet_512'):
        downs = {'unet_32': 5, 'unet_64': 6, 'unet_128': 7, 'unet_256': 8, 'unet_512': 9}[netG]
        net = UnetGenerator(input_nc, output_nc, downs, ngf, norm_layer=norm_layer, use_dropout=use_dropout)
    elif netG == 'unet_512_attention':
        net = AttU_Net(img_ch=input_nc, output_ch=output_nc)       # networks.py:189-190: ngf / norm / dropout are not forwarded
    else:
        raise NotImplementedError('Generator model name [%s] is not on the MI355X hot path' % netG)
    return init_net(net, init_type, init_gain, gpu_ids)


def define_D(input_nc, ndf, netD, n_layers_D=3, norm='batch', init_type='normal', init_gain=0.02, gpu_ids=[]):
    norm_layer = get_norm_layer(norm_type=norm)
    if netD == 'basic':
        net = NLayerDiscriminator(input_nc, ndf, n_layers=3, norm_layer=norm_layer)
    elif netD == 'n_layers':
        net = NLayerDiscriminator(input_nc, ndf, n_layers_D, norm_layer=norm_layer)
    elif netD == 'pixel':
        net = PixelDiscriminator(input_nc, ndf, norm_layer=norm_layer)
    else:
        raise NotImplementedError('Discriminator model name [%s] is not on the MI355X hot path' % netD)
    return init_net(net, init_type, init_gain, gpu_ids)


# -------------------------------------------------------------------------------------------------------------
# VGG19 perceptual loss (networks.py:698-743)
# -------------------------------------------------------------------------------------------------------------
VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']     # torchvision cfg 'E'
VGG19_SLICE_ENDS = (2, 7, 12, 21, 30)       # Vgg19.slice1..5 = features[0:2], [2:7], [7:12], [12:21], [21:30] (networks.py:707-716)


class Vgg19(EngineNet):
    """torchvision.models.vgg19().features[0:30] as the reference slices it (networks.py:698-731), frozen.  `features` has torchvision's
    module indices, so a torchvision `vgg19` state_dict ('features.N.weight' / '.bias'; classifier keys are ignored) loads directly:
    the pretrained weights the reference downloads must be supplied as a FILE here (no network on the box): opt.vgg_weights or
    $DEEPLIIF_VGG19_WEIGHTS.  Conv + ReLU run as one conv kernel with a ReLU epilogue; 2x2 max pooling is dl_maxpool2_*."""

    def __init__(self):
        super().__init__()
        layers: List[nn.Module] = []
        cin = 3
        for v in VGG19_CFG:
            if v == 'M':
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers[:VGG19_SLICE_ENDS[-1]])
        for p in self.parameters():
            p.requires_grad = False

    def load_torchvision_state_dict(self, sd):
        own = self.state_dict()
        picked = {k: v for k, v in sd.items() if k in own}
        missing = [k for k in own if k not in picked]
        if missing:
            raise KeyError(f'VGG19 weight file lacks {missing[:4]}... (expected torchvision vgg19 keys features.N.weight / .bias)')
        self.load_state_dict(picked)

    def _bind(self):
        prog = []
        for i, m in enumerate(self.features):
            if isinstance(m, nn.Conv2d):
                prog.append((i, E.ConvLayer(ConvSpec('conv', m.in_channels, m.out_channels, 3, 1, 1), m.weight, m.bias)))
            elif isinstance(m, nn.MaxPool2d):
                prog.append((i, None))
        return prog

    def run(self, ctx: E.Ctx, x: E.Act):
        """-> [h_relu1 .. h_relu5] (networks.py:722-731)"""
        outs, h = [], x
        for i, layer in self._layers():
            h = E.maxpool2(ctx, h) if layer is None else E.conv(ctx, h, layer, act=L.ACT_RELU)
            if i + 2 in VGG19_SLICE_ENDS:            # the ReLU that closes a slice sits right after this conv
                outs.append(h)
        return outs


class VGGLoss(nn.Module):
    """networks.py:732-743: sum_i w_i * L1(vgg(x)_i, vgg(y)_i.detach()), w = [1/32, 1/16, 1/8, 1/4, 1]."""
    weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]

    def __init__(self, weights_path=None, device=None, precision=None):
        super().__init__()
        self.vgg = Vgg19()
        if device is not None:
            self.vgg.to(device)
        if precision:
            self.vgg.set_precision(precision)
        if weights_path:
            sd = torch.load(weights_path, map_location='cpu')
            self.vgg.load_torchvision_state_dict(sd.get('state_dict', sd) if isinstance(sd, dict) else sd)

    def run(self, ctx: E.Ctx, x: E.Act, y: E.Act, weight: float, loss_out: torch.Tensor, out_scale: float = 1.0, accumulate: bool = False):
        """loss_out[0] (+)= out_scale * the VGG loss (unweighted by `weight`); queues d(weight * loss)/dx on the tape.  The target features carry
        no gradient."""
        fy = self.vgg.run(E.Ctx(ctx.prec, None, training=False), y.detach())
        fx = self.vgg.run(ctx, x)
        for i, (a, b) in enumerate(zip(fx, fy)):
            E.loss_op(ctx, L.LOSS_L1, a, b, 0.0, weight * self.weights[i], loss_out, out_scale=out_scale * self.weights[i], accumulate=accumulate or i > 0)


def vgg_weights_path(opt):
    return getattr(opt, 'vgg_weights', None) or os.environ.get('DEEPLIIF_VGG19_WEIGHTS')


# -------------------------------------------------------------------------------------------------------------
# losses / optimiser / schedule
# -------------------------------------------------------------------------------------------------------------
class GANLoss(nn.Module):
    """networks.py:244-317, modes 'vanilla' (BCE-with-logits) and 'lsgan' (MSE) against a constant target.
    On engine activations use `.kind` / `.target(...)` with engine.loss_op; calling it on a plain tensor returns the loss
    computed by the HIP loss kernel as a 0-dim device tensor (no gradient graph)."""

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0, label_smoothing=0.0):
        super().__init__()
        self.register_buffer('real_label', torch.tensor(target_real_label))
        self.register_buffer('fake_label', torch.tensor(target_fake_label))
        self.gan_mode = gan_mode
        self.label_smoothing = label_smoothing
        # the labels as host floats: target() runs for every loss term of every step, and float(<device buffer>) is a device-to-host copy that waits
        # for the whole queue (and is not allowed while a hipGraph is being captured, models.StepGraph)
        self._labels = (float(target_real_label), float(target_fake_label))
        if gan_mode == 'lsgan':
            self.kind = L.LOSS_MSE
        elif gan_mode == 'vanilla':
            self.kind = L.LOSS_BCE_LOGITS
        elif gan_mode == 'wgangp':
            self.kind = L.LOSS_LINEAR            # -mean(pred) for real, +mean(pred) for fake (networks.py:307-311); no model class adds the gradient penalty
        else:
            raise NotImplementedError('gan mode %s is not on the MI355X hot path' % gan_mode)

    def target(self, target_is_real: bool) -> float:
        if self.gan_mode == 'wgangp':
            return -1.0 if target_is_real else 1.0       # the sign dl_loss multiplies the prediction with (DL_LOSS_LINEAR)
        if target_is_real:
            return self._labels[0] * (1 - self.label_smoothing)
        return self._labels[1] * self.label_smoothing

    def __call__(self, prediction, target_is_real):
        prec = E.Precision.get('fp32_bf16mma')
        a = prediction if isinstance(prediction, E.Act) else E.to_engine(prediction, prec)
        out = torch.zeros(1, dtype=torch.float32, device=a.t.device)
        E.loss_op(E.Ctx(prec, None, False), self.kind, a, None, self.target(target_is_real), 1.0, out)
        return out[0]


def get_optimizer(optimizer_name):
    """networks.py:46-53.  'adam' maps to the fused flat Adam kernel; any other torch.optim class keeps its own (ATen) update rule on
    a flat parameter set (optim.flat_optimizer) -- functional, not the MI355X hot path."""
    from .optim import FusedAdam, flat_optimizer
    if optimizer_name.lower() == 'adam':
        return FusedAdam
    names = {n.lower(): n for n in dir(torch.optim) if n[0].isupper()}
    try:
        return flat_optimizer(getattr(torch.optim, names[optimizer_name.lower()]))
    except KeyError:
        raise NotImplementedError('optimizer [%s] is not found' % optimizer_name)


def get_scheduler(optimizer, opt):
    """networks.py:55-81."""
    if opt.lr_policy == 'linear':
        def lambda_rule(epoch):
            return 1.0 - max(0, epoch + opt.epoch_count - opt.n_epochs) / float(opt.n_epochs_decay + 1)
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule)
    if opt.lr_policy == 'step':
        return lr_scheduler.StepLR(optimizer, step_size=opt.lr_decay_iters, gamma=0.1)
    if opt.lr_policy == 'plateau':
        return lr_scheduler.ReduceLROnPlateau(optimizer, mode='min', factor=0.2, threshold=0.01, patience=5)
    if opt.lr_policy == 'cosine':
        return lr_scheduler.CosineAnnealingLR(optimizer, T_max=opt.n_epochs, eta_min=0)
    raise NotImplementedError('learning rate policy [%s] is not implemented' % opt.lr_policy)
