"""CPU oracle for the DeepLIIF cGAN hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  The product path (deepliif_amd/) never imports it and has no CPU
fallback.

What it is: a functional fp32 restatement (torch CPU ops: F.conv2d /
F.conv_transpose2d + hand-written norm / activation / loss / Adam arithmetic) of
the reference algorithm, written against reference-keyed state_dicts so that the
same weights can be fed to the reference modules, to this oracle and to the HIP
engine.  Each function cites the reference file:line it restates
(paths relative to /root/reference).

Pinning: the reference's own tests hold no numeric vectors for this path
(SURVEY.md 4 / 8c), so the oracle is pinned against outputs of the reference
itself, generated in the build container by tests/golden/make_golden.py
(which imports /root/reference) and committed as tests/golden/*.npz.
tests/test_oracle_golden.py checks every fixture.

Third-party arithmetic: all conv / norm / loss / optimizer arithmetic of the
reference lives in PyTorch (setup.py:23 pins torch==2.8.0; this image has
2.10.0).  The oracle uses the same documented nn.functional semantics.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

EPS = 1e-5          # nn.BatchNorm2d / nn.InstanceNorm2d default eps (networks.py:35-37)
BN_MOMENTUM = 0.1   # nn.BatchNorm2d default momentum


# ----------------------------------------------------------------------------------------------
# norm / activations
# ----------------------------------------------------------------------------------------------
def norm2d(x: torch.Tensor, kind: str, weight: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
           running: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """networks.py:25-44 get_norm_layer.

    'batch'    : BatchNorm2d(affine=True) evaluated on *batch* statistics -- in .train() by definition and
                 in .eval() because disable_batchnorm_tracking_stats nulls the running buffers
                 (deepliif/util/__init__.py:743-755).  Reduction over (N,H,W), biased variance.
                 If `running` is given (training with tracking on) the running stats are updated in place
                 with momentum 0.1 and the unbiased variance, as nn.BatchNorm2d does.
    'instance' : InstanceNorm2d(affine=False, track_running_stats=False): reduction over (H,W) per sample.
    'none'     : identity.
    """
    if kind == 'none':
        return x
    if kind == 'instance':
        dims = (2, 3)
    elif kind == 'batch':
        dims = (0, 2, 3)
    else:
        raise NotImplementedError(kind)
    mean = x.mean(dim=dims, keepdim=True)
    var = x.var(dim=dims, unbiased=False, keepdim=True)
    y = (x - mean) / torch.sqrt(var + EPS)
    if kind == 'batch':
        if running is not None:
            n = x.numel() // x.shape[1]
            with torch.no_grad():
                running['running_mean'].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.reshape(-1))
                running['running_var'].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var.reshape(-1) * n / max(n - 1, 1))
                running['num_batches_tracked'] += 1
        y = y * weight.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    return y


def _norm_from_sd(sd, prefix, x, kind, update_running=False):
    if kind == 'batch':
        running = None
        if update_running:
            running = {k: sd[f'{prefix}.{k}'] for k in ('running_mean', 'running_var', 'num_batches_tracked')}
        return norm2d(x, kind, sd[f'{prefix}.weight'], sd[f'{prefix}.bias'], running)
    return norm2d(x, kind)


def _pad(x, p, mode):
    if p == 0:
        return x
    if mode == 'zero':
        return F.pad(x, (p, p, p, p))
    if mode == 'reflect':
        return F.pad(x, (p, p, p, p), mode='reflect')
    if mode == 'replicate':
        return F.pad(x, (p, p, p, p), mode='replicate')
    raise NotImplementedError(mode)


# ----------------------------------------------------------------------------------------------
# networks
# ----------------------------------------------------------------------------------------------
def resnet_generator(sd: Dict[str, torch.Tensor], x: torch.Tensor, norm: str = 'batch', padding_type: str = 'zero',
                     n_blocks: int = 9, update_running: bool = False, upsample: str = 'convtranspose') -> torch.Tensor:
    """ResnetGenerator.forward, networks.py:357-450 (+ ResnetBlock :453-513), dropout off, upsample='convtranspose'.

    state_dict keys follow nn.Sequential indices of the reference: with no dropout
      model.1 (7x7), model.2 norm, model.4/5 and model.7/8 (downs), model.10..10+n_blocks-1 (blocks,
      conv_block.{0|1}=conv1 ... ), then two ConvTranspose2d and the 7x7 head.
    For padding 'reflect'/'replicate' each block has explicit pad layers, which shifts the indices inside
    conv_block (pad,conv,norm,relu,pad,conv,norm) -> conv at 1 and 5, norm at 2 and 6; with 'zero' conv at 0 and 3,
    norm at 1 and 4 (networks.py:479-508).
    """
    def bias(k):
        return sd.get(k)
    # stem: pad3 + conv7 + norm + relu (networks.py:386-397)
    h = _pad(x, 3, 'reflect' if padding_type == 'reflect' else 'zero')
    h = F.conv2d(h, sd['model.1.weight'], bias('model.1.bias'))
    h = torch.relu(_norm_from_sd(sd, 'model.2', h, norm, update_running))
    # two stride-2 downs, always zero padding=1 (networks.py:400-404)
    idx = 4
    for _ in range(2):
        h = F.conv2d(h, sd[f'model.{idx}.weight'], bias(f'model.{idx}.bias'), stride=2, padding=1)
        h = torch.relu(_norm_from_sd(sd, f'model.{idx + 1}', h, norm, update_running))
        idx += 3
    # residual blocks (networks.py:407-410, 467-513)
    if padding_type == 'zero':
        c1, n1, c2, n2, p = 0, 1, 3, 4, 'zero'
    else:
        c1, n1, c2, n2, p = 1, 2, 5, 6, padding_type
    for b in range(n_blocks):
        pre = f'model.{idx}.conv_block'
        r = F.conv2d(_pad(h, 1, p), sd[f'{pre}.{c1}.weight'], bias(f'{pre}.{c1}.bias'))
        r = torch.relu(_norm_from_sd(sd, f'{pre}.{n1}', r, norm, update_running))
        r = F.conv2d(_pad(r, 1, p), sd[f'{pre}.{c2}.weight'], bias(f'{pre}.{c2}.bias'))
        r = _norm_from_sd(sd, f'{pre}.{n2}', r, norm, update_running)
        h = h + r
        idx += 1
    # two ConvTranspose2d k3 s2 p1 op1 (networks.py:425-436), or with upsample='resize_conv' (:409-415)
    # [Upsample(x2, nearest), ReflectionPad2d(1), Conv2d(k3, bias)] at model.idx .. idx+2, norm at idx+3
    for _ in range(2):
        if upsample == 'resize_conv':
            h = F.interpolate(h, scale_factor=2, mode='nearest')
            h = F.conv2d(_pad(h, 1, 'reflect'), sd[f'model.{idx + 2}.weight'], sd[f'model.{idx + 2}.bias'])
            h = torch.relu(_norm_from_sd(sd, f'model.{idx + 3}', h, norm, update_running))
            idx += 5
            continue
        h = F.conv_transpose2d(h, sd[f'model.{idx}.weight'], bias(f'model.{idx}.bias'), stride=2, padding=1,
                               output_padding=1)
        h = torch.relu(_norm_from_sd(sd, f'model.{idx + 1}', h, norm, update_running))
        idx += 3
    # head: pad3 + conv7 (bias always) + tanh (networks.py:438-444)
    h = _pad(h, 3, 'reflect' if padding_type == 'reflect' else 'zero')
    idx += 1
    h = F.conv2d(h, sd[f'model.{idx}.weight'], sd[f'model.{idx}.bias'])
    return torch.tanh(h)


def unet_generator(sd: Dict[str, torch.Tensor], x: torch.Tensor, norm: str = 'batch', num_downs: int = 9,
                   update_running: bool = False) -> torch.Tensor:
    """UnetGenerator.forward, networks.py:516-615, dropout off.

    Recursion restated as an explicit down list / up loop.  Block d (0 = outermost) lives at key prefix
    'model' + '.model.1' * d  for d = 0 and  'model.model.1' + '.model.3' * (d-1)  for d >= 1 -- i.e. the
    submodule sits at Sequential index 1 in the outermost block (down=[conv]) and at index 3 in the middle blocks
    (down=[lrelu, conv, norm]) (networks.py:583-607).

    In-place LeakyReLU(0.2, True) (networks.py:578) mutates the block input before torch.cat (:615), so the skip
    half of every concat is lrelu(x), not x; the up path's ReLU(True) then sees relu(lrelu(x)).
    """
    def prefix(d):
        if d == 0:
            return 'model'
        return 'model.model.1' + '.model.3' * (d - 1)

    def nrm(key, t):
        return _norm_from_sd(sd, key, t, norm, update_running)

    # ---- down path
    skips: List[torch.Tensor] = []      # skips[d] = (activated) input of block d, d >= 1
    h = F.conv2d(x, sd['model.model.0.weight'], sd.get('model.model.0.bias'), stride=2, padding=1)  # outermost downconv
    for d in range(1, num_downs):
        pre = prefix(d) + '.model'
        a = F.leaky_relu(h, 0.2)        # in-place in the reference: this IS what the skip carries
        skips.append(a)
        h = F.conv2d(a, sd[f'{pre}.1.weight'], sd.get(f'{pre}.1.bias'), stride=2, padding=1)
        if d != num_downs - 1:          # innermost has no down-norm (networks.py:590-596)
            h = nrm(f'{pre}.2', h)
    # ---- up path
    for d in range(num_downs - 1, 0, -1):
        pre = prefix(d) + '.model'
        if d == num_downs - 1:          # innermost: model = [lrelu, conv, relu, convT, norm]
            ci, ni = 3, 4
        else:                           # middle:    model = [lrelu, conv, norm, sub, relu, convT, norm]
            ci, ni = 5, 6
        u = F.conv_transpose2d(torch.relu(h), sd[f'{pre}.{ci}.weight'], sd.get(f'{pre}.{ci}.bias'), stride=2, padding=1)
        u = nrm(f'{pre}.{ni}', u)
        h = torch.cat([skips[d - 1], u], 1)
    # outermost: [downconv, sub, relu, convT(bias), tanh]
    y = F.conv_transpose2d(torch.relu(h), sd['model.model.3.weight'], sd['model.model.3.bias'], stride=2, padding=1)
    return torch.tanh(y)


ATT_DOWN = [(None, 64), (64, 128), (128, 256), (256, 512), (512, 512), (512, 512), (512, 512), (512, 512)]      # Conv1..Conv8 (ch_in of Conv1 = img_ch)
ATT_UP = {8: (512, 512), 7: (512, 512), 6: (512, 512), 5: (512, 512), 4: (512, 256), 3: (256, 128), 2: (128, 64)}     # Up_k: ch_in (x2 when concatenated), ch_out
ATT_GATE = {8: (512, 512), 7: (512, 512), 6: (512, 512), 5: (512, 512), 4: (256, 128), 3: (128, 64), 2: (64, 32)}     # Att_k: F_g = F_l, F_int


def att_unet_generator(sd: Dict[str, torch.Tensor], x: torch.Tensor, update_running: bool = False) -> torch.Tensor:
    """AttU_Net.forward, deepliif/models/att_unet.py:153-199 (`--net-gs unet_512_attention`, networks.py:189-190).

    conv_block (:32-55): Conv2d(k4 s2 p1, bias) [+ BatchNorm2d] + LeakyReLU(0.2); Conv1 has no norm, Conv8 no norm and ReLU.
    up_conv (:57-85): ConvTranspose2d(k4 s2 p1, no bias) + BatchNorm2d + ReLU; Up1 has a bias and ends in Tanh.
    Attention_block (:88-115): psi = sigmoid(BN(conv1x1(relu(BN(conv1x1(g)) + BN(conv1x1(x)))))), one channel; returns x * psi.
    BatchNorm2d is hard-wired (define_G forwards neither norm nor ngf) and evaluated on batch statistics like every norm of the path."""
    def bn(key, t):
        return _norm_from_sd(sd, key, t, 'batch', update_running)

    xs = []
    h = x
    for k in range(1, 9):
        pre = f'Conv{k}.conv'
        h = F.conv2d(h, sd[f'{pre}.0.weight'], sd[f'{pre}.0.bias'], stride=2, padding=1)
        if k == 1:
            h = F.leaky_relu(h, 0.2)
        elif k == 8:
            h = torch.relu(h)
        else:
            h = F.leaky_relu(bn(f'{pre}.1', h), 0.2)
        xs.append(h)
    src = xs[7]
    for k in range(8, 1, -1):
        d = torch.relu(bn(f'Up{k}.up.1', F.conv_transpose2d(src, sd[f'Up{k}.up.0.weight'], None, stride=2, padding=1)))
        skip = xs[k - 2]
        a = f'Att{k}'
        g1 = bn(f'{a}.W_g.1', F.conv2d(d, sd[f'{a}.W_g.0.weight'], sd[f'{a}.W_g.0.bias']))
        x1 = bn(f'{a}.W_x.1', F.conv2d(skip, sd[f'{a}.W_x.0.weight'], sd[f'{a}.W_x.0.bias']))
        psi = torch.sigmoid(bn(f'{a}.psi.1', F.conv2d(torch.relu(g1 + x1), sd[f'{a}.psi.0.weight'], sd[f'{a}.psi.0.bias'])))
        src = torch.cat((skip * psi, d), dim=1)
    return torch.tanh(F.conv_transpose2d(src, sd['Up1.up.0.weight'], sd['Up1.up.0.bias'], stride=2, padding=1))


def nlayer_discriminator(sd: Dict[str, torch.Tensor], x: torch.Tensor, norm: str = 'batch', n_layers: int = 4,
                         update_running: bool = False) -> torch.Tensor:
    """NLayerDiscriminator.forward, networks.py:618-664: conv(k4,s2,p1,bias)+lrelu; (n_layers-1) x [conv k4 s2 + norm
    + lrelu]; conv k4 s1 + norm + lrelu; conv k4 s1 -> 1 channel (bias)."""
    h = F.leaky_relu(F.conv2d(x, sd['model.0.weight'], sd['model.0.bias'], stride=2, padding=1), 0.2)
    idx = 2
    for n in range(1, n_layers + 1):
        stride = 2 if n < n_layers else 1
        h = F.conv2d(h, sd[f'model.{idx}.weight'], sd.get(f'model.{idx}.bias'), stride=stride, padding=1)
        h = F.leaky_relu(_norm_from_sd(sd, f'model.{idx + 1}', h, norm, update_running), 0.2)
        idx += 3
    return F.conv2d(h, sd[f'model.{idx}.weight'], sd[f'model.{idx}.bias'], stride=1, padding=1)


def pixel_discriminator(sd: Dict[str, torch.Tensor], x: torch.Tensor, norm: str = 'batch', update_running: bool = False) -> torch.Tensor:
    """PixelDiscriminator.forward, networks.py:667-696: conv1x1(bias) + lrelu, conv1x1 + norm + lrelu, conv1x1 -> 1 channel."""
    h = F.leaky_relu(F.conv2d(x, sd['net.0.weight'], sd['net.0.bias']), 0.2)
    h = F.conv2d(h, sd['net.2.weight'], sd.get('net.2.bias'))
    h = F.leaky_relu(_norm_from_sd(sd, 'net.3', h, norm, update_running), 0.2)
    return F.conv2d(h, sd['net.5.weight'], sd.get('net.5.bias'))


def run_discriminator(arch: str, sd, x, norm='batch', n_layers=4, update_running=False):
    """define_D dispatch, networks.py:222-238."""
    if arch == 'pixel':
        return pixel_discriminator(sd, x, norm, update_running)
    return nlayer_discriminator(sd, x, norm, 3 if arch == 'basic' else n_layers, update_running)


def run_generator(arch: str, sd, x, norm='batch', padding_type='zero', update_running=False):
    """define_G dispatch, networks.py:175-188.  'resnet_9blocks:resize_conv' = the same generator built with upsample='resize_conv'."""
    if arch.startswith('resnet_'):
        arch, _, ups = arch.partition(':')
        n_blocks = int(arch.split('_')[1].replace('blocks', ''))
        return resnet_generator(sd, x, norm, padding_type, n_blocks, update_running, ups or 'convtranspose')
    table = {'unet_32': 5, 'unet_64': 6, 'unet_128': 7, 'unet_256': 8, 'unet_512': 9}
    if arch in table:
        return unet_generator(sd, x, norm, table[arch], update_running)
    if arch == 'unet_512_attention':
        return att_unet_generator(sd, x, update_running)
    raise NotImplementedError(arch)


# ----------------------------------------------------------------------------------------------
# losses  (networks.py:244-317, DeepLIIF_model.py:121-123)
# ----------------------------------------------------------------------------------------------
def gan_loss(pred: torch.Tensor, target_is_real: bool, mode: str) -> torch.Tensor:
    """GANLoss.__call__ with label_smoothing=0: 'vanilla' = BCEWithLogitsLoss(mean) vs constant 1/0 target,
    'lsgan' = MSELoss(mean)."""
    t = 1.0 if target_is_real else 0.0
    if mode == 'vanilla':
        # max(x,0) - x*t + log(1 + exp(-|x|))
        return (pred.clamp(min=0) - pred * t + torch.log1p(torch.exp(-pred.abs()))).mean()
    if mode == 'lsgan':
        return ((pred - t) ** 2).mean()
    if mode == 'wgangp':                       # networks.py:307-311 (no gradient penalty is ever added by the model classes)
        return -pred.mean() if target_is_real else pred.mean()
    raise NotImplementedError(mode)


def smooth_l1(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """torch.nn.SmoothL1Loss(beta=1, reduction='mean') (DeepLIIF_model.py:123)."""
    d = (a - b).abs()
    return torch.where(d < 1.0, 0.5 * d * d, d - 0.5).mean()


# ----------------------------------------------------------------------------------------------
# seeded weights in the reference's RNG-consumption order (networks.py:84-139)
# ----------------------------------------------------------------------------------------------
# ----------------------------------------------------------------------------------------------
# VGG19 perceptual loss  (networks.py:698-743; torchvision.models.vgg19().features[0:30])
# ----------------------------------------------------------------------------------------------
VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
VGG_SLICE_ENDS = (2, 7, 12, 21, 30)                 # Vgg19.slice1..5 (networks.py:707-716)
VGG_WEIGHTS = (1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0)       # VGGLoss.weights (networks.py:736)


def vgg19_features(sd: Dict[str, torch.Tensor], x: torch.Tensor):
    """[h_relu1 .. h_relu5] of Vgg19.forward (networks.py:722-731) from a torchvision-keyed state_dict ('features.N.weight/bias')."""
    outs, h, idx = [], x, 0
    for v in VGG19_CFG:
        if idx >= VGG_SLICE_ENDS[-1]:
            break
        if v == 'M':
            h = F.max_pool2d(h, kernel_size=2, stride=2)
            idx += 1
        else:
            h = F.relu(F.conv2d(h, sd[f'features.{idx}.weight'], sd[f'features.{idx}.bias'], padding=1))
            idx += 2
        if idx in VGG_SLICE_ENDS:
            outs.append(h)
    return outs


def vgg_loss(sd: Dict[str, torch.Tensor], x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """VGGLoss.forward (networks.py:738-743): sum_i w_i * mean|vgg(x)_i - vgg(y)_i.detach()|."""
    fx, fy = vgg19_features(sd, x), vgg19_features(sd, y)
    return sum(w * (a - b.detach()).abs().mean() for w, a, b in zip(VGG_WEIGHTS, fx, fy))


def random_vgg19_state_dict(generator: Optional[torch.Generator] = None) -> 'OrderedDict[str, torch.Tensor]':
    """Stand-in for the pretrained torchvision weights (a download): torchvision's own conv init, kaiming_normal_(fan_out, relu) with zero
    biases, drawn from `generator` in module order -- keeps the 13-layer stack's activations O(1).  Keys = torchvision's."""
    sd, cin, idx = OrderedDict(), 3, 0
    for v in VGG19_CFG:
        if idx >= VGG_SLICE_ENDS[-1]:
            break
        if v == 'M':
            idx += 1
            continue
        std = math.sqrt(2.0 / (v * 9))
        sd[f'features.{idx}.weight'] = torch.randn(v, cin, 3, 3, generator=generator) * std
        sd[f'features.{idx}.bias'] = torch.zeros(v)
        cin, idx = v, idx + 2
    return sd


def layer_table(arch: str, input_nc: int, output_nc: int = 3, nf: int = 64, norm: str = 'batch',
                padding_type: str = 'zero', n_layers: int = 4):
    """Ordered (kind, key, ...) list of the parameterised layers of one network in nn.Module.apply() visiting order
    (= construction order), which is the order init_weights (networks.py:95-115) redraws them in."""
    use_bias = norm == 'instance'
    out = []
    if arch.startswith('resnet_'):
        arch, _, ups = arch.partition(':')
        n_blocks = int(arch.split('_')[1].replace('blocks', ''))
        out.append(('conv', 'model.1', nf, input_nc, 7, use_bias, False)); out.append(('norm', 'model.2', nf))
        idx = 4
        for i in range(2):
            m = 2 ** i
            out.append(('conv', f'model.{idx}', nf * m * 2, nf * m, 3, use_bias, False)); out.append(('norm', f'model.{idx + 1}', nf * m * 2))
            idx += 3
        c1, n1, c2, n2 = (0, 1, 3, 4) if padding_type == 'zero' else (1, 2, 5, 6)
        for b in range(n_blocks):
            pre = f'model.{idx}.conv_block'
            out.append(('conv', f'{pre}.{c1}', nf * 4, nf * 4, 3, use_bias, False)); out.append(('norm', f'{pre}.{n1}', nf * 4))
            out.append(('conv', f'{pre}.{c2}', nf * 4, nf * 4, 3, use_bias, False)); out.append(('norm', f'{pre}.{n2}', nf * 4))
            idx += 1
        for i in range(2):
            m = 2 ** (2 - i)
            if ups == 'resize_conv':           # [Upsample, ReflectionPad2d(1), Conv2d(k3, bias=True), norm, ReLU] (networks.py:409-415, 434-436)
                out.append(('conv', f'model.{idx + 2}', nf * m // 2, nf * m, 3, True, False)); out.append(('norm', f'model.{idx + 3}', nf * m // 2))
                idx += 5
                continue
            out.append(('conv', f'model.{idx}', nf * m // 2, nf * m, 3, use_bias, True)); out.append(('norm', f'model.{idx + 1}', nf * m // 2))
            idx += 3
        idx += 1
        out.append(('conv', f'model.{idx}', output_nc, nf, 7, True, False))
        return out
    if arch == 'pixel':                        # PixelDiscriminator (networks.py:684-690)
        out.append(('conv', 'net.0', nf, input_nc, 1, True, False))
        out.append(('conv', 'net.2', nf * 2, nf, 1, use_bias, False)); out.append(('norm', 'net.3', nf * 2))
        out.append(('conv', 'net.5', 1, nf * 2, 1, use_bias, False))
        return out
    if arch == 'unet_512_attention':
        # att_unet.py:120-151: registration order Conv1..Conv8, Up8, Att8, Up7, Att7, ..., Up2, Att2, Up1; widths fixed; BatchNorm2d always
        for k, (cin, cout) in enumerate(ATT_DOWN, start=1):
            out.append(('conv', f'Conv{k}.conv.0', cout, input_nc if cin is None else cin, 4, True, False))
            if 1 < k < 8:
                out.append(('norm', f'Conv{k}.conv.1', cout))
        for k in range(8, 1, -1):
            cin, cout = ATT_UP[k]
            out.append(('conv', f'Up{k}.up.0', cout, cin if k == 8 else 2 * cin, 4, False, True)); out.append(('norm', f'Up{k}.up.1', cout))
            f, fi = ATT_GATE[k]
            out.append(('conv', f'Att{k}.W_g.0', fi, f, 1, True, False)); out.append(('norm', f'Att{k}.W_g.1', fi))
            out.append(('conv', f'Att{k}.W_x.0', fi, f, 1, True, False)); out.append(('norm', f'Att{k}.W_x.1', fi))
            out.append(('conv', f'Att{k}.psi.0', 1, fi, 1, True, False)); out.append(('norm', f'Att{k}.psi.1', 1))
        out.append(('conv', 'Up1.up.0', output_nc, 128, 4, True, True))
        return out
    if arch.startswith('unet_'):
        num_downs = {'unet_32': 5, 'unet_64': 6, 'unet_128': 7, 'unet_256': 8, 'unet_512': 9}[arch]
        # channels of block d's inner / outer features
        inner = [nf, nf * 2, nf * 4, nf * 8] + [nf * 8] * (num_downs - 4)      # inner_nc of block d
        outer = [output_nc, nf, nf * 2, nf * 4] + [nf * 8] * (num_downs - 4)   # outer_nc of block d

        # Module.apply visits children before the module itself and children in registration order;
        # within a block the Sequential holds [.., downconv, (downnorm), submodule, .., upconv, (upnorm)].
        def visit(d):
            pre = ('model' if d == 0 else 'model.model.1' + '.model.3' * (d - 1)) + '.model'
            if d == 0:
                out.append(('conv', f'{pre}.0', inner[0], input_nc, 4, use_bias, False))
                visit(1)
                out.append(('conv', f'{pre}.3', outer[0], inner[0] * 2, 4, True, True))
            elif d == num_downs - 1:
                out.append(('conv', f'{pre}.1', inner[d], outer[d], 4, use_bias, False))
                out.append(('conv', f'{pre}.3', outer[d], inner[d], 4, use_bias, True)); out.append(('norm', f'{pre}.4', outer[d]))
            else:
                out.append(('conv', f'{pre}.1', inner[d], outer[d], 4, use_bias, False)); out.append(('norm', f'{pre}.2', inner[d]))
                visit(d + 1)
                out.append(('conv', f'{pre}.5', outer[d], inner[d] * 2, 4, use_bias, True)); out.append(('norm', f'{pre}.6', outer[d]))
        visit(0)
        return out
    if arch in ('n_layers', 'basic'):
        if arch == 'basic':
            n_layers = 3
        out.append(('conv', 'model.0', nf, input_nc, 4, True, False))
        idx, prev = 2, 1
        for n in range(1, n_layers + 1):
            mult = min(2 ** n, 8)
            out.append(('conv', f'model.{idx}', nf * mult, nf * prev, 4, use_bias, False)); out.append(('norm', f'model.{idx + 1}', nf * mult))
            prev = mult
            idx += 3
        out.append(('conv', f'model.{idx}', 1, nf * prev, 4, True, False))
        return out
    raise NotImplementedError(arch)


def random_state_dict(arch: str, input_nc: int, output_nc: int = 3, nf: int = 64, norm: str = 'batch',
                      padding_type: str = 'zero', n_layers: int = 4, gain: float = 0.02,
                      generator: Optional[torch.Generator] = None) -> 'OrderedDict[str, torch.Tensor]':
    """A reference-keyed state_dict with init_weights('normal', gain) *distributions* (networks.py:98-112):
    conv weights ~ N(0, gain), BN gamma ~ N(1, gain), biases 0.  Values are drawn from `generator` (or the global
    RNG) in layer_table order; they are NOT bit-identical to define_G under the same seed (the reference also burns
    RNG in each layer's default reset_parameters) -- tests that need identical weights pass one state_dict to both."""
    sd: 'OrderedDict[str, torch.Tensor]' = OrderedDict()
    for ent in layer_table(arch, input_nc, output_nc, nf, norm, padding_type, n_layers):
        if ent[0] == 'conv':
            _, key, cout, cin, k, with_bias, transposed = ent
            shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
            sd[key + '.weight'] = torch.empty(shape).normal_(0.0, gain, generator=generator)
            if with_bias:
                sd[key + '.bias'] = torch.zeros(cout)
        elif norm == 'batch':
            _, key, c = ent
            sd[key + '.weight'] = torch.empty(c).normal_(1.0, gain, generator=generator)
            sd[key + '.bias'] = torch.zeros(c)
            sd[key + '.running_mean'] = torch.zeros(c)
            sd[key + '.running_var'] = torch.ones(c)
            sd[key + '.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)
    return sd


# ----------------------------------------------------------------------------------------------
# the DeepLIIF model step  (DeepLIIF_model.py:175-467)
# ----------------------------------------------------------------------------------------------
class OracleConfig:
    """The opt fields the step reads (SURVEY 5 'Config / flags')."""

    def __init__(self, modalities_no=4, seg_gen=True, net_g='resnet_9blocks', net_gs='unet_512', norm='batch',
                 padding='zero', ngf=64, ndf=64, n_layers_D=4, gan_mode='vanilla', gan_mode_s='lsgan',
                 lambda_L1=100.0, lr_g=2e-4, lr_d=2e-4, beta1=0.5, seg_weights=None, loss_G_weights=None,
                 loss_D_weights=None, input_nc=3, output_nc=3, lambda_feat=0.0, net_d='n_layers', upsample='convtranspose'):
        self.lambda_feat = lambda_feat
        self.net_d, self.upsample = net_d, upsample        # cli.py:103 --upsample (translation generators only, DeepLIIF_model.py:92-99), --net-d
        self.modalities_no = modalities_no
        self.seg_gen = seg_gen
        self.net_g = net_g
        self.net_gs = net_gs
        self.norm = norm
        self.padding = padding
        self.ngf, self.ndf, self.n_layers_D = ngf, ndf, n_layers_D
        self.gan_mode, self.gan_mode_s = gan_mode, gan_mode_s
        self.lambda_L1 = lambda_L1
        self.lr_g, self.lr_d, self.beta1 = lr_g, lr_d, beta1
        n = modalities_no + 1
        # cli.py:349-371 defaults
        if seg_weights is None:
            seg_weights = [0.25, 0.15, 0.25, 0.1, 0.25] if modalities_no == 4 else [1.0 / n] * n
        if loss_G_weights is None:
            loss_G_weights = [0.2] * 5 if modalities_no == 4 else [1.0 / n] * n
        if loss_D_weights is None:
            loss_D_weights = [0.2] * 5 if modalities_no == 4 else [1.0 / n] * n
        self.seg_weights, self.loss_G_weights, self.loss_D_weights = seg_weights, loss_G_weights, loss_D_weights
        self.input_nc, self.output_nc = input_nc, output_nc

    # network names exactly as DeepLIIF_model.py:49-115 builds them for a fresh training run
    # (mod_id_seg = 'S' when modalities_names is given, input_id = '0'; util/util.py:242-262)
    def names(self, mod_id_seg='S'):
        g = [f'G{i}' for i in range(1, self.modalities_no + 1)]
        d = [f'D{i}' for i in range(1, self.modalities_no + 1)]
        gs = [f'G{mod_id_seg}{i}' for i in range(self.modalities_no + 1)] if self.seg_gen else []
        ds = [f'D{mod_id_seg}{i}' for i in range(self.modalities_no + 1)] if self.seg_gen else []
        return g, gs, d, ds


class AdamState:
    """torch.optim.Adam(lr, betas=(beta1, 0.999), eps=1e-8, weight_decay=0) restated (networks.py:46-53,
    DeepLIIF_model.py:128-147)."""

    def __init__(self, params: Sequence[torch.Tensor], lr: float, beta1: float, beta2: float = 0.999, eps: float = 1e-8):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    def step(self, grads: Sequence[torch.Tensor]):
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        with torch.no_grad():
            for p, g, m, v in zip(self.params, grads, self.m, self.v):
                m.mul_(self.b1).add_(g, alpha=1 - self.b1)
                v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
                p.addcdiv_(m, denom, value=-self.lr / bc1)


class OracleDeepLIIF:
    """Functional DeepLIIFModel: set_input / forward / backward_D / backward_G / optimize_parameters over
    reference-keyed state_dicts (one per network name).  torch autograd supplies the gradients."""

    def __init__(self, cfg: OracleConfig, nets: Dict[str, Dict[str, torch.Tensor]], train_bn_running: bool = True, vgg_sd=None):
        self.cfg = cfg
        self.nets = nets
        self.vgg_sd = vgg_sd                    # torchvision-keyed VGG19 weights; required when cfg.lambda_feat > 0
        self.g_names, self.gs_names, self.d_names, self.ds_names = cfg.names()
        self.train_bn_running = train_bn_running
        self.losses: Dict[str, torch.Tensor] = {}

        def trainable(names):
            ps = []
            for n in names:
                for k, v in nets[n].items():
                    if v.is_floating_point() and not k.endswith(('running_mean', 'running_var')):
                        v.requires_grad_(True)
                        ps.append(v)
            return ps
        self.params_g = trainable(self.g_names + self.gs_names)
        self.params_d = trainable(self.d_names + self.ds_names)
        self.adam_g = AdamState(self.params_g, cfg.lr_g, cfg.beta1)
        self.adam_d = AdamState(self.params_d, cfg.lr_d, cfg.beta1)

    # -- helpers
    def _G(self, name, x, arch, padding):
        if name in self.g_names and self.cfg.upsample != 'convtranspose' and arch.startswith('resnet_'):
            arch = f'{arch}:{self.cfg.upsample}'
        return run_generator(arch, self.nets[name], x, self.cfg.norm, padding, self.train_bn_running)

    def _D(self, name, x):
        return run_discriminator(self.cfg.net_d, self.nets[name], x, self.cfg.norm, self.cfg.n_layers_D, self.train_bn_running)

    def set_input(self, batch):
        """DeepLIIF_model.py:153-173."""
        self.real_A = batch['A']
        self.real_B = list(batch['B'])

    def forward(self):
        """DeepLIIF_model.py:175-203.  Translation generators use opt.padding; seg generators are built by define_G's
        default padding_type='reflect' (:98-99, networks.py:143) which only matters for resnet seg generators."""
        c = self.cfg
        self.fake_B = [self._G(n, self.real_A, c.net_g, c.padding) for n in self.g_names]
        if c.seg_gen:
            self.fake_seg_parts = []
            for i, n in enumerate(self.gs_names):
                src = self.real_A if i == 0 else self.fake_B[i - 1]
                self.fake_seg_parts.append(self._G(n, src, c.net_gs, 'reflect'))
            self.fake_seg = sum(p * w for p, w in zip(self.fake_seg_parts, c.seg_weights))

    def _seg_pred(self, img, detach):
        c = self.cfg
        preds = []
        for i, n in enumerate(self.ds_names):
            cond = self.real_A if i == 0 else self.real_B[i - 1]      # real modalities condition the seg D's (:253-255)
            pair = torch.cat((cond, img), 1)
            preds.append(self._D(n, pair.detach() if detach else pair) * c.seg_weights[i])
        return sum(preds)                                              # weighted sum BEFORE the lsgan loss (:258-262)

    def loss_D(self):
        """DeepLIIF_model.py:205-330 (without the .backward())."""
        c, L = self.cfg, self.losses
        for i, n in enumerate(self.d_names):
            L[f'D_fake_{i + 1}'] = gan_loss(self._D(n, torch.cat((self.real_A, self.fake_B[i]), 1).detach()), False, c.gan_mode)
        if c.seg_gen:
            L['D_fake_S'] = gan_loss(self._seg_pred(self.fake_seg, True), False, c.gan_mode_s)
        for i, n in enumerate(self.d_names):
            L[f'D_real_{i + 1}'] = gan_loss(self._D(n, torch.cat((self.real_A, self.real_B[i]), 1)), True, c.gan_mode)
        if c.seg_gen:
            L['D_real_S'] = gan_loss(self._seg_pred(self.real_B[c.modalities_no], False), True, c.gan_mode_s)
        total = 0.0
        for i in range(c.modalities_no):
            total = total + (L[f'D_fake_{i + 1}'] + L[f'D_real_{i + 1}']) * 0.5 * c.loss_D_weights[i]
        if c.seg_gen:
            total = total + (L['D_fake_S'] + L['D_real_S']) * 0.5 * c.loss_D_weights[c.modalities_no]
        return total

    def loss_G(self):
        """DeepLIIF_model.py:334-421 (VGG term only when cfg.lambda_feat > 0).  The seg term is weighted by
        loss_G_weights[modalities_no - 1]: the reference reuses the stale loop index `i` (:418-421)."""
        c, L = self.cfg, self.losses
        for i, n in enumerate(self.d_names):
            L[f'G_GAN_{i + 1}'] = gan_loss(self._D(n, torch.cat((self.real_A, self.fake_B[i]), 1)), True, c.gan_mode)
        if c.seg_gen:
            L['G_GAN_S'] = gan_loss(self._seg_pred(self.fake_seg, False), True, c.gan_mode_s)
        for i in range(c.modalities_no):
            L[f'G_L1_{i + 1}'] = smooth_l1(self.fake_B[i], self.real_B[i]) * c.lambda_L1
        if c.seg_gen:
            L['G_L1_S'] = smooth_l1(self.fake_seg, self.real_B[c.modalities_no]) * c.lambda_L1
        vgg = [0.0] * c.modalities_no
        if c.lambda_feat > 0:                   # :406-409; the seg image's VGG value (:408-409) never enters loss_G (:418-421)
            vgg = [vgg_loss(self.vgg_sd, self.fake_B[i], self.real_B[i]) * c.lambda_feat for i in range(c.modalities_no)]
            self.vgg_losses = [float(v.detach()) for v in vgg]
        total = 0.0
        for i in range(c.modalities_no):
            total = total + (L[f'G_GAN_{i + 1}'] + L[f'G_L1_{i + 1}'] + vgg[i]) * c.loss_G_weights[i]
        if c.seg_gen:
            total = total + (L['G_GAN_S'] + L['G_L1_S']) * c.loss_G_weights[c.modalities_no - 1]
        return total

    def optimize_parameters(self):
        """DeepLIIF_model.py:431-467."""
        self.forward()
        # D update: grads wrt D params only (fakes are detached)
        gd = torch.autograd.grad(self.loss_D(), self.params_d)
        self.adam_d.step(gd)
        # G update: D params frozen (set_requires_grad False) but D is differentiated through
        gg = torch.autograd.grad(self.loss_G(), self.params_g)
        self.adam_g.step(gg)
        self.last_grads_d, self.last_grads_g = gd, gg

    def current_losses(self):
        return OrderedDict((k, float(v.detach())) for k, v in self.losses.items())


class OracleDeepLIIFExt:
    """Functional DeepLIIFExtModel (deepliif/models/DeepLIIFExt_model.py:160-269): M translation generators on real_A, M seg
    generators on cat(real_A, fake_B[0], fake_B[i]) (9 channels, :173), seg discriminators on
    cat(real_A, real_B[0], real_B[i], seg) (12 channels, :97,186).  Quirks restated: the generator-side seg GAN loss uses
    criterionGAN_mod (:236) while the discriminator side uses criterionGAN_seg (:195,213); seg loss weights are 1/M (:27,30);
    no VGG term (:257-265 commented out)."""

    def __init__(self, cfg: OracleConfig, nets: Dict[str, Dict[str, torch.Tensor]], train_bn_running: bool = True):
        self.cfg, self.nets, self.train_bn_running = cfg, nets, train_bn_running
        M = cfg.modalities_no
        self.g = [f'G_{i + 1}' for i in range(M)]
        self.gs = [f'GS_{i + 1}' for i in range(M)] if cfg.seg_gen else []
        self.d = [f'D_{i + 1}' for i in range(M)]
        self.ds = [f'DS_{i + 1}' for i in range(M)] if cfg.seg_gen else []
        self.losses: Dict[str, torch.Tensor] = {}

        def trainable(names):
            ps = []
            for n in names:
                for k, v in nets[n].items():
                    if v.is_floating_point() and not k.endswith(('running_mean', 'running_var')):
                        v.requires_grad_(True)
                        ps.append(v)
            return ps
        self.params_g = trainable(self.g + self.gs)
        self.params_d = trainable(self.d + self.ds)
        self.adam_g = AdamState(self.params_g, cfg.lr_g, cfg.beta1)
        self.adam_d = AdamState(self.params_d, cfg.lr_d, cfg.beta1)

    def _G(self, name, x, arch, padding):
        return run_generator(arch, self.nets[name], x, self.cfg.norm, padding, self.train_bn_running)

    def _D(self, name, x):
        return nlayer_discriminator(self.nets[name], x, self.cfg.norm, self.cfg.n_layers_D, self.train_bn_running)

    def set_input(self, batch):
        self.real_A, self.real_B, self.real_BS = batch['A'], list(batch['B']), list(batch.get('BS', []))
        self.real_cat = [torch.cat([self.real_A, self.real_B[0], self.real_B[i]], 1) for i in range(self.cfg.modalities_no)]

    def forward(self):
        c = self.cfg
        self.fake_B = [self._G(n, self.real_A, c.net_g, c.padding) for n in self.g]
        self.fake_BS = [self._G(n, torch.cat([self.real_A, self.fake_B[0], self.fake_B[i]], 1), c.net_gs, 'reflect')
                        for i, n in enumerate(self.gs)]

    def loss_D(self):
        c, L, M = self.cfg, self.losses, self.cfg.modalities_no
        for i in range(M):
            L[f'D_fake_{i + 1}'] = gan_loss(self._D(self.d[i], torch.cat((self.real_A, self.fake_B[i]), 1).detach()), False, c.gan_mode)
        for i in range(len(self.ds)):
            L[f'DS_fake_{i + 1}'] = gan_loss(self._D(self.ds[i], torch.cat((self.real_cat[i], self.fake_BS[i]), 1).detach()), False, c.gan_mode_s)
        for i in range(M):
            L[f'D_real_{i + 1}'] = gan_loss(self._D(self.d[i], torch.cat((self.real_A, self.real_B[i]), 1)), True, c.gan_mode)
        for i in range(len(self.ds)):
            L[f'DS_real_{i + 1}'] = gan_loss(self._D(self.ds[i], torch.cat((self.real_cat[i], self.real_BS[i]), 1)), True, c.gan_mode_s)
        total = 0.0
        for i in range(M):
            total = total + (L[f'D_fake_{i + 1}'] + L[f'D_real_{i + 1}']) * 0.5 * c.loss_D_weights[i]
        for i in range(len(self.ds)):
            total = total + (L[f'DS_fake_{i + 1}'] + L[f'DS_real_{i + 1}']) * 0.5 * (1.0 / M)
        return total

    def loss_G(self):
        c, L, M = self.cfg, self.losses, self.cfg.modalities_no
        for i in range(M):
            L[f'G_GAN_{i + 1}'] = gan_loss(self._D(self.d[i], torch.cat((self.real_A, self.fake_B[i]), 1)), True, c.gan_mode)
        for i in range(len(self.ds)):
            L[f'GS_GAN_{i + 1}'] = gan_loss(self._D(self.ds[i], torch.cat((self.real_cat[i], self.fake_BS[i]), 1)), True, c.gan_mode)
        for i in range(M):
            L[f'G_L1_{i + 1}'] = smooth_l1(self.fake_B[i], self.real_B[i]) * c.lambda_L1
        for i in range(len(self.gs)):
            L[f'GS_L1_{i + 1}'] = smooth_l1(self.fake_BS[i], self.real_BS[i]) * c.lambda_L1
        total = 0.0
        for i in range(M):
            total = total + (L[f'G_GAN_{i + 1}'] + L[f'G_L1_{i + 1}']) * c.loss_G_weights[i]
        for i in range(len(self.gs)):
            total = total + (L[f'GS_GAN_{i + 1}'] + L[f'GS_L1_{i + 1}']) * (1.0 / M)
        return total

    def optimize_parameters(self):
        self.forward()
        self.adam_d.step(torch.autograd.grad(self.loss_D(), self.params_d))
        self.adam_g.step(torch.autograd.grad(self.loss_G(), self.params_g))

    def current_losses(self):
        return OrderedDict((k, float(v.detach())) for k, v in self.losses.items())


class OracleSDG(OracleDeepLIIFExt):
    """Functional SDGModel (deepliif/models/SDG_model.py): DeepLIIFExt's translation branch only.  `input['A']` is a LIST of
    input_no modalities concatenated on the channel axis (set_input, SDG_model.py:108-112), so generators take
    input_nc*input_no channels (:59-60) and discriminators input_nc*input_no + output_nc (:66-68); forward / backward_D /
    backward_G (:126-183) are the Ext ones without the seg branch.  The reference also lists a VGG19 term per modality
    (loss_names 'G_VGG_i', :33, :173-175): it is outside this path (zeroed when the fixtures are generated) and is
    reported as 0 so that the loss_names of the seam are complete."""

    def __init__(self, cfg: OracleConfig, nets: Dict[str, Dict[str, torch.Tensor]], train_bn_running: bool = True):
        assert not cfg.seg_gen, 'SDG has no segmentation branch'
        super().__init__(cfg, nets, train_bn_running)

    def set_input(self, batch):
        A = batch['A']
        super().set_input({'A': torch.cat(list(A), 1) if isinstance(A, (list, tuple)) else A, 'B': batch['B']})

    def current_losses(self):
        out = super().current_losses()
        for i in range(self.cfg.modalities_no):
            out[f'G_VGG_{i + 1}'] = 0.0
        return out


# ----------------------------------------------------------------------------------------------
# DeepLIIFKD: DeepLIIF distilled from a frozen teacher  (deepliif/models/DeepLIIFKD_model.py)
# ----------------------------------------------------------------------------------------------
def kldiv_whole_tensor(x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """KLDivLoss(reduction='batchmean')(LogSoftmax(dim=-1)(x.view(1, 1, -1)), Softmax(dim=-1)(t.view(1, 1, -1)))  (DeepLIIFKD_model.py:148-151,
    313-336): ONE softmax over all elements of each tensor; 'batchmean' divides by the size of the view's first dimension, which is 1.
    KL(p || q) = sum_j p_j (log p_j - log q_j) with p = softmax(t), q = softmax(x)."""
    lq = torch.log_softmax(x.reshape(-1), 0)
    lp = torch.log_softmax(t.reshape(-1), 0)
    return (lp.exp() * (lp - lq)).sum()


class OracleDeepLIIFKD(OracleDeepLIIF):
    """Functional DeepLIIFKDModel: OracleDeepLIIF with the seg branch, vanilla / lsgan GAN losses (:130-131), plus the distillation terms
    (:313-349): 10 * KL for every modality image, every seg generator's image and the aggregated seg image, against a frozen teacher that is run
    like run_dask(img=real_A, nets=teacher, use_dask=False, output_tensor=True) (:203): G_i(A), GS_0(A), GS_i(G_i(A)), seg = mean of the seg
    images (run_dask's default weights 1/(M+1), models/__init__.py:310), BatchNorm on the statistics of the batch, nothing updated."""

    def __init__(self, cfg: OracleConfig, nets, teacher_cfg: OracleConfig, teacher_nets, train_bn_running: bool = True, vgg_sd=None):
        assert cfg.seg_gen and teacher_cfg.seg_gen
        cfg.gan_mode, cfg.gan_mode_s = 'vanilla', 'lsgan'
        super().__init__(cfg, nets, train_bn_running, vgg_sd)
        self.tcfg, self.tnets = teacher_cfg, teacher_nets

    def forward(self):
        super().forward()
        tc = self.tcfg
        tg, tgs, _, _ = tc.names()
        with torch.no_grad():
            self.teacher_B = [run_generator(tc.net_g, self.tnets[n], self.real_A, tc.norm, tc.padding, False) for n in tg]
            parts = [run_generator(tc.net_gs, self.tnets[tgs[0]], self.real_A, tc.norm, 'reflect', False)]
            parts += [run_generator(tc.net_gs, self.tnets[n], self.teacher_B[i], tc.norm, 'reflect', False) for i, n in enumerate(tgs[1:])]
            self.teacher_seg_parts = parts
            self.teacher_seg = sum(p * (1.0 / (tc.modalities_no + 1)) for p in parts)

    def loss_G(self):
        total = super().loss_G()
        c, L = self.cfg, self.losses
        for i in range(c.modalities_no):
            L[f'G_KLDiv_{i + 1}'] = kldiv_whole_tensor(self.fake_B[i], self.teacher_B[i])
        L['G_KLDiv_S'] = kldiv_whole_tensor(self.fake_seg, self.teacher_seg)
        for i in range(c.modalities_no + 1):
            L[f'G_KLDiv_S{i}'] = kldiv_whole_tensor(self.fake_seg_parts[i], self.teacher_seg_parts[i])
        for i in range(c.modalities_no):
            total = total + (L[f'G_KLDiv_{i + 1}'] + L[f'G_KLDiv_S{i + 1}']) * 10
        return total + L['G_KLDiv_S'] * 10 + L['G_KLDiv_S0'] * 10            # input_id '0' (:342-343)


# ----------------------------------------------------------------------------------------------
# CycleGAN  (deepliif/models/CycleGAN_model.py, deepliif/util/image_pool.py)
# ----------------------------------------------------------------------------------------------
class OracleImagePool:
    """util/image_pool.py: per image -- fill the pool first; afterwards one random.uniform decides between returning the new image and
    swapping it against a stored one chosen by random.randint (inclusive)."""

    def __init__(self, pool_size):
        self.pool_size, self.images = pool_size, []

    def query(self, images: torch.Tensor) -> torch.Tensor:
        import random
        if self.pool_size == 0:
            return images
        out = []
        for img in images:
            img = img.detach().unsqueeze(0)
            if len(self.images) < self.pool_size:
                self.images.append(img)
                out.append(img)
            elif random.uniform(0, 1) > 0.5:
                k = random.randint(0, self.pool_size - 1)
                out.append(self.images[k].clone())
                self.images[k] = img
            else:
                out.append(img)
        return torch.cat(out, 0)


class OracleCycleGAN:
    """Functional CycleGANModel.  nets: {'GA_i', 'GB_i', 'DA_i', 'DB_i'} reference-keyed state_dicts.  One step (:266-282): forward; generator
    update on  sum_i w_i [lsgan(DA_i(fake_B_i), 1) + lsgan(DB_i(fake_A_i), 1)] + 10/M sum_i [L1(rec_A_i, A) + L1(rec_B_i, B_i)]  (identity terms
    off :213; the VGG terms :232,238 only with vgg_sd); then per discriminator (lsgan(D(real), 1) + lsgan(D(pool(fake)), 0)) * 0.5 * w_i."""

    def __init__(self, cfg: OracleConfig, nets, pool_size=50, gan_mode='lsgan', train_bn_running: bool = True, vgg_sd=None):
        self.cfg, self.nets, self.gan_mode, self.vgg_sd = cfg, nets, gan_mode, vgg_sd
        M = cfg.modalities_no
        self.ga = [f'GA_{i + 1}' for i in range(M)]
        self.gb = [f'GB_{i + 1}' for i in range(M)]
        self.da = [f'DA_{i + 1}' for i in range(M)]
        self.db = [f'DB_{i + 1}' for i in range(M)]
        self.train_bn_running = train_bn_running
        self.losses: Dict[str, torch.Tensor] = {}
        self.pools_A = [OracleImagePool(pool_size) for _ in range(M)]
        self.pools_B = [OracleImagePool(pool_size) for _ in range(M)]

        def trainable(names):
            ps = []
            for n in names:
                for k, v in nets[n].items():
                    if v.is_floating_point() and not k.endswith(('running_mean', 'running_var')):
                        v.requires_grad_(True)
                        ps.append(v)
            return ps
        self.params_g = trainable(self.ga + self.gb)           # optimizer_G: all GA then all GB (:124-128)
        self.params_d = trainable(self.da + self.db)
        self.adam_g = AdamState(self.params_g, cfg.lr_g, cfg.beta1)
        self.adam_d = AdamState(self.params_d, cfg.lr_d, cfg.beta1)

    def _G(self, name, x):
        return run_generator(self.cfg.net_g, self.nets[name], x, self.cfg.norm, self.cfg.padding, self.train_bn_running)

    def _D(self, name, x):
        return nlayer_discriminator(self.nets[name], x, self.cfg.norm, self.cfg.n_layers_D, self.train_bn_running)

    def set_input(self, batch):
        self.real_A, self.real_Bs = batch['A'], list(batch['Bs'])

    def forward(self):
        self.fake_Bs = [self._G(n, self.real_A) for n in self.ga]
        self.rec_As = [self._G(n, f) for n, f in zip(self.gb, self.fake_Bs)]
        self.fake_As = [self._G(n, b) for n, b in zip(self.gb, self.real_Bs)]
        self.rec_Bs = [self._G(n, f) for n, f in zip(self.ga, self.fake_As)]

    def loss_G(self):
        c, L, M = self.cfg, self.losses, self.cfg.modalities_no
        w = c.loss_G_weights
        vgg = (lambda a, b: vgg_loss(self.vgg_sd, a, b)) if self.vgg_sd is not None else (lambda a, b: 0.0)
        L['G_A'] = sum((gan_loss(self._D(n, f), True, self.gan_mode) + vgg(f, b)) * w[i] for i, (n, f, b) in enumerate(zip(self.da, self.fake_Bs, self.real_Bs)))
        L['G_B'] = sum((gan_loss(self._D(n, f), True, self.gan_mode) + vgg(f, self.real_A)) * w[i] for i, (n, f) in enumerate(zip(self.db, self.fake_As)))
        L['cycle_A'] = sum((r - self.real_A).abs().mean() * 10 / M for r in self.rec_As)
        L['cycle_B'] = sum((r - b).abs().mean() * 10 / M for r, b in zip(self.rec_Bs, self.real_Bs))
        L['idt_A'] = L['idt_B'] = torch.zeros(())
        return L['G_A'] + L['G_B'] + L['cycle_A'] + L['cycle_B']

    def loss_D(self):
        c, L = self.cfg, self.losses
        fb = [p.query(f) for p, f in zip(self.pools_B, self.fake_Bs)]            # backward_D_A first (:193-199), then backward_D_B: pool order matters
        L['D_A'] = sum((gan_loss(self._D(n, r), True, self.gan_mode) + gan_loss(self._D(n, f.detach()), False, self.gan_mode)) * 0.5 * c.loss_D_weights[i]
                       for i, (n, r, f) in enumerate(zip(self.da, self.real_Bs, fb)))
        fa = [p.query(f) for p, f in zip(self.pools_A, self.fake_As)]
        L['D_B'] = sum((gan_loss(self._D(n, self.real_A), True, self.gan_mode) + gan_loss(self._D(n, f.detach()), False, self.gan_mode)) * 0.5 * c.loss_D_weights[i]
                       for i, (n, f) in enumerate(zip(self.db, fa)))
        return L['D_A'] + L['D_B']

    def optimize_parameters(self):
        self.forward()
        gg = torch.autograd.grad(self.loss_G(), self.params_g)
        self.adam_g.step(gg)
        gd = torch.autograd.grad(self.loss_D(), self.params_d)
        self.adam_d.step(gd)
        self.last_grads_d, self.last_grads_g = gd, gg

    def current_losses(self):
        order = ['D_A', 'G_A', 'cycle_A', 'idt_A', 'D_B', 'G_B', 'cycle_B', 'idt_B']
        return OrderedDict((k, float(self.losses[k].detach()) if torch.is_tensor(self.losses[k]) else float(self.losses[k])) for k in order)
