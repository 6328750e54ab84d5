"""Host side of the fp16 INFERENCE policy (engine.Precision 'fp16', ops.half_mode) on the CPU emulation: the policy reaches every engine op with float16
tensors, net(x) enters the half mode by itself, the result is nearer to the oracle than the bf16 policy's, and the guards hold (no tape, no engine
context outside the half mode, unknown formats).  The kernels of libdeepliif_hip_f16.so themselves are checked on the GPU (tests/test_gpu_fp16.py)."""
import pytest
import torch

import fake_backend
from deepliif_amd import engine as E
from deepliif_amd import networks as N
from deepliif_amd import ops
from golden_util import seeded_uniform
from oracle import deepliif_oracle as O


@pytest.fixture(autouse=True)
def _fake():
    fb = fake_backend.install()
    yield fb
    fake_backend.uninstall()


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


@pytest.mark.parametrize('arch,cin,norm,pad,shape', [('resnet_9blocks', 3, 'batch', 'zero', (1, 3, 32, 32)), ('resnet_9blocks', 3, 'instance', 'reflect', (1, 3, 40, 24)),
                                                     ('unet_64', 9, 'instance', 'zero', (1, 9, 64, 64))], ids=lambda v: str(v).replace(' ', ''))
def test_fp16_policy_through_net_call(arch, cin, norm, pad, shape, _fake):
    sd = O.random_state_dict(arch, cin, 3, 8, norm, pad, 4, generator=torch.Generator().manual_seed(5))
    net = N.define_G(cin, 3, 8, arch, norm, False, 'normal', 0.02, [], pad)
    net.load_state_dict(sd, strict=True)
    net.eval()
    x = seeded_uniform(shape, 6)
    with torch.no_grad():
        exp = O.run_generator(arch, {k: v.clone() for k, v in sd.items()}, x.clone(), norm, pad)
    seen = []
    orig = _fake.conv_forward

    def spy(packed, xt, out, *a, **k):
        seen.append((xt.dtype, out.dtype, ops.half_format()))
        return orig(packed, xt, out, *a, **k)
    _fake.conv_forward = spy
    err = {}
    for precname in ('fp16', 'bf16'):
        net.set_precision(precname)
        seen.clear()
        with torch.no_grad():
            err[precname] = rel(net(x), exp)
        want = torch.float16 if precname == 'fp16' else torch.bfloat16
        assert seen and all(s[0] == want and s[1] in (want, torch.float32) and s[2] == precname for s in seen), seen      # (fp32: the narrow head's raw row sums)
        assert ops.half_format() == 'bf16'                # the mode ends with the call
    assert err['fp16'] < 1e-2 and err['fp16'] < 0.5 * err['bf16'], err


def test_guards():
    fp16 = E.Precision.get('fp16')
    assert fp16.half == 'fp16' and fp16.is16 and E.Precision.get('half') == fp16
    assert E.Precision.get('bf16').half == 'bf16' and E.Precision.get('fp32').half == 'bf16' and not E.Precision.get('fp32').is16
    with pytest.raises(RuntimeError, match='half_mode'):
        E.Ctx(fp16, None, training=False)
    with ops.half_mode('fp16'):
        E.Ctx(fp16, None, training=False)
        with pytest.raises(ValueError, match='inference policy'):
            E.Ctx(fp16, E.Tape(), training=False)
        with pytest.raises(ValueError, match='inference policy'):
            E.Ctx(fp16, None, training=True)
        with pytest.raises(RuntimeError, match='half_mode'):
            E.Ctx(E.Precision.get('bf16'), None, training=False)
        with ops.half_mode('bf16'):
            assert ops.half_format() == 'bf16'
        assert ops.half_format() == 'fp16'
    with pytest.raises(ValueError):
        with ops.half_mode('fp8'):
            pass
    assert ops.half_format() == 'bf16'


def test_fp16_overflow_is_reported_not_returned():
    """a NaN in an fp16 result (a value that left half's range inside a net) raises at the next host-side wait instead of reaching the caller's images"""
    from deepliif_amd import inference as I
    I._FP16_NAN.clear()
    ok = E.Act(torch.zeros((1, 4, 4, 8), dtype=torch.float16), 3)
    I._fp16_note({'G1': ok})
    I.fp16_check()                                   # clean batch: nothing raised, flag consumed
    assert not I._FP16_NAN
    bad = E.Act(torch.zeros((1, 4, 4, 8), dtype=torch.float16), 3)
    bad.t[0, 1, 2, 0] = float('nan')
    I._fp16_note({'G1': ok, 'G2': bad})
    I._fp16_note({'G1': ok})                         # a later clean batch does not clear the flag
    with pytest.raises(FloatingPointError, match="precision='bf16'"):
        I.fp16_check()
    I.fp16_check()                                   # consumed
    I._fp16_note({'G1': E.Act(torch.full((1, 4, 4, 8), float('nan'), dtype=torch.bfloat16), 3)})      # other policies are not looked at
    assert not I._FP16_NAN


def test_region_inference_on_the_fp16_policy():
    """infer_region (crop + is_empty + generator DAG + stitch) with the nets on 'fp16': float16 tiles from the gather to the paste, the stitched 8-bit images
    within one level of the strict policy's"""
    import types
    import numpy as np
    from deepliif_amd import inference as I
    from golden_util import synth_image
    torch.manual_seed(0)
    opt = types.SimpleNamespace(model='DeepLIIF', modalities_no=1, seg_gen=True, mod_id_seg='S', input_id=0, input_nc=3, output_nc=3, ngf=8,
                                norm='batch', padding='zero', net_g='resnet_9blocks', net_gs='unet_32', input_no=1, scale_size=64,
                                modalities_names=['input1', 'mod1'], background_colors=[(201, 211, 208)], gpu_ids=[])
    nets = I.build_generators(opt, torch.device('cpu'), 'fp32')
    img = synth_image(150, 230, 13)
    img[:64] = 250                               # an empty first tile row
    img = torch.from_numpy(img)
    out = {}
    for p in ('fp32', 'fp16'):
        for net in nets.values():
            net.set_precision(p)
        out[p], band = I.infer_region([img], 64, 4, nets, opt, seg_weights=[0.5, 0.5], batch_size=3)
        assert band == (0, img.shape[0])
    assert set(out['fp16']) == set(out['fp32'])
    for k, ref in out['fp32'].items():
        d = (out['fp16'][k].to(torch.int32) - ref.to(torch.int32)).abs()
        assert int(d.max()) <= 2 and float((d > 1).float().mean()) < 1e-3, (k, int(d.max()))
