"""TorchScript EXPORT (`deepliif serialize`, cli.py:770-830) on CPU: the plain-torch twins of the engine nets against the pinned oracle, the
files `deepliif_amd.export.serialize` writes (names, state_dict keys, BatchNorm statistics dropped like disable_batchnorm_tracking_stats does),
and the round trip through the default inference route (init_nets(dir) reads `<name>.pt`) against the REFERENCE's run_dask bytes."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

import fake_backend
from deepliif_amd import export as X
from deepliif_amd import inference as I
from deepliif_amd import networks as N
from golden_util import synth_image
from oracle import deepliif_oracle as O
from seam_util import Z, build_checkpoint_dir, close_u8


@pytest.fixture(autouse=True)
def _fake(monkeypatch):
    fake_backend.install()
    monkeypatch.setattr(I, '_device_for', lambda opt: torch.device('cpu'))
    I._NETS_CACHE.clear()
    yield
    fake_backend.uninstall()


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize('arch,norm,pad,cin,hw', [
    ('resnet_9blocks', 'batch', 'zero', 3, 64), ('resnet_9blocks', 'instance', 'reflect', 3, 72), ('resnet_6blocks', 'batch', 'reflect', 6, 64),
    ('unet_64', 'batch', 'zero', 3, 64), ('unet_128', 'instance', 'zero', 9, 128),
])
def test_twin_reproduces_the_oracle_forward(arch, norm, pad, cin, hw):
    """the twin is plain ATen over the engine net's own parameter tree; the oracle is pinned to the reference (test_oracle_golden.py)"""
    net = N.define_G(cin, 3, 8, arch, norm, False, 'normal', 0.02, [], pad).eval()
    twin = X.disable_batchnorm_tracking_stats(X.aten_twin(net).eval())
    assert list(twin.state_dict().keys()) == [k for k in net.state_dict().keys() if k.rsplit('.', 1)[-1] not in ('running_mean', 'running_var')]
    x = torch.rand(2, cin, hw, hw, generator=torch.Generator().manual_seed(5)) * 2 - 1
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    with torch.no_grad():
        got = twin(x.clone())
        exp = O.run_generator(arch, sd, x.clone(), norm=norm, padding_type=pad)
    assert got.shape == exp.shape and rel(got, exp) < 1e-5
    # the engine net was not touched: BatchNorm statistics still tracked, parameters are not shared with the twin
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            assert m.track_running_stats and m.running_mean is not None
    p_net, p_twin = next(net.parameters()), next(twin.parameters())
    assert p_net.data_ptr() != p_twin.data_ptr()


def test_attention_unet_twin_reproduces_the_oracle_forward():
    net = N.define_G(3, 3, 64, 'unet_512_attention', 'batch', False, 'normal', 0.02, []).eval()
    twin = X.disable_batchnorm_tracking_stats(X.aten_twin(net).eval())
    x = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(6)) * 2 - 1
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    with torch.no_grad():
        got, exp = twin(x.clone()), O.run_generator('unet_512_attention', sd, x.clone())
    assert rel(got, exp) < 1e-5


def _test_opt(mdir, precision='fp32'):
    opt = I.get_opt(mdir)
    opt.ngf = 8
    opt.precision = precision
    return opt


@pytest.mark.parametrize('tag,files', [('dl_m2', ['G1.pt', 'G2.pt', 'GS0.pt', 'GS1.pt', 'GS2.pt']), ('ext_m2', ['GS_1.pt', 'GS_2.pt', 'G_1.pt', 'G_2.pt'])])
def test_serialize_writes_a_directory_the_default_route_reads(tmp_path, tag, files, capsys):
    mdir = build_checkpoint_dir(tmp_path, tag)
    sdir = str(tmp_path / 'serialized')
    report = X.serialize(mdir, sdir, device='cpu', opt=_test_opt(mdir))
    assert sorted(os.listdir(sdir)) == sorted(files + ['train_opt.txt'])
    assert open(os.path.join(sdir, 'train_opt.txt')).read() == open(os.path.join(mdir, 'train_opt.txt')).read()
    assert sorted(report) == sorted(f[:-3] for f in files) and max(report.values()) <= 1e-3        # eager twin vs traced file: the same ATen ops
    out = capsys.readouterr().out
    assert out.count('PASS') == len(files) and 'testing similarity' in out
    # each file is a stock TorchScript module: reference keys minus the BatchNorm statistics, callable by plain torch, equal to the oracle
    for f in files:
        name = f[:-3]
        ts = torch.jit.load(os.path.join(sdir, f), map_location='cpu')
        sd = torch.load(os.path.join(mdir, f'latest_net_{name}.pth'), map_location='cpu')
        keep = [k for k in sd if k.rsplit('.', 1)[-1] not in ('running_mean', 'running_var')]
        assert list(ts.state_dict().keys()) == keep
        for k in keep:
            assert torch.equal(ts.state_dict()[k], sd[k]), k
        arch, cin, pad = [a for n, a in zip(Z[f'{tag}/model_names'].tolist(), Z[f'{tag}/net_arch'].tolist()) if n == name][0].split('|')
        x = torch.rand(1, int(cin), 64, 64, generator=torch.Generator().manual_seed(9)) * 2 - 1
        with torch.no_grad():
            assert rel(ts(x.clone()), O.run_generator(arch, sd, x.clone(), norm='batch', padding_type=pad)) < 1e-5
    # the serialized directory alone (no .pth) serves inference through the reference's DEFAULT route and reproduces the reference's bytes
    if tag == 'dl_m2':
        tile = Image.fromarray(synth_image(150, 100, 31)).crop((0, 0, 64, 64))
        I._NETS_CACHE.clear()
        res = I.run_dask(tile, model_path=sdir, opt=_test_opt(sdir))
        assert list(res) == Z['dl_m2/run_dask_keys'].tolist()
        for k, v in res.items():
            close_u8(v, Z[f'dl_m2/run_dask/{k}'], 0.01)


def test_serialize_in_place_and_similarity_failure_is_loud(tmp_path, monkeypatch):
    """output_dir defaults to model_dir (cli.py:773); a traced file that disagrees with the original fails like util/__init__.py:741"""
    mdir = build_checkpoint_dir(tmp_path, 'dl_m2')
    X.serialize(mdir, None, device='cpu', opt=_test_opt(mdir))
    assert {'G1.pt', 'latest_net_G1.pth', 'train_opt.txt'} <= set(os.listdir(mdir))
    with pytest.raises(AssertionError, match='larger than threshold'):
        X.diff_original_serialized(lambda t: t, lambda t: t + 1.0, torch.zeros(1, 3, 8, 8))
    with pytest.raises(RuntimeError, match='no GPU'):
        monkeypatch.setattr(torch.cuda, 'is_available', lambda: False)
        X.serialize(mdir, None, device='gpu', opt=_test_opt(mdir))
