"""The checkpoint-directory / inference seam on CPU (emulated ops backend) against what the REFERENCE produced from the same
directories (tests/golden/make_golden_seam.py): train_opt.txt parsing + test-mode defaults, init_nets from '<epoch>_net_<name>.pth',
run_dask(PIL) and inference(PIL) for DeepLIIF / DeepLIIFExt / SDG, transform / tensor_to_pil, save_networks / load_networks file
format, and the four learning-rate policies."""
import os
import types

import numpy as np
import pytest
import torch
from PIL import Image

import fake_backend
from deepliif_amd import inference as I
from deepliif_amd import models as M
from deepliif_amd import networks as N
from golden_util import synth_image
from seam_util import Z, build_checkpoint_dir, close_u8, serialize_checkpoint_dir


@pytest.fixture(autouse=True)
def _fake(monkeypatch):
    fake_backend.install()
    monkeypatch.setattr(I, '_device_for', lambda opt: torch.device('cpu'))
    I._NETS_CACHE.clear()
    yield
    fake_backend.uninstall()


def _test_opt(mdir, precision='fp32'):
    opt = I.get_opt(mdir)
    opt.ngf = 8                       # test-mode Options force ngf = 64 (options/__init__.py:75); the fixture nets are 8 wide
    opt.precision = precision
    return opt


@pytest.mark.parametrize('tag', ['dl_m2', 'ext_m2', 'sdg_m2_in2'])
def test_get_opt_backfills_like_the_reference(tmp_path, tag):
    mdir = build_checkpoint_dir(tmp_path, tag)
    opt = I.get_opt(mdir)
    for item in Z[f'{tag}/test_opt'].tolist():
        k, v = item.split('=', 1)
        assert repr(getattr(opt, k)) == v, (k, repr(getattr(opt, k)), v)


def _images():
    img = Image.fromarray(synth_image(150, 100, 31))
    a = np.asarray(img).copy()
    a[:, :70] = 252
    return img, Image.fromarray(a)


def test_deepliif_run_dask_and_inference_from_a_checkpoint_dir(tmp_path):
    mdir = build_checkpoint_dir(tmp_path, 'dl_m2')
    opt = _test_opt(mdir)
    img, img2 = _images()
    tile = img.crop((0, 0, 64, 64))
    res = I.run_dask(tile, model_path=mdir, eager_mode=True, opt=opt)
    assert list(res) == Z['dl_m2/run_dask_keys'].tolist()
    for k, v in res.items():
        close_u8(v, Z[f'dl_m2/run_dask/{k}'], 0.01)
    res = I.run_dask(tile, model_path=mdir, eager_mode=True, opt=opt, seg_only=True, seg_weights=[0.5, 0.0, 0.5])
    assert list(res) == Z['dl_m2/run_dask_segonly_keys'].tolist()
    for k, v in res.items():
        close_u8(v, Z[f'dl_m2/run_dask_segonly/{k}'], 0.01)
    for name, kw in (('inf', {}), ('inf_seginter', dict(return_seg_intermediate=True)), ('inf_modonly', dict(mod_only=True)), ('inf_segonly', dict(seg_only=True))):
        r = I.inference(img2, 64, 4, mdir, eager_mode=True, opt=opt, batch_size=3, **kw)
        assert list(r) == Z[f'dl_m2/{name}_keys'].tolist(), name
        for k, v in r.items():
            exp = Z[f'dl_m2/{name}/{k}']
            close_u8(v, exp, 0.01)
            assert np.array_equal(np.asarray(v)[:, :60], exp[:, :60]), (name, k)        # the constant (is_empty) tiles: exact


@pytest.mark.parametrize('tag', ['ext_m2', 'sdg_m2_in2'])
def test_ext_and_sdg_inference_from_a_checkpoint_dir(tmp_path, tag):
    """run_dask / inference for the list-valued model families (models/__init__.py:362-388, 567-575): GS_i(cat(tile, G_1, G_i)) with
    9-channel input; SDG reads its input modalities side by side from one wide image"""
    mdir = build_checkpoint_dir(tmp_path, tag)
    opt = _test_opt(mdir)
    img, img2 = _images()
    if tag == 'ext_m2':
        res = I.run_dask(img.crop((0, 0, 64, 64)), model_path=mdir, eager_mode=True, opt=opt)
        assert list(res) == Z[f'{tag}/run_dask_keys'].tolist()
        for k, v in res.items():
            close_u8(v, Z[f'{tag}/run_dask/{k}'], 0.01)
        src = img2
    else:
        src = Image.fromarray(np.concatenate([np.asarray(img2), synth_image(150, 100, 32)], axis=1))
    r = I.inference(src, 64, 4, mdir, eager_mode=True, opt=opt, batch_size=4)
    assert list(r) == Z[f'{tag}/inf_keys'].tolist()
    for k, v in r.items():
        close_u8(v, Z[f'{tag}/inf/{k}'], 0.01)


def test_serialized_model_directory_is_the_default_inference_route(tmp_path):
    """init_nets(model_dir) with the reference's default eager_mode=False reads `<name>.pt` (deepliif/models/__init__.py:216-219): a directory
    holding ONLY the TorchScript files + train_opt.txt (what `deepliif serialize` writes and what is distributed) gives the same bytes as the
    `.pth` directory it was made from; eager_mode=True on it fails like the reference (no checkpoint to load), naming the file that exists"""
    mdir = build_checkpoint_dir(tmp_path, 'dl_m2')
    sdir = serialize_checkpoint_dir(mdir, str(tmp_path / 'serialized'))
    assert sorted(os.listdir(sdir)) == ['G1.pt', 'G2.pt', 'GS0.pt', 'GS1.pt', 'GS2.pt', 'train_opt.txt']
    opt = _test_opt(sdir)
    img, _ = _images()
    tile = img.crop((0, 0, 64, 64))
    res = I.run_dask(tile, model_path=sdir, opt=opt)                       # eager_mode defaults to False, as in the reference
    assert list(res) == Z['dl_m2/run_dask_keys'].tolist()
    for k, v in res.items():
        close_u8(v, Z[f'dl_m2/run_dask/{k}'], 0.01)
    ref = I.run_dask(tile, model_path=mdir, eager_mode=True, opt=_test_opt(mdir))
    for k in res:
        assert np.array_equal(np.asarray(res[k]), np.asarray(ref[k])), k         # same weights -> same bytes
    with pytest.raises(FileNotFoundError, match='G1.pt does'):
        I.init_nets(sdir, eager_mode=True, opt=opt)
    # eager_mode=False on a directory that was never serialized falls back to the checkpoints
    nets = I.init_nets(mdir, eager_mode=False, opt=_test_opt(mdir))
    assert list(nets) == ['G1', 'G2', 'GS0', 'GS1', 'GS2']


def test_init_nets_cache_distinguishes_options(tmp_path):
    """the cache key must include what the reference's lru_cache keys on (opt, eager_mode): a later call with other options must not
    get nets built for the first one"""
    mdir = build_checkpoint_dir(tmp_path, 'dl_m2')
    a = I.init_nets(mdir, eager_mode=True, opt=_test_opt(mdir, 'fp32'))
    b = I.init_nets(mdir, eager_mode=True, opt=_test_opt(mdir, 'bf16'))
    assert a is not b and next(iter(a.values())).precision == 'fp32' and next(iter(b.values())).precision == 'bf16'
    assert I.init_nets(mdir, eager_mode=True, opt=_test_opt(mdir, 'fp32')) is a


def test_ext_tiled_inference_ignores_mod_only_like_run_wrapper(tmp_path):
    """run_wrapper calls run_fn(tile, model_path, None, eager_mode, opt) for DeepLIIFExt / SDG (models/__init__.py:446-452) -- seg_only /
    mod_only never reach the generator DAG, so inference(mod_only=True) still returns Seg_i"""
    mdir = build_checkpoint_dir(tmp_path, 'ext_m2')
    opt = _test_opt(mdir)
    _, img2 = _images()
    r = I.inference(img2, 64, 4, mdir, eager_mode=True, opt=opt, batch_size=4, mod_only=True)
    assert list(r) == Z['ext_m2/inf_keys'].tolist()
    for k, v in r.items():
        close_u8(v, Z[f'ext_m2/inf/{k}'], 0.01)


def test_transform_and_tensor_to_pil_bytes():
    ts = I.transform(Image.fromarray(Z['dl_m2/transform_in']))
    assert torch.equal(ts, torch.from_numpy(Z['dl_m2/transform_out']))
    odd = I.transform(Image.fromarray(Z['dl_m2/transform_odd_in']))             # 70 x 61 -> bicubic 72 x 60
    assert torch.equal(odd, torch.from_numpy(Z['dl_m2/transform_odd_out']))
    assert np.array_equal(np.asarray(I.tensor_to_pil(ts * 0.731)), Z['dl_m2/t2p'])


@pytest.mark.parametrize('policy', ['linear', 'step', 'cosine', 'plateau'])
def test_scheduler_sequences(policy):
    o = types.SimpleNamespace(**{a.split('=')[0]: (a.split('=')[1] if a.split('=')[0] == 'lr_policy' else int(a.split('=')[1])) for a in Z[f'sched/{policy}_args'].tolist()})
    p = torch.nn.Parameter(torch.zeros(4))
    optim = N.get_optimizer('adam')([p], lr=2e-4, betas=(0.5, 0.999))
    sch = N.get_scheduler(optim, o)
    lrs = [optim.param_groups[0]['lr']]
    for e in range(8):
        if policy == 'plateau':
            sch.step(1.0 if e < 1 else 1.0 + 0.01 * e)
        else:
            sch.step()
        lrs.append(optim.param_groups[0]['lr'])
    assert np.allclose(lrs, Z[f'sched/{policy}'], rtol=1e-12, atol=1e-18)


class CpuModel(M.DeepLIIFModel):
    def _device_from_opt(self, opt):
        return torch.device('cpu')

    def _net_gpu_ids(self):
        return []


def test_save_networks_writes_the_reference_file_set_and_round_trips(tmp_path):
    from test_host_model import make_opt
    torch.manual_seed(3)
    opt = make_opt(2, True, 'batch')
    opt.checkpoints_dir, opt.name = str(tmp_path), 'run'
    model = CpuModel(opt)
    model.setup(opt)
    model.save_networks('latest')
    files = sorted(os.listdir(os.path.join(str(tmp_path), 'run')))
    assert files == [f for f in Z['dl_m2/files'].tolist() if f.endswith('.pth')]
    for n in model.model_names:
        sd = torch.load(os.path.join(str(tmp_path), 'run', f'latest_net_{n}.pth'))
        assert list(sd.keys()) == Z[f'dl_m2/sd_keys/{n}'].tolist()
        assert ['x'.join(str(x) for x in v.shape) for v in sd.values()] == Z[f'dl_m2/sd_shapes/{n}'].tolist()
        assert all(v.device.type == 'cpu' for v in sd.values())
    before = {n: {k: v.clone() for k, v in getattr(model, 'net' + n).state_dict().items()} for n in model.model_names}
    torch.manual_seed(4)
    opt2 = make_opt(2, True, 'batch')
    opt2.checkpoints_dir, opt2.name, opt2.continue_train = str(tmp_path), 'run', True
    m2 = CpuModel(opt2)
    m2.setup(opt2)                                    # continue_train -> load_networks('latest') (base_model.py:87-89)
    for n in model.model_names:
        for k, v in getattr(m2, 'net' + n).state_dict().items():
            assert torch.equal(v, before[n][k]), (n, k)
    assert m2.optimizer_G.flat.attached()             # load_state_dict copied INTO the flat buffers


def test_update_learning_rate_and_visuals_follow_the_reference_contract(capsys):
    from test_host_model import make_opt
    from golden_util import seeded_uniform
    torch.manual_seed(0)
    opt = make_opt(1, True, 'batch')
    opt.n_epochs, opt.n_epochs_decay = 1, 2
    model = CpuModel(opt)
    model.setup(opt)
    model.set_input({'A': seeded_uniform((1, 3, 64, 64), 1), 'B': [seeded_uniform((1, 3, 64, 64), 2), seeded_uniform((1, 3, 64, 64), 3)], 'A_paths': ['p']})
    model.calculate_losses()                          # validation path: losses + gradients, no optimizer step (DeepLIIF_model.py:469-507)
    losses = model.get_current_losses()
    assert list(losses) == model.loss_names and all(np.isfinite(v) for v in losses.values())
    vis = model.get_current_visuals()
    assert list(vis)[:3] == ['real_A', 'fake_B_1', 'real_B_1'] and all(v.shape == (1, 3, 64, 64) for v in vis.values())
    assert model.get_image_paths() == ['p']
    lr0 = model.optimizers[0].param_groups[0]['lr']
    model.update_learning_rate()
    model.update_learning_rate()
    assert 'learning rate = ' in capsys.readouterr().out
    assert model.optimizers[0].param_groups[0]['lr'] == pytest.approx(lr0 * (1 - 1 / 3))


def test_optimizer_state_resumes_training_and_loads_into_torch_adam(tmp_path):
    """save_optimizers / load_optimizers (an extension: the reference stores no optimizer state, base_model.py:190-208).  (1) A run that is
    saved after two steps, rebuilt from the files and stepped once more ends on the SAME weights as the uninterrupted run (bitwise: Adam's
    moments and step count came back).  (2) The file is torch.optim.Adam's own layout: torch.optim.Adam over the same parameter list loads it
    and holds the same moments."""
    from test_host_model import make_opt
    from golden_util import seeded_uniform
    batch = {'A': seeded_uniform((1, 3, 64, 64), 1), 'B': [seeded_uniform((1, 3, 64, 64), 2), seeded_uniform((1, 3, 64, 64), 3)], 'A_paths': ['p']}

    def build(seed, cont):
        torch.manual_seed(seed)
        o = make_opt(1, True, 'batch')
        o.checkpoints_dir, o.name, o.continue_train = str(tmp_path), 'run', cont
        m = CpuModel(o)
        m.setup(o)
        return m

    a = build(5, False)
    for _ in range(2):
        a.set_input(batch)
        a.optimize_parameters()
    a.save_networks('latest')
    a.save_optimizers('latest')
    a.set_input(batch)
    a.optimize_parameters()                                     # the uninterrupted third step
    b = build(6, True)                                          # different init; continue_train loads the nets
    b.load_optimizers('latest')
    assert b.optimizer_G.step_count == 2 and b.optimizer_D.step_count == 2
    b.set_input(batch)
    b.optimize_parameters()
    for n in a.model_names:
        for (k, va), vb in zip(getattr(a, 'net' + n).state_dict().items(), getattr(b, 'net' + n).state_dict().values()):
            assert torch.equal(va, vb), (n, k)
    # torch.optim.Adam reads the same file
    sd = torch.load(os.path.join(str(tmp_path), 'run', 'latest_optimizer_0.pth'))
    params = [torch.nn.Parameter(p.detach().clone()) for p in b.optimizers[0].flat.params]
    ta = torch.optim.Adam(params, lr=1.0)
    ta.load_state_dict(sd)
    assert ta.param_groups[0]['lr'] == pytest.approx(sd['param_groups'][0]['lr']) and tuple(ta.param_groups[0]['betas']) == (0.5, 0.999)
    st = ta.state[params[3]]
    assert float(st['step']) == 2.0 and torch.equal(st['exp_avg'], sd['state'][3]['exp_avg'])
    with pytest.raises(ValueError):
        b.optimizers[1].load_state_dict(sd)                     # the generator set's state does not fit the discriminator set
