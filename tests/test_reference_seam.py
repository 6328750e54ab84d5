"""Build-container only (needs /root/reference; skipped where it is absent, e.g. on the GPU box): the reference's REAL Options object --
built from a cli.py-style d_params dict, and re-read from the train_opt.txt it saves -- drives deepliif_amd.models.create_model()
unchanged.  This is the field contract of SURVEY.md section 5: the drop-in reads nothing the reference's option object does not carry."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.skipif(not os.path.isdir('/root/reference/deepliif'), reason='the reference checkout only exists in the build container')


@pytest.fixture()
def ref_options():
    import _ref_import
    _ref_import.install_stubs()
    from deepliif.options import Options, print_options
    return Options, print_options


def _d_params(tmp_path):
    # the locals() of cli.train that reach Options (cli.py:242-386), reduced to the fields that exist there
    return dict(dataroot='x', name='run', gpu_ids=[0], checkpoints_dir=str(tmp_path), model='DeepLIIF', seg_weights=[0.25, 0.25, 0.5], loss_G_weights=[0.3, 0.3, 0.4],
                loss_D_weights=[0.3, 0.3, 0.4], modalities_no=2, modalities_names=[], seg_gen=True, seg_no=1, net_ds='n_layers', net_gs='unet_64', gan_mode='vanilla', gan_mode_s='lsgan',
                input_nc=3, output_nc=3, ngf=8, ndf=8, net_d='n_layers', net_g='resnet_9blocks', n_layers_d=4, norm='batch', init_type='normal', init_gain=0.02,
                no_dropout=True, upsample='convtranspose', label_smoothing=0.0, direction='AtoB', serial_batches=False, num_threads=4, batch_size=1, load_size=64,
                crop_size=64, max_dataset_size=None, preprocess='none', no_flip=True, display_winsize=64, epoch='latest', load_iter=0, verbose=False,
                lambda_identity=0.5, phase='train', n_epochs=1, n_epochs_decay=2, optimizer='adam', beta1=0.5, lr_g=2e-4, lr_d=2e-4, lr_policy='linear',
                lr_decay_iters=50, epoch_count=0, continue_train=False, padding='zero', local_rank=None, seed=None, use_torchrun=None, scale_size=64, input_no=1,
                dataset_mode='aligned', remote=False, debug=False, with_val=False)


def test_reference_options_object_drives_the_drop_in(tmp_path, ref_options):
    import fake_backend
    from deepliif_amd import models as M
    from golden_util import seeded_uniform
    Options, print_options = ref_options
    opt = Options(d_params=_d_params(tmp_path))
    assert opt.is_train and opt.lambda_feat == 100 and opt.n_layers_D == 4 and opt.netG == 'resnet_9blocks'       # what Options adds itself
    opt.allow_no_vgg = True            # lambda_feat = 100 needs a VGG19 weight file (a download in the reference): explicit opt-out here

    class CpuModel(M.DeepLIIFModel):
        def _device_from_opt(self, o):
            return torch.device('cpu')

        def _net_gpu_ids(self):
            return []
    fake_backend.install()
    try:
        torch.manual_seed(0)
        M._MODEL_CLASSES['DeepLIIF'] = CpuModel
        model = M.create_model(opt)
        model.setup(opt)
        assert model.model_names == ['G1', 'D1', 'G2', 'D2', 'GS0', 'DS0', 'GS1', 'DS1', 'GS2', 'DS2']
        B = [seeded_uniform((1, 3, 64, 64), 2 + i) for i in range(3)]
        model.set_input({'A': seeded_uniform((1, 3, 64, 64), 1), 'B': B, 'A_paths': ['a']})
        model.optimize_parameters()
        losses = model.get_current_losses()
        assert list(losses) == model.loss_names and all(torch.isfinite(torch.tensor(v)) for v in losses.values())
        # the sidecar the reference writes, re-read by the drop-in's own parser
        os.makedirs(os.path.join(str(tmp_path), 'run'), exist_ok=True)
        print_options(opt, save=True)
        model.save_networks('latest')
        from deepliif_amd import inference as I
        topt = I.get_opt(os.path.join(str(tmp_path), 'run'))
        assert (topt.model, topt.modalities_no, topt.seg_gen, topt.mod_id_seg, topt.input_id, topt.scale_size) == ('DeepLIIF', 2, True, 'S', 0, 64)
        assert topt.seg_weights == [0.25, 0.25, 0.5] and topt.modalities_names == ['input1', 'mod1', 'mod2']
    finally:
        M._MODEL_CLASSES['DeepLIIF'] = M.DeepLIIFModel
        fake_backend.uninstall()


def test_files_traced_by_the_reference_load_through_the_default_route(tmp_path, ref_options):
    """`deepliif serialize` (cli.py:796-811) on a checkpoint directory: the reference's own init_nets(eager) nets, disable_batchnorm_tracking_stats,
    torch.jit.trace, save -- done here with the reference's code -- then the drop-in's init_nets(dir) (eager_mode=False, the reference's default,
    models/__init__.py:216-219) reads those `<name>.pt` files and reproduces the reference's run_dask bytes for the directory."""
    import numpy as np
    from PIL import Image
    import fake_backend
    from deepliif.models import init_nets as ref_init_nets
    from deepliif.util import disable_batchnorm_tracking_stats
    from deepliif_amd import inference as I
    from golden_util import synth_image
    from seam_util import Z, build_checkpoint_dir, close_u8
    Options, _ = ref_options
    mdir = build_checkpoint_dir(tmp_path, 'dl_m2')
    ropt = Options(path_file=os.path.join(mdir, 'train_opt.txt'), mode='test')
    ropt.ngf, ropt.gpu_ids, ropt.epoch = 8, [], 'latest'
    sdir = os.path.join(str(tmp_path), 'serialized')
    os.makedirs(sdir)
    import shutil
    shutil.copy(os.path.join(mdir, 'train_opt.txt'), os.path.join(sdir, 'train_opt.txt'))
    sample = torch.zeros(1, 3, 64, 64)
    for name, net in ref_init_nets(mdir, eager_mode=True, opt=ropt, phase='test').items():
        net = disable_batchnorm_tracking_stats(net.eval()).cpu()
        torch.jit.trace(net, sample).save(os.path.join(sdir, f'{name}.pt'))
    assert sorted(os.listdir(sdir)) == ['G1.pt', 'G2.pt', 'GS0.pt', 'GS1.pt', 'GS2.pt', 'train_opt.txt']
    fake_backend.install()
    old = I._device_for
    I._device_for = lambda opt: torch.device('cpu')
    I._NETS_CACHE.clear()
    try:
        opt = I.get_opt(sdir)
        opt.ngf, opt.precision = 8, 'fp32'
        tile = Image.fromarray(synth_image(150, 100, 31)).crop((0, 0, 64, 64))
        res = I.run_dask(tile, model_path=sdir, opt=opt)
        assert list(res) == Z['dl_m2/run_dask_keys'].tolist()
        for k, v in res.items():
            close_u8(v, Z[f'dl_m2/run_dask/{k}'], 0.01)
    finally:
        I._device_for = old
        I._NETS_CACHE.clear()
        fake_backend.uninstall()


def test_files_serialized_by_the_engine_load_in_the_reference(tmp_path, ref_options):
    """The other direction of the seam (VERDICT r3 missing #1): `deepliif_amd.export.serialize` writes `<name>.pt` for nets that live on the engine;
    the UNMODIFIED reference reads the directory through its default route -- init_nets(dir, eager_mode=False) = torch.jit.load
    (models/__init__.py:117-121,216-219) -- and its own run_dask reproduces the bytes the reference produced from the checkpoints."""
    import numpy as np
    from PIL import Image
    import fake_backend
    from deepliif.models import init_nets as ref_init_nets
    from deepliif_amd import export as X
    from deepliif_amd import inference as I
    from golden_util import synth_image
    from seam_util import Z, build_checkpoint_dir, close_u8
    Options, _ = ref_options
    mdir = build_checkpoint_dir(tmp_path, 'dl_m2')
    sdir = os.path.join(str(tmp_path), 'serialized')
    fake_backend.install()
    old = I._device_for
    I._device_for = lambda opt: torch.device('cpu')
    I._NETS_CACHE.clear()
    try:
        opt = I.get_opt(mdir)
        opt.ngf, opt.precision = 8, 'fp32'
        X.serialize(mdir, sdir, device='cpu', opt=opt)
    finally:
        I._device_for = old
        I._NETS_CACHE.clear()
        fake_backend.uninstall()
    assert sorted(os.listdir(sdir)) == ['G1.pt', 'G2.pt', 'GS0.pt', 'GS1.pt', 'GS2.pt', 'train_opt.txt']
    ropt = Options(path_file=os.path.join(sdir, 'train_opt.txt'), mode='test')
    ropt.ngf, ropt.gpu_ids, ropt.epoch = 8, [], 'latest'
    nets = ref_init_nets(sdir, eager_mode=False, opt=ropt, phase='test')
    assert sorted(nets) == ['G1', 'G2', 'GS0', 'GS1', 'GS2'] and all(isinstance(n, torch.jit.ScriptModule) for n in nets.values())
    # every module the reference loaded computes what the checkpoint's network computes (the oracle is pinned to the reference's eager nets);
    # the reference's own run_dask needs torchvision.transforms, which this container lacks (tests/golden/make_golden_seam.py carries stand-ins)
    from oracle import deepliif_oracle as O
    x = I.transform(Image.fromarray(synth_image(150, 100, 31)).crop((0, 0, 64, 64)))
    for name, arch in zip(Z['dl_m2/model_names'].tolist(), Z['dl_m2/net_arch'].tolist()):
        if not name.startswith('G'):
            continue
        a, cin, pad = arch.split('|')
        sd = torch.load(os.path.join(mdir, f'latest_net_{name}.pth'), map_location='cpu')
        with torch.no_grad():
            got, exp = nets[name](x.clone()), O.run_generator(a, sd, x.clone(), norm='batch', padding_type=pad)
        assert float((got - exp).abs().max()) < 1e-5 * float(exp.abs().max()), name
