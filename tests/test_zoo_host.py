"""Host logic of the DeepLIIFKD and CycleGAN drop-ins (deepliif_amd/models.py) on CPU with the emulated ops backend, against the trajectories
recorded from the REFERENCE classes (tests/golden/step_kd_m2.npz, step_cyclegan_m2.npz; tests/golden/make_golden_zoo.py): names, losses, generated
images, the teacher's images, updated weights and the consumption of Python's `random` stream by the image pools."""
import random
import types

import numpy as np
import pytest
import torch

import fake_backend
import seam_util
import zoo_util as Z
from deepliif_amd import inference as I
from deepliif_amd import models as M
from golden_util import digest_close
from test_host_model import make_opt


@pytest.fixture(autouse=True)
def _fake(monkeypatch):
    fake_backend.install()
    monkeypatch.setattr(I, '_device_for', lambda opt: torch.device('cpu'))
    monkeypatch.setattr(I, '_NETS_CACHE', {})
    yield
    fake_backend.uninstall()


class _Cpu:
    def _device_from_opt(self, opt):
        return torch.device('cpu')

    def _net_gpu_ids(self):
        return []


class CpuKD(_Cpu, M.DeepLIIFKDModel):
    pass


class CpuCycleGAN(_Cpu, M.CycleGANModel):
    pass


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def flat_weights(net):
    return torch.cat([v.reshape(-1).float() for v in net.state_dict().values() if v.is_floating_point()])


def test_map_model_names_restates_the_reference():
    """deepliif/util/util.py:273-292 (values worked out from the reference's code)"""
    assert M.map_model_names(['G1', 'G2', 'G51', 'G52', 'G53'], '5', '1', 'S', '0') == \
        {'G1': 'G1', 'G2': 'G2', 'G51': 'GS0', 'G52': 'GS1', 'G53': 'GS2', 'G5': 'GS'}
    assert M.map_model_names(['G1', 'GS0', 'GS1'], 'S', '0', 'S', '0') == {'G1': 'G1', 'GS0': 'GS0', 'GS1': 'GS1', 'GS': 'GS'}
    assert M.map_model_names(['GS0', 'GS1'], 'S', '0', '3', '1') == {'GS0': 'G31', 'GS1': 'G32', 'GS': 'G3'}


def test_deepliifkd_two_steps_follow_the_reference(tmp_path):
    z = Z.kd_fixture()
    opt = make_opt(2, True, str(z['meta'][1]), net_gs=str(z['meta'][3]), nf=int(z['meta'][5]))
    opt.model, opt.model_dir_teacher = 'DeepLIIFKD', seam_util.build_kd_teacher_dir(tmp_path)
    model = CpuKD(opt)
    model.setup(opt)
    assert model.loss_names == [str(n) for n in z['loss_names']]
    assert model.model_names == [str(n) for n in z['model_names']]
    assert sorted(f'{k}={v}' for k, v in model.d_mapping_model_name.items()) == sorted(z['teacher_mapping'].tolist())      # (the reference's dict follows its GPU-group order)
    for name, sd in Z.kd_student_state_dicts(z).items():
        getattr(model, 'net' + name).load_state_dict(sd, strict=True)
    for (name, _, _), dg in zip(seam_util.KD_TEACHER_NETS, z['teacher_digest']):
        ok, msg = digest_close(torch.cat([v.reshape(-1).float() for k, v in model.nets_teacher[name].state_dict().items()
                                          if v.is_floating_point()]), dg, 1e-12)
        assert ok, (name, msg)
    A, B = Z.kd_inputs(z)
    S = str(model.mod_id_seg)
    for s in range(int(z['meta'][7])):
        model.set_input({'A': A, 'B': B, 'A_paths': ['x']})
        model.optimize_parameters()
        got = model.get_current_losses()
        assert list(got) == list(dict.fromkeys(model.loss_names))
        if s == 0:
            for i in range(2):
                assert rel(getattr(model, f'fake_B_{i + 1}_teacher')[:, :, ::2, ::2], z[f'teacher/fake_B_{i + 1}']) < 5e-4
            for i in range(3):
                assert rel(getattr(model, f'fake_B_{S}_{i}_teacher')[:, :, ::2, ::2], z[f'teacher/fake_B_S_{i}']) < 5e-4
            assert rel(getattr(model, f'fake_B_{S}_teacher')[:, :, ::2, ::2], z['teacher/fake_B_S']) < 5e-4
        tol = 5e-4 if s == 0 else 5e-3
        for name, exp in zip(model.loss_names, z[f'step{s}/losses']):
            assert abs(got[name] - exp) <= tol * max(abs(exp), 1e-3 if 'KLDiv' in name else 1.0), (s, name, got[name], exp)
        for name, exp in zip(z['extra_loss_names'], z[f'step{s}/extra_losses']):
            v = float(getattr(model, 'loss_' + str(name)))
            assert abs(v - exp) <= tol * max(abs(exp), 1e-3), (s, str(name), v, exp)
        for i in range(2):
            assert rel(getattr(model, f'fake_B_{i + 1}')[:, :, ::2, ::2], z[f'step{s}/fake_B_{i + 1}']) < (5e-4 if s == 0 else 3e-2)
        assert rel(getattr(model, f'fake_B_{S}')[:, :, ::2, ::2], z[f'step{s}/fake_B_S']) < (5e-4 if s == 0 else 3e-2)
        for n in model.model_names:
            ok, msg = digest_close(flat_weights(getattr(model, 'net' + n)), z[f'step{s}/w_digest/{n}'], 1e-3)
            assert ok, f'step {s} weights of {n}: {msg}'
    # the teacher is frozen: its weights still have the seeded digests, and it never entered an optimizer
    for (name, _, _), dg in zip(seam_util.KD_TEACHER_NETS, z['teacher_digest']):
        ok, msg = digest_close(torch.cat([v.reshape(-1).float() for v in model.nets_teacher[name].state_dict().values() if v.is_floating_point()]), dg, 1e-12)
        assert ok, (name, msg)
    owned = {id(p) for o in model.optimizers for p in o.flat.params}
    assert not any(id(p) in owned for net in model.nets_teacher.values() for p in net.parameters())


def cyc_opt(z, precision='fp32'):
    opt = make_opt(2, False, str(z['meta'][1]), nf=int(z['meta'][5]), precision=precision)
    opt.model, opt.net_g, opt.padding = 'CycleGAN', str(z['meta'][3]), str(z['meta'][2])
    opt.gan_mode, opt.pool_size, opt.BtoA, opt.allow_no_vgg = str(z['meta'][9]), int(z['meta'][8]), False, True      # fixture: VGG terms zeroed
    return opt


def test_cyclegan_two_steps_follow_the_reference():
    z = Z.cyc_fixture()
    opt = cyc_opt(z)
    model = CpuCycleGAN(opt)
    model.setup(opt)
    assert model.loss_names == [str(n) for n in z['loss_names']] and model.model_names == [str(n) for n in z['model_names']]
    for name, sd in Z.cyc_state_dicts(z).items():
        model._net(name).load_state_dict(sd, strict=True)
    A, Bs = Z.cyc_inputs(z)
    random.seed(int(z['meta'][10]))
    for s in range(int(z['meta'][7])):
        model.set_input({'A': A, 'Bs': Bs, 'A_paths': ['x']})
        model.optimize_parameters()
        got = model.get_current_losses()
        tol = 5e-4 if s == 0 else 5e-3
        for name, exp in zip(model.loss_names, z[f'step{s}/losses']):
            assert abs(got[name] - exp) <= tol * max(1.0, abs(exp)), (s, name, got[name], exp)
        for fam in ('fake_Bs', 'rec_As', 'fake_As', 'rec_Bs'):
            for i in range(2):
                assert rel(getattr(model, fam)[i][:, :, ::2, ::2], z[f'step{s}/{fam}_{i + 1}']) < (5e-4 if s == 0 else 3e-2), (s, fam, i)
        for n in model.model_names:
            ok, msg = digest_close(flat_weights(model._net(n)), z[f'step{s}/w_digest/{n}'], 1e-3)
            assert ok, f'step {s} weights of {n}: {msg}'
    assert random.random() == float(z['random_after'][0]), 'the image pools must consume exactly the reference\'s draws'
    vis = model.get_current_visuals()
    assert list(vis) == model.visual_names and all(v.shape == A.shape for v in vis.values())


def test_cyclegan_needs_vgg_weights_unless_opted_out():
    z = Z.cyc_fixture()
    opt = cyc_opt(z)
    opt.allow_no_vgg = False
    with pytest.raises(RuntimeError, match='VGG19'):
        CpuCycleGAN(opt)


def test_cyclegan_inference_direction_and_result_names():
    """test time: only one direction is built (CycleGAN_model.py:57-62); run_dask returns the generators' images under the NET names
    (models/__init__.py:362-372, 577-579), black tiles under the same names for empty input (:453-457)"""
    z = Z.cyc_fixture()
    for btoa, names in ((False, ['GA_1', 'GA_2']), (True, ['GB_1', 'GB_2'])):
        opt = cyc_opt(z)
        opt.is_train, opt.BtoA, opt.seg_gen, opt.scale_size = False, btoa, False, 64
        m = CpuCycleGAN(opt)
        assert m.model_names == names and [len(m.netGA), len(m.netGB)] == ([0, 2] if btoa else [2, 0])
        assert I.generator_names(opt) == (names, [])
        nets = I.build_generators(opt, torch.device('cpu'), 'fp32')
        sds = Z.cyc_state_dicts(z)
        for n in names:
            nets[n].load_state_dict(sds[n])
        x = Z.cyc_inputs(z)[0][:1]
        res = I.run_dask(x, nets=nets, opt=opt, output_tensor=True)
        assert list(res) == names and all(v.shape == (1, 3, 64, 64) for v in res.values())
        assert list(I.empty_tile_colors(opt)) == names


class CpuDeepLIIF(_Cpu, M.DeepLIIFModel):
    pass


def test_non_default_cli_options_follow_the_reference():
    """--upsample resize_conv (nearest x2 + reflect-padded 3x3 conv), --net-d pixel (1x1 PatchGAN), --gan-mode wgangp (+-mean of the prediction)"""
    z = Z.opt_fixture()
    opt = make_opt(1, False, str(z['meta'][1]), nf=int(z['meta'][5]))
    opt.upsample, opt.net_d, opt.gan_mode = str(z['meta'][8]), str(z['meta'][9]), str(z['meta'][10])
    model = CpuDeepLIIF(opt)
    model.setup(opt)
    assert type(model.netD1).__name__ == 'PixelDiscriminator' and model.netG1.upsample == 'resize_conv'
    for name, sd in Z.opt_state_dicts(z).items():
        getattr(model, 'net' + name).load_state_dict(sd, strict=True)
    A, B = Z.opt_inputs(z)
    for s in range(int(z['meta'][7])):
        model.set_input({'A': A, 'B': B, 'A_paths': ['x']})
        model.optimize_parameters()
        got = model.get_current_losses()
        tol = 5e-4 if s == 0 else 5e-3
        for name, exp in zip(model.loss_names, z[f'step{s}/losses']):
            assert abs(got[name] - exp) <= tol * max(1.0, abs(exp)), (s, name, got[name], exp)
        assert rel(model.fake_B_1[:, :, ::2, ::2], z[f'step{s}/fake_B_1']) < (5e-4 if s == 0 else 3e-2)
        for n in model.model_names:
            ok, msg = digest_close(flat_weights(getattr(model, 'net' + n)), z[f'step{s}/w_digest/{n}'], 1e-3)
            assert ok, f'step {s} weights of {n}: {msg}'
