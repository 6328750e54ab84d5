"""oracle/deepliif_oracle.py: OracleDeepLIIFKD and OracleCycleGAN against trajectories recorded from the REFERENCE classes
(deepliif/models/DeepLIIFKD_model.py, CycleGAN_model.py; tests/golden/make_golden_zoo.py)."""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
from golden_util import digest_close  # noqa: E402
import zoo_util as Z  # noqa: E402
from oracle import deepliif_oracle as O  # noqa: E402


def rel_err(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_kldiv_restatement_equals_torch_kldivloss():
    """the oracle's formula vs the modules the reference composes (DeepLIIFKD_model.py:148-151)"""
    g = torch.Generator().manual_seed(0)
    x, t = torch.rand(2, 3, 16, 16, generator=g) * 2 - 1, torch.rand(2, 3, 16, 16, generator=g) * 2 - 1
    x, t = x.double(), t.double()
    ref = torch.nn.KLDivLoss(reduction='batchmean')(torch.nn.LogSoftmax(dim=-1)(x.view(1, 1, -1)), torch.nn.Softmax(dim=-1)(t.view(1, 1, -1)))
    assert abs(float(O.kldiv_whole_tensor(x, t)) - float(ref)) < 1e-12 * max(1.0, abs(float(ref)))


def test_deepliifkd_two_step_trajectory():
    z = Z.kd_fixture()
    model = Z.kd_oracle(z)
    A, B = Z.kd_inputs(z)
    names = [str(n) for n in z['loss_names']]
    for s in range(int(z['meta'][7])):
        model.set_input({'A': A, 'B': B})
        model.optimize_parameters()
        got = model.current_losses()
        if s == 0:
            for i in range(2):
                assert rel_err(model.teacher_B[i][:, :, ::2, ::2], z[f'teacher/fake_B_{i + 1}']) < 2e-4
            for i in range(3):
                assert rel_err(model.teacher_seg_parts[i][:, :, ::2, ::2], z[f'teacher/fake_B_S_{i}']) < 2e-4
            assert rel_err(model.teacher_seg[:, :, ::2, ::2], z['teacher/fake_B_S']) < 2e-4
        tol = 2e-4 if s == 0 else 3e-3
        for name, exp in zip(names, z[f'step{s}/losses']):
            assert abs(got[name] - exp) <= tol * max(abs(exp), 1e-3 if 'KLDiv' in name else 1.0), (s, name, got[name], exp)
        for name, exp in zip(z['extra_loss_names'], z[f'step{s}/extra_losses']):
            assert abs(got[str(name)] - exp) <= tol * max(abs(exp), 1e-3), (s, str(name), got[str(name)], exp)
        for i in range(2):
            assert rel_err(model.fake_B[i].detach()[:, :, ::2, ::2], z[f'step{s}/fake_B_{i + 1}']) < (tol if s == 0 else 2e-2)
        assert rel_err(model.fake_seg.detach()[:, :, ::2, ::2], z[f'step{s}/fake_B_S']) < (tol if s == 0 else 2e-2)
        for n in z['model_names']:
            sd = model.nets[str(n)]
            flat = torch.cat([v.detach().reshape(-1).float() for v in sd.values() if v.is_floating_point()])
            ok, msg = digest_close(flat, z[f'step{s}/w_digest/{n}'], 1e-3)
            assert ok, f'step {s} weights of {n}: {msg}'


def test_cyclegan_two_step_trajectory_with_image_pool_draws():
    z = Z.cyc_fixture()
    model = Z.cyc_oracle(z)
    A, Bs = Z.cyc_inputs(z)
    names = [str(n) for n in z['loss_names']]
    random.seed(int(z['meta'][10]))
    for s in range(int(z['meta'][7])):
        model.set_input({'A': A, 'Bs': Bs})
        model.optimize_parameters()
        got = model.current_losses()
        tol = 2e-4 if s == 0 else 3e-3
        for name, exp in zip(names, z[f'step{s}/losses']):
            assert abs(got[name] - exp) <= tol * max(1.0, abs(exp)), (s, name, got[name], exp)
        for fam in ('fake_Bs', 'rec_As', 'fake_As', 'rec_Bs'):
            for i in range(2):
                # step 0 is a pure function of the inputs; step 1 inherits an Adam update whose noise-level gradients take +-lr steps of
                # rounding-determined sign (test_oracle_golden.test_two_step_trajectory).  (A third step was tried: rec_* images, which pass through two
                # twice-updated generators, then differ by up to 1.2e-1 -- no longer a useful check -- so the fixture stops after the step whose
                # discriminator losses depend on the pool draws.)
                assert rel_err(getattr(model, fam)[i].detach()[:, :, ::2, ::2], z[f'step{s}/{fam}_{i + 1}']) < (tol if s == 0 else 2e-2), (s, fam, i)
        for n in z['model_names']:
            sd = model.nets[str(n)]
            flat = torch.cat([v.detach().reshape(-1).float() for v in sd.values() if v.is_floating_point()])
            ok, msg = digest_close(flat, z[f'step{s}/w_digest/{n}'], 1e-3)
            assert ok, f'step {s} weights of {n}: {msg}'
    assert random.random() == float(z['random_after'][0]), 'the image pools must have consumed exactly the reference\'s draws'


def test_non_default_cli_options_two_step_trajectory():
    """--upsample resize_conv, --net-d pixel, --gan-mode wgangp (cli.py:103, 176-182) in one DeepLIIF trajectory of the reference"""
    z = Z.opt_fixture()
    model = Z.opt_oracle(z)
    A, B = Z.opt_inputs(z)
    for s in range(int(z['meta'][7])):
        model.set_input({'A': A, 'B': B})
        model.optimize_parameters()
        got = model.current_losses()
        tol = 2e-4 if s == 0 else 3e-3
        for name, exp in zip(z['loss_names'], z[f'step{s}/losses']):
            assert abs(got[str(name)] - exp) <= tol * max(1.0, abs(exp)), (s, str(name), got[str(name)], exp)
        assert rel_err(model.fake_B[0].detach()[:, :, ::2, ::2], z[f'step{s}/fake_B_1']) < (tol if s == 0 else 2e-2)
        for n in z['model_names']:
            flat = torch.cat([v.detach().reshape(-1).float() for v in model.nets[str(n)].values() if v.is_floating_point()])
            ok, msg = digest_close(flat, z[f'step{s}/w_digest/{n}'], 1e-3)
            assert ok, f'step {s} weights of {n}: {msg}'
