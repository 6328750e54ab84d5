"""The BENCHED training step under the oracle at full size: DeepLIIF (modalities_no 5, no segmentation generators) = 5 x resnet_9blocks
(ngf 64) + 5 x n_layers PatchGAN (ndf 64) on 512 x 512 tiles -- the configuration BASELINE.json configs[2] names and bench.py times -- one
optimize_parameters() (DeepLIIF_model.py:431-467) on the GPU against oracle.OracleDeepLIIF on the CPU, same seeded weights and tiles:

  * strict policy (the parity product), batch 1, norm instance / batch: all 20 step-0 losses within 1e-3 (north_star), and per NETWORK the digests of
    its whole gradient: the L2 norm within 1e-3 of the oracle's autograd (measured <= 1.7e-4); a fixed random projection and the element-wise
    relative L2 distance within 2e-2 (measured 1e-3 ... 8e-3: both measure the DIRECTION error |g - g_o| / |g_o|, which at this depth is set by ReLU-mask
    flips -- the oracle's own gradient moves that much under fp32-sized rounding noise, measured in the last test of this file);
  * the same at batch 2 with everything the shipped defaults switch on (three branch streams, deferred slab reduction, batched weight gradient): the
    stream join, the slab arena and the per-network batches sit INSIDE the oracle comparison;
  * one resnet_9blocks (ngf 64, 1 x 3 x 512 x 512): dL/dx and the weight gradient against the oracle's autograd for both policies -- strict bounded
    like the fixtures (test_gpu_networks.py), the bf16 policy's distance recorded.
Only full-size shapes reach wgrad_w4_kernel / conv_gemm_w4_kernel / the 7x7 patch kernels / the fused stride-2 tiles; this file is what holds their
backward end to end.  Errors go to gpurun_out/parity_errors_fullsize_step.json (copied to profiles/parity_errors_r05.json)."""
import argparse
import json
import os

import pytest
import torch

import bench
from deepliif_amd import engine as E
from deepliif_amd import models as M
from deepliif_amd import networks as N
from deepliif_amd import ops
from golden_util import seeded_uniform
from oracle import deepliif_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'
ERRLOG = {}


@pytest.fixture(autouse=True)
def _log():
    ops._impl = None
    yield
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/parity_errors_fullsize_step.json', 'w') as f:
        json.dump(ERRLOG, f, indent=1, sort_keys=True)


def _step_pair(norm, batch, precision='fp32'):
    args = argparse.Namespace(ngf=64, norm=norm, precision=precision, batch=batch, size=512)
    torch.manual_seed(0)
    opt = bench.make_opt(args, 0)
    model = M.create_model(opt)
    model.setup(opt)
    cfg = O.OracleConfig(modalities_no=5, seg_gen=False, norm=norm, padding='zero', ngf=64, ndf=64)
    nets = {k: {kk: v.detach().cpu().clone() for kk, v in getattr(model, 'net' + k).state_dict().items()} for k in model.model_names}
    om = O.OracleDeepLIIF(cfg, nets)
    A = seeded_uniform((batch, 3, 512, 512), 1234)
    B = [seeded_uniform((batch, 3, 512, 512), 1235 + i) for i in range(5)]
    model.set_input({'A': A.to(DEV), 'B': [b.to(DEV) for b in B], 'A_paths': ['fullsize']})
    model.optimize_parameters()
    torch.cuda.synchronize()
    om.set_input({'A': A, 'B': B})
    om.optimize_parameters()
    return model, om


GRAD_TOL = 1e-3            # L2 norm of a network's whole gradient (measured <= 1.7e-4, profiles/parity_errors_r05.json)
DIR_TOL = 2e-2             # its direction: element-wise relative L2 distance and a random projection (both measure |g - g_oracle| / |g|; measured 1.2e-3 ... 8.3e-3 --
                           # the size of the oracle's own response to fp32 rounding noise at this depth, see test_full_size_resnet_backward_against_the_oracle)


def _check_step(model, om, tag, tol=1e-3):
    got, exp = model.get_current_losses(), om.current_losses()
    assert set(got) == set(exp) and len(exp) == 20
    worst = 0.0
    for k in exp:
        e = abs(got[k] - exp[k]) / max(1.0, abs(exp[k]))
        ERRLOG[f'{tag}/loss/{k}'] = e
        worst = max(worst, e)
    assert worst <= tol, (tag, worst, got, exp)
    # gradients per network: the oracle's tuples follow its parameter lists (state_dict order of every network, running statistics left out)
    og = {}
    it_g, it_d = iter(om.last_grads_g), iter(om.last_grads_d)
    for names, it in ((om.g_names + om.gs_names, it_g), (om.d_names + om.ds_names, it_d)):
        for n in names:
            for k, v in om.nets[n].items():
                if v.is_floating_point() and not k.endswith(('running_mean', 'running_var')):
                    og[(n, k)] = next(it)
    for name in model.model_names:
        named = dict(getattr(model, 'net' + name).named_parameters())
        keys = [k for (n, k) in og if n == name]
        assert sorted(keys) == sorted(named), name
        ge = torch.cat([named[k].grad.detach().reshape(-1).double().cpu() for k in keys])
        go = torch.cat([og[(name, k)].reshape(-1).double() for k in keys])
        r = torch.randn(go.numel(), generator=torch.Generator().manual_seed(99), dtype=torch.float64)
        e_norm = float(abs(ge.norm() - go.norm()) / go.norm())
        e_proj = float(abs(((ge - go) * r).sum()) / go.norm())          # a unit-variance random direction: the projection of g itself is ~ |g|
        e_l2 = float((ge - go).norm() / go.norm())
        ERRLOG[f'{tag}/grad/{name}/norm'], ERRLOG[f'{tag}/grad/{name}/projection'], ERRLOG[f'{tag}/grad/{name}/l2'] = e_norm, e_proj, e_l2
        assert e_norm <= GRAD_TOL, (tag, name, e_norm, e_proj, e_l2)
        assert e_proj <= DIR_TOL and e_l2 <= DIR_TOL, (tag, name, e_norm, e_proj, e_l2)


@pytest.mark.parametrize('norm', ['instance', 'batch'])
def test_benched_step_strict_batch1_against_the_oracle(norm):
    model, om = _step_pair(norm, 1)
    _check_step(model, om, f'step/strict/b1/{norm}')


def test_benched_step_strict_batch2_shipped_defaults_against_the_oracle():
    assert M._N_STREAMS >= 1 and ops._WGRAD_DEFER                      # whatever the environment ships: recorded with the result
    model, om = _step_pair('instance', 2)
    ERRLOG['step/strict/b2/config'] = {'streams': M._N_STREAMS, 'branch_streams_used': model._branch_streams() is not None,
                                       'wgrad_defer': ops._WGRAD_DEFER, 'wgrad_batch': ops._WGRAD_BATCH}
    _check_step(model, om, 'step/strict/b2/instance')


def _host_memory_limit_gb():
    """memory this process may use: the cgroup limit when there is one, else the machine's available memory"""
    lim = None
    for f in ('/sys/fs/cgroup/memory.max', '/sys/fs/cgroup/memory/memory.limit_in_bytes'):
        try:
            v = open(f).read().strip()
            if v.isdigit():
                lim = int(v) / 1e9
                break
        except OSError:
            pass
    try:
        import psutil
        avail = psutil.virtual_memory().available / 1e9
    except Exception:
        avail = lim or 0.0
    return min(lim, avail) if lim else avail


def test_benched_step_strict_batch8_against_the_oracle():
    """VERDICT r5 #3a / weak #7: the batch bench.py times (8 tiles per GPU: 1 024 image rows per layer through wgrad_w4_kernel's row ranges, 18 x 67 MB operands per
    network queue, conv_s2d_kernel's 256 strips) under the oracle with the shipped defaults, same bounds as batch 1 / 2.  The oracle's autograd holds the graphs of
    all five generators: ~13 GB per tile measured (batch 1) -> ~105 GB at batch 8; the GPU boxes allow 322 GB.  Skipped (not failed) where the host has less."""
    need = 150.0
    have = _host_memory_limit_gb()
    if have < need:
        pytest.skip(f'the CPU oracle at batch 8 needs ~105 GB of host memory (limit here: {have:.0f} GB)')
    model, om = _step_pair('instance', 8)
    ERRLOG['step/strict/b8/config'] = {'streams': M._N_STREAMS, 'branch_streams_used': model._branch_streams() is not None,
                                       'wgrad_defer': ops._WGRAD_DEFER, 'wgrad_batch': ops._WGRAD_BATCH}
    _check_step(model, om, 'step/strict/b8/instance')


def l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def test_full_size_resnet_backward_against_the_oracle():
    sd = O.random_state_dict('resnet_9blocks', 3, 3, 64, 'instance', 'zero', 4, generator=torch.Generator().manual_seed(51))
    x = seeded_uniform((1, 3, 512, 512), 52)
    sdo = {k: v.clone() for k, v in sd.items()}
    params = {k: v.requires_grad_(True) for k, v in sdo.items() if v.is_floating_point() and 'running' not in k}
    xo = x.clone().requires_grad_(True)
    yo = O.run_generator('resnet_9blocks', sdo, xo, 'instance', 'zero')
    r = torch.randn(yo.shape, generator=torch.Generator().manual_seed(53))
    grads = torch.autograd.grad((yo * r).sum(), [xo] + list(params.values()))
    dw_o = torch.cat([g.reshape(-1) for g in grads[1:]])
    net = N.define_G(3, 3, 64, 'resnet_9blocks', 'instance', False, 'normal', 0.02, [0], 'zero')
    net.load_state_dict(sd, strict=True)
    net.train()
    # the oracle's OWN sensitivity to rounding noise of the strict policy's size on every conv output (test_gpu_networks.conv_noise): at this depth the
    # gradient moves by ReLU-mask flips, not by the arithmetic -- the engine is held to 4 x the worst of two noise draws (floor 1e-3), as at fixture size
    from test_gpu_networks import GRAD_FLOOR, LAYER_NOISE, conv_noise
    s_dx = s_dw = 0.0
    for seed in (1, 2):
        sdn = {k: v.clone() for k, v in sd.items()}
        pn = {k: v.requires_grad_(True) for k, v in sdn.items() if v.is_floating_point() and 'running' not in k}
        xn = x.clone().requires_grad_(True)
        with conv_noise(LAYER_NOISE['fp32'], seed):
            yn = O.run_generator('resnet_9blocks', sdn, xn, 'instance', 'zero')
        gn = torch.autograd.grad((yn * r).sum(), [xn] + list(pn.values()))
        s_dx = max(s_dx, l2(gn[0], grads[0]))
        s_dw = max(s_dw, l2(torch.cat([g.reshape(-1) for g in gn[1:]]), dw_o))
    ERRLOG['backward/resnet_9blocks-64-512/oracle_sensitivity_dx_l2'], ERRLOG['backward/resnet_9blocks-64-512/oracle_sensitivity_dw_l2'] = s_dx, s_dw
    for precname, bound in (('fp32', True), ('bf16', None)):
        prec = E.Precision.get(precname)
        tape = E.Tape()
        ctx = E.Ctx(prec, tape, training=True)
        xa = E.to_engine(x.to(DEV), prec)
        xa.needs_grad = True
        for p in net.parameters():
            p.grad = torch.zeros_like(p)
        ya = net.run(ctx, xa)
        ya.grad = E.to_engine(r.to(DEV), prec).t
        tape.backward()
        torch.cuda.synchronize()
        dx = E.from_engine(E.Act(xa.grad, xa.C))
        named = dict(net.named_parameters())
        dw = torch.cat([named[k].grad.reshape(-1).cpu() for k in params])
        e_dx, e_dw = l2(dx, grads[0]), l2(dw, dw_o)
        ERRLOG[f'backward/resnet_9blocks-64-512/{precname}/dx_l2'], ERRLOG[f'backward/resnet_9blocks-64-512/{precname}/dw_l2'] = e_dx, e_dw
        if bound is not None:
            assert e_dx <= max(GRAD_FLOOR['fp32'], 4 * s_dx) and e_dw <= max(GRAD_FLOOR['fp32'], 4 * s_dw), (precname, e_dx, e_dw, s_dx, s_dw)
        else:
            assert e_dx < 0.5 and e_dw < 0.5, (precname, e_dx, e_dw)            # sanity only: the bf16 policy is not the parity product
