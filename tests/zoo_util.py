"""Shared by the oracle, host (emulated backend) and GPU tests of the remaining model zoo (DeepLIIFKD, CycleGAN): rebuild the seeded
networks of tests/golden/step_kd_m2.npz / step_cyclegan_m2.npz (tests/golden/make_golden_zoo.py) and the inputs."""
import os

import numpy as np
import torch

from golden_util import seeded_uniform
from oracle import deepliif_oracle as O
import seam_util

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def kd_fixture():
    return np.load(os.path.join(G, 'step_kd_m2.npz'))


def kd_student_state_dicts(z):
    """{fixture network name: reference-keyed state_dict} drawn from the stored seeds (make_golden.seed_model_nets' rules)"""
    _, norm, padding, net_gs, _, nf, _, _, _ = [str(x) for x in z['meta']]
    nets = {}
    for name, seed in zip(z['model_names'], z['net_seeds']):
        name = str(name)
        if name.startswith('D'):
            arch, pad, cin = 'n_layers', 'zero', 6
        elif len(name) == 2:
            arch, pad, cin = 'resnet_9blocks', padding, 3
        else:
            arch, pad, cin = net_gs, 'reflect', 3
        nets[name] = O.random_state_dict(arch, cin, 3, int(nf), norm, pad, 4, generator=torch.Generator().manual_seed(int(seed)))
    return nets


def kd_teacher_state_dicts():
    return {name: O.random_state_dict(arch, 3, 3, 64, 'batch', pad, 4, generator=torch.Generator().manual_seed(800 + j))
            for j, (name, arch, pad) in enumerate(seam_util.KD_TEACHER_NETS)}


def kd_inputs(z):
    size, batch = int(z['meta'][4]), int(z['meta'][6])
    A = seeded_uniform((batch, 3, size, size), 42)
    B = [seeded_uniform((batch, 3, size, size), 43 + i) for i in range(3)]
    return A, B


def kd_oracle(z):
    nf = int(z['meta'][5])
    cfg = O.OracleConfig(modalities_no=2, seg_gen=True, net_g='resnet_9blocks', net_gs=str(z['meta'][3]), norm=str(z['meta'][1]), padding=str(z['meta'][2]),
                         ngf=nf, ndf=nf)
    tcfg = O.OracleConfig(modalities_no=2, seg_gen=True, net_g='resnet_9blocks', net_gs='unet_64', norm='batch', padding='zero', ngf=64, ndf=64)
    return O.OracleDeepLIIFKD(cfg, kd_student_state_dicts(z), tcfg, kd_teacher_state_dicts())


def cyc_fixture():
    return np.load(os.path.join(G, 'step_cyclegan_m2.npz'))


def cyc_state_dicts(z):
    nf = int(z['meta'][5])
    nets = {}
    for name, seed in zip(z['model_names'], z['net_seeds']):
        name = str(name)
        arch = 'n_layers' if name.startswith('D') else str(z['meta'][3])
        nets[name] = O.random_state_dict(arch, 3, 3, nf, str(z['meta'][1]), str(z['meta'][2]), 4, generator=torch.Generator().manual_seed(int(seed)))
    return nets


def cyc_inputs(z):
    size, batch = int(z['meta'][4]), int(z['meta'][6])
    return seeded_uniform((batch, 3, size, size), 52), [seeded_uniform((batch, 3, size, size), 53 + i) for i in range(2)]


def cyc_oracle(z):
    nf = int(z['meta'][5])
    cfg = O.OracleConfig(modalities_no=2, seg_gen=False, net_g=str(z['meta'][3]), norm=str(z['meta'][1]), padding=str(z['meta'][2]), ngf=nf, ndf=nf)
    return O.OracleCycleGAN(cfg, cyc_state_dicts(z), pool_size=int(z['meta'][8]), gan_mode=str(z['meta'][9]))


def opt_fixture():
    return np.load(os.path.join(G, 'step_options_m1.npz'))


def opt_state_dicts(z):
    nf = int(z['meta'][5])
    nets = {}
    for name, seed in zip(z['model_names'], z['net_seeds']):
        name = str(name)
        arch, cin = (str(z['meta'][9]), 6) if name.startswith('D') else (f"{z['meta'][3]}:{z['meta'][8]}", 3)
        nets[name] = O.random_state_dict(arch, cin, 3, nf, str(z['meta'][1]), str(z['meta'][2]), 4, generator=torch.Generator().manual_seed(int(seed)))
    return nets


def opt_inputs(z):
    size, batch = int(z['meta'][4]), int(z['meta'][6])
    return seeded_uniform((batch, 3, size, size), 62), [seeded_uniform((batch, 3, size, size), 63)]


def opt_oracle(z):
    nf = int(z['meta'][5])
    cfg = O.OracleConfig(modalities_no=1, seg_gen=False, net_g=str(z['meta'][3]), norm=str(z['meta'][1]), padding=str(z['meta'][2]), ngf=nf, ndf=nf,
                         gan_mode=str(z['meta'][10]), net_d=str(z['meta'][9]), upsample=str(z['meta'][8]))
    return O.OracleDeepLIIF(cfg, opt_state_dicts(z))
