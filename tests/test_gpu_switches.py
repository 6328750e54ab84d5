"""Kernels behind environment switches, tested in child processes (the library reads its switches once per process).

The fused four-phase tile (csrc/conv_s2f.hip) below its size rule: dl_conv_forward only sends layers with >= 256 phase-grid tiles to it, so the small
kernel-test shapes (and the fused-statistics case, 2 x 16 x 16) never reach it in the main test process.  The library reads DL_CONV_S2F once per process:
the same kernel tests run again in a child process with DL_CONV_S2F=2 (size rule lifted) -- forward + data gradient against the emulated reference, fused
statistics against the stand-alone pass, run-to-run equality."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernel_tests_with_the_fused_stride2_tile_forced():
    env = dict(os.environ, DL_CONV_S2F='2')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_kernels.py'), '-m', 'gpu', '-q', '-x', '-k',
                        'bf16 and (big_tiles or fused_norm_statistics or conv_forward_and_dgrad)'], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout and 'failed' not in r.stdout, r.stdout[-1500:]


def test_the_dispatch_names_the_fused_tile_for_the_generator_layers():
    import torch
    from deepliif_amd import _lib as L, ops
    from deepliif_amd.engine import Precision
    from deepliif_amd.geometry import ConvSpec
    be = ops.impl()
    prec = Precision.get('bf16')
    # up1 (256 -> 128 at 128 x 128, batch 8): the fused four-phase tile; up2 (128 -> 64): since r06 the register-resident-weights kernel (csrc/conv_s2u.hip)
    for cin, cout, n, hw, fused, plain in ((256, 128, 8, 128, 'conv_s2f_kernel', 'conv_gemm_glds_kernel<128,128,64>'), (128, 64, 1, 256, 'conv_s2u_kernel', None)):
        spec = ConvSpec('convT', cin, cout, 3, 2, 1, L.PAD_ZERO, 1)
        w = torch.randn(cin, cout, 3, 3, device='cuda') * 0.02
        pf = ops.PackedWeights(spec.forward_plan(), 'cuda', False)
        be.pack_weights(pf, w)
        x = torch.randn(n, hw, hw, cin, device='cuda').to(prec.dtype)
        out = torch.empty(n, 2 * hw, 2 * hw, cout, device='cuda', dtype=prec.dtype)
        be.conv_forward(pf, x, out, hw, hw, None, L.ACT_NONE, L.ACT_NONE, prec.prec)
        torch.cuda.synchronize()
        if fused == 'conv_s2u_kernel':
            expect = fused if os.environ.get('DL_CONV_S2D') != '0' else ('conv_s2f_kernel' if os.environ.get('DL_CONV_S2F') != '0' else 'conv_gemm_glds_kernel<128,64,64>')
        else:
            expect = plain if os.environ.get('DL_CONV_S2F') == '0' else fused
        assert be.last_conv_kernel == expect, (cin, cout, be.last_conv_kernel, expect)


def test_strict_w4_kernel_under_its_switch():
    """conv_gemm_w4x3_kernel (csrc/conv_w4x3.hip) is opt-in (a measured tie with the 8-phase strict kernel): its parity test under DL_CONV_W4X3=1"""
    env = dict(os.environ, DL_CONV_W4X3='1')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_kernels.py'), '-m', 'gpu', '-q', '-x', '-k',
                        'strict_w4_kernel or split_copy_inputs'], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout and 'failed' not in r.stdout, r.stdout[-1500:]



def test_strict_fused_stride2_tile_below_its_size_rule():
    """conv_s2f_x3_kernel (csrc/conv_s2f_x3.hip) shares the size rule of the bf16 tile: the small stride-2 cases of the split-copy test reach it only with
    DL_CONV_S2F=2 -- forward / data gradient within 2e-6 of the 4-phase strict kernel, weight gradients untouched"""
    env = dict(os.environ, DL_CONV_S2F='2')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_kernels.py'), '-m', 'gpu', '-q', '-x', '-k',
                        'split_copy_inputs'], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout and 'failed' not in r.stdout, r.stdout[-1500:]
