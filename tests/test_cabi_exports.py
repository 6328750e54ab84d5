"""The C-ABI shared library must load and export every symbol include/deepliif_hip.h declares (no compute calls without a
GPU), and the ctypes structure mirrors must have the C structs' sizes."""
import ctypes
import os

import pytest
import re
import subprocess

from deepliif_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'deepliif_hip.h')


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(dl_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    names = declared_symbols()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), f'{n} is declared in include/deepliif_hip.h but not exported by libdeepliif_hip.so'
        assert n in L.SIGNATURES, f'{n} has no ctypes signature in deepliif_amd/_lib.py'
    assert lib.dl_version() == L.DL_VERSION
    assert isinstance(lib.dl_last_error(), bytes)


def test_ctypes_struct_sizes_match_the_header(tmp_path):
    c = tmp_path / 'sz.c'
    c.write_text('#include "%s"\n#include <stdio.h>\nint main(){printf("%%zu %%zu %%zu %%zu %%zu\\n", sizeof(dl_conv_desc), sizeof(dl_wgrad_desc), '
                 'sizeof(dl_pack_desc), sizeof(dl_norm_desc), sizeof(dl_wgrad_reduce_entry));return 0;}\n' % HEADER)
    exe = tmp_path / 'sz'
    subprocess.run(['gcc', str(c), '-o', str(exe)], check=True)
    sizes = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert sizes == [ctypes.sizeof(L.ConvDesc), ctypes.sizeof(L.WgradDesc), ctypes.sizeof(L.PackDesc), ctypes.sizeof(L.NormDesc), ctypes.sizeof(L.WgradReduceEntry)]


def test_product_has_no_cpu_fallback():
    """ops.HipBackend refuses CPU tensors; the package never imports the oracle."""
    import torch
    import pytest
    from deepliif_amd import ops
    ops._impl = None
    be = ops.impl()
    x = torch.zeros(1, 4, 4, 8)
    with pytest.raises(L.HipLibraryError):
        be.act_forward(L.ACT_RELU, x, x.clone())
    pkg = os.path.join(ROOT, 'deepliif_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            assert 'oracle' not in open(os.path.join(pkg, fn)).read().replace('the oracle', ''), fn


def test_empty_problems_are_rejected_before_any_launch():
    """N = 0 (or a zero-sized image) is an error with a message that says so -- decided on the host before the device is touched,
    so this runs without a GPU (the pointers below are never dereferenced)."""
    import ctypes as C
    from deepliif_amd import _lib as L
    lib = L.load()
    dummy = C.c_void_p(0x1000)
    d = L.ConvDesc()
    d.N, d.Hi, d.Wi, d.Ci, d.in_pstride = 0, 32, 32, 8, 8
    d.Ho, d.Wo, d.Co, d.out_pstride, d.Hq, d.Wq = 32, 32, 8, 8, 32, 32
    d.n_phase, d.splitk = 1, 1
    assert lib.dl_conv_forward(C.byref(d), dummy, dummy, None, None, dummy, None, None, None) != 0
    assert b'empty problem' in lib.dl_last_error()
    with pytest.raises(L.HipLibraryError, match='empty problem'):
        L.check(lib.dl_conv_forward(C.byref(d), None, dummy, None, None, None, None, None, None), 'dl_conv_forward')      # as torch hands over an empty tensor
    w = L.WgradDesc()
    w.N, w.Hp, w.Wp, w.CAp, w.Hq, w.Wq, w.CBp, w.splitk = 2, 0, 16, 8, 16, 16, 8, 1
    assert lib.dl_conv_wgrad(C.byref(w), dummy, dummy, dummy, dummy, None) != 0 and b'empty problem' in lib.dl_last_error()
    n = L.NormDesc()
    n.N, n.H, n.W, n.Cp, n.C = 0, 8, 8, 8, 8
    assert lib.dl_norm_forward(C.byref(n), dummy, None, None, None, None, dummy, dummy, dummy, dummy, None, dummy, dummy, None, None) != 0
    assert b'empty problem' in lib.dl_last_error()
