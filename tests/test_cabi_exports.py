"""The C-ABI shared library must load and export every symbol include/deepliif_hip.h declares (no compute calls without a
GPU), and the ctypes structure mirrors must have the C structs' sizes."""
import ctypes
import os
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
    assert lib.dl_version() == 100
    assert isinstance(lib.dl_last_error(), bytes)


def test_ctypes_struct_sizes_match_the_header(tmp_path):
    c = tmp_path / 'sz.c'
    c.write_text('#include "%s"\n#include <stdio.h>\nint main(){printf("%%zu %%zu %%zu %%zu\\n", sizeof(dl_conv_desc), sizeof(dl_wgrad_desc), '
                 'sizeof(dl_pack_desc), sizeof(dl_norm_desc));return 0;}\n' % HEADER)
    exe = tmp_path / 'sz'
    subprocess.run(['gcc', str(c), '-o', str(exe)], check=True)
    sizes = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert sizes == [ctypes.sizeof(L.ConvDesc), ctypes.sizeof(L.WgradDesc), ctypes.sizeof(L.PackDesc), ctypes.sizeof(L.NormDesc)]


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
