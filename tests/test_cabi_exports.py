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


@pytest.mark.parametrize('half', ['bf16', 'fp16'])
def test_library_exports_every_declared_symbol(half):
    """both builds of the sources: libdeepliif_hip.so (bfloat16) and libdeepliif_hip_f16.so (IEEE half, the fp16 inference policy) -- one ABI"""
    lib = L.load(half)
    names = declared_symbols()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), f'{n} is declared in include/deepliif_hip.h but not exported by the {half} library'
        assert n in L.SIGNATURES, f'{n} has no ctypes signature in deepliif_amd/_lib.py'
    assert lib.dl_version() == L.DL_VERSION
    assert lib.dl_half_format() == (L.HALF_FP16 if half == 'fp16' else L.HALF_BF16)
    assert isinstance(lib.dl_last_error(), bytes)


def test_a_library_of_the_wrong_format_is_refused(monkeypatch):
    monkeypatch.setattr(L, 'LIB_PATH_F16', L.LIB_PATH)
    monkeypatch.setattr(L, '_libs', {})
    with pytest.raises(L.HipLibraryError, match='other 16-bit format'):
        L.load('fp16')
    with pytest.raises(ValueError):
        L.load('fp8')


@pytest.mark.parametrize('half', ['bf16', 'fp16'])
def test_shipped_library_has_no_result_changing_switches(half):
    """the default build carries no timing-only ablation (results wrong by construction) and no getenv on a launch path -- the only
    environment variables it knows are the nine documented A/B switches of include/deepliif_hip.h, copied once at load time."""
    lib = L.load(half)
    if os.environ.get('DEEPLIIF_AMD_LIB') or os.environ.get('DEEPLIIF_AMD_LIB_F16'):
        pytest.skip('a non-default library was selected')
    assert lib.dl_dev_build() == 0
    names = [lib.dl_switch_name(i).decode() for i in range(lib.dl_switch_count())]
    assert len(names) == 9 and len(set(names)) == 9
    header = open(HEADER).read()
    for n in names:
        assert n in header, f'{n} is not documented in include/deepliif_hip.h'
    blob = open(L.LIB_PATH if half == 'bf16' else L.LIB_PATH_F16, 'rb').read()
    found = sorted(set(m.decode() for m in re.findall(rb'DL_[A-Z0-9_]{3,}', blob)))
    assert not [f for f in found if 'ABL' in f], found           # DL_CONV_ABLATE, DL_W4_ABLATE, DL_C4_ABL ... are gone
    assert sorted(f for f in found if f in names) == sorted(names)
    extra = [f for f in found if f not in names]
    assert not extra, f'undocumented switch strings in the shipped library: {extra}'
    # every getenv in the sources goes through the load-time table (error.cpp) or the dev-only macro
    csrc = os.path.join(ROOT, 'deepliif_amd', 'csrc')
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith(('.hip', '.h', '.cpp')):
            continue
        for i, line in enumerate(open(os.path.join(csrc, fn)), 1):
            code = line.split('//')[0]
            if 'getenv(' in code and fn != 'error.cpp' and '#define DL_DEV_ENV' not in code:
                raise AssertionError(f'{fn}:{i}: getenv outside the switch table: {line.strip()}')


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
