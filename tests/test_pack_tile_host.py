"""Host check of the tiled weight packing (csrc/pack_tile.h): the two device functions of the tiled form are run thread by thread on the CPU
(csrc/pack_tile_check.cpp, compiled host-only) for the pack descriptors of every conv shape the networks have, and every bit of the hi / lo images is
compared with the element-wise decode.  The GPU twin is test_gpu_kernels.py::test_pack_weights_batch_matches_single_packs."""
import ctypes as C
import os
import shutil
import subprocess

import pytest

from deepliif_amd import _lib as L
from deepliif_amd.geometry import ConvSpec, fill_pack_desc

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'deepliif_amd', 'csrc')
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'

# (spec, how many of its images must take the tiled form)
SPECS = [
    (ConvSpec('conv', 3, 64, 7, 1, 3), 0),                          # stem: 3 channels
    (ConvSpec('conv', 64, 128, 3, 2, 1), 2),                        # Resnet-9 downs
    (ConvSpec('conv', 128, 256, 3, 2, 1), 2),
    (ConvSpec('conv', 256, 256, 3, 1, 1), 2),                       # ResnetBlock
    (ConvSpec('convT', 256, 128, 3, 2, 1, L.PAD_ZERO, 1), 2),       # ups: 4 phases of 1 / 2 / 2 / 4 taps
    (ConvSpec('convT', 128, 64, 3, 2, 1, L.PAD_ZERO, 1), 2),
    (ConvSpec('conv', 64, 3, 7, 1, 3), 0),                          # head: 7x7
    (ConvSpec('conv', 6, 64, 4, 2, 1), None),                       # PatchGAN input layer (6 channels one way, 64 the other)
    (ConvSpec('conv', 64, 128, 4, 2, 1), 2),
    (ConvSpec('conv', 256, 512, 4, 1, 1), 2),
    (ConvSpec('conv', 512, 1, 4, 1, 1), None),                      # PatchGAN head: one real row
    (ConvSpec('conv', 64, 128, 4, 2, 1), 2),                        # UNet-512 downs / ups (4x4, stride 2)
    (ConvSpec('conv', 512, 512, 4, 2, 1), 2),
    (ConvSpec('convT', 512, 512, 4, 2, 1), 2),
    (ConvSpec('convT', 1024, 512, 4, 2, 1), 2),
    (ConvSpec('convT', 128, 3, 4, 2, 1), None),
    (ConvSpec('conv', 256, 128, 1, 1, 0), None),                    # 1x1 (attention gates)
]


@pytest.fixture(scope='module')
def checker(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not found')
    exe = str(tmp_path_factory.mktemp('pack') / 'pack_tile_check')
    r = subprocess.run([HIPCC, '--cuda-host-only', '-O2', '-std=c++17', '-x', 'hip', os.path.join(CSRC, 'pack_tile_check.cpp'), '-o', exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def _descs():
    out = []
    for spec, want in SPECS:
        A, B = (spec.cout, spec.cin) if spec.kind == 'conv' else (spec.cin, spec.cout)
        plans = [spec.forward_plan(), spec.dgrad_plan()]
        if spec.is_narrow():
            plans.append(spec.narrow_forward_plan())
        for plan in plans:
            d = fill_pack_desc(plan, A, B, spec.k)
            d.KH = spec.k
            out.append((spec, want, d))
    return out


def test_tiled_packing_writes_the_bits_of_the_elementwise_decode(checker, tmp_path):
    descs = _descs()
    blob = tmp_path / 'descs.bin'
    blob.write_bytes(b''.join(bytes(d) for _, _, d in descs))
    r = subprocess.run([checker, str(blob)], capture_output=True, text=True, timeout=600)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0, r.stdout[-2000:]
    assert len(lines) == len(descs) + 1 and lines[-1].endswith(' 0 bad')
    tiled_of = {}
    for (spec, want, _), line in zip(descs, lines):
        assert ' bad descriptor' not in line
        tiled_of.setdefault(id(spec), [want, 0])[1] += ' tiled ' in line
    for want, got in tiled_of.values():
        assert want is None or got == want, (want, got, lines)
    assert int(lines[-1].split()[0]) >= 24          # the shapes that carry the parameters all take the tiled form


def test_pack_descriptor_layout_matches_the_header():
    assert C.sizeof(L.PackDesc) == 4 * (4 + 1 + 2 + 2 + 1 + 5 + 4) + 2 * 64 + 4 + 4


def test_block_table_of_the_shipped_library(monkeypatch):
    """dl_pack_job_fill / dl_pack_batch_blocks are host-only (nothing is launched): the table the SHIPPED library builds marks the eligible images as
    tiled -- one entry per 8-row x 64-channel tile -- and DL_PACK_TILED=0 puts every image back into chunks of 1024 x 16 bytes."""
    lib = L.load()
    descs = _descs()
    jb = int(lib.dl_pack_job_bytes())
    host = (C.c_ubyte * (jb * len(descs)))()
    base = C.addressof(host)
    for i, (_, _, d) in enumerate(descs):           # the pointers are only recorded
        L.check(lib.dl_pack_job_fill(C.byref(d), C.c_void_p(0x10000), C.c_void_p(0x20000), C.c_void_p(0x30000), C.c_void_p(base + i * jb)), 'dl_pack_job_fill')

    def table():
        n = int(lib.dl_pack_batch_blocks(C.c_void_p(base), len(descs), None))
        assert n > 0
        tab = (C.c_int32 * (2 * n))()
        assert int(lib.dl_pack_batch_blocks(C.c_void_p(base), len(descs), C.c_void_p(C.addressof(tab)))) == n
        return [(tab[2 * i], tab[2 * i + 1]) for i in range(n)]

    # the library copies its eight runtime switches out of the environment when it is loaded: a test that flips one re-reads the table
    monkeypatch.delenv('DL_PACK_TILED', raising=False)
    lib.dl_switches_reload()
    tiled = table()
    monkeypatch.setenv('DL_PACK_TILED', '0')
    lib.dl_switches_reload()
    try:
        chunk = table()
    finally:
        monkeypatch.delenv('DL_PACK_TILED', raising=False)
        lib.dl_switches_reload()
    flag = 1 << 30
    assert not any(j & flag for j, _ in chunk)
    for i, (_, _, d) in enumerate(descs):
        chunks = d.rows_pad * (d.kstride // 8)
        mine_c = [y for j, y in chunk if j == i]
        assert mine_c == list(range(0, chunks, 1024))                       # first chunk of each workgroup
        mine_t = [(j, y) for j, y in tiled if (j & ~flag) == i]
        if mine_t[0][0] & flag:
            assert d.Cc == d.Cc_pad and d.Cc_pad >= 64 and d.KH * d.KW <= 16 and not d.stack_kw
            assert [y for _, y in mine_t] == list(range((d.rows_pad // 8) * (d.Cc_pad // 64)))
        else:
            assert [y for _, y in mine_t] == mine_c
    assert sum(1 for j, _ in tiled if j & flag) > 0 and [j & ~flag for j, _ in tiled] == sorted(j & ~flag for j, _ in tiled)
