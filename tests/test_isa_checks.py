"""Contracts the compiler cannot see, checked on the code it generates (hipcc cross-compiles without a GPU).

wgrad_w4_kernel (csrc/wgrad_w4.h) issues its transposing LDS reads as inline asm (`ds_read_b64_tr_b16`: behind the builtin hipcc inserted a vmcnt(0) that
serialised the kernel's DMA pipeline) and waits for them with its own `s_waitcnt lgkmcnt(0)`.  The compiler believes an inline-asm output is valid
as soon as the statement has executed, so nothing stops it from copying or re-using a destination register before the data has landed.  The kernel
is written so that it has no reason to; this test makes that a checked property of every build: between a transposing read and the next `s_waitcnt lgkmcnt(0)`
no instruction may read or write the read's destination registers -- not inside the row loop (fragment double buffer) and not after it (the prefetch of the
last sub-step must land before the epilogue re-uses the registers)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'deepliif_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def _vregs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which('hipcc')), reason='hipcc not available')
@pytest.mark.parametrize('flags', [[], ['-DDL_H16_FP16']], ids=['bf16', 'fp16'])
def test_wgrad_w4_fragment_registers_are_untouched_until_the_wait(tmp_path, flags):
    asm = tmp_path / 'wgrad.s'
    subprocess.run([HIPCC if os.path.exists(HIPCC) else 'hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', *flags,
                    os.path.join(CSRC, 'wgrad.hip'), '-o', str(asm)], check=True, capture_output=True, timeout=600)
    lines = asm.read_text().split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\d+wgrad_w4_kernel.*:', l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
    pending, reads, violations = set(), 0, []
    for ln in lines[start:end]:
        s = ln.split(';')[0].strip()
        if not s or s.startswith('.') or s.endswith(':'):
            continue
        op = s.split()[0]
        args = [a.strip().split()[0] for a in s[len(op):].split(',') if a.strip()]
        if op == 'ds_read_b64_tr_b16':
            pending |= _vregs(args[0])
            reads += 1
        elif op == 's_waitcnt' and 'lgkmcnt(0)' in s:
            pending = set()
        elif pending and set().union(*[_vregs(a) for a in args] or [set()]) & pending:
            violations.append(s)
    assert reads >= 100, f'only {reads} transposing reads found: the kernel no longer uses the inline-asm reads?'
    assert not violations, f'{len(violations)} instruction(s) touch a fragment register before its read was waited for, e.g. {violations[:4]}'
