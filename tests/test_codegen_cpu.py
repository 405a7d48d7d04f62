"""Code-generation guards (CPU, cross-compile only): properties of the emitted gfx950 ISA that the
measured performance depends on and that hipcc silently broke once."""
import os
import re
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, 'few-shot-vit_amd', 'csrc')


@pytest.fixture(scope='module')
def v2_asm(tmp_path_factory):
    d = tmp_path_factory.mktemp('isa')
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', CSRC, '-c',
                    os.path.join(CSRC, 'conv_gemm_v2.hip'), '-o', str(d / 'v2.o'), '-save-temps'],
                   check=True, cwd=str(d), stderr=subprocess.DEVNULL)
    return open(d / 'conv_gemm_v2-hip-amdgcn-amd-amdhsa-gfx950.s').read()


def _kernel(asm, prefix):
    m = re.search(r'^(' + re.escape(prefix) + r'\w*):', asm, re.M)
    assert m, prefix
    a = m.start()
    return asm[a:asm.index('.Lfunc_end', a)].split('\n')


@pytest.mark.parametrize('sym', ['_ZN5fsvit19conv_gemm_v2_kernelIDF16bLi128ELi128ELi2ELi2ELi2E',     # bf16 128x128
                                 '_ZN5fsvit19conv_gemm_v2_kernelIfLi128ELi128ELi2ELi2ELi2E'])
def test_inner_k_loop_does_not_drain_lds_dma(v2_asm, sym):
    body = _kernel(v2_asm, sym)
    mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
    assert mf, 'no MFMA emitted'
    j = mf[0]
    while 'global_load_lds_dwordx4' not in body[j]:
        j -= 1
        assert j > 0
    between = body[j:mf[0]]
    assert not any('s_waitcnt' in l and 'vmcnt' in l for l in between), \
        'hipcc placed a vmcnt wait between the LDS-DMA issue and the first MFMA of the k-step'
    assert any('ds_read_b128' in l for l in between)
    # the DMA is 16 bytes per lane and nothing spills
    assert not any('global_load_lds_dword ' in l for l in body)
    m = re.search(r'\.name:\s+' + re.escape(sym) + r'\w*\n.*?\.vgpr_spill_count:\s+(\d+)', v2_asm, re.S)
    assert m and int(m.group(1)) == 0
