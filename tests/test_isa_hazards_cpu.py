"""Static ISA check of the kernels that issue MFMAs from inline asm (hipcc pads nothing around those): no VALU / v_accvgpr_write feeding an MFMA operand within
two wait states, no non-MFMA access to an MFMA's destination before its passes are over (tools/check_mfma_hazard.py; cdna_hip_programming.md 5.7 item 2).
Found in round 6: hipcc copied a VGPR quad to an AGPR directly in front of an asm MFMA (NaNs in a prototype), and placed the broadcast of a bias into an
accumulator one state before the MFMA reading it in two shipped kernels."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, 'few-shot-vit_amd', 'csrc', 'build')
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'


@pytest.mark.parametrize('obj', ['mlp_rows', 'mlp_rows.f16', 'stage1_w4', 'stage1_w4.f16', 'qkv_attn', 'stage1_ring', 'conv3x3_halo', 'gemm256', 'wgrad3x3'])
def test_no_unpadded_mfma_hazard_in_the_built_objects(obj):
    path = os.path.join(BUILD, obj + '.o')
    if not os.path.exists(path) or not os.path.exists(OBJDUMP):
        pytest.skip('no build objects / llvm tools here')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_mfma_hazard.py'), path, '.'], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert 'MFMAs:' in r.stdout
