"""The row-wise kernels of the end of round 2 (mlp_rows at C = 384, ln_gemm_rows, qkv_attn_rows, patch_embed2 on the rows kernel) replaced launches
that stay in the library as the general path (other widths, more than 32 tokens per image at stage 3, FSVIT_* switches).  The switches are read
once per process, so the general path runs in a child process; both must agree with each other within the bf16 mode's own noise."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from fewshot_vit_amd import models, synthetic
name, img, B, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
m = models.make(name, numerics='bf16')
shapes = {'encoder.' + k: tuple(v.shape) for k, v in m.state_dict().items()}
sd = synthetic.synthetic_checkpoint_sd(shapes) if name.startswith('visformer') else synthetic.procedural_state_dict(shapes)
m.load_state_dict({k[len('encoder.'):]: v for k, v in sd.items()}, strict=True)
m = m.cuda().eval()
x = torch.randn(B, 3, img, img, generator=torch.Generator().manual_seed(5))
with torch.no_grad():
    f = m(x.cuda()).float().cpu()
torch.save(f, out)
''' % ROOT


def _run(tmp_path, name, img, B, env_extra, tag):
    out = str(tmp_path / f'{tag}.pt')
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, '-c', CHILD, name, str(img), str(B), out], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    return torch.load(out)


@pytest.mark.parametrize('name,img,B,tol', [('visformer_micro_80', 80, 6, 0.015), ('deit_small_patch16_224', 224, 3, 0.04)])
def test_general_path_agrees_with_row_kernels(tmp_path, name, img, B, tol):
    fused = _run(tmp_path, name, img, B, {}, 'fused')
    general = _run(tmp_path, name, img, B, {'FSVIT_LN_GEMM_ROWS': '0', 'FSVIT_QKV_ATTN_ROWS': '0', 'FSVIT_MLP_ROWS': '7'}, 'general')
    assert torch.isfinite(fused).all() and torch.isfinite(general).all()
    d = (fused - general).abs().max().item() / max(1.0, float(general.abs().max()))
    print(f'{name}: max rel |fused - general| = {d:.3e}')
    assert 0.0 < d <= tol        # different kernels (not bit-identical), same mathematics
