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


def test_deit_unfused_attention_switch(tmp_path):
    """FSVIT_VIT_ATTN_ROWS=0: LayerNorm + qkv (`ln_gemm_rows`) and `attention_v2` as two launches instead of `vit_attn_rows`."""
    fused = _run(tmp_path, 'deit_small_patch16_224', 224, 3, {}, 'fused')
    two = _run(tmp_path, 'deit_small_patch16_224', 224, 3, {'FSVIT_VIT_ATTN_ROWS': '0'}, 'two')
    d = (fused - two).abs().max().item() / max(1.0, float(two.abs().max()))
    print(f'deit_small_patch16_224: max rel |vit_attn_rows - two launches| = {d:.3e}')
    assert torch.isfinite(two).all() and d <= 0.04


# Every dispatch switch the shipped library still reads (each one is a product configuration).  The experiment-only knobs
# (FSVIT_GEMM_TILE, FSVIT_GEMM256_X2, FSVIT_GEMM256_MIN_AI, FSVIT_ATTN_BWD_VALU) were removed; the ones below select a general kernel instead
# of a fused one and must give the same features within the bf16 mode's own noise.
EVAL_SWITCHES = [{'FSVIT_HALO': '0'}, {'FSVIT_STEM_CONV1': '0'}, {'FSVIT_NO_FUSE': '1'}, {'FSVIT_GEMM256': '0'}, {'FSVIT_QKV_ATTN': '0'},
                 {'FSVIT_STAGE1_W4': '0'}, {'FSVIT_MLP_ROWS': '0'}, {'FSVIT_MLP_ROWS': '3'}]


def test_every_eval_dispatch_switch_agrees_with_the_default_path(tmp_path):
    fused = _run(tmp_path, 'visformer_micro_80', 80, 6, {}, 'fused')
    assert torch.isfinite(fused).all()
    for i, env in enumerate(EVAL_SWITCHES):
        other = _run(tmp_path, 'visformer_micro_80', 80, 6, env, f'sw{i}')
        d = (fused - other).abs().max().item() / max(1.0, float(fused.abs().max()))
        print(f'{env}: max rel |default - switched| = {d:.3e}')
        assert torch.isfinite(other).all() and d <= 0.015, env


TRAIN_CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from fewshot_vit_amd import models, synthetic
from fewshot_vit_amd.utils import few_shot as fs
out = sys.argv[1]
m = models.make('meta-baseline', encoder='visformer_micro_80', encoder_args={'numerics': 'bf16', 'drop_path_rate': 0.0})
m.load_state_dict(synthetic.synthetic_checkpoint_sd({k: tuple(v.shape) for k, v in m.state_dict().items()}), strict=True)
m = m.cuda().train()
x = synthetic.synthetic_episodes(5, 2, 5, 1, 3)
xs, xq = fs.split_shot_query(x, 5, 1, 3, 2)
loss = torch.nn.functional.cross_entropy(m(xs.cuda(), xq.cuda()).view(-1, 5), fs.make_nk_label(5, 3, 2).cuda())
loss.backward()
torch.cuda.synchronize()
torch.save({k: p.grad.float().cpu() for k, p in m.named_parameters()}, out)
''' % ROOT


def test_every_training_dispatch_switch_agrees_with_the_default_path(tmp_path):
    def run(env, tag):
        out = str(tmp_path / f'{tag}.pt')
        r = subprocess.run([sys.executable, '-c', TRAIN_CHILD, out], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env), cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        return torch.load(out)
    base = run({}, 'base')
    # (switch, gate): routes that sum in the same order are gated at fp32-rounding level (measured 0 ... 6e-7); routes that ROUND at different points
    # - a BatchNorm folded into conv1 / fc1's weights, BatchNorm statistics of the fp32 accumulators instead of the stored 16-bit map - move bf16
    # gradients by 3 ... 6e-2 of their norm (measured next to each) and get 1.5 x that (one 0.1 gate for every route hid the difference)
    routes = [({'FSVIT_WGRAD3X3': '0'}, 5e-6), ({'FSVIT_WGRAD1X1': '0'}, 5e-6), ({'FSVIT_GCONV3X3': '0'}, 1e-6), ({'FSVIT_WGRAD_SIDE_STREAM': '0'}, 1e-6),
              ({'FSVIT_BN_ROWS': '0'}, 1e-6),
              ({'FSVIT_STAGE1_TRAIN_FUSED': '0'}, 0.06),              # 3.8e-2
              ({'FSVIT_STAGE1_BLOCK_FUSED': '0'}, 0.06),              # 3.8e-2
              ({'FSVIT_BN_PRODUCER_STATS': '0'}, 0.09),               # 6.0e-2
              ({'FSVIT_STAGE1_TRAIN_FUSED': '0', 'FSVIT_GCONV3X3': '0'}, 0.06)]      # (the grouped-conv kernel only runs on the three-launch route)
    for i, (env, gate) in enumerate(routes):
        other = run(env, f'tr{i}')
        # (the bias of a PatchEmbed conv and of its BatchNorm have a structurally ZERO gradient - every consumer of the residual stream starts with a
        # BatchNorm that removes a per-channel constant - what the kernels return there is rounding noise, different on every route)
        dead = ('patch_embed2.proj.bias', 'patch_embed3.proj.bias', 'patch_embed2.norm.bn.bias', 'patch_embed3.norm.bn.bias')
        worst = max(float((other[k] - v).norm() / (v.norm() + 1e-12)) for k, v in base.items() if float(v.norm()) > 1e-5 and not k.endswith(dead))
        print(f'{env}: worst gradient rel difference to the default path = {worst:.3e} (gate {gate:.1e})')
        assert worst <= gate, env


WROUND_CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from fewshot_vit_amd import models, synthetic
mode, numerics, out = sys.argv[1], sys.argv[2], sys.argv[3]
m = models.make('visformer_micro_80', numerics=numerics)
shapes = {'encoder.' + k: tuple(v.shape) for k, v in m.state_dict().items()}
sd = synthetic.synthetic_checkpoint_sd(shapes) if mode == 'calibrated' else synthetic.procedural_state_dict(shapes)      # 'default': running_mean 0 / running_var 1 as after __init__
m.load_state_dict({k[len('encoder.'):]: v for k, v in sd.items()}, strict=True)
m = m.cuda().eval()
x = torch.randn(6, 3, 80, 80, generator=torch.Generator().manual_seed(5))
with torch.no_grad():
    f = m(x.cuda()).float().cpu()
torch.save(f, out)
''' % ROOT


@pytest.mark.parametrize('mode', ['calibrated', 'default'])
def test_weight_rounding_correction_switch_and_stale_statistics(tmp_path, mode):
    """FSVIT_WROUND=0 builds the 16-bit weight images without the calibration-free bias correction.  The correction must not move the
    bf16 features AWAY from the exact-fp32 ones - neither with statistics that match the data ('calibrated') nor with the default running statistics
    of a freshly constructed network ('default': mean 0 / var 1, data that does not match them)."""
    def run(numerics, env, tag):
        out = str(tmp_path / f'{mode}_{tag}.pt')
        r = subprocess.run([sys.executable, '-c', WROUND_CHILD, mode, numerics, out], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env), cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        return torch.load(out)
    exact = run('parity', {}, 'parity')
    on = run('bf16', {}, 'on')
    off = run('bf16', {'FSVIT_WROUND': '0'}, 'off')
    assert torch.isfinite(on).all() and torch.isfinite(off).all()
    scale = max(1.0, float(exact.abs().max()))
    d_on, d_off = (on - exact).abs().max().item() / scale, (off - exact).abs().max().item() / scale
    print(f'{mode}: max rel |bf16 - fp32| with the correction {d_on:.3e}, without {d_off:.3e}')
    assert not torch.equal(on, off)                  # the switch is live
    if mode == 'calibrated':
        assert d_on <= 1.0 * d_off + 1e-3, (d_on, d_off)          # statistics that describe the data: the correction does not hurt
        assert d_off <= 0.03                                      # the pre-correction tolerance of the bf16 mode on these features
    else:
        # Measured (round 5): with statistics that do NOT describe the data the network itself is degenerate (bf16 already deviates by 12 % of the
        # feature scale) and the correction's E[operand] estimates are wrong: 0.119 without, 0.154 with.  The correction is a bet on calibrated
        # statistics; this gate pins how much it can cost when the bet is lost, and FSVIT_WROUND=0 is the way out.
        assert d_on <= 1.5 * d_off, (d_on, d_off)


STATS_CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from fewshot_vit_amd import models, synthetic
from fewshot_vit_amd.utils import few_shot as fs
E, out = int(sys.argv[1]), sys.argv[2]
m = models.make('meta-baseline', encoder='visformer_micro_80', encoder_args={'numerics': 'bf16', 'drop_path_rate': 0.0})
m.load_state_dict(synthetic.synthetic_checkpoint_sd({k: tuple(v.shape) for k, v in m.state_dict().items()}), strict=True)
m = m.cuda().train()
x = synthetic.synthetic_episodes(5, E, 5, 1, 3)
xs, xq = fs.split_shot_query(x, 5, 1, 3, E)
with torch.no_grad():
    m(xs.cuda(), xq.cuda())
torch.cuda.synchronize()
torch.save({k: v.float().cpu() for k, v in m.state_dict().items() if k.startswith('encoder.stem') and 'running' in k}, out)
''' % ROOT


@pytest.mark.parametrize('episodes', [1, 7, 40])
def test_producer_statistics_are_per_channel_exact(tmp_path, episodes):
    """the BatchNorm statistics the stem's GEMM / halo kernels sum in their epilogues (conv_gemm_v2<...,STATS>, conv3x3_halo<...,STATS>: one partial
    row per persistent workgroup, valid only while (grid / 8) %% tiles_n == 0) against the reduce pass over the stored map (FSVIT_BN_PRODUCER_STATS=0), channel
    by channel, at three launch geometries (20 / 140 / 800 images).  The two routes sum the same channel's values - fp32 accumulators vs their bf16-rounded
    stores - so the running means agree to 1e-4 of a standard deviation and the variances to 2e-4 (measured: 6e-6 / 9e-6); a partial row credited to the wrong channel
    would be off by the spread of the channel means (~ a standard deviation)."""
    def run(env, tag):
        out = str(tmp_path / f'{tag}.pt')
        r = subprocess.run([sys.executable, '-c', STATS_CHILD, str(episodes), out], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env), cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        return torch.load(out)
    a, b = run({}, 'producer'), run({'FSVIT_BN_PRODUCER_STATS': '0'}, 'reduce')
    assert set(a) == set(b) and len(a) == 8
    worst_m = worst_v = 0.0
    for k in a:
        if k.endswith('running_mean'):
            sd = b[k.replace('running_mean', 'running_var')].sqrt()
            worst_m = max(worst_m, float(((a[k] - b[k]).abs() / sd).max()))
        else:
            worst_v = max(worst_v, float(((a[k] - b[k]).abs() / b[k]).max()))
    print(f'{episodes} episodes: worst |d running_mean| / std = {worst_m:.2e}, worst relative d running_var = {worst_v:.2e}')
    assert worst_m <= 1e-4 and worst_v <= 2e-4          # measured 2 .. 6e-6 / 2 .. 9e-6


def test_fp32_attention_backward_fallback_switch():
    """FSVIT_ATTN_BWD_F32_MFMA=0 keeps the FMA-loop attention backward for the Visformer head shapes (the kernel the fp32 MFMA one replaced in
    round 3): the same operator tests must pass through it."""
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_train.py'), '-q', '-x', '-k', 'attention_backward_vs_torch'],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, FSVIT_ATTN_BWD_F32_MFMA='0'), cwd=ROOT)
    assert r.returncode == 0 and ' passed' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_layernorm_backward_fallback_switch():
    """FSVIT_LN_BWD_ROWS=0 keeps the two-pass LayerNorm backward kernel (the register-resident one replaced it in round 3): the ViT training-step
    tests must pass through it as well."""
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_deit.py'), '-q', '-x', '-k', 'vit_train_step_vs_oracle_autograd_small'],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, FSVIT_LN_BWD_ROWS='0'), cwd=ROOT)
    assert r.returncode == 0 and ' passed' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
