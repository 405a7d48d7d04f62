"""End-to-end parity of the HIP Visformer + MetaBaseline path (through the C-ABI) against the
oracle on the same seeded inputs, and against the committed golden vectors of the reference.

north_star tolerance: logits within 1e-3 of the reference CPU path -> enforced for the `parity`
numerics mode (exact fp32 MFMA).  The `bf16` mode is the throughput mode; its measured deviation
is bounded loosely here (SURVEY.md 7 measured 1.6e-2..3.2e-2 for bf16 operands) and arg-max
agreement is required."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LOGIT_TOL_PARITY = 1e-3        # BASELINE.json north_star
# bf16 throughput mode against the fp32 reference golden: 1.5 x the measured 4.9e-2 (5-shot) / 5.0e-2 (1-shot) of round 6 (table GELU on the bf16-rounded
# pre-activation in stage1_w4 / mlp_rows; round 4: 4.2e-2 / 5.7e-2 with gelu_sig on the fp32 one).  Rounds 1-3 measured
# 9.0e-2 .. 1.07e-1: the bf16 rounding of the (BN-folded) WEIGHTS - a fixed perturbation of the model that does not average out over tokens
# (ablation on the rounding-point emulator: DESIGN.md 2, tools/emul_ablation.py).  Its MEAN effect is now folded into the fp32 biases at pack
# time from the checkpoint's own BatchNorm statistics (engine.hip `WRound`); what is left is the per-image part and the activation rounding.
LOGIT_TOL_BF16 = 0.076
# ... and against the oracle that rounds where the kernels round (oracle/visformer_emul.py): only accumulation order and the
# softmax / GELU instruction sequences differ - one-ulp flips of stored bf16 activations (2^-8 relative) that then propagate.
# 1.5 x the measured 3.1e-2 / 3.5e-2 (logits, mean 7e-3; round 6: 3.5e-2 / 4.0e-2, mean 9e-3 - the same gate holds) and 1.2e-2 (taps, relative to the tap's max).
LOGIT_TOL_BF16_EMUL = 5.3e-2
TAP_TOL_BF16_EMUL = 1.8e-2


@pytest.fixture(scope='module')
def full_sd():
    from fewshot_vit_amd import synthetic
    from oracle import visformer_oracle as vo
    shapes = vo.state_dict_shapes(vo.VisformerCfg(), prefix='encoder.')
    shapes['temp'] = ()
    return synthetic.synthetic_checkpoint_sd(shapes)


def _model(sd, numerics):
    from fewshot_vit_amd import models
    m = models.make('meta-baseline', encoder='visformer_micro_80', encoder_args={'numerics': numerics})
    m.load_state_dict(sd, strict=True)
    return m.cuda().eval()


def _episode(seed, shot):
    from fewshot_vit_amd import synthetic
    from fewshot_vit_amd.utils import few_shot as fs
    x = synthetic.synthetic_episodes(seed, 1, 5, shot, 15)
    return fs.split_shot_query(x, 5, shot, 15, 1)


@pytest.mark.parametrize('name,seed,shot', [('5shot', 11, 5), ('1shot', 12, 1)])
def test_logits_parity_mode_vs_reference_golden(full_sd, golden_dir, name, seed, shot):
    z = np.load(os.path.join(golden_dir, 'full_visformer_micro_80.npz'))
    m = _model(full_sd, 'parity')
    xs, xq = _episode(seed, shot)
    with torch.no_grad():
        logits = m(xs.cuda(), xq.cuda())
    torch.cuda.synchronize()
    err = np.abs(logits.cpu().numpy() - z[f'logits_{name}']).max()
    print(f'[parity] {name} max|dlogit| vs reference golden = {err:.3e}')
    assert logits.shape == (1, 75, 5)
    assert err <= LOGIT_TOL_PARITY


@pytest.mark.parametrize('numerics', ['bf16x2', 'f16x2'])
@pytest.mark.parametrize('name,seed,shot', [('5shot', 11, 5), ('1shot', 12, 1)])
def test_logits_two_limb_modes_vs_reference_golden(full_sd, golden_dir, name, seed, shot, numerics):
    """The 16-bit-MFMA modes that also meet the north-star tolerance: fp32 storage, every GEMM with both operands split into hi + lo
    16-bit limbs (conv_gemm_v2.hip x2_split; VERDICT r01 #1c's split-operand path)."""
    z = np.load(os.path.join(golden_dir, 'full_visformer_micro_80.npz'))
    m = _model(full_sd, numerics)
    xs, xq = _episode(seed, shot)
    with torch.no_grad():
        logits = m(xs.cuda(), xq.cuda())
    torch.cuda.synchronize()
    err = np.abs(logits.cpu().numpy() - z[f'logits_{name}']).max()
    print(f'[{numerics}] {name} max|dlogit| vs reference golden = {err:.3e}')
    assert err <= LOGIT_TOL_PARITY


@pytest.mark.parametrize('name,seed,shot', [('5shot', 11, 5), ('1shot', 12, 1)])
def test_logits_bf16_mode_vs_reference_golden(full_sd, golden_dir, name, seed, shot):
    z = np.load(os.path.join(golden_dir, 'full_visformer_micro_80.npz'))
    ref = z[f'logits_{name}']
    m = _model(full_sd, 'bf16')
    xs, xq = _episode(seed, shot)
    with torch.no_grad():
        logits = m(xs.cuda(), xq.cuda()).cpu().numpy()
    err = np.abs(logits - ref).max()
    agree = (logits.argmax(-1) == ref.argmax(-1)).mean()
    print(f'[bf16] {name} max|dlogit| vs reference golden = {err:.3e}, argmax agreement = {agree:.4f}')
    assert err <= LOGIT_TOL_BF16
    assert agree == 1.0


# f16 mode (fp16 storage + MFMA, same kernels compiled for _Float16): the rounding-point oracle predicts 1.3e-2 / 9e-3 against the fp32
# reference for these two episodes; set from the GPU measurement (1.5 x).
LOGIT_TOL_F16 = 1.1e-2          # 1.5 x the measured 7.2e-3 (5-shot) / 6.4e-3 (1-shot) with the weight-rounding correction (round 3: 1.18e-2 / 9.4e-3)
LOGIT_TOL_F16_EMUL = 7.5e-3     # 1.5 x the measured 4.9e-3 / 4.4e-3


@pytest.mark.parametrize('name,seed,shot', [('5shot', 11, 5), ('1shot', 12, 1)])
def test_logits_f16_mode_vs_reference_golden_and_rounding_point_oracle(full_sd, golden_dir, name, seed, shot):
    """The `f16` numerics mode: every 16-bit kernel of the engine compiled for _Float16 (namespace fsvit_f16).  Against the reference's
    golden logits it must sit ~8 x closer than bf16 (3 more mantissa bits; the bf16 deviation is weight rounding), and
    against the oracle that rounds to fp16 where the kernels do, closer still."""
    from oracle import visformer_emul as ve
    from oracle import visformer_oracle as vo
    z = np.load(os.path.join(golden_dir, 'full_visformer_micro_80.npz'))
    ref = z[f'logits_{name}']
    m = _model(full_sd, 'f16')
    xs, xq = _episode(seed, shot)
    with torch.no_grad():
        logits = m(xs.cuda(), xq.cuda()).cpu().numpy()
    err = np.abs(logits - ref).max()
    emu = ve.meta_baseline_forward_emul(full_sd, xs, xq, vo.VisformerCfg(), residual='bf16', storage=torch.float16).numpy()
    err_e = np.abs(logits - emu).max()
    print(f'[f16] {name} max|dlogit| vs reference golden = {err:.3e}, vs fp16 rounding-point oracle = {err_e:.3e}')
    assert err <= LOGIT_TOL_F16
    assert err_e <= LOGIT_TOL_F16_EMUL
    assert (logits.argmax(-1) == ref.argmax(-1)).mean() == 1.0


def test_f16_mode_is_launch_size_invariant_and_bf16_engine_is_untouched(full_sd):
    """Two engines of different 16-bit types side by side in one process (two kernel namespaces in one library): each keeps its own
    results, and the f16 engine is batch-size invariant like the bf16 one."""
    from fewshot_vit_amd import synthetic
    from fewshot_vit_amd.utils import few_shot as fs
    mb, mf = _model(full_sd, 'bf16'), _model(full_sd, 'f16')
    x = synthetic.synthetic_episodes(78, 3, 5, 1, 15)
    xs, xq = fs.split_shot_query(x, 5, 1, 15, 3)
    with torch.no_grad():
        b0 = mb(xs.cuda(), xq.cuda()).cpu()
        f_all = mf(xs.cuda(), xq.cuda()).cpu()
        b1 = mb(xs.cuda(), xq.cuda()).cpu()
        f_single = torch.cat([mf(xs[e:e + 1].cuda(), xq[e:e + 1].cuda()).cpu() for e in range(3)])
    assert torch.equal(b0, b1) and torch.equal(f_all, f_single)
    assert not torch.equal(b0, f_all) and (b0 - f_all).abs().max() < 0.3


@pytest.mark.parametrize('name,seed,shot', [('5shot', 11, 5), ('1shot', 12, 1)])
def test_logits_bf16_mode_vs_rounding_point_oracle(full_sd, name, seed, shot):
    """a bf16 gate that means something.  The emulated oracle applies bf16 rounding exactly where the engine
    stores / feeds MFMA operands, so a kernel regression (a wrong fragment, a dropped K slice, a missing bias) cannot hide inside the
    0.1 logit deviation that bf16 weights cause against the fp32 reference."""
    from oracle import visformer_emul as ve
    from oracle import visformer_oracle as vo
    m = _model(full_sd, 'bf16')
    xs, xq = _episode(seed, shot)
    ref = ve.meta_baseline_forward_emul(full_sd, xs, xq, vo.VisformerCfg(), residual='bf16').numpy()
    with torch.no_grad():
        logits = m(xs.cuda(), xq.cuda()).cpu().numpy()
    err = np.abs(logits - ref).max()
    print(f'[bf16] {name} max|dlogit| vs rounding-point oracle = {err:.3e} (mean {np.abs(logits - ref).mean():.3e})')
    assert err <= LOGIT_TOL_BF16_EMUL
    assert (logits.argmax(-1) == ref.argmax(-1)).mean() >= 0.98


def test_residual_stream_taps_bf16_vs_rounding_point_oracle(full_sd):
    """Every residual-stream checkpoint of the bf16 engine against the rounding-point oracle (relative to the tap's max)."""
    from fewshot_vit_amd import synthetic
    from oracle import visformer_emul as ve
    from oracle import visformer_oracle as vo
    cfg = vo.VisformerCfg()
    m = _model(full_sd, 'bf16')
    eng = m.encoder.engine()
    x = synthetic.synthetic_episodes(31, 1, 3, 1, 1)            # 6 images
    B = x.shape[0]
    shapes = {'stem': (B, 20, 20, 128), 'patch_embed2': (B, 10, 10, 256), 'patch_embed3': (B, 5, 5, 512)}
    shapes.update({f'stage1.{i}': (B, 20, 20, 128) for i in range(4)})
    shapes.update({f'stage2.{i}': (B, 10, 10, 256) for i in range(2)})
    shapes.update({f'stage3.{i}': (B, 5, 5, 512) for i in range(3)})
    bufs = {k: eng.set_tap(k, s) for k, s in shapes.items()}
    with torch.no_grad():
        feat = m.encoder(x.cuda())
    torch.cuda.synchronize()
    taps = {}
    with torch.no_grad():
        pooled = ve.visformer_forward_emul(full_sd, x, cfg, prefix='encoder.', residual='bf16', taps=taps)
    worst = {}
    for k, buf in bufs.items():
        got = buf.float().cpu().permute(0, 3, 1, 2)
        worst[k] = ((got - taps[k]).abs().max() / max(1.0, float(taps[k].abs().max()))).item()
    print('[bf16 vs rounding-point oracle] tap rel errors:', {k: f'{v:.2e}' for k, v in worst.items()})
    perr = (feat.cpu() - pooled).abs().max().item() / max(1.0, float(pooled.abs().max()))
    print(f'[bf16 vs rounding-point oracle] pooled feature rel err = {perr:.3e}')
    for k, v in worst.items():
        assert v <= TAP_TOL_BF16_EMUL, (k, v)
    assert perr <= TAP_TOL_BF16_EMUL


@pytest.mark.parametrize('numerics,tol', [('parity', 2e-4), ('bf16', 0.027)])     # bf16: 1.5 x the measured 1.8e-2
def test_residual_stream_taps_vs_oracle(full_sd, numerics, tol):
    """Every residual-stream checkpoint of the encoder against the oracle's NCHW taps."""
    from fewshot_vit_amd import synthetic
    from oracle import visformer_oracle as vo
    cfg = vo.VisformerCfg()
    m = _model(full_sd, numerics)
    eng = m.encoder.engine()
    x = synthetic.synthetic_episodes(31, 1, 3, 1, 1)            # 6 images
    B = x.shape[0]
    shapes = {'stem': (B, 20, 20, 128), 'patch_embed2': (B, 10, 10, 256), 'patch_embed3': (B, 5, 5, 512)}
    for i in range(4):
        shapes[f'stage1.{i}'] = (B, 20, 20, 128)
    for i in range(2):
        shapes[f'stage2.{i}'] = (B, 10, 10, 256)
    for i in range(3):
        shapes[f'stage3.{i}'] = (B, 5, 5, 512)
    bufs = {k: eng.set_tap(k, s) for k, s in shapes.items()}
    with torch.no_grad():
        feat = m.encoder(x.cuda())
    torch.cuda.synchronize()
    taps = {}
    enc_sd = {k[len('encoder.'):]: v for k, v in full_sd.items() if k.startswith('encoder.')}
    with torch.no_grad():
        pooled = vo.visformer_forward(enc_sd, x, cfg, taps=taps)
    worst = {}
    for k, buf in bufs.items():
        ref = taps[k]
        if k == 'stem':
            ref = ref + enc_sd['pos_embed1']
        elif k.startswith('patch_embed'):
            ref = ref + enc_sd['pos_embed' + k[-1]]
        got = buf.float().cpu().permute(0, 3, 1, 2)
        worst[k] = ((got - ref).abs().max() / max(1.0, float(ref.abs().max()))).item()
    print(f'[{numerics}] tap rel errors:', {k: f'{v:.2e}' for k, v in worst.items()})
    perr = (feat.cpu() - pooled).abs().max().item()
    print(f'[{numerics}] pooled feature max abs err = {perr:.3e}')
    for k, v in worst.items():
        assert v <= tol, (k, v)
    # the pooled feature is a maximum over 5 x 512 values after the final norm; in bf16 it sits within 10 % of the tap bound and moves with the
    # summation order of a single layer (0.0482 with the 256-tile qkv GEMM in stage 3, 0.0515 with the row-wise one): 1.25 x the tap bound
    assert perr <= 1.25 * tol * max(1.0, float(pooled.abs().max()))


def test_batched_episodes_equal_single_episodes(full_sd):
    """Batching episodes per launch must not change any episode's logits (eval BN = running stats)."""
    from fewshot_vit_amd import synthetic
    from fewshot_vit_amd.utils import few_shot as fs
    m = _model(full_sd, 'bf16')
    x = synthetic.synthetic_episodes(77, 3, 5, 1, 15)
    xs, xq = fs.split_shot_query(x, 5, 1, 15, 3)
    with torch.no_grad():
        all_logits = m(xs.cuda(), xq.cuda()).cpu()
        singles = torch.cat([m(xs[e:e + 1].cuda(), xq[e:e + 1].cuda()).cpu() for e in range(3)])
    assert torch.equal(all_logits, singles)


def test_rejects_wrong_image_size_and_cpu_tensors(full_sd):
    m = _model(full_sd, 'bf16')
    with pytest.raises(AssertionError):
        m.encoder(torch.zeros(1, 3, 84, 84, device='cuda'))
    with pytest.raises(RuntimeError):
        m.encoder.engine().forward(torch.zeros(1, 3, 80, 80))     # CPU tensor: no fallback
    with pytest.raises(KeyError):
        from fewshot_vit_amd import models
        models.make('no-such-model')


def test_strict_state_dict_and_repacking(full_sd):
    from fewshot_vit_amd import models
    m = models.make('meta-baseline', encoder='visformer_micro_80', encoder_args={})
    bad = dict(full_sd)
    bad.pop('encoder.stage2.0.attn.qkv.weight')
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad, strict=True)
    m.load_state_dict(full_sd)
    m.cuda().eval()
    xs, xq = _episode(11, 1)
    with torch.no_grad():
        a = m(xs.cuda(), xq.cuda()).cpu()
        with torch.no_grad():
            m.encoder.norm.bn.weight.mul_(1.5)       # in-place edit bumps the version -> engine re-packs
        b = m(xs.cuda(), xq.cuda()).cpu()
    assert not torch.equal(a, b)


def test_ragged_batches_and_chunking(full_sd):
    """Image counts that are not multiples of the tile / chunk sizes, chunked vs unchunked, 10-way episodes."""
    from fewshot_vit_amd import synthetic
    from fewshot_vit_amd.engine import VisformerEngine
    enc_sd = {k[len('encoder.'):]: v for k, v in full_sd.items() if k.startswith('encoder.')}
    cfg = dict(img_size=80, init_channels=64, embed_dim=256, depth=(4, 2, 3), num_heads=6, mlp_ratio=4.0, group=8)
    x = synthetic.synthetic_episodes(3, 1, 1, 7, 0).cuda()              # 7 images
    big = VisformerEngine(cfg, enc_sd, numerics='bf16', chunk_images=1600)
    small = VisformerEngine(cfg, enc_sd, numerics='bf16', chunk_images=3)   # 3 + 3 + 1
    a, b = big.forward(x), small.forward(x)
    one = torch.cat([big.forward(x[i:i + 1]) for i in range(7)])
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a, one)
    assert big.forward(x[:0]).shape == (0, 512)                          # empty input
    m = _model(full_sd, 'bf16')
    x10 = synthetic.synthetic_episodes(4, 2, 10, 5, 5).cuda()           # SUN-M train geometry: 10-way 5-shot 5-query
    from fewshot_vit_amd.utils import few_shot as fs
    xs, xq = fs.split_shot_query(x10, 10, 5, 5, 2)
    with torch.no_grad():
        logits = m(xs, xq)
    assert logits.shape == (2, 50, 10) and torch.isfinite(logits).all()


def test_full_size_step_equals_small_launches(full_sd):
    """BASELINE configs[1] at bench size (64 episodes x 100 images in ONE encoder pass: every persistent kernel walks many tiles per
    workgroup, tile tails included) against the same episodes run 8 at a time: bit-identical logits.  Size-independent property:
    eval-mode episodes are independent and no kernel's arithmetic may depend on where a row sits in the launch."""
    from fewshot_vit_amd import synthetic
    from fewshot_vit_amd.utils import few_shot as fs
    m = _model(full_sd, 'bf16')
    E = 64
    x = synthetic.synthetic_episodes(2024, E, 5, 5, 15)
    xs, xq = fs.split_shot_query(x, 5, 5, 15, E)
    with torch.no_grad():
        big = m(xs.cuda(), xq.cuda()).cpu()
        small = torch.cat([m(xs[e:e + 8].cuda(), xq[e:e + 8].cuda()).cpu() for e in range(0, E, 8)])
    assert big.shape == (E, 75, 5) and torch.isfinite(big).all()
    assert torch.equal(big, small)


def test_image_size_64_runs_the_ring_stage1_block():
    """A 64 x 64 model (stage-1 map 16 x 16: chunks of 64 pixels straddle images in the stage-1 block kernel; stage 2 has
    64 tokens, stage 3 sixteen): parity features vs the oracle, bf16 features vs parity, launch-size invariance."""
    from fewshot_vit_amd import synthetic
    from fewshot_vit_amd.engine import VisformerEngine
    from oracle import visformer_oracle as vo
    ocfg = vo.VisformerCfg(img_size=64)
    full = synthetic.synthetic_checkpoint_sd(vo.state_dict_shapes(ocfg, prefix='encoder.'))       # (the BN calibration is per channel: any image size)
    sd = {k[len('encoder.'):]: v for k, v in full.items()}
    cfg = dict(img_size=64, init_channels=64, embed_dim=256, depth=(4, 2, 3), num_heads=6, mlp_ratio=4.0, group=8)
    x = synthetic.synthetic_episodes(5, 1, 3, 2, 1, img=64)            # 9 images
    with torch.no_grad():
        ref = vo.visformer_forward(sd, x, ocfg)
    par = VisformerEngine(cfg, sd, numerics='parity').forward(x.cuda()).cpu()
    eng = VisformerEngine(cfg, sd, numerics='bf16')
    bf = eng.forward(x.cuda()).cpu()
    one = torch.cat([eng.forward(x[i:i + 1].cuda()) for i in range(x.shape[0])]).cpu()
    e_par, e_bf = (par - ref).abs().max().item(), (bf - par).abs().max().item()
    print(f'64 x 64: parity vs oracle {e_par:.3e}, bf16 vs parity {e_bf:.3e} (max |feat| {ref.abs().max():.2f})')
    assert e_par <= 1e-3
    assert e_bf <= 0.05 * max(1.0, ref.abs().max().item())
    assert torch.equal(bf, one)


@pytest.mark.parametrize('depth', [(0, 1, 1), (1, 0, 1), (1, 1, 0)])
def test_stage_without_blocks_builds_and_matches_parity_in_the_16_bit_modes(depth):
    """the weight-rounding correction handed the next PatchEmbed an operand-mean vector of the wrong size (or none) when a stage
    has no blocks - a heap out-of-bounds read in engine.hip's pack_layer.  The vector is now seeded / re-seeded per stage: a Visformer with an empty
    stage builds in bf16 and f16 and its features stay within the modes' tolerance of the exact-fp32 engine."""
    from fewshot_vit_amd import synthetic
    from fewshot_vit_amd.models.visformer import Visformer
    x = torch.randn(4, 3, 80, 80, generator=torch.Generator().manual_seed(11))
    feats = {}
    for numerics in ('parity', 'bf16', 'f16'):
        m = Visformer(img_size=80, init_channels=64, embed_dim=256, depth=list(depth), num_heads=6, mlp_ratio=4., group=8, numerics=numerics)
        sd = synthetic.procedural_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
        for k in sd:                       # benign running statistics (the shipped calibration is for depth (4, 2, 3))
            if k.endswith('running_var'):
                sd[k] = torch.ones_like(sd[k])
            if k.endswith('running_mean'):
                sd[k] = torch.zeros_like(sd[k])
        m.load_state_dict(sd, strict=True)
        m = m.cuda().eval()
        with torch.no_grad():
            feats[numerics] = m(x.cuda()).float().cpu()
        assert torch.isfinite(feats[numerics]).all(), (depth, numerics)
    scale = max(1.0, float(feats['parity'].abs().max()))
    for numerics, tol in (('bf16', 0.05), ('f16', 0.01)):
        d = float((feats[numerics] - feats['parity']).abs().max()) / scale
        print(f'depth {depth} {numerics}: max rel |d feature| vs parity = {d:.3e}')
        assert d <= tol, (depth, numerics, d)
