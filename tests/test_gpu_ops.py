"""Operator-level parity of the HIP kernels (through the C-ABI) against plain fp32 torch on CPU.

Tolerances: f32 storage uses exact-fp32 MFMA -> 2e-4 abs on O(1..10) outputs (accumulation
order only).  bf16 storage: inputs/weights are pre-rounded to bf16 on the reference side, so the
remaining error is the bf16 rounding of the OUTPUT (rel 2^-8) plus fp32 accumulation order.
"""
import math
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {'f32': torch.float32, 'bf16': torch.bfloat16}


def _tol(dt, ref):
    scale = float(ref.abs().max())
    return 2e-4 * max(1.0, scale) if dt == 'f32' else 1.2e-2 * max(1.0, scale)


def pack_w(w, groups, dtype):
    """[O, Ig, KH, KW] -> [groups][N][Kw] with k = (ky*KW+kx)*Ig + c, zero-padded to the K slice."""
    O, Ig, KH, KW = w.shape
    N = O // groups
    K = KH * KW * Ig
    bke = 32 if dtype == torch.float32 else 64
    Kw = (K + bke - 1) // bke * bke
    p = torch.zeros(groups, N, Kw)
    p[:, :, :K] = w.permute(0, 2, 3, 1).reshape(groups, N, K)
    return p.to(dtype)


def q(t, dtype):
    """round-trip through the storage dtype (identity for fp32)."""
    return t.to(dtype).float()


CASES = [
    # name,            B, H,  W,  Cin_tot, O,   KH, s, p, groups, act, res, res_first, bias, pos
    ('1x1_gelu',        2, 20, 20, 128,    256, 1, 1, 0, 1, 1, False, 0, True, False),
    ('3x3_lrelu_res',   2, 40, 40, 64,     128, 3, 1, 1, 1, 2, True, 1, True, False),
    ('3x3_grouped',     3, 20, 20, 256,    256, 3, 1, 1, 8, 1, False, 0, False, False),
    ('3x3_grouped_rag', 5, 7,  9,  256,    256, 3, 1, 1, 8, 0, False, 0, False, False),    # chunks straddling images, a ragged last chunk
    ('k2s2_pos',        2, 20, 20, 128,    256, 2, 2, 0, 1, 0, False, 0, True, True),
    ('1x1_tails',       1, 5,  5,  288,    96,  1, 1, 0, 1, 0, True, 0, False, False),
    ('1x1_qkv_like',    3, 10, 10, 256,    864, 1, 1, 0, 1, 0, False, 0, True, False),
    ('k32_im2col',      2, 40, 40, 32,     64,  1, 1, 0, 1, 2, False, 0, True, False),
    ('1x1_bigK',        5, 5,  5,  2048,   512, 1, 1, 0, 1, 0, True, 0, False, False),
    ('3x3_small_n32',   1, 12, 12, 32,     32,  3, 1, 1, 1, 1, False, 0, True, False),
    # stem geometry (40x40, 128 output channels): conv3x3_halo.hip in bf16 (several 8-row tiles per image, image borders)
    ('halo_conv2_c64',  5, 40, 40, 64,     128, 3, 1, 1, 1, 2, False, 0, True, False),
    ('halo_c128_gelu',  3, 40, 40, 128,    128, 3, 1, 1, 1, 1, False, 0, True, False),
    # dense bf16 layers large enough for gemm256.hip (M >= 1024, N >= 192, K % 64 == 0): M / N tails, every epilogue
    ('g256_fc1_gelu',   13, 10, 10, 256,   1024, 1, 1, 0, 1, 1, False, 0, True, False),
    ('g256_fc2_res',    11, 10, 10, 1024,  256, 1, 1, 0, 1, 0, True, 0, True, False),
    ('g256_qkv_tailN',  12, 10, 10, 256,   1152, 1, 1, 0, 1, 0, False, 0, True, False),
    ('g256_qkv_n864',   12, 10, 10, 256,   864, 1, 1, 0, 1, 0, False, 0, True, False),     # 48-wide heads: 197.5 flop/B, the lowest gemm256 takes
    ('g256_proj_k384',  41, 5,  5,  384,   704, 1, 1, 0, 1, 2, True, 1, False, False),
    ('g256_k128',       3, 20, 20, 128,    320, 1, 1, 0, 1, 0, False, 0, False, False),
    ('g256_bigK',       45, 5,  5,  2048,  512, 1, 1, 0, 1, 0, True, 0, True, False),
]


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_conv_gemm(case, dt):
    from fewshot_vit_amd.engine import ops
    name, B, H, W, Cin, O, KH, s, p, groups, act, use_res, res_first, use_bias, use_pos = case
    dtype = DT[dt]
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 100000)
    Ig = Cin // groups
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = q(torch.randn(O, Ig, KH, KH, generator=g) / math.sqrt(Ig * KH * KH), dtype)
    bias = torch.randn(O, generator=g) * 0.3 if use_bias else None
    OH = (H + 2 * p - KH) // s + 1
    res = q(torch.randn(B, O, OH, OH, generator=g), dtype) if use_res else None
    pos = torch.randn(OH * OH, O, generator=g) * 0.2 if use_pos else None

    ref = F.conv2d(x, w, bias, stride=s, padding=p, groups=groups)
    if use_res and res_first:
        ref = ref + res
    ref = {0: lambda t: t, 1: F.gelu, 2: lambda t: F.leaky_relu(t, 0.1)}[act](ref)
    if use_res and not res_first:
        ref = ref + res
    if use_pos:
        ref = ref + pos.t().reshape(1, O, OH, OH)

    dev = 'cuda'
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev, dtype)
    wd = pack_w(w, groups, dtype).to(dev)
    y = ops.conv_gemm(xd, wd, bias.to(dev) if use_bias else None,
                      res.permute(0, 2, 3, 1).contiguous().to(dev, dtype) if use_res else None,
                      pos.to(dev) if use_pos else None,
                      B, H, W, Ig, KH, KH, s, p, O // groups, groups, act, res_first)
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs().max().item()
    assert err <= _tol(dt, ref), (name, dt, err)


X2_CASES = ['1x1_gelu', '3x3_lrelu_res', '3x3_grouped', '3x3_grouped_rag', 'k2s2_pos', '1x1_tails', '1x1_qkv_like', 'k32_im2col', '1x1_bigK', '3x3_small_n32',
            # M >= 1024, N >= 192: gemm256_x2_kernel (256 x 256 two-limb tile): M / N tails, every epilogue
            'g256_fc1_gelu', 'g256_fc2_res', 'g256_qkv_tailN', 'g256_qkv_n864', 'g256_proj_k384', 'g256_k128', 'g256_bigK']


@pytest.mark.parametrize('numerics', ['bf16x2', 'f16x2'])
@pytest.mark.parametrize('case', [c for c in CASES if c[0] in X2_CASES], ids=[c[0] for c in CASES if c[0] in X2_CASES])
def test_conv_gemm_two_limb(case, numerics):
    """fp32 storage, two-limb 16-bit MFMA arithmetic (conv_gemm_v2_kernel<f32x2l,...>): UN-rounded fp32 operands against an fp64
    convolution.  Operand precision 2^-16 (bf16 limbs; the activation's hi limb is a truncation) / 2^-21 (fp16 limbs)."""
    from fewshot_vit_amd.engine import ops
    name, B, H, W, Cin, O, KH, s, p, groups, act, use_res, res_first, use_bias, use_pos = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 100000)
    Ig = Cin // groups
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(O, Ig, KH, KH, generator=g) / math.sqrt(Ig * KH * KH)
    bias = torch.randn(O, generator=g) * 0.3 if use_bias else None
    OH = (H + 2 * p - KH) // s + 1
    res = torch.randn(B, O, OH, OH, generator=g) if use_res else None
    pos = torch.randn(OH * OH, O, generator=g) * 0.2 if use_pos else None
    ref = F.conv2d(x.double(), w.double(), bias.double() if use_bias else None, stride=s, padding=p, groups=groups)
    if use_res and res_first:
        ref = ref + res
    ref = {0: lambda t: t, 1: F.gelu, 2: lambda t: F.leaky_relu(t, 0.1)}[act](ref)
    if use_res and not res_first:
        ref = ref + res
    if use_pos:
        ref = ref + pos.t().reshape(1, O, OH, OH)
    dev = 'cuda'
    wd = ops.x2_limbs(pack_w(w, groups, torch.float32), numerics).to(dev)
    y = ops.conv_gemm(x.permute(0, 2, 3, 1).contiguous().to(dev), wd, bias.to(dev) if use_bias else None,
                      res.permute(0, 2, 3, 1).contiguous().to(dev) if use_res else None, pos.to(dev) if use_pos else None,
                      B, H, W, Ig, KH, KH, s, p, O // groups, groups, act, res_first, numerics=numerics)
    torch.cuda.synchronize()
    err = (y.cpu().permute(0, 3, 1, 2).double() - ref).abs().max().item()
    print(f'conv_gemm[{numerics}] {name}: max err {err:.3e} (max |y| {ref.abs().max():.2f})')
    assert err <= (1e-4 if numerics == 'bf16x2' else 2e-5) * max(1.0, float(ref.abs().max())), (name, numerics, err)


@pytest.mark.parametrize('dt,B', [('bf16', 53), ('bf16', 3), ('f32', 2)])
def test_stem_tail_conv3_downsample_lrelu_maxpool_pos(dt, B):
    """The most expensive instantiation of the bench, conv3x3_halo_kernel<128,true>: conv3 + bn3 with the
    downsample conv + bn_d as a tail K slice over the im2col rows, LeakyReLU(0.1), MaxPool2d(2) over 2x2-window-major rows, + pos_embed1
    (visformer.py:213-237,431).  Reference: fp32 torch on operands pre-rounded to the storage dtype (the kernel's rounding points: x, x2,
    w in, y out).  B = 53 -> 265 tiles > 256 persistent workgroups: some workgroups walk two tiles (halo prefetch under the epilogue);
    image borders (zero padding) and every window position are covered by comparing the whole map.  f32 runs the same contract on
    conv_gemm_v2 (x2 / pool2 / pos)."""
    from fewshot_vit_amd.engine import ops
    dtype = DT[dt]
    g = torch.Generator().manual_seed(1234 + B)
    H = W = 40
    Cin = N = 128
    bke = 32 if dt == 'f32' else 64
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)                                   # c2 = output of stem conv2
    img = q(torch.randn(B, 3, 2 * H, 2 * W, generator=g), dtype)                           # the network input, for the downsample conv
    w3 = q(torch.randn(N, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin), dtype)             # bn3 folded
    wd = q(torch.randn(N, 3, 3, 3, generator=g) / math.sqrt(27), dtype)                    # bn_d folded
    bias = torch.randn(N, generator=g) * 0.3
    pos = torch.randn((H // 2) * (W // 2), N, generator=g) * 0.2
    ref = F.conv2d(x, w3, None, padding=1) + F.conv2d(img, wd, None, stride=2, padding=1) + bias.view(1, -1, 1, 1)
    ref = F.max_pool2d(F.leaky_relu(ref, 0.1), 2) + pos.t().reshape(1, N, H // 2, W // 2)
    # im2col rows of the 3x3 / stride-2 / pad-1 downsample conv: [B*H*W][32], k = (ky*3 + kx)*3 + c (what stem.hip writes)
    cols = F.unfold(img, 3, padding=1, stride=2).view(B, 3, 9, H * W).permute(0, 3, 2, 1).reshape(B * H * W, 27)
    x2 = torch.zeros(B * H * W, 32)
    x2[:, :27] = cols
    Kmain = 9 * Cin
    wp = torch.zeros(N, Kmain + bke)
    wp[:, :Kmain] = w3.permute(0, 2, 3, 1).reshape(N, Kmain)
    wp[:, Kmain:Kmain + 27] = wd.permute(0, 2, 3, 1).reshape(N, 27)
    dev = 'cuda'
    y = ops.conv_stem_tail(x.permute(0, 2, 3, 1).contiguous().to(dev, dtype), wp.to(dev, dtype), bias.to(dev), pos.to(dev),
                           x2.to(dev, dtype), 32)
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs()
    if dt == 'f32':
        assert err.max().item() <= 2e-4 * max(1.0, float(ref.abs().max()))
    else:       # output rounding of bf16 (2^-9 relative) + accumulation order
        bound = 2.0 ** -8 * ref.abs() + 2e-3
        assert bool((err <= bound).all()), (float(err.max()), float((err - bound).max()))


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('S,heads,hd,pad', [(100, 6, 42, 0), (100, 6, 42, 16), (25, 6, 85, 0), (100, 2, 10, 0), (25, 3, 21, 0), (197, 3, 40, 16)])
def test_attention(S, heads, hd, pad, dt):
    """pad = 16: head dim padded to a multiple of 16 only (bf16: 42 -> 48 = one and a half MFMA K chunks, the engine's choice for
    Visformer-S stage 2); pad = 0: to the full 64-byte chunk."""
    from fewshot_vit_amd.engine import ops
    dtype = DT[dt]
    kch = pad or (16 if dt == 'f32' else 32)
    hdp = (hd + kch - 1) // kch * kch
    B = 3
    g = torch.Generator().manual_seed(S * 7 + hd)
    qkv = q(torch.randn(B, S, 3, heads, hd, generator=g), dtype)
    scale = hd ** -0.5
    qq, kk, vv = [qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3)]        # [B,heads,S,hd]
    ref = ((qq @ kk.transpose(-1, -2)) * scale).softmax(-1) @ vv
    ref = ref.permute(0, 2, 1, 3)                                            # [B,S,heads,hd]
    padded = torch.zeros(B, S, 3, heads, hdp)
    padded[..., :hd] = qkv
    ctx = ops.attention(padded.reshape(B * S, 3 * heads * hdp).to('cuda', dtype), B, S, heads, hdp, scale)
    torch.cuda.synchronize()
    got = ctx.float().cpu().reshape(B, S, heads, hdp)
    assert got[..., hd:].abs().max().item() == 0.0                           # padded head dims stay exactly 0
    err = (got[..., :hd] - ref).abs().max().item()
    assert err <= (2e-5 if dt == 'f32' else 2e-2), (dt, err)


def test_stem_conv1_fused_equals_im2col_plus_gemm():
    """stem_conv1_kernel (im2col + conv1 + LeakyReLU in one pass) against the two-launch path of the same library: the patch rows
    must be bit-identical, conv1 within bf16 rounding of torch's conv on the bf16-rounded operands."""
    from fewshot_vit_amd.engine import ops
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(11)
    B = 5
    x = torch.randn(B, 3, 80, 80, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) / math.sqrt(27)
    bias = torch.randn(64, generator=g) * 0.2
    wp = torch.zeros(64, 64)
    wp[:, :27] = w.permute(0, 2, 3, 1).reshape(64, 27)                          # K order (ky, kx, c)
    wp = q(wp, bf)
    patches, c1 = ops.stem_conv1(x.cuda(), wp.to('cuda', bf), bias.cuda())
    ref_p = ops.im2col27(x.cuda(), bf)
    torch.cuda.synchronize()
    assert torch.equal(patches, ref_p)
    ref = F.conv2d(q(x, bf), q(wp[:, :27].reshape(64, 3, 3, 3).permute(0, 3, 1, 2), bf), bias, stride=2, padding=1)
    ref = F.leaky_relu(ref, 0.1).permute(0, 2, 3, 1).reshape(B * 1600, 64)
    err = (c1.float().cpu() - ref).abs().max().item()
    assert err <= 3e-2, err


@pytest.mark.parametrize('B,S,use_bias', [(3, 100, True), (1, 100, False), (300, 100, True), (5, 37, True), (2, 112, True), (7, 97, True), (4, 1, True)])
def test_qkv_attention_fused_matches_unfused_math(B, S, use_bias):
    """qkv_attn.hip (qkv conv + attention in one launch, qkv kept on chip) against the unfused math with the same bf16 rounding
    points (q / k / v and P rounded to bf16, fp32 accumulation), Visformer-S stage-2 geometry; ragged token counts, one workgroup
    walking several images (B > 256), the single-token edge."""
    from fewshot_vit_amd.engine import ops
    bf = torch.bfloat16
    C, heads, hd, hdp = 256, 6, 42, 48
    g = torch.Generator().manual_seed(B * 131 + S)
    x = q(torch.randn(B * S, C, generator=g), bf)
    w = torch.zeros(3, heads, hdp, C)
    w[:, :, :hd] = torch.randn(3, heads, hd, C, generator=g) / math.sqrt(C)
    w = q(w, bf).reshape(3 * heads * hdp, C)
    bias = torch.zeros(3, heads, hdp)
    if use_bias:
        bias[:, :, :hd] = torch.randn(3, heads, hd, generator=g) * 0.3
    bias = bias.reshape(-1)
    scale = hd ** -0.5
    qkv = q(x @ w.t() + bias, bf).reshape(B, S, 3, heads, hdp)
    qq, kk, vv = [qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3)]
    p = ((qq @ kk.transpose(-1, -2)) * scale).softmax(-1)
    ref = (p @ vv).permute(0, 2, 1, 3).reshape(B * S, heads * hdp)
    got = ops.qkv_attention(x.to('cuda', bf), w.to('cuda', bf), bias.cuda() if use_bias else None, B, S, heads, hdp, scale)
    torch.cuda.synchronize()
    got = got.float().cpu()
    assert torch.isfinite(got).all()
    assert got.reshape(B * S, heads, hdp)[..., hd:].abs().max().item() == 0.0       # padded head dims stay exactly 0
    err = (got - ref).abs()
    assert err.max().item() <= 3e-2 * max(1.0, float(ref.abs().max())), (B, S, err.max().item())
    assert err.mean().item() <= 3e-3, (B, S, err.mean().item())
    # and against the two-launch path of the same library (identical rounding points; the fp32 accumulation order differs)
    two = ops.attention(ops.conv_gemm(x.to('cuda', bf).reshape(B, S, 1, C), w.to('cuda', bf).reshape(1, 3 * heads * hdp, C), bias.cuda() if use_bias else None,
                                      None, None, B, S, 1, C, 1, 1, 1, 0, 3 * heads * hdp, 1, 0, 0).reshape(B * S, 3 * heads * hdp), B, S, heads, hdp, scale)
    d = (got - two.float().cpu()).abs()
    assert d.max().item() <= 2e-2 * max(1.0, float(ref.abs().max())) and d.mean().item() <= 1e-3, (d.max().item(), d.mean().item())


@pytest.mark.parametrize('B,S', [(3, 25), (1031, 25), (640, 16), (9, 32), (5, 1)])
@pytest.mark.parametrize('use_bias', [True, False])
def test_qkv_attention_rows_stage3_matches_unfused_math(B, S, use_bias):
    """qkv_attn_rows (mlp_rows.hip: qkv conv + attention with one image per wave, V computed transposed by swapping the MFMA operands) at the
    Visformer stage-3 geometry (C = 512, 6 heads x 85 padded to 96) against the unfused math with the same bf16 rounding points; partial last
    workgroup, full 32-token maps, the single-token edge; repeats bit-identical."""
    from fewshot_vit_amd.engine import ops
    bf = torch.bfloat16
    C, heads, hd, hdp = 512, 6, 85, 96
    g = torch.Generator().manual_seed(B * 131 + S)
    x = q(torch.randn(B * S, C, generator=g), bf)
    w = torch.zeros(3, heads, hdp, C)
    w[:, :, :hd] = torch.randn(3, heads, hd, C, generator=g) / math.sqrt(C)
    w = q(w, bf).reshape(3 * heads * hdp, C)
    bias = torch.zeros(3, heads, hdp)
    if use_bias:
        bias[:, :, :hd] = torch.randn(3, heads, hd, generator=g) * 0.3
    bias = bias.reshape(-1)
    scale = hd ** -0.5
    qkv = q(x @ w.t() + bias, bf).reshape(B, S, 3, heads, hdp)
    qq, kk, vv = [qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3)]
    p = ((qq @ kk.transpose(-1, -2)) * scale).softmax(-1)
    ref = (p @ vv).permute(0, 2, 1, 3).reshape(B * S, heads * hdp)
    args = (x.to('cuda', bf), w.to('cuda', bf), bias.cuda() if use_bias else None, B, S, heads, hdp, scale)
    got_d = ops.qkv_attention(*args)
    torch.cuda.synchronize()
    got = got_d.float().cpu()
    assert torch.isfinite(got).all()
    assert got.reshape(B * S, heads, hdp)[..., hd:].abs().max().item() == 0.0       # padded head dims stay exactly 0
    err = (got - ref).abs()
    assert err.max().item() <= 3e-2 * max(1.0, float(ref.abs().max())), (B, S, err.max().item())
    assert err.mean().item() <= 3e-3, (B, S, err.mean().item())
    for _ in range(5):
        assert torch.equal(ops.qkv_attention(*args), got_d)


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
def test_im2col_maxpool_pool(dt):
    from fewshot_vit_amd.engine import ops
    dtype = DT[dt]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 3, 80, 80, generator=g)
    cols = ops.im2col27(x.cuda(), dtype).float().cpu()                        # [B*1600, 32]
    ref = F.unfold(x, 3, padding=1, stride=2)                                 # [B, 27 (c,ky,kx), 1600]
    ref = ref.reshape(3, 3, 9, 1600).permute(0, 3, 2, 1).reshape(3 * 1600, 27)   # k = (ky*3+kx)*3 + c
    assert (cols[:, 27:] == 0).all()
    assert (cols[:, :27] - q(ref, dtype)).abs().max().item() == 0.0

    a = q(torch.randn(2, 16, 40, 40, generator=g), dtype)
    pos = torch.randn(400, 16, generator=g)
    mp = ops.maxpool2_pos(a.permute(0, 2, 3, 1).contiguous().to('cuda', dtype), pos.cuda()).float().cpu()
    refmp = F.max_pool2d(a, 2) + pos.t().reshape(1, 16, 20, 20)
    assert (mp.permute(0, 3, 1, 2) - refmp).abs().max().item() <= (1e-6 if dt == 'f32' else 4e-2)

    xx = q(torch.randn(4, 25, 512, generator=g), dtype)
    sc, sh = torch.rand(512, generator=g) + 0.5, torch.randn(512, generator=g)
    pooled = ops.pool_affine(xx.to('cuda', dtype), sc.cuda(), sh.cuda()).cpu()
    assert (pooled - (xx.mean(1) * sc + sh)).abs().max().item() <= 1e-5


@pytest.mark.parametrize('method', ['cos', 'sqr'])
@pytest.mark.parametrize('E,way,shot,Qper,D', [(3, 5, 5, 15, 512), (2, 5, 1, 15, 512), (1, 10, 5, 5, 128)])
def test_proto_head(E, way, shot, Qper, D, method):
    from fewshot_vit_amd.engine import ops
    from oracle import fewshot_oracle as fo
    from oracle import visformer_oracle as vo
    g = torch.Generator().manual_seed(E * 100 + way)
    fs = torch.randn(E, way, shot, D, generator=g)
    fq = torch.randn(E, way * Qper, D, generator=g) + fs.mean(2).repeat_interleave(Qper, dim=1) * 0.7
    temp = 10.0 if method == 'cos' else 0.05
    ref = vo.meta_baseline_head(fs, fq, method=method, temp=temp)
    logits, acc, loss = ops.proto_head(fs.cuda(), fq.cuda(), temp, method)
    torch.cuda.synchronize()
    assert (logits.cpu() - ref).abs().max().item() <= 1e-4 * max(1.0, float(ref.abs().max()))
    label = fo.make_nk_label(way, Qper, 1)
    for e in range(E):
        assert acc[e].item() == pytest.approx(fo.compute_acc(ref[e].numpy(), label), abs=1e-6)
        assert loss[e].item() == pytest.approx(fo.cross_entropy(ref[e].numpy(), label), rel=1e-4, abs=1e-4)


def test_compute_logits_gpu_matches_reference_semantics():
    from fewshot_vit_amd import utils
    from oracle import visformer_oracle as vo
    g = torch.Generator().manual_seed(9)
    feat, proto = torch.randn(2, 7, 64, generator=g), torch.randn(2, 3, 64, generator=g)
    for metric, temp in (('cos', 10.0), ('sqr', 1.0), ('dot', 2.0)):
        got = utils.compute_logits(feat.cuda(), proto.cuda(), metric, temp).cpu()
        ref = vo.compute_logits(feat, proto, metric, temp)
        assert (got - ref).abs().max().item() <= 1e-4 * max(1.0, float(ref.abs().max())), metric


def test_stage1_fused_block_matches_unfused_math():
    """Fused stage-1 block (fsvit_stage1_block: the 16-wave ring kernel at 20 x 20) vs fp32 torch with the same bf16 roundings of x, weights and the
    two hidden maps."""
    from fewshot_vit_amd.engine import ops
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(123)
    B = 5
    x = q(torch.randn(B, 128, 20, 20, generator=g), bf)
    w1 = q(torch.randn(256, 128, 1, 1, generator=g) / math.sqrt(128), bf)
    b1 = torch.randn(256, generator=g) * 0.2
    w2 = q(torch.randn(256, 32, 3, 3, generator=g) / math.sqrt(288), bf)
    w3 = q(torch.randn(128, 256, 1, 1, generator=g) / math.sqrt(256), bf)
    h1 = q(F.gelu(F.conv2d(x, w1, b1)), bf)
    h2 = q(F.gelu(F.conv2d(h1, w2, padding=1, groups=8)), bf)
    ref = x + F.conv2d(h2, w3)
    y = ops.stage1_block(x.permute(0, 2, 3, 1).contiguous().to('cuda', bf), pack_w(w1, 1, bf)[0].cuda(), b1.cuda(),
                         pack_w(w2, 8, bf).cuda(), pack_w(w3, 1, bf)[0].cuda())
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs()
    # hidden maps are rounded to bf16 at slightly different fp32 values on the two sides: allow a few bf16 ulps of the output
    assert err.max().item() <= 3e-2 * max(1.0, float(ref.abs().max())), err.max().item()
    assert err.mean().item() <= 2e-3
    # halo / border structure: the error must not concentrate on the middle rows or the image border
    assert err[:, :, 8:12].mean().item() <= 3 * err.mean().item() + 1e-6
    border = torch.cat([err[:, :, 0].flatten(), err[:, :, -1].flatten(), err[:, :, :, 0].flatten(), err[:, :, :, -1].flatten()])
    assert border.mean().item() <= 3 * err.mean().item() + 1e-6


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('B,HW', [(5, 20), (3, 16), (7, 10), (2, 4), (130, 20)])
def test_stage1_ring_block_matches_unfused_math(B, HW, dtype):
    """fsvit_stage1_block_hw (stage1_w4.hip: weights in registers, pixel rings) vs fp32 torch with the
    same 16-bit roundings of x, the weights and the two hidden maps - several map sizes (chunks of 64 pixels straddle images and the batch end), and
    against the 16-wave ring kernel (fsvit_stage1_block) at 20 x 20."""
    from fewshot_vit_amd.engine import ops
    g = torch.Generator().manual_seed(1000 * B + HW)
    x = q(torch.randn(B, 128, HW, HW, generator=g), dtype)
    w1 = q(torch.randn(256, 128, 1, 1, generator=g) / math.sqrt(128), dtype)
    b1 = torch.randn(256, generator=g) * 0.2
    w2 = q(torch.randn(256, 32, 3, 3, generator=g) / math.sqrt(288), dtype)
    w3 = q(torch.randn(128, 256, 1, 1, generator=g) / math.sqrt(256), dtype)
    # the bf16 build of stage1_w4.hip looks its GELUs up in a table indexed by the bf16-ROUNDED pre-activation (round 6): one more rounding point
    pre = (lambda z: q(z, dtype)) if dtype == torch.bfloat16 else (lambda z: z)
    h1 = q(F.gelu(pre(F.conv2d(x, w1, b1))), dtype)
    h2 = q(F.gelu(pre(F.conv2d(h1, w2, padding=1, groups=8))), dtype)
    ref = x + F.conv2d(h2, w3)
    xd = x.permute(0, 2, 3, 1).contiguous().to('cuda', dtype)
    args = (pack_w(w1, 1, dtype)[0].cuda(), b1.cuda(), pack_w(w2, 8, dtype).cuda(), pack_w(w3, 1, dtype)[0].cuda())
    y = ops.stage1_block_hw(xd, *args)
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs()
    print(f'stage1 ring B={B} {HW}x{HW} {dtype}: max err {err.max():.3e}, mean {err.mean():.3e}')
    assert err.max().item() <= 3e-2 * max(1.0, float(ref.abs().max())), err.max().item()
    assert err.mean().item() <= 2e-3
    border = torch.cat([err[:, :, 0].flatten(), err[:, :, -1].flatten(), err[:, :, :, 0].flatten(), err[:, :, :, -1].flatten()])
    assert border.mean().item() <= 3 * err.mean().item() + 1e-6
    assert torch.equal(y, ops.stage1_block_hw(xd, *args))                                   # deterministic
    if HW == 20 and dtype == torch.bfloat16:                                                 # the 16-wave ring kernel computes the same block
        y0 = ops.stage1_block(xd, *args).float().cpu().permute(0, 3, 1, 2)
        assert (y0 - got).abs().max().item() <= 3e-2 * max(1.0, float(ref.abs().max()))
        assert (y0 - got).abs().mean().item() <= 2e-3      # (measured 1.4e-3: the ring kernel keeps gelu_sig on the fp32 pre-activation; 8e-6 before the table)


@pytest.mark.parametrize('C', [256, 512])
@pytest.mark.parametrize('M', [256, 1000, 70000 + 37])
def test_mlp_rows_fused_matches_unfused_math(M, C):
    """Fused row Mlp (mlp_rows.hip: hidden map kept in registers) vs fp32 torch with the same bf16 roundings of x, the weights
    and the hidden map; M tails, several tiles per persistent workgroup, 10 repeats bit-identical (race screen of the LDS-DMA ring)."""
    from fewshot_vit_amd.engine import ops
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(M + C)
    HID = 4 * C
    x = q(torch.randn(M, C, generator=g), bf)
    w1 = q(torch.randn(HID, C, generator=g) / math.sqrt(C), bf)
    b1 = torch.randn(HID, generator=g) * 0.3
    w2 = q(torch.randn(C, HID, generator=g) / math.sqrt(HID), bf)
    b2 = torch.randn(C, generator=g) * 0.3
    hdn = q(F.gelu(q(x @ w1.t() + b1, bf)), bf)        # (round 6: mlp_rows looks its GELU up by the bf16-rounded pre-activation - one more rounding point)
    xd, w1d, w2d = x.to('cuda', bf), w1.to('cuda', bf), w2.to('cuda', bf)
    for use_b2 in (False, True):
        ref = x + hdn @ w2.t() + (b2 if use_b2 else 0.0)
        y0 = ops.mlp_rows(xd, w1d, b1.cuda(), w2d, b2.cuda() if use_b2 else None)
        torch.cuda.synchronize()
        err = (y0.float().cpu() - ref).abs()
        assert err.max().item() <= 3e-2 * max(1.0, float(ref.abs().max())), (M, use_b2, err.max().item())
        assert err.mean().item() <= 3e-3, (M, use_b2, err.mean().item())
        for _ in range(10):
            assert torch.equal(ops.mlp_rows(xd, w1d, b1.cuda(), w2d, b2.cuda() if use_b2 else None), y0)
    # in place (the engine's use): y aliases x
    lib_y = xd.clone()
    from fewshot_vit_amd import _lib
    from fewshot_vit_amd.engine import _ptr, _stream_ptr
    _lib.check(_lib.load().fsvit_mlp_rows(_ptr(lib_y), _ptr(lib_y), _ptr(w1d), C, _ptr(b1.cuda()), _ptr(w2d), HID, None, M, C, HID, _stream_ptr(xd.device)))
    torch.cuda.synchronize()
    assert torch.equal(lib_y, ops.mlp_rows(xd, w1d, b1.cuda(), w2d, None))


@pytest.mark.parametrize('C,KC', [(256, 288), (512, 576)])
@pytest.mark.parametrize('M', [300, 40000 + 11])
def test_proj_mlp_rows_fused_matches_unfused_math(M, C, KC):
    """mlp_rows with the attention block's proj conv + residual as a prologue: x1 = bf16(x + ctx Wp^T), y = x1 + W2 GELU(W1 x1 + b1);
    fp32 torch with the same bf16 roundings; 5 repeats bit-identical."""
    from fewshot_vit_amd.engine import ops
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(M + C)
    HID = 4 * C
    x = q(torch.randn(M, C, generator=g), bf)
    ctx = q(torch.randn(M, KC, generator=g), bf)
    wp = q(torch.randn(C, KC, generator=g) / math.sqrt(KC), bf)
    w1 = q(torch.randn(HID, C, generator=g) / math.sqrt(C), bf)
    b1 = torch.randn(HID, generator=g) * 0.3
    w2 = q(torch.randn(C, HID, generator=g) / math.sqrt(HID), bf)
    x1 = q(x + ctx @ wp.t(), bf)
    ref = x1 + q(F.gelu(q(x1 @ w1.t() + b1, bf)), bf) @ w2.t()
    args = [t.to('cuda', bf) for t in (x, ctx, wp, w1)] + [b1.cuda(), w2.to('cuda', bf)]
    y0 = ops.proj_mlp_rows(*args)
    torch.cuda.synchronize()
    err = (y0.float().cpu() - ref).abs()
    # x1 is rounded to bf16 at slightly different fp32 values on the two sides (different summation order): a flipped rounding of x1
    # moves y by one bf16 ulp of x1 through the residual
    assert err.max().item() <= 4e-2 * max(1.0, float(ref.abs().max())), (M, C, err.max().item())
    assert err.mean().item() <= 4e-3, (M, C, err.mean().item())
    for _ in range(5):
        assert torch.equal(ops.proj_mlp_rows(*args), y0)


@pytest.mark.parametrize('M', [197, 128 * 40 + 77])
def test_vit_block_tail_matches_unfused_math(M):
    """mlp_rows at the DeiT-S geometry with proj + bias + residual, LayerNorm and both Mlp biases fused (deit.py:69-72) against fp32 torch
    with the same bf16 roundings (x1, LN(x1), GELU output); 5 repeats bit-identical; in place."""
    from fewshot_vit_amd.engine import ops
    bf = torch.bfloat16
    C, KC, HID, eps = 384, 384, 1536, 1e-6
    g = torch.Generator().manual_seed(M)
    x = q(torch.randn(M, C, generator=g) * 2.0 + 0.5, bf)
    ctx = q(torch.randn(M, KC, generator=g), bf)
    wp = q(torch.randn(C, KC, generator=g) / math.sqrt(KC), bf)
    bp = torch.randn(C, generator=g) * 0.3
    w1 = q(torch.randn(HID, C, generator=g) / math.sqrt(C), bf)
    b1 = torch.randn(HID, generator=g) * 0.3
    w2 = q(torch.randn(C, HID, generator=g) / math.sqrt(HID), bf)
    b2 = torch.randn(C, generator=g) * 0.3
    x1 = q(x + ctx @ wp.t() + bp, bf)
    xn = q(F.layer_norm(x1, (C,), eps=eps), bf)
    ref = x1 + q(F.gelu(q(xn @ w1.t() + b1, bf)), bf) @ w2.t() + b2
    args = [x.to('cuda', bf), ctx.to('cuda', bf), wp.to('cuda', bf), bp.cuda(), w1.to('cuda', bf), b1.cuda(), w2.to('cuda', bf), b2.cuda()]
    y0 = ops.vit_block_tail(*args, eps=eps)
    torch.cuda.synchronize()
    err = (y0.float().cpu() - ref).abs()
    assert err.max().item() <= 4e-2 * max(1.0, float(ref.abs().max())), (M, err.max().item())
    assert err.mean().item() <= 4e-3, (M, err.mean().item())
    for _ in range(5):
        assert torch.equal(ops.vit_block_tail(*args, eps=eps), y0)
    from fewshot_vit_amd import _lib
    from fewshot_vit_amd.engine import _ptr, _stream_ptr
    yi = args[0].clone()
    _lib.check(_lib.load().fsvit_vit_block_tail(_ptr(yi), _ptr(yi), _ptr(args[1]), _ptr(args[2]), KC, KC, _ptr(args[3]), _ptr(args[4]), C, _ptr(args[5]),
                                                _ptr(args[6]), HID, _ptr(args[7]), M, C, HID, eps, _stream_ptr(yi.device)))
    torch.cuda.synchronize()
    assert torch.equal(yi, y0)


@pytest.mark.parametrize('M,N', [(197, 1152), (128 * 70 + 5, 1152), (3000, 96)])
def test_ln_linear_rows_matches_unfused_math(M, N):
    """ln_gemm_rows (DeiT norm1 + qkv, deit.py:40-47,:69) against fp32 torch with the normalised rows rounded to bf16; repeats bit-identical."""
    from fewshot_vit_amd.engine import ops
    bf = torch.bfloat16
    C, eps = 384, 1e-6
    g = torch.Generator().manual_seed(M + N)
    x = q(torch.randn(M, C, generator=g) * 2.0 + 0.5, bf)
    w = q(torch.randn(N, C, generator=g) / math.sqrt(C), bf)
    b = torch.randn(N, generator=g) * 0.3
    ref = q(F.layer_norm(x, (C,), eps=eps), bf) @ w.t() + b
    args = [x.to('cuda', bf), w.to('cuda', bf), b.cuda()]
    y0 = ops.ln_linear_rows(*args, eps=eps)
    torch.cuda.synchronize()
    err = (y0.float().cpu() - ref).abs()
    assert err.max().item() <= 4e-2 * max(1.0, float(ref.abs().max())), (M, N, err.max().item())
    assert err.mean().item() <= 4e-3, (M, N, err.mean().item())
    for _ in range(8):
        assert torch.equal(ops.ln_linear_rows(*args, eps=eps), y0)


@pytest.mark.parametrize('M,N,bias', [(49 * 5, 1728, True), (128 * 90 + 33, 1728, False)])
def test_linear_rows_c512_matches_fp32(M, N, bias):
    """The row-wise GEMM without LayerNorm at the Visformer stage-3 qkv geometry (C = 512, N = 3 x 6 x 96) against fp32 torch."""
    from fewshot_vit_amd.engine import ops
    bf = torch.bfloat16
    C = 512
    g = torch.Generator().manual_seed(M + N)
    x = q(torch.randn(M, C, generator=g), bf)
    w = q(torch.randn(N, C, generator=g) / math.sqrt(C), bf)
    b = torch.randn(N, generator=g) * 0.3 if bias else None
    ref = x @ w.t() + (b if bias else 0.0)
    args = [x.to('cuda', bf), w.to('cuda', bf), b.cuda() if bias else None]
    y0 = ops.ln_linear_rows(*args)
    torch.cuda.synchronize()
    err = (y0.float().cpu() - ref).abs()
    assert err.max().item() <= 2e-2 * max(1.0, float(ref.abs().max())), (M, N, err.max().item())
    assert err.mean().item() <= 3e-3, (M, N, err.mean().item())
    for _ in range(8):
        assert torch.equal(ops.ln_linear_rows(*args), y0)


@pytest.mark.parametrize('B,H,N', [(3, 20, 256), (411, 20, 256), (70, 12, 96)])
def test_patch_embed2x2_rows_vs_conv2d(B, H, N):
    """The 2 x 2 / stride-2 patch embedding + pos_embed on the rows kernel (visformer.py:266-288, :437-447) against torch conv2d in fp32 on the
    same bf16 operands; partial last workgroup; repeats bit-identical."""
    from fewshot_vit_amd.engine import ops
    bf = torch.bfloat16
    Ci = 128
    g = torch.Generator().manual_seed(B + H + N)
    x = q(torch.randn(B, Ci, H, H, generator=g), bf)
    w = q(torch.randn(N, Ci, 2, 2, generator=g) / math.sqrt(4 * Ci), bf)
    bias = torch.randn(N, generator=g) * 0.3
    pos = torch.randn(N, H // 2, H // 2, generator=g) * 0.5
    ref = (F.conv2d(x, w, bias, stride=2) + pos).permute(0, 2, 3, 1).reshape(-1, N)
    xd = x.permute(0, 2, 3, 1).contiguous().to('cuda', bf)
    wd = w.permute(0, 2, 3, 1).reshape(N, 4 * Ci).contiguous().to('cuda', bf)          # K order (ky, kx, c)
    posd = pos.permute(1, 2, 0).reshape(-1, N).contiguous().cuda()
    y0 = ops.patch_embed2x2(xd, wd, bias.cuda(), posd)
    torch.cuda.synchronize()
    err = (y0.float().cpu() - ref).abs()
    assert err.max().item() <= 2e-2 * max(1.0, float(ref.abs().max())), (B, H, N, err.max().item())
    assert err.mean().item() <= 3e-3, err.mean().item()
    for _ in range(6):
        assert torch.equal(ops.patch_embed2x2(xd, wd, bias.cuda(), posd), y0)


def test_gemm256_large_shapes_repeatable_and_correct():
    """Race screen of the pipelined 256x256 kernel (counted-vmcnt LDS-DMA ring, cdna_hip_programming.md: a misplaced wait shows up as
    rare wrong tiles): several persistent items per workgroup, tails in M and N, 25 repeats must be bit-identical and match fp32."""
    from fewshot_vit_amd.engine import ops
    g = torch.Generator().manual_seed(11)
    for (M, N, K) in ((70000, 1152, 256), (20000, 2048, 512), (9000, 512, 2048)):
        x = torch.randn(M, K, generator=g).bfloat16()
        w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16()
        xd, wd = x.cuda().view(1, M, 1, K), w.cuda().view(1, N, K)
        first = None
        for rep in range(25):
            y = ops.conv_gemm(xd, wd, None, None, None, 1, M, 1, K, 1, 1, 1, 0, N, 1, 0, False).view(M, N)
            if first is None:
                first = y.clone()
            else:
                assert torch.equal(y, first), f'run {rep} differs from run 0 at {(M, N, K)}'
        torch.cuda.synchronize()
        rows = torch.randint(0, M, (512,), generator=g)
        ref = x[rows].float() @ w.float().t()
        err = (first[rows.cuda()].float().cpu() - ref).abs().max().item()
        assert err <= 2e-2 * max(1.0, ref.abs().max().item()), (M, N, K, err)


@pytest.mark.parametrize('B,S', [(3, 197), (2, 64), (1, 33), (5, 224), (2, 1)])
def test_vit_ln_qkv_attention_rows_matches_unfused_math(B, S):
    """vit_attn_rows (mlp_rows.hip): norm1 + qkv Linear + attention core of a DeiT-S block in one launch - a workgroup of 8 waves owns an image,
    wave = 32-token row block, K and V^T fragments exchanged through LDS, flash-style loop over the key blocks - against the unfused math with
    the same bf16 rounding points (LayerNorm output, q / k / v, probabilities); token counts that end inside / at a row block, the 197 tokens
    of DeiT-S/16, a single token; repeats bit-identical."""
    from fewshot_vit_amd.engine import ops
    bf = torch.bfloat16
    C, heads, hd = 384, 6, 64
    g = torch.Generator().manual_seed(B * 1000 + S)
    x = q(torch.randn(B * S, C, generator=g) * 1.5 + 0.3, bf)
    w = q(torch.randn(3 * heads * hd, C, generator=g) / math.sqrt(C), bf)
    bias = torch.randn(3 * heads * hd, generator=g) * 0.3
    scale = hd ** -0.5
    mu, var = x.mean(-1, keepdim=True), x.var(-1, unbiased=False, keepdim=True)
    xn = q((x - mu) * torch.rsqrt(var + 1e-6), bf)
    qkv = q(xn @ w.t() + bias, bf).reshape(B, S, 3, heads, hd)
    qq, kk, vv = [qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3)]
    p = ((qq @ kk.transpose(-1, -2)) * scale).softmax(-1)
    ref = (p @ vv).permute(0, 2, 1, 3).reshape(B * S, heads * hd)
    args = (x.to('cuda', bf), w.to('cuda', bf), bias.cuda(), B, S, heads, hd, scale)
    got_d = ops.vit_ln_qkv_attention(*args)
    torch.cuda.synchronize()
    got = got_d.float().cpu()
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    print(f'vit_ln_qkv_attention B={B} S={S}: max err {err.max().item():.3e}, mean {err.mean().item():.3e} (max |ref| {ref.abs().max().item():.2f})')
    assert err.max().item() <= 3e-2 * max(1.0, float(ref.abs().max())), (B, S, err.max().item())
    assert err.mean().item() <= 3e-3, (B, S, err.mean().item())
    for _ in range(3):
        assert torch.equal(ops.vit_ln_qkv_attention(*args), got_d)
