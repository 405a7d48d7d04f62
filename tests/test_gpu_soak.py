"""Determinism soak (bench-size launches, repeated): the kernels that order LDS-DMA data with counted vmcnt + barriers (mlp_rows,
qkv_attn) must return bit-identical results run after run.  A wait that is one short shows up as a rare differing tile, which the
small op tests cannot see (tools/soak.py is the long version)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _repeat(fn, n):
    ref = fn()
    torch.cuda.synchronize()
    assert torch.isfinite(ref.float()).all()
    bad = sum(0 if torch.equal(fn(), ref) else 1 for _ in range(n))
    torch.cuda.synchronize()
    return bad


@pytest.mark.parametrize('C,KC,M', [(256, 288, 640000), (512, 576, 160000)])
def test_mlp_rows_is_deterministic_at_bench_size(C, KC, M):
    from fewshot_vit_amd.engine import ops
    bf = torch.bfloat16
    g = torch.Generator(device='cuda').manual_seed(C)
    rn = lambda *s: torch.randn(*s, device='cuda', generator=g)
    x, ctx = rn(M, C).to(bf), rn(M, KC).to(bf)
    wp = (rn(C, KC) / math.sqrt(KC)).to(bf)
    w1, w2 = (rn(4 * C, C) / math.sqrt(C)).to(bf), (rn(C, 4 * C) / math.sqrt(4 * C)).to(bf)
    b1 = rn(4 * C) * 0.3
    assert _repeat(lambda: ops.proj_mlp_rows(x, ctx, wp, w1, b1, w2, None), 40) == 0
    assert _repeat(lambda: ops.mlp_rows(x, w1, b1, w2, None), 40) == 0


def test_qkv_attention_is_deterministic_at_bench_size():
    from fewshot_vit_amd.engine import ops
    bf = torch.bfloat16
    B, S, heads, hd, hdp = 6400, 100, 6, 42, 48
    g = torch.Generator(device='cuda').manual_seed(7)
    rn = lambda *s: torch.randn(*s, device='cuda', generator=g)
    x = rn(B * S, 256).to(bf)
    w = torch.zeros(3, heads, hdp, 256, device='cuda')
    w[:, :, :hd] = rn(3, heads, hd, 256) / 16.0
    w = w.reshape(3 * heads * hdp, 256).to(bf)
    bias = rn(3 * heads * hdp) * 0.3
    assert _repeat(lambda: ops.qkv_attention(x, w, bias, B, S, heads, hdp, hd ** -0.5), 40) == 0


def test_vit_block_tail_is_deterministic_at_bench_size():
    """mlp_rows at C = 384 (24-fragment ring slots, LayerNorm variant): 30 bench-size launches bit-identical and finite."""
    from fewshot_vit_amd.engine import ops
    bf = torch.bfloat16
    M, C, KC, HID = 3200 * 197, 384, 384, 1536
    g = torch.Generator(device='cuda').manual_seed(11)
    rn = lambda *s: torch.randn(*s, device='cuda', generator=g)
    x, ctx = rn(M, C).to(bf), rn(M, KC).to(bf)
    wp = (rn(C, KC) / math.sqrt(KC)).to(bf)
    w1, w2 = (rn(HID, C) / math.sqrt(C)).to(bf), (rn(C, HID) / math.sqrt(HID)).to(bf)
    bp, b1, b2 = rn(C) * 0.3, rn(HID) * 0.3, rn(C) * 0.3
    assert _repeat(lambda: ops.vit_block_tail(x, ctx, wp, bp, w1, b1, w2, b2), 30) == 0


@pytest.mark.parametrize('C,N,M', [(384, 1152, 3200 * 197), (512, 1728, 12800 * 25)])
def test_ln_linear_rows_is_deterministic_at_bench_size(C, N, M):
    """ln_gemm_rows (two workgroups per CU on one LDS-DMA ring each): 40 bench-size launches bit-identical and finite."""
    from fewshot_vit_amd.engine import ops
    bf = torch.bfloat16
    g = torch.Generator(device='cuda').manual_seed(C)
    rn = lambda *s: torch.randn(*s, device='cuda', generator=g)
    x = rn(M, C).to(bf)
    w = (rn(N, C) / math.sqrt(C)).to(bf)
    b = rn(N) * 0.3
    assert _repeat(lambda: ops.ln_linear_rows(x, w, b), 40) == 0


def test_qkv_attention_rows_is_deterministic_at_bench_size():
    """qkv_attn_rows (stage-3 geometry, 12800 images of 25 tokens): 40 launches bit-identical and finite."""
    from fewshot_vit_amd.engine import ops
    bf = torch.bfloat16
    B, S, C, heads, hd, hdp = 12800, 25, 512, 6, 85, 96
    g = torch.Generator(device='cuda').manual_seed(9)
    rn = lambda *s: torch.randn(*s, device='cuda', generator=g)
    x = rn(B * S, C).to(bf)
    w = torch.zeros(3, heads, hdp, C, device='cuda')
    w[:, :, :hd] = rn(3, heads, hd, C) / math.sqrt(C)
    w = w.reshape(3 * heads * hdp, C).to(bf)
    bias = rn(3 * heads * hdp) * 0.3
    assert _repeat(lambda: ops.qkv_attention(x, w, bias, B, S, heads, hdp, hd ** -0.5), 40) == 0


def test_vit_ln_qkv_attention_rows_is_deterministic_at_bench_size():
    """vit_attn_rows (DeiT-S/16 geometry, 3200 images of 197 tokens: 8 waves per image, K / V^T fragments through LDS behind the ring barriers, the
    next image's rows in flight under the last head): 30 launches bit-identical and finite - a missed barrier or wait shows up as a flicker."""
    from fewshot_vit_amd.engine import ops
    bf = torch.bfloat16
    B, S, C, heads, hd = 3200, 197, 384, 6, 64
    g = torch.Generator(device='cuda').manual_seed(11)
    rn = lambda *s: torch.randn(*s, device='cuda', generator=g)
    x = (rn(B * S, C) * 1.5 + 0.3).to(bf)
    w = (rn(3 * heads * hd, C) / math.sqrt(C)).to(bf)
    bias = rn(3 * heads * hd) * 0.3
    assert _repeat(lambda: ops.vit_ln_qkv_attention(x, w, bias, B, S, heads, hd, hd ** -0.5), 30) == 0


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_stage1_w4_block_is_deterministic_at_bench_size(dtype):
    """stage1_w4.hip orders its LDS hand-offs (h1 / h2 / x rings, 192-slot residual window) with two hand-written barriers per 64-pixel chunk and
    counted waits: 6400 images (40 000 chunks, 157 per workgroup) repeated 30 times must be bit-identical, and agree with the 16-wave ring kernel."""
    from fewshot_vit_amd.engine import ops
    from test_gpu_ops import pack_w
    g = torch.Generator().manual_seed(77)
    B = 6400
    x = torch.randn(B, 20, 20, 128, generator=g).to('cuda', dtype)
    w1 = torch.randn(256, 128, 1, 1, generator=g) / math.sqrt(128)
    w2 = torch.randn(256, 32, 3, 3, generator=g) / math.sqrt(288)
    w3 = torch.randn(128, 256, 1, 1, generator=g) / math.sqrt(256)
    b1 = (torch.randn(256, generator=g) * 0.2).cuda()
    args = (pack_w(w1, 1, dtype)[0].cuda(), b1, pack_w(w2, 8, dtype).cuda(), pack_w(w3, 1, dtype)[0].cuda())
    assert _repeat(lambda: ops.stage1_block_hw(x, *args), 30) == 0
    if dtype == torch.bfloat16:
        y, y0 = ops.stage1_block_hw(x, *args).float(), ops.stage1_block(x, *args).float()
        d = (y - y0).abs()
        # (the ring kernel rounds its hidden maps from gelu_sig of the fp32 pre-activation, the w4 kernel from the table GELU of the bf16-rounded one: 1.4e-3)
        assert d.max().item() <= 3e-2 * max(1.0, float(y0.abs().max())) and d.mean().item() <= 2e-3
