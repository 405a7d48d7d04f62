"""Row-wise Mlp of the stage-2 / 3 blocks in the meta-tuning step (csrc/mlp_train.hip) through the C-ABI, against the unfused math in torch
(test_phase/models/visformer.py:146-150, :262 in train mode and its autograd): operands rounded to bf16 where the kernel rounds them, fp32 sums."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16).float()


def _mk(M, C, hid, seed):
    g = torch.Generator().manual_seed(seed)
    xa = (torch.randn(M, C, generator=g) * 1.5).to(torch.bfloat16)
    w1 = torch.randn(hid, C, generator=g) * (1.0 / C ** 0.5)
    w2 = torch.randn(C, hid, generator=g) * (1.0 / hid ** 0.5)
    sa = 0.5 + torch.rand(C, generator=g)
    sb = torch.randn(C, generator=g) * 0.3
    return xa, w1, w2, sa, sb


# (M): whole tiles, a ragged tail whose last waves hold no valid row, more tiles than workgroups (the persistent loop and the wrapped prefetch)
@pytest.mark.parametrize('C,hid,M,rows_per_img', [(256, 1024, 256, 100), (256, 1024, 300, 100), (256, 1024, 40000, 100), (512, 2048, 20000, 25),
                                                   (512, 2048, 33, 25), (512, 2048, 36001, 25)])
def test_mlp_train_forward_matches_unfused_math(C, hid, M, rows_per_img):
    from fewshot_vit_amd.engine import ops
    dev = torch.device('cuda:0')
    xa, w1, w2, sa, sb = _mk(M, C, hid, 100 + M)
    n_img = (M + rows_per_img - 1) // rows_per_img
    scale = torch.where(torch.rand(n_img, generator=torch.Generator().manual_seed(5)) < 0.5, 0.0, 2.0)
    out, xn, h, g = ops.mlp_train_forward(xa.to(dev), w1.to(dev), w2.to(dev), sa.to(dev), sb.to(dev), scale.to(dev), rows_per_img)
    torch.cuda.synchronize()
    x = xa.to(dev).float()
    w1d, w2d, sad, sbd = w1.to(dev), w2.to(dev), sa.to(dev), sb.to(dev)
    z1 = x @ _bf(w1d * sad[None, :]).t() + (w1d @ sbd)[None, :]
    z1 = z1.detach().requires_grad_(True)
    href = torch.nn.functional.gelu(z1)
    gref, = torch.autograd.grad(href.sum(), z1)
    y = _bf(href.detach()) @ _bf(w2d).t()
    sc = scale.to(dev).repeat_interleave(rows_per_img)[:M, None]
    oref = x + sc * y
    assert torch.isfinite(out.float()).all()
    # bf16 outputs: half an ulp of the value + the 2.6e-5 of the sigmoid-form GELU + the fp32 summation order
    assert (h.float() - href.detach()).abs().max().item() <= 2e-2 * href.detach().abs().max().item() / 2 + 1e-3, 'h'
    assert ((h.float() - href.detach()).abs() <= 4e-3 * href.detach().abs() + 2e-3).all(), 'h elementwise'
    assert ((g.float() - gref).abs() <= 4e-3 * gref.abs() + 2e-3).all(), 'g'
    assert ((xn.float() - (x * sad + sbd)).abs() <= 4e-3 * (x * sad + sbd).abs() + 1e-6).all(), 'xn'
    err = (out.float() - oref).abs()
    assert (err <= 4e-3 * oref.abs() + 2e-2).all(), ('out', err.max().item())
    dropped = sc[:, 0] == 0
    assert torch.equal(out[dropped], xa.to(dev)[dropped]), 'a dropped image passes its rows through unchanged'


@pytest.mark.parametrize('C,hid,M', [(256, 1024, 256), (256, 1024, 300), (256, 1024, 40000), (512, 2048, 20000), (512, 2048, 33), (512, 2048, 36001)])
def test_mlp_train_backward_matches_unfused_math(C, hid, M):
    from fewshot_vit_amd.engine import ops
    dev = torch.device('cuda:0')
    _, w1, w2, _, _ = _mk(8, C, hid, 7)
    gen = torch.Generator().manual_seed(900 + M)
    dz = torch.randn(M, C, generator=gen).to(torch.bfloat16).to(dev)
    g = (torch.rand(M, hid, generator=gen) * 1.2 - 0.1).to(torch.bfloat16).to(dev)
    dh, dxn = ops.mlp_train_backward(dz, w1.to(dev), w2.to(dev), g)
    torch.cuda.synchronize()
    w1d, w2d = w1.to(dev), w2.to(dev)
    dh_ref = (dz.float() @ _bf(w2d)) * g.float()
    dxn_ref = _bf(dh_ref) @ _bf(w1d)
    assert torch.isfinite(dh.float()).all() and torch.isfinite(dxn.float()).all()
    assert ((dh.float() - dh_ref).abs() <= 4e-3 * dh_ref.abs() + 1e-3).all(), ('dh', (dh.float() - dh_ref).abs().max().item())
    e = (dxn.float() - dxn_ref).abs()
    assert (e <= 4e-3 * dxn_ref.abs() + 2e-2).all(), ('dxn', e.max().item())


@pytest.mark.parametrize('C,hid,M', [(256, 1024, 40000), (512, 2048, 36001)])
def test_mlp_train_is_deterministic_and_row_local(C, hid, M):
    """A row's result does not depend on the launch it is part of (tile position, workgroup, tail handling): rows of a multi-tile-per-workgroup launch
    equal the same rows computed in a 256-row launch, bit for bit, and ten identical launches agree bit for bit in both directions (the weight streams
    run on counted vmcnt waits across barriers and tiles: a race shows up as a rare wrong tile, not as a crash)."""
    from fewshot_vit_amd.engine import ops
    dev = torch.device('cuda:0')
    xa, w1, w2, sa, sb = _mk(M, C, hid, 3)
    a = [t.to(dev) for t in (xa, w1, w2, sa, sb)]
    o1 = ops.mlp_train_forward(*a)
    b1 = ops.mlp_train_backward(o1[0], a[1], a[2], o1[3])
    for _ in range(10):
        o2 = ops.mlp_train_forward(*a)
        for u, v in zip(o1, o2):
            assert torch.equal(u, v)
        b2 = ops.mlp_train_backward(o1[0], a[1], a[2], o1[3])
        for u, v in zip(b1, b2):
            assert torch.equal(u, v)
    lo = M - 1000
    o3 = ops.mlp_train_forward(a[0][lo:lo + 256], *a[1:])
    for u, v in zip(o1, o3):
        assert torch.equal(u[lo:lo + 256], v)
    b3 = ops.mlp_train_backward(o1[0][lo:lo + 256], a[1], a[2], o1[3][lo:lo + 256])
    for u, v in zip(b1, b3):
        assert torch.equal(u[lo:lo + 256], v)
