import sys, math, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from fewshot_vit_amd.engine import ops
from test_gpu_ops import pack_w, q
dtype = torch.bfloat16
def tail(B):
    g = torch.Generator().manual_seed(1234 + B)
    H = W = 40; Cin = N = 128; bke = 64
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    img = q(torch.randn(B, 3, 2 * H, 2 * W, generator=g), dtype)
    w3 = q(torch.randn(N, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin), dtype)
    wd = q(torch.randn(N, 3, 3, 3, generator=g) / math.sqrt(27), dtype)
    bias = torch.randn(N, generator=g) * 0.3
    pos = torch.randn((H // 2) * (W // 2), N, generator=g) * 0.2
    ref = F.conv2d(x, w3, None, padding=1) + F.conv2d(img, wd, None, stride=2, padding=1) + bias.view(1, -1, 1, 1)
    ref = F.max_pool2d(F.leaky_relu(ref, 0.1), 2) + pos.t().reshape(1, N, H // 2, W // 2)
    cols = F.unfold(img, 3, padding=1, stride=2).view(B, 3, 9, H * W).permute(0, 3, 2, 1).reshape(B * H * W, 27)
    x2 = torch.zeros(B * H * W, 32); x2[:, :27] = cols
    Kmain = 9 * Cin
    wp = torch.zeros(N, Kmain + bke)
    wp[:, :Kmain] = w3.permute(0, 2, 3, 1).reshape(N, Kmain)
    wp[:, Kmain:Kmain + 27] = wd.permute(0, 2, 3, 1).reshape(N, 27)
    y = ops.conv_stem_tail(x.permute(0, 2, 3, 1).contiguous().cuda().to(dtype), wp.cuda().to(dtype), bias.cuda(), pos.cuda(), x2.cuda().to(dtype), 32)
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs()
    # per (image, pooled-row-group of 4 = tile) max error
    e = err.amax(dim=1).view(B, 5, 4, 20).amax(dim=(2, 3))
    bad = (e > 0.1).nonzero().tolist()
    print('tail B=%d: max err %.3f; bad (image, tile) count %d of %d:' % (B, err.max(), len(bad), B * 5), bad[:40])
def plain(B, Cin):
    g = torch.Generator().manual_seed(99 + B)
    H = W = 40; N = 128
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = q(torch.randn(N, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin), dtype)
    bias = torch.randn(N, generator=g) * 0.3
    ref = F.leaky_relu(F.conv2d(x, w, bias, padding=1), 0.1)
    y = ops.conv_gemm(x.permute(0, 2, 3, 1).contiguous().cuda().to(dtype), pack_w(w, 1, dtype).cuda(), bias.cuda(), None, None, B, H, W, Cin, 3, 3, 1, 1, N, 1, 2, 0)
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs()
    e = err.amax(dim=1).view(B, 5, 8, 40).amax(dim=(2, 3))
    bad = (e > 0.1).nonzero().tolist()
    print('plain Cin=%d B=%d: max err %.3f; bad (image, tile) %d of %d:' % (Cin, B, err.max(), len(bad), B * 5), bad[:40])
for B in (3, 53, 110):
    tail(B)
for B in (3, 53, 110):
    plain(B, 64); plain(B, 128)
