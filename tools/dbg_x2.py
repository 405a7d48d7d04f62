"""debug: two-limb conv_gemm on structured inputs"""
import torch, sys
sys.path.insert(0, '.')
from fewshot_vit_amd.engine import ops
torch.manual_seed(0)
for numerics in ('bf16x2', 'f16x2', None):
    B, H, W, C, N = 1, 4, 4, 32, 32
    x = torch.zeros(B, H, W, C); w = torch.zeros(1, N, 32)
    for k in range(32):
        x[..., k] = 1.0 + k / 64.0
    for n in range(N):
        w[0, n, n] = 1.0 + 1.0 / 1024            # identity-like: y[m][n] = x[m][n] * w[n][n]
    wd = ops.x2_limbs(w, numerics).cuda() if numerics else w.cuda()
    y = ops.conv_gemm(x.cuda(), wd, None, None, None, B, H, W, C, 1, 1, 1, 0, N, 1, 0, 0, numerics=numerics)
    torch.cuda.synchronize()
    ref = x[0, 0, 0] * (1.0 + 1.0 / 1024)
    print(numerics, 'got', y[0, 0, 0, :8].cpu().tolist(), '\n   ref', ref[:8].tolist(), 'maxerr', (y[0, 0, 0].cpu() - ref).abs().max().item())
    xr = torch.randn(B, H, W, C); wr = torch.randn(1, N, 32) / 6
    wd = ops.x2_limbs(wr, numerics).cuda() if numerics else wr.cuda()
    y = ops.conv_gemm(xr.cuda(), wd, None, None, None, B, H, W, C, 1, 1, 1, 0, N, 1, 0, 0, numerics=numerics)
    ref = xr.double().reshape(-1, C) @ wr[0].double().t()
    print(numerics, 'random maxerr', (y.cpu().double().reshape(-1, N) - ref).abs().max().item())
