import math, sys, os, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch.nn.functional as F
from fewshot_vit_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from fewshot_vit_amd.engine import ops
from test_gpu_ops import pack_w, q
for B, HW, dtype in ((5, 20, torch.bfloat16), (130, 20, torch.bfloat16), (7, 10, torch.bfloat16)):
    g = torch.Generator().manual_seed(1000 * B + HW)
    x = q(torch.randn(B, 128, HW, HW, generator=g), dtype)
    w1 = q(torch.randn(256, 128, 1, 1, generator=g) / math.sqrt(128), dtype)
    b1 = torch.randn(256, generator=g) * 0.2
    w2 = q(torch.randn(256, 32, 3, 3, generator=g) / math.sqrt(288), dtype)
    w3 = q(torch.randn(128, 256, 1, 1, generator=g) / math.sqrt(256), dtype)
    h1 = q(F.gelu(F.conv2d(x, w1, b1)), dtype)
    h2 = q(F.gelu(F.conv2d(h1, w2, padding=1, groups=8)), dtype)
    ref = x + F.conv2d(h2, w3)
    xd = x.permute(0, 2, 3, 1).contiguous().to('cuda', dtype)
    args = (pack_w(w1, 1, dtype)[0].cuda(), b1.cuda(), pack_w(w2, 8, dtype).cuda(), pack_w(w3, 1, dtype)[0].cuda())
    y = ops.stage1_block_hw(xd, *args)
    got = y.float().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs()
    same = all(torch.equal(y, ops.stage1_block_hw(xd, *args)) for _ in range(20))
    line = f'B={B} HW={HW}: vs fp32 torch max {err.max():.3e} mean {err.mean():.3e}; deterministic {same}'
    if HW == 20:
        y0 = ops.stage1_block(xd, *args).float().cpu().permute(0, 3, 1, 2)
        line += f'; vs ring kernel max {(y0-got).abs().max():.3e} mean {(y0-got).abs().mean():.3e}; ring vs torch mean {(y0-ref).abs().mean():.3e}'
    print(line)
