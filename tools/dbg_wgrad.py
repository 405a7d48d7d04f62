import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fewshot_vit_amd.engine import ops
B, H, W, O, Ig, groups = [int(a) for a in sys.argv[1:7]] if len(sys.argv) > 6 else (1, 12, 16, 128, 64, 1)
dtype = torch.bfloat16
g = torch.Generator().manual_seed(1)
x = torch.randn(B, groups * Ig, H, W, generator=g).to(dtype).float()
dz = (torch.randn(B, O, H, W, generator=g) * 0.1).to(dtype).float()
wref = torch.zeros(O, Ig, 3, 3, requires_grad=True)
F.conv2d(x, wref, padding=1, groups=groups).backward(dz)
ref = wref.grad
for it in range(3):
    got = ops.conv3x3_wgrad(x.permute(0, 2, 3, 1).contiguous().to('cuda', dtype), dz.permute(0, 2, 3, 1).contiguous().to('cuda', dtype), O, Ig, groups).cpu()
    e = (got - ref).abs()
    print('run', it, 'max err', e.max().item(), 'per tap', [round(e[:, :, t // 3, t % 3].max().item(), 4) for t in range(9)])
    print('   per o-block', [round(e[o:o + 32].max().item(), 4) for o in range(0, O, 32)], 'per i-block', [round(e[:, i:i + 32].max().item(), 4) for i in range(0, Ig, 32)])
    bad = (e > 1e-2).nonzero()
    print('   bad count', len(bad), 'first', bad[:6].tolist())
