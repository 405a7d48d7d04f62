import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fewshot_vit_amd.engine import ops
B, H, W, O, Ig, groups = 1, 12, 16, 128, 64, 1
dtype = torch.bfloat16
# x[i][y][x] = y*16 + x + i/64 (exactly representable pieces): identify the pixel read per tap
yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
x = (yy * 16 + xx).float()[None, None].repeat(B, Ig, 1, 1)
dz = torch.zeros(B, O, H, W)
for (oy, ox) in [(5, 7), (0, 0), (11, 15), (3, 15), (4, 0)]:
    dz.zero_(); dz[0, 0, oy, ox] = 1.0
    got = ops.conv3x3_wgrad(x.permute(0, 2, 3, 1).contiguous().to('cuda', dtype), dz.permute(0, 2, 3, 1).contiguous().to('cuda', dtype), O, Ig, groups).cpu()
    exp = [[(oy + ky - 1) * 16 + ox + kx - 1 if 0 <= oy + ky - 1 < H and 0 <= ox + kx - 1 < W else 0 for kx in range(3)] for ky in range(3)]
    print('pixel', (oy, ox), 'm', oy * W + ox, 'got', got[0, 0].tolist(), 'expected', exp, ' i=40:', got[0, 40].tolist())
