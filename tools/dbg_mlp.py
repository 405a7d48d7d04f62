import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from fewshot_vit_amd import _lib
if len(sys.argv) > 4: _lib.LIB_PATH = os.path.abspath(sys.argv[4])
from fewshot_vit_amd.engine import ops
bf = torch.bfloat16
C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M = int(sys.argv[2]) if len(sys.argv) > 2 else 256
HID = 4 * C
g = torch.Generator().manual_seed(1)
q = lambda t: t.to(bf).float()
x = q(torch.randn(M, C, generator=g)); w1 = q(torch.randn(HID, C, generator=g) / math.sqrt(C)); b1 = torch.randn(HID, generator=g) * 0.3
w2 = q(torch.randn(C, HID, generator=g) / math.sqrt(HID))
hdn = q(F.gelu(x @ w1.t() + b1)); ref = x + hdn @ w2.t()
y = ops.mlp_rows(x.to('cuda', bf), w1.to('cuda', bf), b1.cuda(), w2.to('cuda', bf), None).float().cpu()
err = (y - ref).abs()
bad = ~(err < (float(sys.argv[3]) if len(sys.argv) > 3 else 0.2))
print("mean err", err.mean().item(), "max", err.max().item())
print('bad fraction', bad.float().mean().item(), 'nonfinite', (~torch.isfinite(y)).float().mean().item())
print('bad rows (count per 32-row block):', bad.any(1).view(-1, 32).sum(1).tolist()[:16])
print('bad cols (count per 32-col block):', bad.any(0).view(-1, 32).sum(1).tolist())
r = bad.any(1).nonzero().flatten().tolist()[:8]; print('first bad rows', r)
if r:
    c = bad[r[0]].nonzero().flatten().tolist(); print('bad cols in row', r[0], c[:40], len(c))
    print('y', y[r[0], c[:6]].tolist(), 'ref', ref[r[0], c[:6]].tolist())
# hidden-only probe: w2 = identity-like to expose hidden chunk errors
