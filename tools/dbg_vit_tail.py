"""Stage-wise debug of the fused DeiT block tail (mlp_rows LN variant): zeroes parts of the problem to localise an error."""
import math, os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fewshot_vit_amd.engine import ops
bf = torch.bfloat16
q = lambda t: t.to(bf).float()
C, KC, HID, eps, M = 384, 384, 1536, 1e-6, 256
g = torch.Generator().manual_seed(1)
def run(tag, zero_ctx=False, zero_w2=False, zero_w1=False, zero_b=False):
    x = q(torch.randn(M, C, generator=g) * 2.0 + 0.5)
    ctx = q(torch.randn(M, KC, generator=g)) * (0 if zero_ctx else 1)
    wp = q(torch.randn(C, KC, generator=g) / math.sqrt(KC))
    bp = torch.randn(C, generator=g) * 0.3 * (0 if zero_b else 1)
    w1 = q(torch.randn(HID, C, generator=g) / math.sqrt(C)) * (0 if zero_w1 else 1)
    b1 = torch.randn(HID, generator=g) * 0.3
    w2 = q(torch.randn(C, HID, generator=g) / math.sqrt(HID)) * (0 if zero_w2 else 1)
    b2 = torch.randn(C, generator=g) * 0.3 * (0 if zero_b else 1)
    x1 = q(x + ctx @ wp.t() + bp)
    xn = q(F.layer_norm(x1, (C,), eps=eps))
    ref = x1 + q(F.gelu(xn @ w1.t() + b1)) @ w2.t() + b2
    y = ops.vit_block_tail(x.to('cuda', bf), ctx.to('cuda', bf), wp.to('cuda', bf), bp.cuda(), w1.to('cuda', bf), b1.cuda(), w2.to('cuda', bf), b2.cuda(), eps=eps)
    torch.cuda.synchronize()
    err = (y.float().cpu() - ref).abs()
    print(f'{tag:28s} max {err.max().item():.4g} mean {err.mean().item():.4g} nan {int(torch.isnan(y).sum())} | rows with err>0.1: {(err.max(1).values > 0.1).sum().item()} cols: {(err.max(0).values > 0.1).sum().item()}')
    bad = (err.max(0).values > 0.1).nonzero().flatten().tolist()
    print('   bad cols', bad[:40])
    if bad:
        c = bad[0]
        print('   row0..5 got', y.float().cpu()[:6, c].tolist(), 'want', ref[:6, c].tolist(), 'x', x[:6, c].tolist())
    return err
run('w2=0,ctx=0,b=0 (y=x)', zero_ctx=True, zero_w2=True, zero_b=True)
run('w2=0,ctx=0 (y=x+bp+b2)', zero_ctx=True, zero_w2=True)
run('w2=0 (proj)', zero_w2=True)
run('w1=0,ctx=0 (gelu(b1) W2)', zero_ctx=True, zero_w1=True)
run('ctx=0 (ln+mlp)', zero_ctx=True)
e = run('full')
print(e.max(1).values[:40])
