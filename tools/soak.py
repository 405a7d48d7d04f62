#!/usr/bin/env python3
"""Determinism soak of the kernels that order LDS-DMA data by counted vmcnt + barriers (mlp_rows, qkv_attn): the same inputs N times at
bench size, every result compared bit for bit with the first one.  A wait that is one short shows up as a rare differing tile.
   python3 tools/soak.py [iterations]"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewshot_vit_amd.engine import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bf = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(3)
rn = lambda *s: torch.randn(*s, device='cuda', generator=g)
bad = 0

def soak(name, fn):
    global bad
    ref = fn()
    torch.cuda.synchronize()
    assert torch.isfinite(ref.float()).all(), name
    n = 0
    for _ in range(N):
        out = fn()
        if not torch.equal(out, ref):
            n += 1
    torch.cuda.synchronize()
    print(f'{name:<28} {N} runs, {n} differ from the first', flush=True)
    bad += n

for C, KC, M in ((256, 288, 640000), (512, 576, 160000)):
    HID = 4 * C
    x = rn(M, C).to(bf); ctx = rn(M, KC).to(bf)
    wp = (rn(C, KC) / math.sqrt(KC)).to(bf)
    w1 = (rn(HID, C) / math.sqrt(C)).to(bf); w2 = (rn(C, HID) / math.sqrt(HID)).to(bf)
    b1 = rn(HID) * 0.3
    soak(f'proj_mlp_rows C={C}', lambda: ops.proj_mlp_rows(x, ctx, wp, w1, b1, w2, None))
    soak(f'mlp_rows C={C}', lambda: ops.mlp_rows(x, w1, b1, w2, None))
B, S, heads, hd, hdp = 6400, 100, 6, 42, 48
x = rn(B * S, 256).to(bf)
w = torch.zeros(3, heads, hdp, 256, device='cuda')
w[:, :, :hd] = rn(3, heads, hd, 256) / 16.0
w = w.reshape(3 * heads * hdp, 256).to(bf)
bias = rn(3 * heads * hdp) * 0.3
soak('qkv_attention', lambda: ops.qkv_attention(x, w, bias, B, S, heads, hdp, hd ** -0.5))
print('SOAK', 'FAILED' if bad else 'OK')
sys.exit(1 if bad else 0)
