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
# the rows kernels of the end of round 2 (24-fragment slots, two workgroups per CU, one image per wave)
M, C, KC, HID = 3200 * 197, 384, 384, 1536
x = rn(M, C).to(bf); ctx = rn(M, KC).to(bf)
wp = (rn(C, KC) / math.sqrt(KC)).to(bf)
w1 = (rn(HID, C) / math.sqrt(C)).to(bf); w2 = (rn(C, HID) / math.sqrt(HID)).to(bf)
bp, b1, b2 = rn(C) * 0.3, rn(HID) * 0.3, rn(C) * 0.3
soak('vit_block_tail C=384', lambda: ops.vit_block_tail(x, ctx, wp, bp, w1, b1, w2, b2))
wq = (rn(1152, C) / math.sqrt(C)).to(bf); bq = rn(1152) * 0.3
soak('ln_linear_rows C=384', lambda: ops.ln_linear_rows(x, wq, bq))
B, S, heads, hd, hdp = 12800, 25, 6, 85, 96
x5 = rn(B * S, 512).to(bf)
w5 = torch.zeros(3, heads, hdp, 512, device='cuda')
w5[:, :, :hd] = rn(3, heads, hd, 512) / math.sqrt(512)
w5 = w5.reshape(3 * heads * hdp, 512).to(bf)
b5 = rn(3 * heads * hdp) * 0.3
soak('linear_rows C=512', lambda: ops.ln_linear_rows(x5, w5, b5))
soak('qkv_attention rows S=25', lambda: ops.qkv_attention(x5, w5, b5, B, S, heads, hdp, hd ** -0.5))
xe = rn(6400, 20, 20, 128).to(bf)
we = (rn(256, 512) / math.sqrt(512)).to(bf)
pe = rn(100, 256) * 0.5
soak('patch_embed2x2', lambda: ops.patch_embed2x2(xe, we, rn(256) * 0 + 0.1, pe))
print('SOAK', 'FAILED' if bad else 'OK')
sys.exit(1 if bad else 0)
