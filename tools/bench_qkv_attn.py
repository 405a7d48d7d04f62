#!/usr/bin/env python3
"""Runs the fused qkv + attention operator (qkv_attn.hip) on the stage-2 shape of the bench step (for rocprofv3 kernel traces):
   python3 tools/bench_qkv_attn.py [images] [libfsvit variant .so]"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewshot_vit_amd import _lib
if len(sys.argv) > 2:
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
from fewshot_vit_amd.engine import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6400
S, C, heads, hd, hdp = 100, 256, 6, 42, 48
bf = torch.bfloat16
x = torch.randn(B * S, C, device='cuda').to(bf)
w = torch.zeros(3, heads, hdp, C, device='cuda')
w[:, :, :hd] = torch.randn(3, heads, hd, C, device='cuda') / math.sqrt(C)
w = w.reshape(3 * heads * hdp, C).to(bf)
bias = torch.randn(3 * heads * hdp, device='cuda') * 0.3
for _ in range(8):
    ops.qkv_attention(x, w, bias, B, S, heads, hdp, hd ** -0.5)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.qkv_attention(x, w, bias, B, S, heads, hdp, hd ** -0.5)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100.0
flop = B * (2.0 * S * C * 3 * heads * hd + 4.0 * S * S * hd * heads)
print('qkv_attn %d images: %.1f us per launch, %.1f algorithmic TFLOP/s%s' % (B, us, flop / us / 1e6, (' (' + os.path.basename(sys.argv[2]) + ')') if len(sys.argv) > 2 else ''))
