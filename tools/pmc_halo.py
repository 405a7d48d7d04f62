#!/usr/bin/env python3
"""Runs the stem 3x3 convs (conv3x3_halo.hip: conv2 64 -> 128, a conv3-like 128 -> 128, and the fused conv3 + downsample + pool tail, 40x40)
a few times - for rocprofv3 passes and for the -DH_CLK cycle-breakdown variant:  python tools/pmc_halo.py [images] [libfsvit variant .so]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewshot_vit_amd import _lib
if len(sys.argv) > 2: _lib.LIB_PATH = os.path.abspath(sys.argv[2])
from fewshot_vit_amd.engine import ops
from bench_ops import pack_w, time_it
bf = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6400
reps = 5
for cin in (64, 128):
    x = torch.randn(B, 40, 40, cin, device='cuda').to(bf)
    w = pack_w(128, cin, 3, 1, bf)
    b = torch.randn(128, device='cuda')
    us = time_it(lambda: ops.conv_gemm(x, w, b, None, None, B, 40, 40, cin, 3, 3, 1, 1, 128, 1, 2, 0), n=reps)
    print(f'conv3x3 {cin}->128  {us:9.1f} us  {2.0 * B * 1600 * 128 * 9 * cin / us / 1e6:8.1f} TF/s')
x = torch.randn(B, 40, 40, 128, device='cuda').to(bf)
wt = (torch.randn(128, 9 * 128 + 64, device='cuda') / 34.0).to(bf)
x2 = torch.randn(B * 1600, 32, device='cuda').to(bf)
pos = torch.randn(400, 128, device='cuda')
b = torch.randn(128, device='cuda')
us = time_it(lambda: ops.conv_stem_tail(x, wt, b, pos, x2, 32), n=reps)
print(f'stem tail (conv3 + downsample + pool + pos)  {us:9.1f} us  {2.0 * B * 1600 * 128 * (9 * 128 + 27) / us / 1e6:8.1f} TF/s')
torch.cuda.synchronize()
