#!/usr/bin/env python3
"""Runs the stem 3x3 convs (conv3x3_halo.hip: conv2 64 -> 128 and a conv3-like 128 -> 128, 40x40) a few times for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewshot_vit_amd.engine import ops
from bench_ops import pack_w
bf = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
for cin in (64, 128):
    x = torch.randn(B, 40, 40, cin, device='cuda').to(bf)
    w = pack_w(128, cin, 3, 1, bf)
    b = torch.randn(128, device='cuda')
    for _ in range(3):
        ops.conv_gemm(x, w, b, None, None, B, 40, 40, cin, 3, 3, 1, 1, 128, 1, 2, 0)
torch.cuda.synchronize()
