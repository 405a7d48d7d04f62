#!/usr/bin/env python3
"""Runs a few representative conv_gemm shapes once each (for rocprofv3 --pmc passes)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewshot_vit_amd.engine import ops
from bench_ops import pack_w

def run(B, H, Cin, O, KH, s, p, groups, act, res, bias=True, n=3):
    dtype = torch.bfloat16
    Ig = Cin // groups
    x = torch.randn(B, H, H, Cin, device='cuda').to(dtype)
    w = pack_w(O, Ig, KH, groups, dtype)
    b = torch.randn(O, device='cuda') if bias else None
    OH = (H + 2 * p - KH) // s + 1
    r = torch.randn(B, OH, OH, O, device='cuda').to(dtype) if res else None
    for _ in range(n):
        ops.conv_gemm(x, w, b, r, None, B, H, H, Ig, KH, KH, s, p, O // groups, groups, act, 0)
    torch.cuda.synchronize()

if __name__ == '__main__':
    run(800, 40, 128, 128, 3, 1, 1, 1, 2, False)      # stem.conv3-like: M=1.28M, K=1152, N=128
    run(1600, 5, 512, 2048, 1, 1, 0, 1, 1, False)     # s3.fc1: M=40000, K=512, N=2048
    run(1600, 10, 256, 1024, 1, 1, 0, 1, 1, False)    # s2.fc1
