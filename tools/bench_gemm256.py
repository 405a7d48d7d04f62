#!/usr/bin/env python3
"""Micro-benchmark of the dense GEMM layers (gemm256.hip vs conv_gemm_v2): python tools/bench_gemm256.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewshot_vit_amd.engine import ops

SHAPES = [('s3.fc2', 80000, 512, 2048), ('s3.fc1', 80000, 2048, 512), ('s2.fc1', 320000, 1024, 256), ('s2.fc2', 320000, 256, 1024),
          ('s2.qkv', 320000, 1152, 256), ('s2.proj', 320000, 256, 384), ('deit.fc1', 157600, 1536, 384), ('sq4k', 4096, 4096, 4096)]

def run(name, M, N, K, reps=10):
    x = torch.randn(M, K, device='cuda').bfloat16().view(1, M, 1, K)
    w = (torch.randn(N, K, device='cuda') / K ** 0.5).bfloat16().view(1, N, K)
    f = lambda: ops.conv_gemm(x, w, None, None, None, 1, M, 1, K, 1, 1, 1, 0, N, 1, 0, False)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, 2.0 * M * N * K / ms / 1e9

if __name__ == '__main__':
    tag = 'var=%s g256=%s' % (os.environ.get('FSVIT_G256_VAR', '0'), os.environ.get('FSVIT_GEMM256', '1'))
    out = []
    for s in SHAPES:
        ms, tf = run(*s)
        out.append(f'{s[0]} {ms*1e3:.0f}us {tf:.0f}TF')
    print(tag, ' | '.join(out))
