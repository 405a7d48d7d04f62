#!/usr/bin/env python3
"""Micro-benchmarks of single fsvit operators on the GPU (torch events on the current stream)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewshot_vit_amd.engine import ops

def pack_w(O, Ig, KH, groups, dtype):
    N = O // groups; K = KH * KH * Ig
    bke = 32 if dtype == torch.float32 else 64
    Kw = (K + bke - 1) // bke * bke
    return (torch.randn(groups, N, Kw, device='cuda') / math.sqrt(K)).to(dtype)

def time_it(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us

def conv_case(name, B, H, Cin, O, KH, s, p, groups, act, res, dtype=torch.bfloat16, bias=True):
    Ig = Cin // groups
    x = torch.randn(B, H, H, Cin, device='cuda').to(dtype)
    w = pack_w(O, Ig, KH, groups, dtype)
    b = torch.randn(O, device='cuda') if bias else None
    OH = (H + 2 * p - KH) // s + 1
    r = torch.randn(B, OH, OH, O, device='cuda').to(dtype) if res else None
    us = time_it(lambda: ops.conv_gemm(x, w, b, r, None, B, H, H, Ig, KH, KH, s, p, O // groups, groups, act, 0))
    fl = 2.0 * B * OH * OH * O * Ig * KH * KH
    print(f'{name:<34} {us:9.1f} us  {fl / us / 1e6:8.1f} TF/s')

if __name__ == '__main__':
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    for act in (0, 1, 2):
        conv_case(f's1.conv1 K128 N256 act{act}', B, 20, 128, 256, 1, 1, 0, 1, act, False)
    for act in (0, 1):
        conv_case(f's2.fc1 K256 N1024 act{act}', B, 10, 256, 1024, 1, 1, 0, 1, act, False)
    conv_case('s2.fc2 K1024 N256 res', B, 10, 1024, 256, 1, 1, 0, 1, 0, True, bias=False)
    conv_case('s3.fc1 K512 N2048 gelu', B, 5, 512, 2048, 1, 1, 0, 1, 1, False)
    conv_case('s3.fc2 K2048 N512 res', B, 5, 2048, 512, 1, 1, 0, 1, 0, True, bias=False)
    conv_case('stem.conv2 3x3 64->128', B, 40, 64, 128, 3, 1, 1, 1, 2, False)
    conv_case('stem.conv3 3x3 128->128', B, 40, 128, 128, 3, 1, 1, 1, 2, True)
    conv_case('s1.conv2 grouped 3x3', B, 20, 256, 256, 3, 1, 1, 8, 1, False, bias=False)
    for act in (0, 1):
        conv_case(f's1.conv2 grouped act{act}', B, 20, 256, 256, 3, 1, 1, 8, act, False, bias=False)
    conv_case('s1.conv3 K256 N128 res', B, 20, 256, 128, 1, 1, 0, 1, 0, True, bias=False)
    conv_case('s2.qkv K256 N1152', B, 10, 256, 1152, 1, 1, 0, 1, 0, False)
    conv_case('big 1x1 K2048 N2048 (M=B*400)', B, 20, 2048, 2048, 1, 1, 0, 1, 0, False, bias=False)
