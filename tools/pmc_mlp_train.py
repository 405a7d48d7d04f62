#!/usr/bin/env python3
"""Runs the row-wise training Mlp (mlp_train.hip) forward + data gradient on the stage-2 and stage-3 shapes of the 800-image step a few times
(for rocprofv3 --kernel-trace / --pmc passes):   python3 tools/pmc_mlp_train.py [launches]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewshot_vit_amd.engine import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for (M, C, HID, rpi) in ((80000, 256, 1024, 100), (20000, 512, 2048, 25)):
    g = torch.Generator(device='cuda').manual_seed(1)
    xa = torch.randn(M, C, device='cuda', generator=g).bfloat16()
    w1 = torch.randn(HID, C, device='cuda', generator=g) / C ** 0.5
    w2 = torch.randn(C, HID, device='cuda', generator=g) / HID ** 0.5
    sa = 0.5 + torch.rand(C, device='cuda', generator=g)
    sb = torch.randn(C, device='cuda', generator=g) * 0.3
    for _ in range(n):
        out, xn, h, gp = ops.mlp_train_forward(xa, w1, w2, sa, sb, None, rpi)
        dh, dxn = ops.mlp_train_backward(out, w1, w2, gp)
torch.cuda.synchronize()
