#!/usr/bin/env python3
"""Runs the stage-1 block operator (fsvit_stage1_block_hw: stage1_w4.hip) a few times for rocprofv3 --pmc passes."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewshot_vit_amd import _lib
if len(sys.argv) > 2: _lib.LIB_PATH = os.path.abspath(sys.argv[2])
from fewshot_vit_amd.engine import ops
from bench_ops import pack_w
bf = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3200
x = torch.randn(B, 20, 20, 128, device='cuda').to(bf)
w1 = pack_w(256, 128, 1, 1, bf)[0]; w2 = pack_w(256, 32, 3, 8, bf); w3 = pack_w(128, 256, 1, 1, bf)[0]
b1 = torch.randn(256, device='cuda')
import time
for _ in range(3):
    ops.stage1_block_hw(x, w1, b1, w2, w3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    ops.stage1_block_hw(x, w1, b1, w2, w3)
torch.cuda.synchronize()
print("stage1_block_hw B=%d: %.1f us per launch" % (B, (time.perf_counter() - t0) * 1e5))
