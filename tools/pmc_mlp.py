#!/usr/bin/env python3
"""Runs the fused row Mlp (mlp_rows.hip) on the stage-2 (default) or stage-3 shape a few times (for rocprofv3 --pmc passes):
   python3 tools/pmc_mlp.py [launches] [2|3]"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fewshot_vit_amd import _lib
from fewshot_vit_amd.engine import _ptr, _stream_ptr

M, C, HID = (160000, 512, 2048) if len(sys.argv) > 2 and sys.argv[2] == '3' else (320000, 256, 1024)
x = torch.randn(M, C, device='cuda').bfloat16()
w1 = (torch.randn(HID, C, device='cuda') / math.sqrt(C)).bfloat16()
w2 = (torch.randn(C, HID, device='cuda') / math.sqrt(HID)).bfloat16()
b1 = torch.randn(HID, device='cuda') * 0.3
y = torch.empty_like(x)
lib = _lib.load()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    _lib.check(lib.fsvit_mlp_rows(_ptr(x), _ptr(y), _ptr(w1), C, _ptr(b1), _ptr(w2), HID, None, M, C, HID, _stream_ptr(x.device)))
torch.cuda.synchronize()
