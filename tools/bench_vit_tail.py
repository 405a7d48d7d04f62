"""Launches the fused DeiT block tail (fsvit_vit_block_tail, mlp_rows at C = 384) a few times for rocprofv3 kernel timing (tools/prof_kernel.sh):
python tools/bench_vit_tail.py [M [variant.so]]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fewshot_vit_amd import _lib            # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] else 6400 * 197
if len(sys.argv) > 2:
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
from fewshot_vit_amd.engine import ops      # noqa: E402
bf = torch.bfloat16
C, KC, HID = 384, 384, 1536
g = torch.Generator().manual_seed(1)
x = torch.randn(M, C, generator=g).to('cuda', bf)
ctx = torch.randn(M, KC, generator=g).to('cuda', bf)
wp = (torch.randn(C, KC, generator=g) / math.sqrt(KC)).to('cuda', bf)
w1 = (torch.randn(HID, C, generator=g) / math.sqrt(C)).to('cuda', bf)
w2 = (torch.randn(C, HID, generator=g) / math.sqrt(HID)).to('cuda', bf)
bp, b1, b2 = torch.randn(C, generator=g).cuda(), torch.randn(HID, generator=g).cuda(), torch.randn(C, generator=g).cuda()
for _ in range(5):
    y = ops.vit_block_tail(x, ctx, wp, bp, w1, b1, w2, b2)
torch.cuda.synchronize()
print('ok', float(y.float().abs().mean()))
