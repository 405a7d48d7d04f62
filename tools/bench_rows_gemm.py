"""Launches the row-wise GEMM operator (fsvit_ln_linear_rows) a few times for rocprofv3 kernel timing (tools/prof_kernel.sh):
python tools/bench_rows_gemm.py C [M [variant.so]]   (C = 384: LayerNorm + qkv of DeiT-S, N = 1152; C = 512: Visformer stage-3 qkv, N = 1728)"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fewshot_vit_amd import _lib            # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 384
N = 1152 if C == 384 else 1728
M = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] else (6400 * 197 if C == 384 else 12800 * 49)
if len(sys.argv) > 3:
    _lib.LIB_PATH = os.path.abspath(sys.argv[3])
from fewshot_vit_amd.engine import ops      # noqa: E402
bf = torch.bfloat16
g = torch.Generator().manual_seed(1)
x = torch.randn(M, C, generator=g).to('cuda', bf)
w = (torch.randn(N, C, generator=g) / math.sqrt(C)).to('cuda', bf)
b = torch.randn(N, generator=g).cuda()
for _ in range(6):
    y = ops.ln_linear_rows(x, w, b)
torch.cuda.synchronize()
print('ok', float(y.float().abs().mean()))
