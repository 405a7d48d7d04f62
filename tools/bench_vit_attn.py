"""Times fsvit_vit_ln_qkv_attention's kernel (mlp_rows.hip vit_attn_rows: norm1 + qkv + attention of a DeiT-S block in one launch) at the bench size.
python tools/bench_vit_attn.py [images [variant.so]]   (a variant built by tools/build_variant.sh, e.g. with -DVAR_DIAG=1 / 2 timing ablations)"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, '.')
from fewshot_vit_amd import _lib            # noqa: E402
from fewshot_vit_amd.engine import ops      # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
if len(sys.argv) > 2:
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
S, C, heads, hd = 197, 384, 6, 64
g = torch.Generator().manual_seed(1)
x = torch.randn(B * S, C, generator=g).to('cuda', torch.bfloat16)
w = (torch.randn(3 * heads * hd, C, generator=g) / math.sqrt(C)).to('cuda', torch.bfloat16)
bias = (torch.randn(3 * heads * hd, generator=g) * 0.3).cuda()
for _ in range(2):
    ops.vit_ln_qkv_attention(x, w, bias, B, S, heads, hd, hd ** -0.5)
torch.cuda.synchronize()
N = 5
t0 = time.perf_counter()
for _ in range(N):
    ops.vit_ln_qkv_attention(x, w, bias, B, S, heads, hd, hd ** -0.5)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N            # (includes the operator's per-call weight pack + sync: ~0.1 ms)
fl = 2.0 * B * S * 3 * C * C + 4.0 * B * heads * S * S * hd
print(f'vit_attn_rows, {B} images x {S} tokens: {dt * 1e3:.3f} ms per call, {fl / dt / 1e12:.1f} TFLOP/s (algorithmic)')
