"""Times the fused stage-1 block (fsvit_stage1_block through ops.stage1_block is synchronous: this calls the launcher through the engine op in a loop
with HIP events).  python tools/bench_stage1.py [images [variant.so]]      FSVIT_STAGE1_RING=1 selects the ring kernel."""
import math
import sys
import ctypes as C

import torch

sys.path.insert(0, '.')
from fewshot_vit_amd import _lib            # noqa: E402
from fewshot_vit_amd.engine import _ptr, _stream_ptr     # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
bf = torch.bfloat16
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 20, 20, 128, generator=g).to('cuda', bf)
w1 = (torch.randn(256, 128, generator=g) / math.sqrt(128)).to('cuda', bf)
b1 = (torch.randn(256, generator=g) * 0.2).cuda()
w2 = torch.zeros(256, 320)
w2[:, :288] = torch.randn(256, 288, generator=g) / math.sqrt(288)
w2 = w2.to('cuda', bf)
w3 = (torch.randn(128, 256, generator=g) / math.sqrt(256)).to('cuda', bf)
y = torch.empty_like(x)
import os
if len(sys.argv) > 2:                    # a variant library (tools/build_variant.sh)
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
lib = _lib.load()
st = _stream_ptr(x.device)
for _ in range(2):
    _lib.check(lib.fsvit_stage1_block(_ptr(x), _ptr(y), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(w3), B, st))
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
N = 10
for _ in range(N):
    _lib.check(lib.fsvit_stage1_block(_ptr(x), _ptr(y), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(w3), B, st))      # (the op synchronises the stream itself)
dt = (time.perf_counter() - t0) / N
fl = 2.0 * B * 400 * (256 * 128 + 256 * 9 * 32 + 128 * 256)
print(f'stage1 block, {B} images: {dt * 1e3:.3f} ms per call (incl. the weight-image pack + sync of the op), {fl / dt / 1e12:.1f} TFLOP/s')
