"""Times the fused stage-1 block operators: fsvit_stage1_block_hw (the engines' launch: stage1_w4.hip) or,
with BENCH_STAGE1_RING=0, fsvit_stage1_block (always the 16-wave kernel of stage1_ring.hip).
python tools/bench_stage1.py [images [variant.so]]"""
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
RING = os.environ.get('BENCH_STAGE1_RING', '1') != '0'


def call():
    if RING:
        _lib.check(lib.fsvit_stage1_block_hw(_ptr(x), _ptr(y), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(w3), B, 20, 20, _lib.BF16, st))
    else:
        _lib.check(lib.fsvit_stage1_block(_ptr(x), _ptr(y), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(w3), B, st))

st = _stream_ptr(x.device)
for _ in range(2):
    call()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
N = 10
for _ in range(N):
    call()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
if len(sys.argv) > 2 and RING:           # variant: results must equal the shipped library's (same arithmetic, different schedule)
    base = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), 'libfsvit.so'))
    base.fsvit_stage1_block_hw.restype = C.c_int
    base.fsvit_stage1_block_hw.argtypes = lib.fsvit_stage1_block_hw.argtypes
    y0 = torch.empty_like(x)
    assert base.fsvit_stage1_block_hw(_ptr(x), _ptr(y0), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(w3), B, 20, 20, _lib.BF16, st) == 0
    torch.cuda.synchronize()
    d = (y.float() - y0.float()).abs().max().item()
    print(f'variant vs shipped library: max |dy| = {d:.3e} ({"bit-identical" if torch.equal(y, y0) else "DIFFERENT"})')
    for hw, bb in ((16, 7), (10, 33), (4, 5)):
        xs = torch.randn(bb, hw, hw, 128, generator=g).to('cuda', bf)
        ya, yb = torch.empty_like(xs), torch.empty_like(xs)
        assert lib.fsvit_stage1_block_hw(_ptr(xs), _ptr(ya), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(w3), bb, hw, hw, _lib.BF16, st) == 0
        assert base.fsvit_stage1_block_hw(_ptr(xs), _ptr(yb), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(w3), bb, hw, hw, _lib.BF16, st) == 0
        torch.cuda.synchronize()
        print(f'  {bb} x {hw} x {hw}: {"bit-identical" if torch.equal(ya, yb) else "DIFFERENT max %.3e" % (ya.float() - yb.float()).abs().max().item()}')
fl = 2.0 * B * 400 * (256 * 128 + 256 * 9 * 32 + 128 * 256)
print(f'stage1 block ({"ring" if RING else "half-image"} kernel), {B} images: {dt * 1e3:.3f} ms per call, {fl / dt / 1e12:.1f} TFLOP/s')
