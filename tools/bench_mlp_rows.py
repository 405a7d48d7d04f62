#!/usr/bin/env python3
"""Times the fused row Mlp (mlp_rows.hip) on the stage-2 / stage-3 shapes of the bench step and checks it against torch:
   python3 tools/bench_mlp_rows.py [path/to/libfsvit_variant.so]"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from fewshot_vit_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from fewshot_vit_amd.engine import _ptr, _stream_ptr
lib = _lib.load()
bf = torch.bfloat16


def case(name, M, C, HID):
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(M, C, device='cuda', generator=g).to(bf)
    w1 = (torch.randn(HID, C, device='cuda', generator=g) / math.sqrt(C)).to(bf)
    w2 = (torch.randn(C, HID, device='cuda', generator=g) / math.sqrt(HID)).to(bf)
    b1 = torch.randn(HID, device='cuda', generator=g) * 0.3
    y = torch.empty_like(x)
    run = lambda: _lib.check(lib.fsvit_mlp_rows(_ptr(x), _ptr(y), _ptr(w1), C, _ptr(b1), _ptr(w2), HID, None, M, C, HID, _stream_ptr(x.device)))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    n = 4096
    h = F.gelu(x[:n].float() @ w1.float().t() + b1).to(bf).float()
    ref = x[:n].float() + h @ w2.float().t()
    err = (y[:n].float() - ref).abs().max().item()
    tail = (y[-n:].float() - (x[-n:].float() + F.gelu(x[-n:].float() @ w1.float().t() + b1).to(bf).float() @ w2.float().t())).abs().max().item()
    print(f'{name:<22} {us:9.1f} us  {4.0 * M * C * HID / us / 1e6:8.1f} TF/s   max|err| head {err:.3e} tail {tail:.3e}')


case('stage2 C256 H1024', 640000, 256, 1024)
case('stage3 C512 H2048', 160000, 512, 2048)
