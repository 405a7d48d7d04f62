"""Times fsvit_conv3x3_wgrad at the shapes of the 800-image training step:  python tools/bench_wgrad3x3.py [variant.so]"""
import os
import sys
import torch
sys.path.insert(0, '.')
from fewshot_vit_amd import _lib            # noqa: E402
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from fewshot_vit_amd.engine import ops      # noqa: E402

for name, (B, H, O, Ig, groups) in {'stage-1 grouped 256 -> 256 (g 8), 20 x 20': (800, 20, 256, 32, 8), 'stem conv2 64 -> 128, 40 x 40': (800, 40, 128, 64, 1),
                                    'stem conv3 128 -> 128, 40 x 40': (800, 40, 128, 128, 1)}.items():
    x = torch.randn(B, H, H, groups * Ig, device='cuda', dtype=torch.bfloat16)
    dz = torch.randn(B, H, H, O, device='cuda', dtype=torch.bfloat16)
    for _ in range(3):
        ops.conv3x3_wgrad(x, dz, O, Ig, groups)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv3x3_wgrad(x, dz, O, Ig, groups)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gb = (x.numel() + dz.numel()) * 2 / 1e9
    fl = 2.0 * B * H * H * O * Ig * 9
    print(f'wgrad3x3 {name}: {ms * 1e3:.1f} us (incl. finalize), {gb / ms:.2f} TB/s of unique operand bytes, {fl / ms / 1e9:.0f} TFLOP/s')
