"""Times fsvit_attention_backward at the Visformer head shapes of an 800-image training step:  python tools/bench_attn_bwd.py [f32|bf16] [variant .so]"""
import sys
import torch
import os
sys.path.insert(0, '.')
from fewshot_vit_amd import _lib            # noqa: E402
if len(sys.argv) > 2:                       # a variant library
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
from fewshot_vit_amd.engine import ops      # noqa: E402

dt = torch.float32 if (len(sys.argv) < 2 or sys.argv[1] == 'f32') else torch.bfloat16
for S, hd, hdp, heads in ((100, 42, 48, 6), (25, 85, 96, 6)):
    if dt == torch.bfloat16 and hdp % 32:
        hdp = (hdp + 31) // 32 * 32
    B = 800
    qkv = torch.zeros(B, S, 3, heads, hdp)
    qkv[..., :hd] = torch.randn(B, S, 3, heads, hd)
    dctx = torch.zeros(B, S, heads, hdp)
    dctx[..., :hd] = torch.randn(B, S, heads, hd)
    q, d = qkv.to(dt).cuda().view(B * S, -1), dctx.to(dt).cuda().view(B * S, -1)
    for _ in range(3):
        ops.attention_backward(q, d, B, S, heads, hd, hdp, hd ** -0.5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.attention_backward(q, d, B, S, heads, hd, hdp, hd ** -0.5)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = B * heads * 5 * 2.0 * S * S * hd
    print(f'attention backward {dt} B={B} S={S} hd={hd}: {ms * 1e3:.1f} us, {fl / ms / 1e9:.1f} TFLOP/s')
