#!/usr/bin/env python3
"""Synthetic-episode difficulty calibration (GPU box): accuracy and bf16-vs-parity agreement as a function of the noise level, for
bench.py's generator (x = mu_c + noise * eps) and the driver's 'synthetic-episodes' dataset.  The reference's 5-way 5-shot accuracy on
miniImageNet is 83.25 %: the synthetic episodes should sit there, not at chance, or arg-max agreement measures coin flips."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from fewshot_vit_amd import models, synthetic  # noqa: E402

dev = torch.device('cuda', 0)
engs = {}
for num in ('bf16', 'parity'):
    m = models.make('meta-baseline', encoder='visformer_micro_80', encoder_args={'numerics': num})
    sd = synthetic.synthetic_checkpoint_sd({k: tuple(v.shape) for k, v in m.state_dict().items()}, calib='visformer_micro_80')
    m.load_state_dict(sd, strict=True)
    engs[num] = m.to(dev).eval().encoder.engine()
way, shot, query, E = 5, 5, 15, 64
label = torch.arange(way, device=dev).repeat_interleave(query)
for noise in [float(a) for a in sys.argv[1:]] or [0.5, 0.75, 1.0, 1.25, 1.5]:
    lg = {k: [] for k in engs}
    for i in range(4):
        g = torch.Generator(device=dev).manual_seed(777 + i)
        mu = torch.randn(E, way, 1, 3, 80, 80, device=dev, generator=g)
        x = mu + noise * torch.randn(E, way, shot + query, 3, 80, 80, device=dev, generator=g)
        xs, xq = x[:, :, :shot].contiguous(), x[:, :, shot:].contiguous().view(E, way * query, 3, 80, 80)
        for k, e in engs.items():
            lg[k].append(e.meta_baseline_forward(xs, xq, 10.0, 'cos'))
    b, p = torch.cat(lg['bf16']), torch.cat(lg['parity'])
    top2 = p.topk(2, dim=-1).values
    print('noise %.2f: acc bf16 %.4f parity %.4f | argmax agreement %.5f | max|dlogit| %.3f mean %.4f | median top-2 logit gap %.3f, frac gap<0.1: %.4f' % (
        noise, (b.argmax(-1) == label).float().mean().item(), (p.argmax(-1) == label).float().mean().item(),
        (b.argmax(-1) == p.argmax(-1)).float().mean().item(), (b - p).abs().max().item(), (b - p).abs().mean().item(),
        (top2[..., 0] - top2[..., 1]).median().item(), ((top2[..., 0] - top2[..., 1]) < 0.1).float().mean().item()))
