"""Synthetic checkpoints and episodes (no dataset or published checkpoint is reachable offline).

* `procedural_state_dict(shapes)` fills every state-dict entry BY KEY NAME, so any party that
  knows the key/shape table regenerates bit-identical weights independent of module
  construction order (SURVEY.md 8c "full-size goldens with procedural weights").
* `load_bn_calibration(name)` returns BatchNorm running statistics that were calibrated once
  on a synthetic batch (tests/golden/make_golden.py) and are shipped as package data; without
  them a random-weight Visformer is numerically degenerate ("Hard parts").
* `synthetic_episodes(...)` draws class-structured episodes x = mu_c + 0.5*eps in the
  class-major order `fs.split_shot_query` expects.
"""
import math
import os
import zlib
from typing import Dict

import numpy as np
import torch

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')


def _gen(key: str) -> torch.Generator:
    return torch.Generator().manual_seed(zlib.crc32(key.encode()))


def procedural_tensor(key: str, shape) -> torch.Tensor:
    shape = tuple(shape)
    leaf = key.rsplit('.', 1)[-1]
    g = _gen(key)
    if leaf == 'num_batches_tracked':
        return torch.tensor(1, dtype=torch.long)
    if leaf == 'temp':
        return torch.tensor(10.0)
    if leaf == 'running_mean':
        return 0.1 * torch.randn(shape, generator=g)
    if leaf == 'running_var':
        return 0.5 + torch.rand(shape, generator=g)
    if leaf == 'weight' and len(shape) == 1:           # BatchNorm / LayerNorm gain
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if leaf == 'weight':                               # conv [O,I,kh,kw] / linear [O,I]
        fan_in = int(np.prod(shape[1:]))
        return torch.randn(shape, generator=g) / math.sqrt(fan_in)
    # biases, BN shifts, pos_embed*, cls_token
    return 0.1 * torch.randn(shape, generator=g)


def procedural_state_dict(shapes: Dict[str, tuple]) -> Dict[str, torch.Tensor]:
    return {k: procedural_tensor(k, s) for k, s in shapes.items()}


def load_bn_calibration(name: str = 'visformer_micro_80') -> Dict[str, torch.Tensor]:
    """running_mean / running_var entries (keys as in the meta-baseline state dict)."""
    path = os.path.join(_DATA_DIR, f'bn_calib_{name}.npz')
    with np.load(path) as z:
        return {k: torch.from_numpy(z[k].copy()) for k in z.files}


def synthetic_checkpoint_sd(shapes: Dict[str, tuple], calib: str = 'visformer_micro_80'):
    """Procedural weights + shipped BN calibration = the synthetic 'checkpoint' used by the
    parity tests, the benchmark and the CPU baseline alike."""
    sd = procedural_state_dict(shapes)
    if calib is not None:
        stats = load_bn_calibration(calib)
        for k, v in stats.items():
            if k not in sd or tuple(sd[k].shape) != tuple(v.shape):
                raise KeyError(f'BN calibration entry {k} does not match the model')
            sd[k] = v
    return sd


def synthetic_episodes(seed: int, n_ep: int, way: int, shot: int, query: int,
                       img: int = 80, noise: float = 0.5) -> torch.Tensor:
    """float32 [n_ep*way*(shot+query), 3, img, img], class-major inside each episode
    (image i = e*way*(S+Q) + c*(S+Q) + j, first S of each class are the shots)."""
    g = torch.Generator().manual_seed(seed)
    per = shot + query
    mu = torch.randn(n_ep, way, 1, 3, img, img, generator=g)
    eps = torch.randn(n_ep, way, per, 3, img, img, generator=g)
    x = mu + noise * eps
    return x.reshape(n_ep * way * per, 3, img, img).contiguous()
