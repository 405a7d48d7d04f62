#!/usr/bin/env python3
"""Golden vectors of the eval image transform, produced by Pillow itself (the reference's transform is
torchvision.transforms.Resize on PIL images == Image.resize(..., BILINEAR); torchvision is not installed here, Pillow is):
    python tests/golden/make_transform_golden.py   ->  tests/golden/transform_pil.npz  (inputs + Pillow outputs)"""
import os

import numpy as np
from PIL import Image

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(2024)
    imgs = rng.integers(0, 256, size=(4, 84, 84, 3), dtype=np.uint8)
    imgs[1] = (np.linspace(0, 255, 84)[None, :, None] * np.ones((84, 1, 3))).astype(np.uint8)      # smooth ramp
    imgs[2, ::2] = 255                                                                            # saturating stripes
    imgs[2, 1::2] = 0
    out = {'images': imgs}
    out['resize88'] = np.stack([np.asarray(Image.fromarray(x).resize((88, 88), Image.BILINEAR)) for x in imgs])
    out['resize80'] = np.stack([np.asarray(Image.fromarray(x).resize((80, 80), Image.BILINEAR)) for x in imgs])
    import PIL
    out['pillow_version'] = np.array(PIL.__version__)
    np.savez_compressed(os.path.join(OUT, 'transform_pil.npz'), **out)
    print('wrote transform_pil.npz', {k: getattr(v, 'shape', None) for k, v in out.items()})


if __name__ == '__main__':
    main()
