"""'image-folder' / 'cifar-fs' (VERDICT r01 missing #2, #3; test_phase/datasets/image_folder.py:13-66, cifar_fs.py:25-108) on tiny
directory trees written by the test: class / label enumeration, split files, and the host half of the transform against the
Pillow-pinned numpy oracle (oracle/transform_oracle.py)."""
import json
import os

import numpy as np
import pytest
import torch

Image = pytest.importorskip('PIL.Image')


def _tree(root, classes, sizes, seed=0, ext='png'):
    rng = np.random.default_rng(seed)
    imgs = {}
    for c in classes:
        os.makedirs(os.path.join(root, c))
        for j, (h, w) in enumerate(sizes):
            a = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
            Image.fromarray(a).save(os.path.join(root, c, f'img{j}.{ext}'))
            imgs[(c, j)] = a
    return imgs


def test_image_folder_matches_reference_transform(tmp_path):
    from fewshot_vit_amd import datasets
    from oracle import transform_oracle as to
    root = str(tmp_path / 'images')
    sizes = [(96, 64), (50, 120), (64, 64), (70, 33)]                       # portrait, landscape, square, crop larger than the short side after resize
    imgs = _tree(root, ['b_cls', 'a_cls', 'c_cls'], sizes)
    with open(tmp_path / 'split.json', 'w') as f:
        json.dump({'test': ['c_cls', 'a_cls']}, f)
    ds = datasets.make('image-folder', root_path=root, image_size=48, box_size=56, device='cpu')
    assert ds.n_classes == 3 and len(ds) == 12 and ds.label == [0] * 4 + [1] * 4 + [2] * 4     # sorted classes: a, b, c
    sub = datasets.make('image-folder', root_path=root, image_size=48, box_size=56, split='test', device='cpu')
    assert sub.n_classes == 2 and [os.path.basename(os.path.dirname(p)) for p in sub.filepaths[::4]] == ['a_cls', 'c_cls']
    for idx, (c, j) in enumerate([('a_cls', j) for j in range(4)]):
        a = imgs[(c, j)]
        h, w = a.shape[:2]
        if w <= h:
            nw, nh = 56, int(56 * h / w)
        else:
            nh, nw = 56, int(56 * w / h)
        r = to.pil_resize_bilinear(a, nh, nw)                              # == Pillow (pinned), == torchvision Resize(56) on a PIL image
        top, left = int(round((nh - 48) / 2.0)), int(round((nw - 48) / 2.0))
        r = r[top:top + 48, left:left + 48].astype(np.float32) / np.float32(255.0)
        exp = ((r - to.MEAN) / to.STD).transpose(2, 0, 1)
        x, y = ds[idx]
        assert y == 0 and x.shape == (3, 48, 48)
        assert np.abs(x.numpy() - exp).max() <= 1e-6
    back = ds.convert_raw(ds[0][0])
    assert float(back.min()) >= -1e-6 and float(back.max()) <= 1.0 + 1e-6


def test_cifar_fs_loads_split_and_keeps_uint8_images(tmp_path):
    from fewshot_vit_amd import datasets
    root = str(tmp_path / 'cifar-fs')
    imgs = _tree(os.path.join(root, 'meta-test'), ['dog', 'cat'], [(32, 32)] * 3, seed=3)
    _tree(os.path.join(root, 'meta-train'), ['x'], [(32, 32)], seed=4)
    ds = datasets.make('cifar-fs', root_path=root, split='test', device='cpu')
    assert len(ds) == 6 and ds.n_classes == 2 and ds.label == [0, 0, 0, 1, 1, 1]                 # sorted: cat, dog
    assert tuple(ds.images.shape) == (6, 32, 32, 3) and ds.images.dtype == torch.uint8
    assert np.array_equal(ds.images[0].numpy(), imgs[('cat', 0)])
    assert (ds.resize, ds.crop) == ((80, 80), 80) and ds.mean == (0.5071, 0.4866, 0.4409)
    with pytest.raises(ValueError):
        datasets.make('cifar-fs', root_path=root, split='nope', device='cpu')
    with pytest.raises(NotImplementedError):
        datasets.make('cifar-fs', root_path=root, split='test', augment='resize', device='cpu')
    with pytest.raises(RuntimeError):
        ds.gather(torch.tensor([0]))                                       # the transform itself runs on the GPU: no CPU fallback


def test_image_folder_cache_is_bounded_lru(tmp_path):
    """the decoded-image cache is capped in bytes (least recently used images are evicted), results unchanged."""
    from PIL import Image
    from fewshot_vit_amd import datasets
    root = tmp_path / 'imgs'
    rng = np.random.default_rng(0)
    for c in range(2):
        (root / f'c{c}').mkdir(parents=True)
        for i in range(4):
            Image.fromarray(rng.integers(0, 256, size=(40, 48, 3), dtype=np.uint8)).save(root / f'c{c}' / f'{i}.png')
    ds = datasets.make('image-folder', root_path=str(root), image_size=32, box_size=36, device='cpu', cache_bytes=3 * 32 * 32 * 3)
    first = ds._load_u8(0).clone()
    for i in range(8):
        ds._load_u8(i)
    assert len(ds._cache) == 3 and ds._cache_bytes <= 3 * 32 * 32 * 3
    assert 0 not in ds._cache and 7 in ds._cache
    assert torch.equal(ds._load_u8(0), first)
