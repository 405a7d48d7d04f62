"""CPU checks of the eval image transform: the oracle restates Pillow's BILINEAR resampling
(third-party arithmetic behind torchvision.transforms.Resize, mini_imagenet.py:49-52) bit for bit - pinned against
vectors produced by Pillow (tests/golden/transform_pil.npz) and, when Pillow is importable, against Pillow directly on
random sizes; the product's host-side coefficient tables equal the oracle's."""
import os

import numpy as np
import pytest

from oracle import transform_oracle as to


def test_oracle_resize_matches_pillow_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, 'transform_pil.npz'))
    for i, img in enumerate(z['images']):
        assert np.array_equal(to.pil_resize_bilinear(img, 88, 88), z['resize88'][i]), i      # upscale, ksize 3
        assert np.array_equal(to.pil_resize_bilinear(img, 80, 80), z['resize80'][i]), i      # downscale (antialias), ksize 5


def test_oracle_resize_matches_pillow_live():
    Image = pytest.importorskip('PIL.Image')
    rng = np.random.default_rng(7)
    for (h, w, oh, ow) in ((84, 84, 88, 88), (84, 84, 80, 80), (32, 32, 88, 88), (96, 64, 40, 72), (84, 84, 84, 84)):
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(to.pil_resize_bilinear(img, oh, ow), ref), (h, w, oh, ow)


def test_product_tables_equal_oracle_tables():
    from fewshot_vit_amd.datasets.transforms import pil_bilinear_tables
    for a, b in ((84, 88), (84, 80), (84, 84), (32, 88), (100, 40), (7, 3)):
        for x, y in zip(pil_bilinear_tables(a, b), to.bilinear_coeffs(a, b)):
            assert np.array_equal(x, y), (a, b)


def test_eval_transform_shapes_and_normalisation(golden_dir):
    z = np.load(os.path.join(golden_dir, 'transform_pil.npz'))
    t = to.eval_transform(z['images'][0], 88, 80)
    assert t.shape == (3, 80, 80) and t.dtype == np.float32
    crop = z['resize88'][0][4:84, 4:84].astype(np.float32) / np.float32(255)
    np.testing.assert_array_equal(t, ((crop - to.MEAN) / to.STD).transpose(2, 0, 1))


def test_image_datasets_read_the_reference_file_formats(tmp_path):
    """mini-imagenet pickle / tiered-imagenet npz+pkl layouts (mini_imagenet.py:29-44, tiered_imagenet.py:16-30): labels are
    re-based to 0, tiered flips BGR->RGB; the transform itself needs the GPU and says so."""
    import pickle
    import torch
    from fewshot_vit_amd import datasets
    rng = np.random.default_rng(0)
    data = rng.integers(0, 256, size=(12, 84, 84, 3), dtype=np.uint8)
    labels = [64 + i // 3 for i in range(12)]
    with open(tmp_path / 'miniImageNet_category_split_test.pickle', 'wb') as f:
        pickle.dump({'data': data, 'labels': labels}, f)
    ds = datasets.make('mini-imagenet', root_path=str(tmp_path), split='test')
    assert len(ds) == 12 and ds.n_classes == 4 and ds.label[:4] == [0, 0, 0, 1]
    assert torch.equal(ds.images, torch.from_numpy(data))
    np.savez(tmp_path / 'val_images.npz', images=data)
    with open(tmp_path / 'val_labels.pkl', 'wb') as f:
        pickle.dump({'labels': labels}, f)
    dt = datasets.make('tiered-imagenet', root_path=str(tmp_path), split='val')
    assert torch.equal(dt.images, torch.from_numpy(data[..., ::-1].copy()))
    with pytest.raises(NotImplementedError):
        datasets.make('mini-imagenet', root_path=str(tmp_path), split='test', augment='crop')
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            ds.gather(torch.tensor([0, 1]))
