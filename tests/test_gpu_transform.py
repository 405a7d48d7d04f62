"""Device-resident dataset transform through the C-ABI (fsvit_image_transform_gather) vs the oracle (bit-exact: the uint8
resize is integer work, the normalisation is two correctly-rounded fp32 operations) and vs Pillow's golden vectors."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('resize,crop', [(88, 80), (80, 80), (84, 84)])
def test_transform_gather_bit_exact_vs_oracle(golden_dir, resize, crop):
    from fewshot_vit_amd.datasets.transforms import DeviceTransform
    from oracle import transform_oracle as to
    z = np.load(os.path.join(golden_dir, 'transform_pil.npz'))
    rng = np.random.default_rng(1)
    imgs = np.concatenate([z['images'], rng.integers(0, 256, size=(28, 84, 84, 3), dtype=np.uint8)])
    dev = torch.device('cuda', 0)
    tf = DeviceTransform((84, 84), resize, crop, dev)
    index = torch.tensor([3, 0, 31, 7, 7, 1, 2, 30], device=dev)
    out = tf(torch.from_numpy(imgs).to(dev), index).cpu().numpy()
    ref = np.stack([to.eval_transform(imgs[i], resize, crop) for i in index.tolist()])
    assert out.shape == ref.shape == (8, 3, crop, crop)
    assert np.array_equal(out, ref), float(np.abs(out - ref).max())
    if (resize, crop) == (88, 80):            # straight against Pillow's own output for the committed images
        pil = z['resize88'][3][4:84, 4:84].astype(np.float32) / np.float32(255)
        assert np.array_equal(out[0], ((pil - to.MEAN) / to.STD).transpose(2, 0, 1))


def test_mini_imagenet_files_through_the_eval_driver(tmp_path):
    """A miniImageNet-format pickle -> 'mini-imagenet' dataset -> test_few_shot.evaluate: episodes are gathered and
    transformed on the GPU; per-batch accuracies equal those computed from host-transformed images (oracle transform)."""
    import pickle
    from fewshot_vit_amd import datasets, test_few_shot
    from oracle import transform_oracle as to
    rng = np.random.default_rng(5)
    n_cls, per = 6, 20
    mu = rng.integers(0, 256, size=(n_cls, 1, 84, 84, 3))
    data = np.clip(mu + rng.normal(0, 60, size=(n_cls, per, 84, 84, 3)), 0, 255).astype(np.uint8).reshape(-1, 84, 84, 3)
    labels = [80 + i // per for i in range(n_cls * per)]
    with open(tmp_path / 'miniImageNet_category_split_test.pickle', 'wb') as f:
        pickle.dump({'data': data, 'labels': labels}, f)
    cfg = dict(dataset='mini-imagenet', dataset_args=dict(split='test', root_path=str(tmp_path)), synthetic_checkpoint='visformer_micro_80')
    got = test_few_shot.evaluate(cfg, shot=1, n_batch=4, launch_batches=2, numerics='parity', log=lambda *a: None)

    class HostDataset:                         # the same images through the oracle transform, fed the classic way
        label = [x - 80 for x in labels]

        def __getitem__(self, i):
            return torch.from_numpy(to.eval_transform(data[i], 88, 80)), self.label[i]
    datasets.register('_host_mini')(lambda **kw: HostDataset())
    ref = test_few_shot.evaluate(dict(dataset='_host_mini', dataset_args={}, synthetic_checkpoint='visformer_micro_80'), shot=1, n_batch=4,
                                 launch_batches=2, numerics='parity', log=lambda *a: None)
    assert got['va_lst'] == ref['va_lst'] and got['n'] == 4
    assert got['loss'] == pytest.approx(ref['loss'], abs=1e-6)


def test_cifar_fs_tree_upsampled_on_the_gpu_bit_exact(tmp_path):
    """'cifar-fs' (cifar_fs.py:25-108): 32 x 32 PNG tree -> uint8 table -> Resize(80) (Pillow BILINEAR upsampling, 3 taps) + ToTensor +
    Normalize(CIFAR statistics) in the device transform, bit-exact against the Pillow-pinned oracle."""
    Image = pytest.importorskip('PIL.Image')
    from fewshot_vit_amd import datasets
    from oracle import transform_oracle as to
    rng = np.random.default_rng(11)
    raw = {}
    for c in ('apple', 'bus', 'cup'):
        os.makedirs(tmp_path / 'meta-val' / c)
        for j in range(4):
            a = rng.integers(0, 256, size=(32, 32, 3), dtype=np.uint8)
            Image.fromarray(a).save(tmp_path / 'meta-val' / c / f'{j}.png')
            raw[(c, j)] = a
    ds = datasets.make('cifar-fs', root_path=str(tmp_path), split='val')
    idx = torch.tensor([0, 5, 11, 5])
    out = ds.gather(idx).cpu().numpy()
    assert out.shape == (4, 3, 80, 80)
    mean, std = np.array(ds.mean, np.float32), np.array(ds.std, np.float32)
    for k, (c, j) in enumerate([('apple', 0), ('bus', 1), ('cup', 3), ('bus', 1)]):
        r = to.pil_resize_bilinear(raw[(c, j)], 80, 80).astype(np.float32) / np.float32(255.0)
        assert np.array_equal(out[k], ((r - mean) / std).transpose(2, 0, 1)), (c, j)
    x0, y0 = ds[5]
    assert y0 == 1 and np.array_equal(x0.cpu().numpy(), out[1])


def test_image_folder_batches_reach_a_224_encoder(tmp_path):
    """'image-folder' (image_folder.py:13-66), the data route of BASELINE configs[4]: variable-size files -> Resize(256) -> CenterCrop(224)
    on the host (Pillow), uint8 upload, normalisation on the device; `gather` == stacking `__getitem__`, and the batch runs through
    deit_small_patch16_224."""
    Image = pytest.importorskip('PIL.Image')
    from fewshot_vit_amd import datasets, models
    rng = np.random.default_rng(12)
    for c in ('n01', 'n02'):
        os.makedirs(tmp_path / 'images' / c)
        for j, (h, w) in enumerate(((300, 260), (256, 400), (224, 224))):
            Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).save(tmp_path / 'images' / c / f'{j}.jpg', quality=95)
    ds = datasets.make('image-folder', root_path=str(tmp_path / 'images'))
    assert len(ds) == 6 and ds.n_classes == 2
    batch = ds.gather(torch.arange(6))
    assert batch.shape == (6, 3, 224, 224) and batch.is_cuda
    one = torch.stack([ds[i][0] for i in range(6)])
    assert torch.equal(batch.cpu(), one.cpu())
    enc = models.make('deit_small_patch16_224', numerics='bf16').eval()
    with torch.no_grad():
        feat = enc(batch)
    assert feat.shape == (6, 384) and torch.isfinite(feat).all()
