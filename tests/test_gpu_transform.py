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
