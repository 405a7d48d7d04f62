"""'mini-imagenet' / 'tiered-imagenet' with the reference's file formats and constructor surface
(test_phase/datasets/mini_imagenet.py:27-44, tiered_imagenet.py:13-50), MI355X-first: the whole split is uploaded ONCE as a
uint8 [N,84,84,3] tensor (mini test split 254 MB; all 60 000 images 1.27 GB of the 288 GB) and episodes are gathered +
transformed on the GPU by index (`gather`, fsvit_image_transform_gather) instead of 8 DataLoader workers running PIL.
Only the eval transform (`augment=None`) is built; the train-time augmentations of the supervised phase are out of scope."""
import os
import pickle

import numpy as np
import torch

from .datasets import register
from .transforms import IMAGENET_MEAN, IMAGENET_STD, DeviceTransform


class _DeviceImageDataset:
    resize, crop = (88, 88), 80

    def _finish(self, data: np.ndarray, label, device):
        if data.dtype != np.uint8 or data.ndim != 4 or data.shape[-1] != 3:
            raise ValueError('expected uint8 images [N,H,W,3]')
        min_label = min(label)
        self.label = [int(x) - int(min_label) for x in label]
        self.n_classes = max(self.label) + 1
        self.device = torch.device(device if device is not None else ('cuda' if torch.cuda.is_available() else 'cpu'))
        self.images = torch.from_numpy(np.ascontiguousarray(data))
        self._on_device = None
        self._transform = None

    def __len__(self):
        return len(self.label)

    def device_images(self):
        if self._on_device is None:
            if self.device.type != 'cuda':
                raise RuntimeError('fsvit: the dataset transform runs on an MI355X (no CPU fallback)')
            self._on_device = self.images.to(self.device)
            self._transform = DeviceTransform(tuple(self.images.shape[1:3]), self.resize, self.crop, self.device,
                                              mean=getattr(self, 'mean', IMAGENET_MEAN), std=getattr(self, 'std', IMAGENET_STD))
        return self._on_device

    def gather(self, index) -> torch.Tensor:
        """index: LongTensor of dataset indices (one sampler batch) -> float32 [len, 3, 80, 80] on the GPU."""
        imgs = self.device_images()
        return self._transform(imgs, torch.as_tensor(index))

    def __getitem__(self, i):
        return self.gather(torch.tensor([int(i)]))[0], self.label[i]


@register('mini-imagenet')
class MiniImageNet(_DeviceImageDataset):
    resize, crop = (88, 88), 80                         # Resize((88, 88)) -> CenterCrop(80), mini_imagenet.py:49-52

    def __init__(self, root_path, split='train', augment=None, device=None, **kwargs):
        if augment is not None:
            raise NotImplementedError('fsvit: only the eval transform (augment=None) is built')
        split_tag = 'train_phase_train' if split == 'train' else split
        with open(os.path.join(root_path, 'miniImageNet_category_split_{}.pickle'.format(split_tag)), 'rb') as f:
            pack = pickle.load(f, encoding='latin1')
        self._finish(np.asarray(pack['data']), pack['labels'], device)


@register('tiered-imagenet')
class TieredImageNet(_DeviceImageDataset):
    resize, crop = (80, 80), 80                         # Resize(80) on square images, tiered_imagenet.py:53-57

    def __init__(self, root_path, split='train', mini=False, augment=None, device=None, **kwargs):
        if augment is not None:
            raise NotImplementedError('fsvit: only the eval transform (augment=None) is built')
        data = np.load(os.path.join(root_path, '{}_images.npz'.format(split)), allow_pickle=True)['images']
        data = data[:, :, :, ::-1]                      # BGR -> RGB, tiered_imagenet.py:21
        with open(os.path.join(root_path, '{}_labels.pkl'.format(split)), 'rb') as f:
            label = pickle.load(f)['labels']
        if mini:                                        # tiered_imagenet.py:33-50
            min_label = min(label)
            label = [x - min_label for x in label]
            np.random.seed(0)
            c = np.random.choice(max(label) + 1, 64, replace=False).tolist()
            cnt = {x: 0 for x in c}
            ind = {x: i for i, x in enumerate(c)}
            keep, label_ = [], []
            for i in range(len(data)):
                y = int(label[i])
                if y in cnt and cnt[y] < 600:
                    keep.append(i)
                    label_.append(ind[y])
                    cnt[y] += 1
            data, label = data[keep], label_
        self._finish(np.ascontiguousarray(data), label, device)
