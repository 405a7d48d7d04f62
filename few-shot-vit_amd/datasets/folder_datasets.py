"""'image-folder' and 'cifar-fs': the reference's directory-tree datasets (test_phase/datasets/image_folder.py:13-66,
cifar_fs.py:25-108) on the device-resident data path.

* 'cifar-fs' (two of the six shipped SUN-M configs: meta_tuning_sun_m/configs/train_meta_cifarfs_visformer_{1,5}shot.yaml): every
  image is 32 x 32, so the split is decoded ONCE on the host (Pillow, exactly `Image.open(path).convert('RGB')`), uploaded as a uint8
  [N,32,32,3] tensor, and a sampler batch is gathered + transformed by index on the GPU with the kernel behind
  fsvit_image_transform_gather: Resize(80) (= Pillow BILINEAR 32 -> 80, bit-exact) -> ToTensor -> Normalize(CIFAR mean / std).
* 'image-folder' (the only route to 224-pixel inputs with real data - BASELINE configs[4], ViT-S/16 on tieredImageNet): images have
  arbitrary sizes, so Resize(box_size) -> CenterCrop(image_size) runs on the host with Pillow itself (the reference's own arithmetic:
  torchvision's Resize on a PIL image IS `Image.resize(..., BILINEAR)`), each image once, cached as uint8 [image_size, image_size, 3];
  a batch is stacked, uploaded and normalised on the GPU.  `gather(index)` is the batch interface the drivers use.

Only the eval transform (`augment` unset) is built, like the other image datasets.  Class order: the reference enumerates
`os.listdir` (cifar-fs) / `sorted(os.listdir)` (image-folder); listdir order is file-system dependent, so cifar-fs sorts too - class
identity never reaches the few-shot episode (labels are re-made per episode, utils/few_shot.py:11-16)."""
import json
import os

import numpy as np
import torch

from .datasets import register
from .image_datasets import _DeviceImageDataset
from .transforms import IMAGENET_MEAN, IMAGENET_STD

CIFAR_MEAN = (0.5071, 0.4866, 0.4409)            # cifar_fs.py:62-63
CIFAR_STD = (0.2009, 0.1984, 0.2023)


def _open_rgb(path):
    from PIL import Image
    with Image.open(path) as im:
        return im.convert('RGB')


@register('cifar-fs')
class CifarFS(_DeviceImageDataset):
    resize, crop = (80, 80), 80                 # transforms.Resize(80) on square 32 x 32 images (cifar_fs.py:70-76)
    mean, std = CIFAR_MEAN, CIFAR_STD

    def __init__(self, root_path, split='train', augment=None, device=None, **kwargs):
        if augment is not None:
            raise NotImplementedError("fsvit: only the eval transform (augment=None) is built ('resize' / 'cropaug' are train-time)")
        sub = {'train': 'meta-train', 'val': 'meta-val', 'test': 'meta-test'}.get(split)
        if sub is None:
            raise ValueError('Unkown setname.')                              # cifar_fs.py:42
        base = os.path.join(root_path, sub)
        folders = [os.path.join(base, d) for d in sorted(os.listdir(base)) if os.path.isdir(os.path.join(base, d))]
        data, label = [], []
        for idx, folder in enumerate(folders):
            for name in sorted(os.listdir(folder)):
                data.append(np.asarray(_open_rgb(os.path.join(folder, name)), dtype=np.uint8))
                label.append(idx)
        shapes = {a.shape for a in data}
        if len(shapes) != 1:
            raise ValueError(f'cifar-fs: images of different sizes {sorted(shapes)} (expected 32 x 32 everywhere)')
        self.num_class = len(set(label))
        self._finish(np.stack(data), label, device)


@register('image-folder')
class ImageFolder:
    """root_path/<class>/<image files>; optional `split` + `split_file` (json: {split: [class names]}, default
    <parent of root_path>/split.json) restrict the classes (image_folder.py:22-34)."""

    def __init__(self, root_path, image_size=224, box_size=256, device=None, **kwargs):
        if kwargs.get('augment'):
            raise NotImplementedError('fsvit: only the eval transform (no augment) is built')
        if box_size is None:
            box_size = image_size
        self.image_size, self.box_size = int(image_size), int(box_size)
        classes = sorted(os.listdir(root_path))
        if kwargs.get('split'):
            path = kwargs.get('split_file')
            if path is None:
                path = os.path.join(os.path.dirname(root_path.rstrip('/')), 'split.json')
            with open(path, 'r') as f:
                classes = sorted(json.load(f)[kwargs['split']])
        self.filepaths, self.label = [], []
        for i, c in enumerate(classes):
            for filename in sorted(os.listdir(os.path.join(root_path, c))):
                self.filepaths.append(os.path.join(root_path, c, filename))
                self.label.append(i)
        self.n_classes = max(self.label) + 1
        self.device = torch.device(device if device is not None else ('cuda' if torch.cuda.is_available() else 'cpu'))
        # decoded-image cache, bounded in bytes (LRU): a tieredImageNet-sized split at 224 px is 150 KB per image - 30 ... 67 GB if every
        # image ever touched stayed resident, per rank.  `cache_bytes` (default 8 GiB; 0 disables) caps it.
        from collections import OrderedDict
        self._cache = OrderedDict()
        self._cache_bytes = 0
        self._cache_limit = int(kwargs.get('cache_bytes', 8 << 30))
        self._mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
        self._std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)

    def __len__(self):
        return len(self.filepaths)

    def _load_u8(self, i) -> torch.Tensor:
        """Resize(box_size) -> CenterCrop(image_size) of image i as uint8 [S,S,3] (torchvision semantics on a PIL image: the SHORTER
        side becomes box_size, the other int(box_size * long / short); Pillow BILINEAR; crop offsets round((dim - S) / 2))."""
        i = int(i)
        hit = self._cache.get(i)
        if hit is not None:
            self._cache.move_to_end(i)
            return hit
        from PIL import Image
        im = _open_rgb(self.filepaths[i])
        w, h = im.size
        short, long_ = (w, h) if w <= h else (h, w)
        if short != self.box_size:
            new_short, new_long = self.box_size, int(self.box_size * long_ / short)
            im = im.resize((new_short, new_long) if w <= h else (new_long, new_short), Image.BILINEAR)
        w, h = im.size
        S = self.image_size
        if w < S or h < S:                                                  # torchvision pads when the image is smaller than the crop
            canvas = Image.new('RGB', (max(w, S), max(h, S)))
            canvas.paste(im, ((max(w, S) - w) // 2, (max(h, S) - h) // 2))
            im, (w, h) = canvas, canvas.size
        top, left = int(round((h - S) / 2.0)), int(round((w - S) / 2.0))
        t = torch.from_numpy(np.asarray(im.crop((left, top, left + S, top + S)), dtype=np.uint8).copy())
        if self._cache_limit > 0:
            self._cache[i] = t
            self._cache_bytes += t.numel()
            while self._cache_bytes > self._cache_limit and len(self._cache) > 1:
                _, old = self._cache.popitem(last=False)
                self._cache_bytes -= old.numel()
        return t

    def _normalise(self, u8: torch.Tensor) -> torch.Tensor:
        x = u8.permute(0, 3, 1, 2).to(torch.float32).div_(255.0)           # ToTensor
        return x.sub_(self._mean.to(x.device)).div_(self._std.to(x.device))  # Normalize

    def gather(self, index) -> torch.Tensor:
        """index: dataset indices of one sampler batch -> float32 [len, 3, S, S] on the GPU (uint8 upload, normalisation on the device)."""
        u8 = torch.stack([self._load_u8(i) for i in torch.as_tensor(index).tolist()])
        return self._normalise(u8.to(self.device, non_blocking=True))

    def __getitem__(self, i):
        return self._normalise(self._load_u8(i).unsqueeze(0).to(self.device))[0], self.label[i]       # same device arithmetic as `gather`

    def convert_raw(self, x):
        return x * self._std[0].type_as(x) + self._mean[0].type_as(x)
