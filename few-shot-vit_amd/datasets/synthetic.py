"""'synthetic-episodes': a stand-in for the miniImageNet test split (20 classes x 600 images of
3x80x80 after the eval transform, test_phase/datasets/mini_imagenet.py:27-56) for machines without
the pickles.  Image i of class c is mu_c + noise * eps_i, generated deterministically from (seed, i),
so every rank and every run sees the same data for the same index."""
import numpy as np
import torch

from .datasets import register


@register('synthetic-episodes')
class SyntheticEpisodes:
    def __init__(self, root_path=None, split='test', n_classes=20, n_per_class=600, image_size=80, noise=1.5,
                 seed=0, **unused):
        self.n_classes = n_classes
        self.image_size = image_size
        self.noise = noise
        self.seed = seed
        self.label = np.repeat(np.arange(n_classes), n_per_class).tolist()
        g = torch.Generator().manual_seed(seed)
        self.mu = torch.randn(n_classes, 3, image_size, image_size, generator=g)

        self._table = None

    def __len__(self):
        return len(self.label)

    def gather(self, idx, device=None):
        """Device-resident form of `torch.stack([self[i][0] for i in idx])` (same values: the table is built once from
        `__getitem__` and uploaded), so that a 2000-episode evaluation is not bound by per-image host generation."""
        device = device or torch.device('cuda', torch.cuda.current_device())
        if self._table is None or self._table.device != device:
            self._table = torch.stack([self[i][0] for i in range(len(self))]).to(device)
        return self._table.index_select(0, torch.as_tensor(idx, dtype=torch.long).to(device))

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + int(i) + 1)
        c = self.label[i]
        x = self.mu[c] + self.noise * torch.randn(3, self.image_size, self.image_size, generator=g)
        return x, c
