"""Episode index sampler (surface and RNG stream of test_phase/datasets/samplers.py:5-35)."""
import numpy as np
import torch


class CategoriesSampler:
    """Yields, per batch, a flat LongTensor of ep_per_batch * n_cls * n_per dataset indices in
    class-major order.  Draws from the GLOBAL legacy numpy RNG (`np.random.choice`) in the main
    process, exactly like the reference, so `np.random.seed(s)` reproduces its episode stream.

    `rank` / `world_size` (extension, defaults = reference behaviour): every rank draws the SAME
    global stream and keeps batches `rank::world_size`, which is how episodes are sharded over the
    GPUs of a node without any data-path collective."""

    def __init__(self, label, n_batch, n_cls, n_per, ep_per_batch=1, rank=0, world_size=1):
        self.n_batch = n_batch
        self.n_cls = n_cls
        self.n_per = n_per
        self.ep_per_batch = ep_per_batch
        self.rank, self.world_size = rank, world_size
        label = np.array(label)
        self.catlocs = [np.argwhere(label == c).reshape(-1) for c in range(max(label) + 1)]

    def __len__(self):
        return len(range(self.rank, self.n_batch, self.world_size))

    def __iter__(self):
        for i_batch in range(self.n_batch):
            batch = []
            for _ in range(self.ep_per_batch):
                classes = np.random.choice(len(self.catlocs), self.n_cls, replace=False)
                episode = [torch.from_numpy(np.random.choice(self.catlocs[c], self.n_per, replace=False))
                           for c in classes]
                batch.append(torch.stack(episode))
            if i_batch % self.world_size != self.rank:
                continue                      # drawn (keeps the stream aligned) but owned by another rank
            yield torch.stack(batch).view(-1)  # bs * n_cls * n_per
