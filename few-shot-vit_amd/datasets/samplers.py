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
        # The legacy-RNG call sequence is the reference's (one choice of classes, then one choice per class, per episode); the indices are
        # assembled in one numpy array per batch (the reference's per-class torch.from_numpy + two torch.stack cost more host time than the
        # draws themselves, and at ~4000 episodes/s the host has 250 us per episode).
        n_cat, choice = len(self.catlocs), np.random.choice
        for i_batch in range(self.n_batch):
            batch = np.empty((self.ep_per_batch, self.n_cls, self.n_per), dtype=np.int64)
            for e in range(self.ep_per_batch):
                classes = choice(n_cat, self.n_cls, replace=False)
                for j, c in enumerate(classes):
                    batch[e, j] = choice(self.catlocs[c], self.n_per, replace=False)
            if i_batch % self.world_size != self.rank:
                continue                      # drawn (keeps the stream aligned) but owned by another rank
            yield torch.from_numpy(batch).view(-1)  # bs * n_cls * n_per
