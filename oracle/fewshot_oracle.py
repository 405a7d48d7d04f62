"""ORACLE (test infrastructure, NOT product code).

numpy restatement of the host-side episodic helpers of the reference hot path.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this.

Parity status: PINNED against the known answers captured from the imported reference
(`tests/golden/host_known_answers.json`, written by
`tests/golden/make_golden.py`).
"""
import numpy as np
import scipy.stats


def split_shot_query(data, way, shot, query, ep_per_batch=1):
    """utils/few_shot.py:4-10.  data [E*way*(shot+query), ...] class-major ->
    x_shot [E,way,shot,...], x_query [E,way*query,...]."""
    data = np.asarray(data)
    img_shape = data.shape[1:]
    data = data.reshape(ep_per_batch, way, shot + query, *img_shape)
    x_shot = np.ascontiguousarray(data[:, :, :shot])
    x_query = np.ascontiguousarray(data[:, :, shot:]).reshape(ep_per_batch, way * query, *img_shape)
    return x_shot, x_query


def make_nk_label(n, k, ep_per_batch=1):
    """utils/few_shot.py:13-16."""
    return np.tile(np.repeat(np.arange(n), k), ep_per_batch)


def categories_sampler(label, n_batch, n_cls, n_per, ep_per_batch=1):
    """datasets/samplers.py:5-35.  Uses the GLOBAL legacy numpy RNG exactly as the reference
    does (np.random.choice), so the caller seeds with np.random.seed(...)."""
    label = np.array(label)
    catlocs = [np.argwhere(label == c).reshape(-1) for c in range(max(label) + 1)]
    for _ in range(n_batch):
        batch = []
        for _ in range(ep_per_batch):
            classes = np.random.choice(len(catlocs), n_cls, replace=False)
            episode = [np.random.choice(catlocs[c], n_per, replace=False) for c in classes]
            batch.append(np.stack(episode))
        yield np.stack(batch).reshape(-1)          # bs * n_cls * n_per


def compute_acc(logits, label):
    """utils/__init__.py:104-109 (reduction='mean')."""
    return float((np.argmax(logits, axis=1) == label).astype(np.float32).mean())


def cross_entropy(logits, label):
    """F.cross_entropy(logits, label), mean reduction (test_few_shot.py:89)."""
    logits = np.asarray(logits, dtype=np.float64)
    m = logits.max(axis=1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(logits - m).sum(axis=1))
    return float((lse - logits[np.arange(len(label)), label]).mean())


def mean_confidence_interval(data, confidence=0.95):
    """test_few_shot.py:20-25."""
    a = 1.0 * np.array(data)
    n = len(a)
    se = scipy.stats.sem(a)
    return se * scipy.stats.t.ppf((1 + confidence) / 2., n - 1)


class Averager:
    """utils/__init__.py:28-39."""

    def __init__(self):
        self.n = 0.0
        self.v = 0.0

    def add(self, v, n=1.0):
        self.v = (self.v * self.n + v * n) / (self.n + n)
        self.n += n

    def item(self):
        return self.v
