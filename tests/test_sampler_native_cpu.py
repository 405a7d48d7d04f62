"""fsvit_sampler_draw (csrc/episode_sampler.hip, host only) against numpy's legacy generator: the same index stream as the reference's
CategoriesSampler (test_phase/datasets/samplers.py:19-35) AND the same generator state afterwards."""
import json
import os

import numpy as np
import pytest
import torch

from fewshot_vit_amd.datasets.samplers import CategoriesSampler

HERE = os.path.dirname(os.path.abspath(__file__))


def _labels(n_cls=20, per=600, ragged=False):
    if not ragged:
        return np.repeat(np.arange(n_cls), per).tolist()
    rng = np.random.default_rng(3)
    sizes = rng.integers(21, 700, size=n_cls)
    lab = np.concatenate([np.full(s, c) for c, s in enumerate(sizes)])
    rng.shuffle(lab)                       # class members scattered over the dataset, as in the folder datasets
    return lab.tolist()


@pytest.mark.parametrize('ragged', [False, True])
@pytest.mark.parametrize('n_batch,ep,n_cls,n_per', [(7, 1, 5, 20), (3, 4, 5, 16), (2, 1, 20, 21), (130, 1, 5, 6)])
def test_native_stream_and_state_equal_numpy(ragged, n_batch, ep, n_cls, n_per):
    lab = _labels(ragged=ragged)
    np.random.seed(12345)
    a = list(CategoriesSampler(lab, n_batch, n_cls, n_per, ep, native=False))
    sa = np.random.get_state()
    tail_a = np.random.randint(0, 1 << 30, size=5)
    np.random.seed(12345)
    b = list(CategoriesSampler(lab, n_batch, n_cls, n_per, ep, native=True))
    sb = np.random.get_state()
    tail_b = np.random.randint(0, 1 << 30, size=5)
    assert len(a) == len(b) == n_batch
    for u, v in zip(a, b):
        assert u.dtype == torch.int64 and torch.equal(u, v)
    assert sa[0] == sb[0] and np.array_equal(sa[1], sb[1]) and sa[2] == sb[2]
    assert np.array_equal(tail_a, tail_b)          # whatever draws next from the global generator sees the same stream


@pytest.mark.parametrize('native', [True, False])
def test_stream_equals_reference_known_answers(native):
    """The reference's own sampler stream (tests/golden/host_known_answers.json, written by importing test_phase/datasets/samplers.py)."""
    with open(os.path.join(HERE, 'golden', 'host_known_answers.json')) as f:
        ka = json.load(f)
    label = np.repeat(np.arange(20), 600).tolist()
    np.random.seed(12345)
    assert [b.tolist() for b in CategoriesSampler(label, 3, 5, 16, 1, native=native)] == ka['sampler_seed12345_20x600_3x5x16']
    np.random.seed(0)
    assert [int(b.sum()) for b in CategoriesSampler(label, 2, 5, 20, 4, native=native)] == ka['sampler_seed0_20x600_2x(4ep)x5x20_sum']


def test_native_is_faster_than_numpy():
    import time
    label = np.repeat(np.arange(20), 600).tolist()
    t = {False: float('inf'), True: float('inf')}
    for _ in range(3):                       # best of three: a timing comparison on a shared CI host
        for native in (False, True):
            np.random.seed(1)
            t0 = time.perf_counter()
            n = sum(1 for _ in CategoriesSampler(label, 300, 5, 20, 1, native=native))
            t[native] = min(t[native], time.perf_counter() - t0)
            assert n == 300
    print('sampler: numpy %.1f us / episode, native %.1f us / episode' % (1e6 * t[False] / 300, 1e6 * t[True] / 300))
    assert t[True] < t[False]


def test_native_crosses_generator_refills_mid_shuffle():
    """A 600-item shuffle consumes ~1.2 generator refills (624 words each): positions 0 .. 623 at the start of a draw all give numpy's stream."""
    lab = _labels(8, 600)
    for burn in (0, 1, 311, 623, 624, 1000):
        np.random.seed(99)
        np.random.randint(0, 10, size=burn)
        a = list(CategoriesSampler(lab, 2, 5, 20, native=False))
        np.random.seed(99)
        np.random.randint(0, 10, size=burn)
        b = list(CategoriesSampler(lab, 2, 5, 20, native=True))
        assert all(torch.equal(u, v) for u, v in zip(a, b)), burn


def test_native_rejects_short_class():
    lab = [0] * 30 + [1] * 3
    with pytest.raises(Exception):
        list(CategoriesSampler(lab, 1, 2, 5, native=True))


def test_native_slab_one_keeps_the_generator_where_the_lazy_draws_leave_it_and_errors_fall_back():
    """the native path reads ahead in slabs - the generator matches the reference's lazy per-batch draws only at slab boundaries.
    `native_slab=1` restores the exact interleaving (an early break, np.random between batches); under native=None a native error (here: a
    class shorter than n_per that is never drawn... numpy would only raise when it IS drawn) falls back to the numpy path instead of failing up front."""
    from fewshot_vit_amd.datasets.samplers import CategoriesSampler
    label = np.repeat(np.arange(20), 600).tolist()
    np.random.seed(7)
    ref = CategoriesSampler(label, 9, 5, 6, 2, native=False)
    it = iter(ref)
    want = [next(it).tolist() for _ in range(3)]
    want_state = np.random.get_state()[1].copy()                 # after exactly three lazy batches
    np.random.seed(7)
    it = iter(CategoriesSampler(label, 9, 5, 6, 2, native=True, native_slab=1))
    got = [next(it).tolist() for _ in range(3)]
    assert got == want and np.array_equal(np.random.get_state()[1], want_state)
    # a label set whose class 19 has 3 items < n_per = 6: the native draw refuses the table; native=None must still run (numpy raises only if class 19 is drawn)
    short = np.repeat(np.arange(19), 600).tolist() + [19] * 3
    np.random.seed(11)
    try:
        a = [b.tolist() for b in CategoriesSampler(short, 2, 5, 6, 1, native=False)]
    except ValueError:
        a = None
    np.random.seed(11)
    try:
        b = [x.tolist() for x in CategoriesSampler(short, 2, 5, 6, 1, native=None)]
    except ValueError:
        b = None
    assert a == b
