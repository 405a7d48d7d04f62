"""world_size=2 gloo tests (CPU) of the episode-parallel path: sampler sharding + the single
gather / all-reduce exchange reproduce the single-process statistics exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_batch, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from fewshot_vit_amd import parallel, utils
    from fewshot_vit_amd.datasets.samplers import CategoriesSampler
    r, w, _ = parallel.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    label = np.repeat(np.arange(20), 600).tolist()
    np.random.seed(12345)
    sampler = CategoriesSampler(label, n_batch, 5, 6, 1, rank=r, world_size=w)
    # a deterministic stand-in for "accuracy of this batch": a hash of its indices
    mine = torch.tensor([[float(int(b.sum()) % 97) / 97.0, float(int(b[0]) % 13)] for b in sampler], dtype=torch.float64)
    allv = parallel.gather_in_stream_order(mine, n_batch, r, w)
    acc = allv[:, 0]
    mean, var, cnt = parallel.allreduce_mean_stats(float(mine[:, 0].sum()), float((mine[:, 0] ** 2).sum()), float(len(mine)), 'cpu')
    q.put((rank, allv.numpy(), mean, var, cnt, float(utils.mean_confidence_interval(acc.tolist()))))
    dist.destroy_process_group()


@pytest.mark.parametrize('n_batch', [10, 7])
def test_two_rank_gather_matches_single_process(n_batch):
    from fewshot_vit_amd import utils
    from fewshot_vit_amd.datasets.samplers import CategoriesSampler
    label = np.repeat(np.arange(20), 600).tolist()
    np.random.seed(12345)
    ref = np.array([[float(int(b.sum()) % 97) / 97.0, float(int(b[0]) % 13)] for b in CategoriesSampler(label, n_batch, 5, 6, 1)])
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, allv, mean, var, cnt, ci in results:
        np.testing.assert_array_equal(allv, ref)                      # stream order restored on every rank
        assert cnt == n_batch
        assert mean == pytest.approx(ref[:, 0].mean(), abs=1e-12)
        assert var == pytest.approx(ref[:, 0].var(ddof=1), abs=1e-12)
        assert ci == pytest.approx(float(utils.mean_confidence_interval(ref[:, 0].tolist())), abs=1e-12)
