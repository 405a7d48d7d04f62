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


def _grad_worker(rank, world, port, q, use_bucket=False):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from fewshot_vit_amd import parallel
    from fewshot_vit_amd.datasets.samplers import CategoriesSampler
    parallel.init_from_env(backend='gloo')
    params, data, loss_fn = _toy_problem()
    label = np.repeat(np.arange(20), 600).tolist()
    np.random.seed(1)
    idx = next(iter(CategoriesSampler(label, 1, 5, 4, ep_per_batch=4)))              # same stream on every rank
    mine = parallel.shard_episode_axis(idx, 4, rank, world)
    if use_bucket:                      # persistent flat buffer, `.grad` = views into it, ONE in-place all-reduce
        bucket = parallel.GradBucket(params)
        for it in range(2):             # second step: zero_grad(set_to_none) dropped the views, the bucket collects and re-attaches
            for p in params:
                p.grad = None if it else p.grad
            loss_fn(params, data[mine % len(data)].view(4 // world, -1, data.shape[1])).backward()
            bucket.allreduce_mean()
            assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, bucket.views))
    else:
        loss_fn(params, data[mine % len(data)].view(4 // world, -1, data.shape[1])).backward()
        parallel.allreduce_mean_grads(params)
    q.put((rank, [p.grad.numpy().copy() for p in params], mine.tolist()))
    dist.destroy_process_group()


def _toy_problem():
    g = torch.Generator().manual_seed(0)
    params = [torch.randn(6, 3, generator=g).requires_grad_(True), torch.randn(3, generator=g).requires_grad_(True)]
    data = torch.randn(64, 6, generator=g)

    def loss_fn(ps, x):                 # x [episodes, items, 6]: per-episode mean loss, averaged over the episodes
        return ((x @ ps[0] + ps[1]).tanh() ** 2).mean(dim=(1, 2)).mean()
    return params, data, loss_fn


@pytest.mark.parametrize('use_bucket', [False, True])
def test_two_rank_gradient_allreduce_matches_single_process_step(use_bucket):
    """train_meta.py data parallelism: ranks keep slices of the batch's episode axis; the all-reduced mean of their
    gradients equals the gradient of the whole batch's mean loss."""
    from fewshot_vit_amd import parallel
    from fewshot_vit_amd.datasets.samplers import CategoriesSampler
    params, data, loss_fn = _toy_problem()
    label = np.repeat(np.arange(20), 600).tolist()
    np.random.seed(1)
    idx = next(iter(CategoriesSampler(label, 1, 5, 4, ep_per_batch=4)))
    assert torch.equal(torch.cat([parallel.shard_episode_axis(idx, 4, r, 2) for r in range(2)]), idx)
    with pytest.raises(ValueError):
        parallel.shard_episode_axis(idx, 4, 0, 3)
    loss_fn(params, data[idx % len(data)].view(4, -1, data.shape[1])).backward()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q, use_bucket)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, grads, mine in results:
        for g, p in zip(grads, params):
            np.testing.assert_allclose(g, p.grad.numpy(), rtol=1e-6, atol=1e-7)


def test_grad_bucket_hands_unused_parameters_back_without_a_gradient():
    """a parameter that received no gradient must stay `.grad = None` after the bucket's exchange, so that the optimizer skips it
    (no momentum buffer, no weight decay) exactly as in a single-process run; the bucket only zeroes its slice for the collective."""
    from fewshot_vit_amd import parallel
    used, unused = torch.nn.Parameter(torch.ones(4)), torch.nn.Parameter(torch.ones(3))
    bucket = parallel.GradBucket([used, unused])
    for p in (used, unused):
        p.grad = None                                     # optimizer.zero_grad()
    (used * 2.0).sum().backward()
    bucket.flat[4:] = 7.0                                 # stale bytes in the unused slice must not reach the collective
    bucket.allreduce_mean()
    assert unused.grad is None and torch.equal(bucket.flat[4:], torch.zeros(3))
    assert used.grad.data_ptr() == bucket.views[0].data_ptr() and torch.equal(used.grad, torch.full((4,), 2.0))
    opt = torch.optim.SGD([used, unused], lr=0.1, momentum=0.9, weight_decay=0.1)
    opt.step()
    assert torch.equal(unused.detach(), torch.ones(3)) and len(opt.state[unused]) == 0


# ---------------------------------------------------------------- round 5: every rank draws only what it needs
def _stream_worker(rank, world, port, n_batch, shard, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from fewshot_vit_amd import parallel
    from fewshot_vit_amd.datasets.samplers import CategoriesSampler
    r, w, _ = parallel.init_from_env(backend='gloo')
    label = np.repeat(np.arange(20), 600).tolist()
    np.random.seed(12345)
    before = np.random.get_state()[1].copy()
    # shard=None is 'replay' ('scatter' hangs a loop that not every rank runs); the drivers ask for what parallel.sampler_shard says
    if shard == 'driver':
        shard = parallel.sampler_shard(w)
        assert shard == 'scatter'
    sampler = CategoriesSampler(label, n_batch, 5, 6, 2, rank=r, world_size=w, shard=shard)
    assert sampler._shard_mode() == ('scatter' if shard == 'scatter' else 'replay')
    epochs = [[b.tolist() for b in sampler] for _ in range(2)]           # the second epoch continues the same generator (test_epochs > 1)
    touched = not np.array_equal(before, np.random.get_state()[1])
    # the exchange at the end of an epoch on n_batch rows that do not divide over the ranks
    mine = torch.tensor([[float(sum(b) % 97)] for b in epochs[0]], dtype=torch.float64).view(-1, 1)
    allv = parallel.gather_in_stream_order(mine, n_batch, r, w)
    q.put((rank, epochs, touched, allv.view(-1).tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,n_batch,shard', [(2, 9, None), (2, 9, 'driver'), (4, 10, 'scatter'), (8, 21, 'driver'), (8, 21, 'replay'), (8, 5, 'scatter')])
def test_ranks_reassemble_the_global_stream_bit_for_bit(world, n_batch, shard):
    """Rank r's batches are global batches r, r + world, ... of the ONE stream a single process draws - for the replayed stream and for the
    table rank 0 draws alone and broadcasts ('scatter': the other ranks' generators are never touched); n_batch does not divide over the ranks."""
    from fewshot_vit_amd.datasets.samplers import CategoriesSampler
    label = np.repeat(np.arange(20), 600).tolist()
    np.random.seed(12345)
    single = CategoriesSampler(label, n_batch, 5, 6, 2)
    ref = [[b.tolist() for b in single] for _ in range(2)]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stream_worker, args=(r, world, port, n_batch, shard, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in procs:
        rank, epochs, touched, allv = q.get(timeout=240)
        results[rank] = (epochs, touched, allv)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    scatter = shard in ('driver', 'scatter')
    for e in range(2):
        glob = [None] * n_batch
        for rank, (epochs, _, _) in results.items():
            assert len(epochs[e]) == len(range(rank, n_batch, world))
            for k, b in enumerate(epochs[e]):
                glob[rank + k * world] = b
        assert glob == ref[e]
    for rank, (_, touched, allv) in results.items():
        assert touched == (rank == 0 or not scatter)            # scatter: only rank 0 draws
        assert allv == [float(sum(b) % 97) for b in ref[0]]     # gather_in_stream_order at world 8 with a ragged tail
