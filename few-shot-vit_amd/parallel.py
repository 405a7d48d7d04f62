"""Episode-parallel evaluation across the GPUs of one node: one process per GPU, no data-path
collective.  Every rank replays the same seeded sampler stream and keeps batches rank::world
(datasets/samplers.py); the only exchange is the per-batch statistics at the end, over
torch.distributed (backend 'nccl' == RCCL over xGMI on the GPU box, 'gloo' in the CPU tests).

The reference does this with nn.DataParallel scatter on the episode axis (test_few_shot.py:65-66);
its CI is computed over per-batch accuracies in stream order (test_few_shot.py:94,114-116), so the
gather below restores exactly that order and the CI is bit-identical to a single-process run."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def gather_in_stream_order(values: torch.Tensor, n_total: int, rank: int, world: int) -> torch.Tensor:
    """values: this rank's per-batch statistics for global batches rank, rank+world, ... (1-D or
    [n, k]).  Returns all n_total rows in global stream order on every rank (ONE all-gather)."""
    if world == 1:
        return values
    per = (n_total + world - 1) // world
    shape = (per,) + tuple(values.shape[1:])
    pad = torch.zeros(shape, dtype=values.dtype, device=values.device)
    pad[:values.shape[0]] = values
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    stacked = torch.stack(out, dim=1)                     # [per, world, ...]: row-major == global order
    return stacked.reshape((per * world,) + tuple(values.shape[1:]))[:n_total]


def allreduce_mean_stats(sum_acc: float, sum_sq: float, n: float, device) -> tuple:
    """The 3 x fp64 all-reduce of the north star: (sum acc, sum acc^2, n) -> mean, unbiased var, n."""
    t = torch.tensor([sum_acc, sum_sq, n], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t)
    s, q, cnt = [float(v) for v in t]
    mean = s / cnt
    var = max(0.0, (q - cnt * mean * mean) / max(1.0, cnt - 1.0))
    return mean, var, cnt


def allreduce_mean_grads(params) -> None:
    """The one exchange of a data-parallel training step: every rank holds the gradient of ITS episodes' mean loss;
    flatten all of them into one bucket, all-reduce (sum) once over RCCL, divide by the world size and scatter the views
    back.  One 50 MB bucket for Visformer-S: large enough to run the xGMI ring at bandwidth, one launch instead of 150."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat)
    flat /= dist.get_world_size()
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def shard_episode_axis(idx: torch.Tensor, ep_per_batch: int, rank: int, world: int) -> torch.Tensor:
    """One sampler batch = ep_per_batch episodes of equal length, concatenated (samplers.py).  Rank r keeps episodes
    [r * ep_per_batch / world, (r + 1) * ep_per_batch / world) - the slice nn.DataParallel would scatter to GPU r."""
    if ep_per_batch % world:
        raise ValueError(f'ep_per_batch={ep_per_batch} must divide over {world} ranks')
    ep_local = ep_per_batch // world
    return idx.view(ep_per_batch, -1)[rank * ep_local:(rank + 1) * ep_local].reshape(-1)
