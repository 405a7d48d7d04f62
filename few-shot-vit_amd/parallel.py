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


def sampler_shard(world):
    """CategoriesSampler(shard=...) for a loop that EVERY rank runs in lockstep: 'scatter' (rank 0 draws, one broadcast per epoch) when
    torch.distributed is up with this world size, else None (= 'replay': every rank draws the whole stream)."""
    return 'scatter' if world > 1 and dist.is_available() and dist.is_initialized() and dist.get_world_size() == world else None


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC (RCCL between processes on this driver); effective if no HIP call was made yet
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
    dev = values.device
    if values.is_cuda and dist.get_backend() == 'gloo':   # gloo ranks that compute on a GPU (tests: two ranks on one device): gather on the host
        values = values.cpu()
    pad = torch.zeros(shape, dtype=values.dtype, device=values.device)
    pad[:values.shape[0]] = values
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    stacked = torch.stack(out, dim=1)                     # [per, world, ...]: row-major == global order
    return stacked.reshape((per * world,) + tuple(values.shape[1:]))[:n_total].to(dev)


def allreduce_mean_stats(sum_acc: float, sum_sq: float, n: float, device) -> tuple:
    """The 3 x fp64 all-reduce of the north star: (sum acc, sum acc^2, n) -> mean, unbiased var, n."""
    t = torch.tensor([sum_acc, sum_sq, n], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t)
    s, q, cnt = [float(v) for v in t]
    mean = s / cnt
    var = max(0.0, (q - cnt * mean * mean) / max(1.0, cnt - 1.0))
    return mean, var, cnt


class GradBucket:
    """The one exchange of a data-parallel training step, in place: every rank holds the gradient of ITS episodes' mean loss in ONE
    persistent flat fp32 buffer (50 MB for Visformer-S: large enough to run the xGMI ring at bandwidth, one collective instead of 150) and
    `allreduce_mean()` is a single all-reduce over that buffer - no flatten (`cat`) before it and no scatter after it.

    `params`: the model's parameters (an iterable, or an nn.Module).  Each parameter's `.grad` becomes a view into the bucket.  The HIP
    trainer writes its gradients straight into those views (autograd.VisformerTrainFn, `trainer.grad_sink`): with one forward / backward
    per step (train_meta.py:161-170: zero_grad, backward, step) no torch kernel touches the gradients between the backward and the
    collective.  Gradients that reach a parameter through plain autograd (the head's `temp`, a zero_grad(set_to_none=True) in between) are
    copied into their view by `allreduce_mean()` - correct either way, the sink only removes the copy."""

    def __init__(self, params, bucket_dtype=None):
        import torch.nn as nn
        self.module = params if isinstance(params, nn.Module) else None
        ps = list(params.parameters()) if self.module is not None else list(params)
        self.params = [p for p in ps if p.requires_grad]
        if not self.params:
            raise ValueError('GradBucket: no trainable parameters')
        dev = self.params[0].device
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
        self.wire_dtype = bucket_dtype               # e.g. torch.bfloat16: halves the bytes on the xGMI ring, gradients are rounded once
        self.views, off = [], 0
        for p in self.params:
            n = p.numel()
            self.views.append(self.flat[off:off + n].view(p.shape))
            off += n
        self.attach()

    def attach(self):
        """Point every `.grad` at its view and hand the encoder trainers their sink (name -> (parameter, view))."""
        for p, v in zip(self.params, self.views):
            p.grad = v
        if self.module is not None:
            by_id = {id(p): v for p, v in zip(self.params, self.views)}
            for m in self.module.modules():
                if hasattr(m, 'trainer') and callable(m.trainer) and hasattr(m, 'named_parameters'):
                    m._grad_sink = {k: (p, by_id[id(p)]) for k, p in m.named_parameters() if id(p) in by_id}

    def detach(self):
        if self.module is not None:
            for m in self.module.modules():
                if hasattr(m, '_grad_sink'):
                    m._grad_sink = None

    def _collect(self):
        """Every gradient into its view.  Returns the parameters that HAVE no gradient (unused in this graph; every rank runs the same
        graph): their views are zeroed for the collective, and allreduce_mean() hands them back as `.grad = None`, so that the optimizer
        skips them (no momentum buffer, no weight decay) exactly as it does in a single-process run."""
        unused = []
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                v.zero_()
                unused.append(p)
                continue
            if g.data_ptr() != v.data_ptr():
                v.copy_(g)
            p.grad = v
        return unused

    def allreduce_mean(self) -> None:
        unused = self._collect()
        if dist.is_initialized() and dist.get_world_size() > 1:
            world = dist.get_world_size()
            if self.wire_dtype is not None and self.wire_dtype != torch.float32:
                wire = self.flat.to(self.wire_dtype)
                dist.all_reduce(wire)
                self.flat.copy_(wire).div_(world)
            else:
                dist.all_reduce(self.flat)
                self.flat.div_(world)
        for p in unused:
            p.grad = None


def allreduce_mean_grads(params) -> None:
    """Gradient all-reduce (mean) for loops that do not keep a GradBucket (train_classifier.py, offline.py): flatten, ONE all-reduce over
    RCCL, divide, scatter the views back."""
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
