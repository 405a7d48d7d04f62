"""Host glue with the reference's names (test_phase/utils/__init__.py:15-153)."""
import os
import shutil
import time

import numpy as np
import scipy.stats
import torch
import torch.nn.functional as F

from . import few_shot  # noqa: F401

class _LogSink:
    """Where `log` mirrors its lines (reference contract: utils.set_log_path(dir) then utils.log(obj[, filename]),
    test_phase/utils/__init__.py:15-25)."""
    directory = None


def set_log_path(path):
    _LogSink.directory = path


def log(obj, filename='log.txt'):
    """Print `obj`; when a log directory is set, append the same line to `<directory>/<filename>`."""
    line = str(obj)
    print(line)
    if _LogSink.directory is None:
        return
    with open(os.path.join(_LogSink.directory, filename), 'a') as sink:
        sink.write(line + '\n')


class Averager:
    """Weighted running mean with the reference's interface `add(value, weight)` / `item()` (utils/__init__.py:28-39), kept as a
    weighted sum and a total weight."""

    def __init__(self):
        self._wsum = 0.0
        self._w = 0.0

    def add(self, v, n=1.0):
        self._wsum += float(v) * float(n)
        self._w += float(n)

    @property
    def n(self):
        return self._w

    @property
    def v(self):
        return self.item()

    def item(self):
        return self._wsum / self._w if self._w > 0 else 0.0


class Timer:
    """Stopwatch: `s()` restarts, `t()` = seconds since the last (re)start (utils/__init__.py:42-50)."""

    def __init__(self):
        self.s()

    def s(self):
        self._t0 = time.monotonic()

    def t(self):
        return time.monotonic() - self._t0


def set_gpu(gpu):
    """Restrict the visible devices (utils/__init__.py:53-55); ROCm honours CUDA_VISIBLE_DEVICES like HIP_VISIBLE_DEVICES."""
    os.environ['CUDA_VISIBLE_DEVICES'] = str(gpu)
    print('set gpu:', gpu)


def ensure_path(path, remove=True, interactive=False):
    """Make `path` an existing directory; an existing one is emptied when `remove` is set (utils/__init__.py:58-67).  The
    reference asks on stdin unless the directory name starts with '_'; here that prompt is opt-in (`interactive=True`)
    because the drivers run unattended under torchrun."""
    if not os.path.isdir(path):
        os.makedirs(path)
        return
    if not remove:
        return
    name = os.path.basename(os.path.normpath(path))
    if interactive and not name.startswith('_'):
        if input('{} exists, remove? ([y]/n): '.format(path)).strip().lower() == 'n':
            return
    shutil.rmtree(path)
    os.makedirs(path)


def time_str(t):
    """Seconds -> '12.3s' / '4.5m' / '1.2h' (utils/__init__.py:70-75)."""
    for unit, span in (('h', 3600.0), ('m', 60.0)):
        if t >= span:
            return '%.1f%s' % (t / span, unit)
    return '%.1f%s' % (t, 's')


def compute_logits(feat, proto, metric='dot', temp=1.0):
    """utils/__init__.py:78-101.  On GPU tensors of the episodic form [E,Q,D] x [E,way,D] the
    'dot' / 'cos' / 'sqr' metrics run in the fsvit head kernel; small host-side (CPU) calls keep
    torch semantics for config plumbing."""
    assert feat.dim() == proto.dim()
    if feat.is_cuda and feat.dim() == 3 and metric in ('cos', 'sqr', 'dot'):
        from ..engine import ops
        return ops.proto_head(proto.unsqueeze(2), feat, temp, metric)[0]
    if metric == 'cos':
        feat, proto, metric = F.normalize(feat, dim=-1), F.normalize(proto, dim=-1), 'dot'
    if metric == 'dot':
        logits = feat @ proto.transpose(-1, -2)
    elif metric == 'sqr':
        logits = -(feat.unsqueeze(-2) - proto.unsqueeze(-3)).pow(2).sum(dim=-1)
    else:
        raise ValueError(metric)
    return logits * temp


def compute_acc(logits, label, reduction='mean'):
    ret = (torch.argmax(logits, dim=1) == label).float()
    if reduction == 'none':
        return ret.detach()
    return ret.mean().item()


def compute_n_params(model, return_str=True):
    tot = sum(int(np.prod(p.shape)) for p in model.parameters())
    if return_str:
        return '{:.1f}M'.format(tot / 1e6) if tot >= 1e6 else '{:.1f}K'.format(tot / 1e3)
    return tot


def mean_confidence_interval(data, confidence=0.95):
    """95 % CI half-width over per-batch accuracies (test_few_shot.py:20-25)."""
    a = 1.0 * np.array(data)
    n = len(a)
    se = scipy.stats.sem(a)
    return se * scipy.stats.t.ppf((1 + confidence) / 2., n - 1)


class FsvitSGD(torch.optim.Optimizer):
    """torch.optim.SGD(params, lr, momentum, weight_decay) semantics (no dampening / nesterov) with the update done by
    the HIP kernel behind fsvit_sgd_step; exposes param_groups so MultiStepLR drives `lr` as in the reference."""

    def __init__(self, params, lr, momentum=0.9, weight_decay=0.):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        from ..engine import ops
        for group in self.param_groups:
            batches = {True: ([], [], []), False: ([], [], [])}            # first step of a tensor (no momentum buffer yet) or not
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                first = 'momentum_buffer' not in st
                if first:
                    st['momentum_buffer'] = torch.empty_like(p, memory_format=torch.contiguous_format)
                data = p.data
                if not data.is_contiguous() or data.dtype != torch.float32 or not data.is_cuda:
                    raise RuntimeError('FsvitSGD: parameters must be contiguous fp32 CUDA tensors')
                ps, gs, bs = batches[first]
                ps.append(data.view(-1)); gs.append(p.grad.contiguous().view(-1)); bs.append(st['momentum_buffer'].view(-1))
            for first, (ps, gs, bs) in batches.items():
                ops.sgd_step_multi(ps, gs, bs, group['lr'], group['momentum'], group['weight_decay'], first)      # one launch per param_group


def make_optimizer(params, name, lr, weight_decay=None, milestones=None, gamma=0.1):
    """utils/__init__.py:128-139 of the reference's meta_tuning_sun_m: SGD(momentum 0.9) or Adam, optional MultiStepLR."""
    if weight_decay is None:
        weight_decay = 0.
    if name == 'sgd':
        optimizer = FsvitSGD(params, lr, momentum=0.9, weight_decay=weight_decay)
    elif name == 'adam':
        optimizer = torch.optim.Adam(params, lr, weight_decay=weight_decay)
    else:
        raise ValueError(name)
    lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones, gamma=gamma) if milestones else None
    return optimizer, lr_scheduler


def freeze_bn(model):
    """utils/__init__.py:150-153: BatchNorm2d modules to eval(); the HIP trainer then normalises with the running statistics
    (fsvit_visformer_trainer_set_freeze_bn)."""
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
