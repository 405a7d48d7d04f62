"""Epoch-indexed learning-rate schedulers used by the reference's warm-up / distillation drivers
(meta_tuning_sun_m/train_meta_warmup.py:19,140-141,217: `MultiStepLRScheduler(optimizer, milestones, decay_rate=0.5,
warmup_lr_init=1e-5, warmup_t=3)` stepped with `step(epoch - 1)`; sun_meta_training/offline.py:234 and
train_classifier.py:130-132: `CosineLRScheduler(optimizer, warmup_lr_init=..., t_initial=max_epoch, cycle_decay=0.1,
warmup_t=...)`).  The reference imports both from `timm.scheduler` - a third-party dependency that is absent from
/root/reference and from this image, unpinned by the reference (README.md:13 "timm (latest)"): PARITY UNPINNED.  What is
restated here is timm's published algorithm (timm/scheduler/scheduler.py, multistep_lr.py, cosine_lr.py, 0.6 - 1.0 series):

  * construction stores every param group's `lr` as its base value (`initial_lr`) and, when `warmup_t > 0`, immediately
    sets every group to `warmup_lr_init`;
  * `step(epoch)` sets the groups to `_get_lr(epoch)` (noise, k-decay != 1 and per-update stepping are not restated: the
    reference uses none of them);
  * warm-up: t < warmup_t  ->  warmup_lr_init + t * (base - warmup_lr_init) / warmup_t;
  * multi-step:            ->  base * decay_rate ** bisect_right(decay_t, t + 1);
  * cosine (cycle_mul = 1): i = t // t_initial, t_curr = t - i * t_initial, lr_max = base * cycle_decay ** i,
    i < cycle_limit ? lr_min + 0.5 (lr_max - lr_min) (1 + cos(pi * t_curr / t_initial)) : lr_min
    (`warmup_prefix=True` would shift t by warmup_t first; the reference leaves it False).
"""
import bisect
import math


class _Scheduler:
    def __init__(self, optimizer, warmup_t=0, warmup_lr_init=0.0):
        self.optimizer = optimizer
        for group in optimizer.param_groups:
            group.setdefault('initial_lr', group['lr'])
        self.base_values = [group['initial_lr'] for group in optimizer.param_groups]
        self.warmup_t = warmup_t
        self.warmup_lr_init = warmup_lr_init
        self.warmup_steps = [(v - warmup_lr_init) / warmup_t for v in self.base_values] if warmup_t else [1.0 for _ in self.base_values]
        if warmup_t:
            self._update([warmup_lr_init for _ in self.base_values])
        else:
            self._update(self.base_values)

    def _update(self, values):
        for group, v in zip(self.optimizer.param_groups, values):
            group['lr'] = v

    def _get_lr(self, t):
        raise NotImplementedError

    def get_epoch_values(self, epoch):
        return self._get_lr(epoch)

    def step(self, epoch, metric=None):
        self._update(self._get_lr(epoch))

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != 'optimizer'}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


class MultiStepLRScheduler(_Scheduler):
    def __init__(self, optimizer, decay_t, decay_rate=1.0, warmup_t=0, warmup_lr_init=0.0, t_in_epochs=True):
        self.decay_t = list(decay_t)
        self.decay_rate = decay_rate
        super().__init__(optimizer, warmup_t, warmup_lr_init)

    def _get_lr(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * s for s in self.warmup_steps]
        k = bisect.bisect_right(self.decay_t, t + 1)
        return [v * (self.decay_rate ** k) for v in self.base_values]


class CosineLRScheduler(_Scheduler):
    def __init__(self, optimizer, t_initial, lr_min=0.0, cycle_mul=1.0, cycle_decay=1.0, cycle_limit=1, warmup_t=0,
                 warmup_lr_init=0.0, warmup_prefix=False, t_in_epochs=True):
        assert t_initial > 0 and lr_min >= 0
        if cycle_mul != 1.0:
            raise NotImplementedError('cycle_mul != 1 is not used by the reference and not restated')
        self.t_initial, self.lr_min = t_initial, lr_min
        self.cycle_decay, self.cycle_limit, self.warmup_prefix = cycle_decay, cycle_limit, warmup_prefix
        super().__init__(optimizer, warmup_t, warmup_lr_init)

    def _get_lr(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * s for s in self.warmup_steps]
        if self.warmup_prefix:
            t = t - self.warmup_t
        i = t // self.t_initial
        t_curr = t - self.t_initial * i
        gamma = self.cycle_decay ** i
        if i < self.cycle_limit:
            return [self.lr_min + 0.5 * (v * gamma - self.lr_min) * (1 + math.cos(math.pi * t_curr / self.t_initial)) for v in self.base_values]
        return [self.lr_min for _ in self.base_values]
