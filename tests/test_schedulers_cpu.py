"""timm-style epoch schedulers restated in utils/schedulers.py (train_meta_warmup.py:140-141,217; offline.py:234).
timm is absent (parity unpinned): the known answers below are evaluated by hand from timm's published formulas
(multistep_lr.py: base * decay_rate ** bisect_right(decay_t, t + 1); cosine_lr.py with cycle_mul = 1; linear warm-up)."""
import math

import pytest
import torch


def _opt(lr=0.1, n_groups=1):
    ps = [torch.nn.Parameter(torch.zeros(1)) for _ in range(n_groups)]
    return torch.optim.SGD([{'params': [p], 'lr': lr * (i + 1)} for i, p in enumerate(ps)], lr=lr)


def test_multistep_with_warmup_matches_reference_usage():
    from fewshot_vit_amd.utils.schedulers import MultiStepLRScheduler
    opt = _opt(1e-3)
    s = MultiStepLRScheduler(opt, [20, 40, 60, 80], decay_rate=0.5, warmup_lr_init=1e-5, warmup_t=3)
    assert opt.param_groups[0]['lr'] == pytest.approx(1e-5)                 # construction drops every group to warmup_lr_init
    assert opt.param_groups[0]['initial_lr'] == pytest.approx(1e-3)
    want = {0: 1e-5, 1: 1e-5 + (1e-3 - 1e-5) / 3, 2: 1e-5 + 2 * (1e-3 - 1e-5) / 3, 3: 1e-3, 18: 1e-3,
            19: 5e-4,                                                       # bisect_right([20,..], 19 + 1) = 1
            38: 5e-4, 39: 2.5e-4, 59: 1.25e-4, 79: 6.25e-5, 99: 6.25e-5}
    for t, lr in want.items():
        s.step(t)
        assert opt.param_groups[0]['lr'] == pytest.approx(lr, rel=1e-12), t
    # the reference steps with (epoch - 1) at the END of epoch `epoch`: epochs 1 and 2 both train at 1e-5
    opt = _opt(1e-3)
    s = MultiStepLRScheduler(opt, [20, 40], decay_rate=0.5, warmup_lr_init=1e-5, warmup_t=3)
    lrs = []
    for epoch in range(1, 6):
        lrs.append(opt.param_groups[0]['lr'])
        s.step(epoch - 1)
    assert lrs == pytest.approx([1e-5, 1e-5, 1e-5 + (1e-3 - 1e-5) / 3, 1e-5 + 2 * (1e-3 - 1e-5) / 3, 1e-3])


def test_cosine_with_warmup_and_cycle_decay():
    from fewshot_vit_amd.utils.schedulers import CosineLRScheduler
    opt = _opt(5e-4, n_groups=2)                                            # groups at 5e-4 and 1e-3
    s = CosineLRScheduler(opt, t_initial=100, cycle_decay=0.1, warmup_t=5, warmup_lr_init=1e-6)
    assert [g['lr'] for g in opt.param_groups] == pytest.approx([1e-6, 1e-6])
    s.step(2)
    assert opt.param_groups[0]['lr'] == pytest.approx(1e-6 + 2 * (5e-4 - 1e-6) / 5)
    assert opt.param_groups[1]['lr'] == pytest.approx(1e-6 + 2 * (1e-3 - 1e-6) / 5)
    for t in (5, 50, 99):                                                    # no warm-up prefix: the cosine is indexed by t itself
        s.step(t)
        assert opt.param_groups[0]['lr'] == pytest.approx(0.5 * 5e-4 * (1 + math.cos(math.pi * t / 100)))
    s.step(100)                                                              # second cycle is past cycle_limit = 1 -> lr_min
    assert opt.param_groups[0]['lr'] == 0.0
    s2 = CosineLRScheduler(_opt(1.0), t_initial=10, cycle_decay=0.1, cycle_limit=2, lr_min=0.01)
    assert s2.get_epoch_values(10)[0] == pytest.approx(0.01 + 0.5 * (0.1 - 0.01) * 2)          # cycle 1 starts at base * cycle_decay
    assert s2.get_epoch_values(15)[0] == pytest.approx(0.01 + 0.5 * (0.1 - 0.01))
    sd = s.state_dict()
    assert 'optimizer' not in sd and sd['t_initial'] == 100
