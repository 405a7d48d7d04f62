"""Oracle of the distillation head (oracle/token_label_oracle.py) against goldens produced by the reference's own definitions of
TokenLabelOffline.forward, generate_softlabel and SoftTargetCrossEntropy (tests/golden/make_token_label_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import token_label_oracle as tlo

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'token_label.npz'))


def _tok(x):                                     # [B, C, H, W] -> [B, T, C]
    B, C = x.shape[:2]
    return np.ascontiguousarray(x.reshape(B, C, -1).transpose(0, 2, 1))


@pytest.mark.parametrize('k,bp', [(3, 10), (5, 7), (1, 0)])
def test_generate_softlabel_matches_reference(k, bp):
    soft = tlo.generate_softlabel(_tok(G['teacher_logits_token']), k=k, bp=bp)
    ref = G[f'soft_k{k}_bp{bp}']
    assert soft.shape == ref.shape == (150, 65)
    assert np.array_equal(soft, ref)             # two values only: bit-exact
    on = soft > 0.5
    assert (on.sum(axis=1) == 1).reshape(6, 25).sum(axis=1).tolist() == ([bp] * 6 if k > 1 else [25] * 6)     # background rows: one hot (column 1)
    assert not on[:, 64].any()                   # the extra column is never 'on' in the reference (offline.py:61,71)


def test_soft_target_cross_entropy_matches_reference():
    loss, grad = tlo.soft_target_cross_entropy(G['student_logits_token'], G['soft_k3_bp10'])
    assert loss == pytest.approx(float(G['soft_ce_loss']), rel=1e-6)
    assert np.abs(grad - G['soft_ce_dlogits']).max() <= 1e-7


def test_token_label_forward_matches_reference():
    sd = {k[len('tl_sd.'):]: G[k] for k in G.files if k.startswith('tl_sd.')}
    for tag, teacher in (('student', False), ('teacher', True)):
        y_token, y, x1 = tlo.token_label_forward(G['tl_map'], G['tl_pooled'], sd, teacher)
        assert np.abs(y_token - _tok(G[f'tl_{tag}_y_token'])).max() <= 1e-5
        assert np.abs(y - G[f'tl_{tag}_y']).max() <= 1e-5
    assert y_token.shape == (4, 25, 10) and _tok(G['tl_student_y_token']).shape == (4, 25, 11)


def test_adamw_matches_torch():
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(257, generator=g)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    pn, m, v = p0.numpy().astype(np.float64), np.zeros(257), np.zeros(257)
    for step in range(1, 6):
        grad = torch.randn(257, generator=g)
        p.grad = grad.clone()
        opt.step()
        pn, m, v = tlo.adamw_step(pn, grad.numpy().astype(np.float64), m, v, step, 3e-3, weight_decay=0.05)
        assert np.abs(pn - p.detach().numpy()).max() <= 2e-6
