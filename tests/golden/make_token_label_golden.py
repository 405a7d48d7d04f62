#!/usr/bin/env python3
"""Golden vectors for the distillation head of sun_meta_training/offline.py (build container only; needs /root/reference).

offline.py cannot be imported (timm, tensorboardX), but `generate_softlabel` (:57-76) and `SoftTargetCrossEntropy` (:34-45) are
plain torch: their definitions are extracted from the reference file's AST at generation time and executed AS THEY ARE on seeded
inputs; `TokenLabelOffline.forward` (models/token_label.py:48-60) is run the same way with the encoder replaced by the seeded
(map, pooled) pair and `models.make('linear-classifier')` by the reference's own LinearClassifier definition (:27-34).
Only inputs and outputs are stored:  tests/golden/token_label.npz.   Run from the repo root: python tests/golden/make_token_label_golden.py
"""
import ast
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = '/root/reference/sun_meta_training'


def extract(path, names):
    src = open(path).read()
    tree = ast.parse(src)
    ns = {'torch': torch, 'nn': nn, 'F': F}
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            node.decorator_list = []                      # @register(...) needs the registry module
            code = compile(ast.Module(body=[node], type_ignores=[]), path, 'exec')
            exec(code, ns)
    return ns


def main():
    off = extract(f'{REF}/offline.py', {'generate_softlabel', 'SoftTargetCrossEntropy'})
    tl = extract(f'{REF}/models/token_label.py', {'LinearClassifier', 'TokenLabelOffline'})
    g = torch.Generator().manual_seed(20221001)
    out = {}
    # ---- generate_softlabel: teacher token logits [B, 64, 5, 5]
    B, C, H = 6, 64, 5
    # (as in the reference, the [B, C, 5, 5] tensor is a permuted view of the classifier's contiguous [B, 5, 5, C] output: :69's .view needs that)
    lt = (torch.randn(B, H, H, C, generator=g) * 2.0).permute(0, 3, 1, 2)
    out['teacher_logits_token'] = lt.contiguous().numpy()
    for k, bp in ((3, 10), (5, 7), (1, 0)):
        out[f'soft_k{k}_bp{bp}'] = off['generate_softlabel'](lt, k=k, bp=bp, device='cpu').numpy()
    # ---- SoftTargetCrossEntropy on student token logits [B*25, 65]
    ls = torch.randn(B * H * H, C + 1, generator=g)
    out['student_logits_token'] = ls.numpy()
    soft = torch.from_numpy(out['soft_k3_bp10'])
    ls_req = ls.clone().requires_grad_(True)
    loss = off['SoftTargetCrossEntropy']()(ls_req, soft)
    loss.backward()
    out['soft_ce_loss'] = np.array(loss.item(), dtype=np.float64)
    out['soft_ce_dlogits'] = ls_req.grad.numpy()
    # ---- TokenLabelOffline.forward with a stub encoder returning the seeded (map, pooled) pair
    D = 32
    fmap = torch.randn(4, D, H, H, generator=g)
    pooled = fmap.mean(dim=(2, 3))

    class Enc(nn.Module):
        out_dim = D

        def forward(self, x):
            return fmap, pooled

    class _Models:
        @staticmethod
        def make(name, **kw):
            if name == 'stub-encoder':
                return Enc()
            assert name == 'linear-classifier'
            return tl['LinearClassifier'](**kw)

    tl['models'] = _Models
    tl['TokenLabelOffline'].__init__.__globals__['models'] = _Models
    torch.manual_seed(7)
    m = tl['TokenLabelOffline']('stub-encoder', {}, 'linear-classifier', {'n_classes': 10})
    out['tl_map'], out['tl_pooled'] = fmap.numpy(), pooled.numpy()
    for k, v in m.state_dict().items():
        out['tl_sd.' + k] = v.numpy()
    with torch.no_grad():
        for teacher in (False, True):
            y_token, y, x1 = m(torch.zeros(4, 3, 8, 8), teacher)
            tag = 'teacher' if teacher else 'student'
            out[f'tl_{tag}_y_token'], out[f'tl_{tag}_y'] = y_token.numpy(), y.numpy()
    np.savez_compressed(os.path.join(REPO, 'tests', 'golden', 'token_label.npz'), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
