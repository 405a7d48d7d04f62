"""The episodic eval driver (reference test_few_shot.py surface) on one GPU."""
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _config():
    return yaml.safe_load(open(os.path.join(REPO, 'few-shot-vit_amd', 'configs', 'test_synthetic.yaml')))


def test_driver_matches_oracle_and_is_launch_size_invariant():
    from fewshot_vit_amd import datasets, synthetic, test_few_shot
    from fewshot_vit_amd.datasets.samplers import CategoriesSampler
    from oracle import fewshot_oracle as fo
    from oracle import visformer_oracle as vo
    cfg = _config()
    logs = []
    a = test_few_shot.evaluate(cfg, shot=1, n_batch=6, launch_batches=4, numerics='bf16', log=logs.append)
    b = test_few_shot.evaluate(cfg, shot=1, n_batch=6, launch_batches=1, numerics='bf16', log=logs.append)
    assert a['va_lst'] == b['va_lst'] and a['n'] == 6            # batching per launch does not change any episode
    assert any(l.startswith('test epoch 1: acc=') for l in logs)
    # oracle on the first two episodes of the same seeded stream
    ds = datasets.make(cfg['dataset'], **cfg['dataset_args'])
    vcfg = vo.VisformerCfg()
    shapes = vo.state_dict_shapes(vcfg, prefix='encoder.')
    shapes['temp'] = ()
    sd = synthetic.synthetic_checkpoint_sd(shapes)
    np.random.seed(12345)
    label = fo.make_nk_label(5, 15, 1)
    for e, idx in enumerate(CategoriesSampler(ds.label, 2, 5, 16, 1)):
        x = torch.stack([ds[int(i)][0] for i in idx])
        xs, xq = fo.split_shot_query(x.numpy(), 5, 1, 15, 1)
        logits = vo.meta_baseline_forward(sd, torch.from_numpy(xs), torch.from_numpy(xq), vcfg)[0].numpy()
        assert a['va_lst'][e] == pytest.approx(fo.compute_acc(logits, label), abs=0.0401)   # <= 3 of 75 near-tie flips in bf16
    p = test_few_shot.evaluate(cfg, shot=1, n_batch=2, launch_batches=2, numerics='parity', log=logs.append)
    np.random.seed(12345)
    for e, idx in enumerate(CategoriesSampler(ds.label, 2, 5, 16, 1)):
        x = torch.stack([ds[int(i)][0] for i in idx])
        xs, xq = fo.split_shot_query(x.numpy(), 5, 1, 15, 1)
        logits = vo.meta_baseline_forward(sd, torch.from_numpy(xs), torch.from_numpy(xq), vcfg)[0].numpy()
        assert p['va_lst'][e] == pytest.approx(fo.compute_acc(logits, label), abs=1e-6)      # exact-fp32 mode: same arg-max
