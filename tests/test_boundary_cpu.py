"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/fsvit.h declares (no compute calls: there is no GPU here), the host-side mirror of the
reference interface behaves like the reference, and the product refuses to run without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from fewshot_vit_amd import _lib
    header = open(os.path.join(REPO, 'include', 'fsvit.h')).read()
    declared = set(re.findall(r'\b(fsvit_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.fsvit_version().decode().startswith('fsvit ')
    assert 'gfx950' in lib.fsvit_version().decode()


def test_registry_and_state_dict_contract():
    from fewshot_vit_amd import models
    from oracle import visformer_oracle as vo
    assert models.make(None) is None
    with pytest.raises(KeyError):
        models.make('definitely-not-registered')
    m = models.make('meta-baseline', encoder='visformer_micro_80', encoder_args={'drop_path_rate': 0.5})
    sd = m.state_dict()
    sh = vo.state_dict_shapes(vo.VisformerCfg(), 'encoder.')
    sh['temp'] = ()
    assert set(sd) == set(sh)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(sh[k]), k
    assert m.encoder.out_dim == 512
    assert float(m.temp) == 10.0
    assert sum(p.numel() for p in m.parameters()) == 12531393
    # checkpoint dict round trip (train_meta.py:241-257 schema -> models.load)
    ck = {'model': 'meta-baseline', 'model_args': {'encoder': 'visformer_micro_80', 'encoder_args': {}}, 'model_sd': sd}
    m2 = models.load(ck)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    # `load_encoder:` path keeps only .encoder (test_few_shot.py:61-63)
    fresh = models.make('meta-baseline', encoder=None)
    fresh.encoder = m2.encoder
    assert fresh.encoder.out_dim == 512


@pytest.mark.skipif(torch.cuda.is_available(), reason='CPU-only behaviour')
def test_no_cpu_fallback():
    from fewshot_vit_amd import models
    m = models.make('meta-baseline', encoder='visformer_micro_80', encoder_args={}).eval()
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(torch.zeros(1, 5, 1, 3, 80, 80), torch.zeros(1, 75, 3, 80, 80))
    m.train()                      # the training step has no CPU path either
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(torch.zeros(1, 5, 1, 3, 80, 80), torch.zeros(1, 75, 3, 80, 80))
    from fewshot_vit_amd import utils
    utils.freeze_bn(m)             # frozen-BN training (train_meta.py:156-157) runs on the HIP trainer too: still no CPU path
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(torch.zeros(1, 5, 1, 3, 80, 80), torch.zeros(1, 75, 3, 80, 80))
    m.train()
    m.encoder.stem.bn1.eval()      # a MIX of frozen and live BatchNorm layers is not built: say so rather than train with the wrong statistics
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 5, 1, 3, 80, 80), torch.zeros(1, 75, 3, 80, 80))


def test_host_helpers_match_known_answers(golden_dir):
    import json
    from fewshot_vit_amd import utils
    from fewshot_vit_amd.datasets.samplers import CategoriesSampler
    from fewshot_vit_amd.utils import few_shot as fs
    ka = json.load(open(os.path.join(golden_dir, 'host_known_answers.json')))
    label = np.repeat(np.arange(20), 600).tolist()
    np.random.seed(12345)
    eps = [b.tolist() for b in CategoriesSampler(label, 3, 5, 16, 1)]
    assert eps == ka['sampler_seed12345_20x600_3x5x16']
    np.random.seed(0)
    assert [int(b.sum()) for b in CategoriesSampler(label, 2, 5, 20, 4)] == ka['sampler_seed0_20x600_2x(4ep)x5x20_sum']
    assert fs.make_nk_label(5, 3, 2).tolist() == ka['make_nk_label_5_3_2']
    xs, xq = fs.split_shot_query(torch.arange(20), 5, 1, 3, 1)
    assert xs.flatten().tolist() == ka['split_arange20_5_1_3']['shot']
    assert xq.flatten().tolist() == ka['split_arange20_5_1_3']['query']
    assert abs(utils.mean_confidence_interval([.6, .8, .7, .9, .5]) - ka['mci_.6_.8_.7_.9_.5']) < 1e-12
    c = ka['compute_logits']
    feat, proto = torch.tensor(c['feat']), torch.tensor(c['proto'])
    for key, metric, temp in (('dot_t2', 'dot', 2.0), ('cos_t10', 'cos', 10.0), ('sqr_t1', 'sqr', 1.0)):
        assert (utils.compute_logits(feat, proto, metric, temp) - torch.tensor(c[key])).abs().max() < 2e-6
    a = ka['acc_ce']
    assert utils.compute_acc(torch.tensor(a['logits']), torch.tensor(a['label'])) == pytest.approx(a['acc'])
    av = utils.Averager()
    for v, n in ((0.5, 80), (0.75, 100), (0.2, 3)):
        av.add(v, n)
    assert av.item() == pytest.approx(ka['averager'], abs=1e-12)


def test_sampler_rank_sharding_partitions_the_reference_stream():
    from fewshot_vit_amd.datasets.samplers import CategoriesSampler
    label = np.repeat(np.arange(20), 600).tolist()
    np.random.seed(12345)
    full = [b.tolist() for b in CategoriesSampler(label, 10, 5, 6, 1)]
    for world in (2, 4):
        got = {}
        for r in range(world):
            np.random.seed(12345)
            s = CategoriesSampler(label, 10, 5, 6, 1, rank=r, world_size=world)
            mine = [b.tolist() for b in s]
            assert len(mine) == len(s)
            for j, b in enumerate(mine):
                got[r + j * world] = b
        assert [got[i] for i in range(10)] == full


def _register_tiny():
    """The registry name the reference-written checkpoints carry (tests/golden/make_golden.py checkpoint_golden registers the same factory on the
    reference side): tiny Visformer, one block per stage."""
    from fewshot_vit_amd.models import register
    from fewshot_vit_amd.models.visformer import Visformer
    register('tiny_visformer')(lambda **kw: Visformer(img_size=80, init_channels=8, embed_dim=64, depth=[1, 1, 1], num_heads=6, mlp_ratio=4., group=8, **kw))


def test_reference_written_checkpoints_rebuild_through_models_load(golden_dir):
    """The checkpoints in tests/golden/ref_ckpt_*.pth were written by the REFERENCE (torch.save of the dict of meta_tuning_sun_m/train_meta.py:241-257 from
    a live reference model, `model: meta-baseline` and `model: classifier`).  `models.load` (models.py:21-26) must rebuild both with strict key / shape
    agreement, keep every tensor bit for bit, and the `load_encoder:` route of test_few_shot.py:61-63 must take `.encoder` of the classifier one."""
    import os
    from fewshot_vit_amd import models
    _register_tiny()
    for name, kind in (('ref_ckpt_meta_baseline.pth', 'meta-baseline'), ('ref_ckpt_classifier.pth', 'classifier')):
        ck = torch.load(os.path.join(golden_dir, name), map_location='cpu', weights_only=False)
        assert {'file', 'config', 'model', 'model_args', 'model_sd', 'training'} <= set(ck) and ck['model'] == kind
        assert {'epoch', 'optimizer', 'optimizer_args', 'optimizer_sd'} <= set(ck['training'])
        m = models.load(ck)
        sd = m.state_dict()
        assert set(sd) == set(ck['model_sd'])
        for k, v in ck['model_sd'].items():
            assert torch.equal(sd[k].cpu(), v), k
        assert m.encoder.out_dim == 128
        if kind == 'meta-baseline':
            assert float(m.temp) == 7.5
        else:
            assert tuple(sd['classifier.linear.weight'].shape) == (7, 128)
            fresh = models.make('meta-baseline', encoder=None)
            fresh.encoder = m.encoder
            assert float(fresh.temp) == 10.0 and fresh.encoder.out_dim == 128
