"""Distillation head on the GPU (C-ABI kernels behind the `token-label` / `linear-classifier` models, generate_softlabel,
SoftTargetCrossEntropy, AdamW) against the reference-pinned oracle and the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'token_label.npz'))


@pytest.mark.parametrize('k,bp', [(3, 10), (5, 7), (1, 0)])
def test_generate_softlabel_bit_exact_vs_reference_golden(k, bp):
    from fewshot_vit_amd.models.classifier import generate_softlabel
    lt = torch.from_numpy(G['teacher_logits_token']).cuda()
    soft = generate_softlabel(lt, k=k, bp=bp).cpu().numpy()
    assert np.array_equal(soft, G[f'soft_k{k}_bp{bp}'])


def test_soft_target_ce_loss_and_gradient():
    from fewshot_vit_amd.models.classifier import SoftTargetCrossEntropy
    z = torch.from_numpy(G['student_logits_token']).cuda().requires_grad_(True)
    t = torch.from_numpy(G['soft_k3_bp10']).cuda()
    loss = SoftTargetCrossEntropy()(z, t)
    (2.0 * loss).backward()
    assert float(loss) == pytest.approx(float(G['soft_ce_loss']), rel=2e-6)
    assert (z.grad.cpu().numpy() - 2.0 * G['soft_ce_dlogits']).__abs__().max() <= 2e-7


def test_token_label_model_matches_reference_and_oracle_gradients():
    from fewshot_vit_amd import models
    from oracle import token_label_oracle as tlo
    fmap, pooled = torch.from_numpy(G['tl_map']).cuda(), torch.from_numpy(G['tl_pooled']).cuda()

    class Enc(nn.Module):
        out_dim = 32

        def forward(self, x):
            return fmap_req, pooled_req

    fmap_req, pooled_req = fmap.clone().requires_grad_(True), pooled.clone().requires_grad_(True)
    m = models.make('token-label', encoder=Enc(), encoder_args={}, classifier='linear-classifier', classifier_args={'n_classes': 10})
    sd = {k[len('tl_sd.'):]: torch.from_numpy(G[k]) for k in G.files if k.startswith('tl_sd.')}
    m.load_state_dict(sd, strict=True)                                            # reference key names / shapes
    for tag, teacher in (('student', False), ('teacher', True)):
        y_token, y, x1 = m(torch.zeros(4, 3, 8, 8, device='cuda'), teacher)
        assert (y_token.detach().cpu().numpy() - G[f'tl_{tag}_y_token']).__abs__().max() <= 1e-5
        assert (y.detach().cpu().numpy() - G[f'tl_{tag}_y']).__abs__().max() <= 1e-5
    # backward of the student pass: every gradient against torch autograd of the same math on the CPU
    y_token, y, _ = m(torch.zeros(4, 3, 8, 8, device='cuda'), False)
    gt = torch.Generator().manual_seed(5)
    wt, wy = torch.randn(y_token.shape, generator=gt), torch.randn(y.shape, generator=gt)
    ((y_token * wt.cuda()).sum() + (y * wy.cuda()).sum()).backward()
    fm_c, po_c = fmap.cpu().clone().requires_grad_(True), pooled.cpu().clone().requires_grad_(True)
    ps = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yt_c = torch.nn.functional.linear(fm_c.permute(0, 2, 3, 1), ps['classifier_local.linear.weight'], ps['classifier_local.linear.bias']).permute(0, 3, 1, 2)
    y_c = torch.nn.functional.linear(po_c, ps['classifier.linear.weight'], ps['classifier.linear.bias'])
    ((yt_c * wt).sum() + (y_c * wy).sum()).backward()
    assert (fmap_req.grad.cpu() - fm_c.grad).abs().max() <= 1e-5
    assert (pooled_req.grad.cpu() - po_c.grad).abs().max() <= 1e-5
    for k, p in m.named_parameters():
        assert (p.grad.cpu() - ps[k].grad).abs().max() <= 2e-5 * max(1.0, float(ps[k].grad.abs().max())), k


def test_adamw_kernel_matches_torch():
    from fewshot_vit_amd.models.classifier import FsvitAdamW
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(1000, generator=g)
    pa, pb = nn.Parameter(p0.clone().cuda()), nn.Parameter(p0.clone())
    oa = FsvitAdamW([pa], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    ob = torch.optim.AdamW([pb], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    for _ in range(7):
        grad = torch.randn(1000, generator=g)
        pa.grad, pb.grad = grad.cuda(), grad.clone()
        oa.step()
        ob.step()
    assert (pa.detach().cpu() - pb.detach()).abs().max() <= 2e-6


def test_distillation_step_reduces_loss():
    """offline.py:283-303 on a stub encoder: cls_loss + 0.5 * token_loss, AdamW, a few steps on one batch."""
    from fewshot_vit_amd import models
    from fewshot_vit_amd.models.classifier import FsvitAdamW, SoftTargetCrossEntropy, generate_softlabel
    torch.manual_seed(0)

    class Enc(nn.Module):
        out_dim = 64

        def __init__(self):
            super().__init__()
            self.scale = nn.Parameter(torch.ones(64, 1, 1))

        def forward(self, x):
            fm = x * self.scale
            return fm, fm.mean(dim=(2, 3))

    x = torch.randn(16, 64, 5, 5, device='cuda')
    label = torch.randint(0, 12, (16,), device='cuda')
    model = models.make('token-label', encoder=Enc(), encoder_args={}, classifier='linear-classifier', classifier_args={'n_classes': 12})
    teacher = models.make('token-label', encoder=Enc(), encoder_args={}, classifier='linear-classifier', classifier_args={'n_classes': 12}).eval()
    opt = FsvitAdamW(model.parameters(), lr=5e-3, weight_decay=0.05)
    crit = SoftTargetCrossEntropy()
    losses = []
    for _ in range(8):
        logits_token, logits, _ = model(x)
        with torch.no_grad():
            lt_t, _, _ = teacher(x, True)
            soft = generate_softlabel(lt_t, k=3, bp=10)
        loss = torch.nn.functional.cross_entropy(logits, label) + 0.5 * crit(logits_token.permute(0, 2, 3, 1).reshape(-1, 13), soft)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


def test_encoder_map_and_token_gradient_vs_oracle():
    """visformer_micro_80(return_map=True) = the distillation phase's encoder contract `return x, pooled`
    (sun_meta_training/models/visformer.py:464): eval map / pooled against the oracle, and in train mode the gradient that reaches the
    parameters through BOTH outputs against torch.autograd of the oracle (parity numerics)."""
    from fewshot_vit_amd import models, synthetic
    from oracle import visformer_oracle as vo
    cfg = vo.VisformerCfg()
    shapes = vo.state_dict_shapes(cfg, prefix='')
    sd = synthetic.synthetic_checkpoint_sd({'encoder.' + k: v for k, v in shapes.items()})
    sd = {k[len('encoder.'):]: v for k, v in sd.items()}
    x = synthetic.synthetic_episodes(5, 1, 5, 1, 1)                             # 10 images [10, 3, 80, 80]
    enc = models.make('visformer_micro_80', numerics='parity', return_map=True)
    enc.load_state_dict(sd, strict=True)
    enc = enc.cuda().eval()
    with torch.no_grad():
        fmap, pooled = enc(x.cuda())
    ref_map, ref_pooled = vo.visformer_forward(sd, x, cfg, return_map=True)
    assert fmap.shape == (10, 512, 5, 5)
    assert (fmap.cpu() - ref_map).abs().max() <= 1e-3 and (pooled.cpu() - ref_pooled).abs().max() <= 1e-4
    assert (fmap.mean(dim=(2, 3)) - pooled).abs().max() <= 1e-4
    # train mode: loss through the map AND the pooled feature
    g = torch.Generator().manual_seed(9)
    wm, wp = torch.randn(10, 512, 5, 5, generator=g), torch.randn(10, 512, generator=g)
    enc.train()
    fmap, pooled = enc(x.cuda())
    ((fmap * wm.cuda()).sum() + (pooled * wp.cuda()).sum()).backward()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))}
    full = {k: v.clone() for k, v in sd.items()}
    full.update(params)
    rm, rp = vo.visformer_forward(full, x, cfg, mode='train', return_map=True)
    ((rm * wm).sum() + (rp * wp).sum()).backward()
    assert (fmap.detach().cpu() - rm.detach()).abs().max() <= 2e-3
    got = dict(enc.named_parameters())
    for k in ('stage3.2.mlp.conv3.weight', 'norm.bn.weight', 'stage2.0.attn.qkv.weight', 'stem.conv1.weight'):
        r = params[k].grad
        assert ((got[k].grad.cpu() - r).norm() / r.norm()).item() <= 5e-3, k


def test_offline_driver_two_epochs(tmp_path):
    """sun_meta_training/offline.py surface end to end on a small schedule: student / teacher `token-label` over visformer_micro_80,
    distillation steps on the HIP trainer, few-shot val episodes, cosine schedule with warm-up stepped with (epoch - 1), checkpoint
    schema readable by models.load."""
    from fewshot_vit_amd import models, offline
    config = dict(train_dataset='synthetic-episodes', train_dataset_args=dict(split='train', n_classes=8, n_per_class=20, noise=1.0, seed=1),
                  val_dataset='synthetic-episodes', val_dataset_args=dict(split='val', n_classes=6, n_per_class=30, noise=1.0, seed=2),
                  model='token-label', model_args=dict(encoder='visformer_micro_80', encoder_args=dict(drop_path_rate=0.1),
                                                       classifier='linear-classifier', classifier_args=dict(n_classes=8)),
                  synthetic_checkpoint='visformer_micro_80', batch_size=16, train_batches=3, eval_batches=2, max_epoch=2,
                  n_way=5, n_shot=1, n_query=3, ep_per_batch=2, tl_soft_k=3, bg_token_num=10, optimizer='adamw',
                  optimizer_args=dict(lr=5e-4, weight_decay=0.05, warmup_lr=1e-6, warmup=1), save_epoch=1)
    lines = []
    trlog = offline.main(config, name='o', device=torch.device('cuda', 0), log=lines.append, save_root=str(tmp_path))
    assert len(trlog['tl']) == 2 and all(np.isfinite(trlog[k]).all() for k in trlog)
    assert any(l.startswith('epoch 2, train') for l in lines)
    ck = torch.load(os.path.join(str(tmp_path), 'o', 'epoch-last.pth'), map_location='cpu')
    assert ck['model'] == 'token-label' and 'classifier_local.linear.weight' in ck['model_sd'] and ck['model_sd']['classifier_local.linear.weight'].shape == (9, 512)
    lr_expected = 0.5 * (5e-4 * 16 / 512) * (1 + np.cos(np.pi * 1 / 2))          # after epoch 2: cosine at t = 1 of t_initial = 2 (warm-up 1 epoch)
    assert ck['training']['optimizer_sd']['param_groups'][0]['lr'] == pytest.approx(lr_expected, rel=1e-6)


def test_train_classifier_driver(tmp_path):
    """sun_meta_training/train_classifier.py surface: `classifier` (encoder + linear head) trained with CE / AdamW / cosine schedule,
    supervised val pass, few-shot episodes through a meta-baseline that shares the encoder, checkpoint usable as offline.py's teacher."""
    from fewshot_vit_amd import models, train_classifier
    config = dict(train_dataset='synthetic-episodes', train_dataset_args=dict(split='train', n_classes=8, n_per_class=20, noise=1.0, seed=1),
                  val_dataset='synthetic-episodes', val_dataset_args=dict(split='train', n_classes=8, n_per_class=4, noise=1.0, seed=1),
                  fs_dataset='synthetic-episodes', fs_dataset_args=dict(split='test', n_classes=6, n_per_class=30, noise=1.0, seed=0),
                  eval_fs_epoch=2, fs_batches=1,
                  model='classifier', model_args=dict(encoder='visformer_micro_80', encoder_args=dict(drop_path_rate=0.1),
                                                      classifier='linear-classifier', classifier_args=dict(n_classes=8)),
                  synthetic_checkpoint='visformer_micro_80', batch_size=16, train_batches=3, max_epoch=2, optimizer='adamw',
                  optimizer_args=dict(lr=5e-4, weight_decay=0.05, warmup_lr=1e-6, warmup=1), save_epoch=1)
    lines = []
    trlog = train_classifier.main(config, name='c', device=torch.device('cuda', 0), log=lines.append, save_root=str(tmp_path))
    assert len(trlog['tl']) == 2 and np.isfinite(trlog['tl']).all() and np.isfinite(trlog['vl']).all()
    assert 0.0 <= trlog['fsa-1'][-1] <= 1.0 and 0.0 <= trlog['fsa-5'][-1] <= 1.0
    assert any('fs 1:' in l for l in lines)
    ck = torch.load(os.path.join(str(tmp_path), 'c', 'epoch-last.pth'), map_location='cpu')
    m = models.load(ck)
    assert ck['model'] == 'classifier' and m.classifier.linear.weight.shape == (8, 512)


def test_wide_heads_tiered_imagenet_class_counts():
    """351 / 352 classes (tieredImageNet pre-training, offline.py classifier_local): Linear backward and the soft-target CE must not be
    limited to 256 / 128 columns."""
    from fewshot_vit_amd.engine import ops
    g = torch.Generator().manual_seed(17)
    M, K = 75, 512
    for N in (351, 352, 600):
        x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / 22.0, torch.randn(N, generator=g)
        dy = torch.randn(M, N, generator=g)
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y_ref = torch.nn.functional.linear(xr, wr, br)
        y_ref.backward(dy)
        y = ops.linear(x.cuda(), w.cuda(), b.cuda())
        dx, dw, db = ops.linear_backward(dy.cuda(), x.cuda(), w.cuda())
        assert (y.cpu() - y_ref.detach()).abs().max() <= 1e-4
        assert (dx.cpu() - xr.grad).abs().max() <= 1e-4 * float(xr.grad.abs().max())
        assert (dw.cpu() - wr.grad).abs().max() <= 1e-4 * float(wr.grad.abs().max())
        assert (db.cpu() - br.grad).abs().max() <= 1e-4 * float(br.grad.abs().max())
    R, Cc = 50, 352
    z = torch.randn(R, Cc, generator=g) * 3.0
    t = torch.rand(R, Cc, generator=g) * (torch.rand(R, Cc, generator=g) < 0.02)
    zr = z.clone().requires_grad_(True)
    loss_ref = torch.sum(-t * torch.nn.functional.log_softmax(zr, dim=-1), dim=-1)
    loss_ref.sum().backward()
    row, dz = ops.soft_target_ce(z.cuda(), t.cuda(), grad_scale=1.0)
    assert (row.cpu() - loss_ref.detach()).abs().max() <= 2e-5 * max(1.0, float(loss_ref.abs().max()))
    assert (dz.cpu() - zr.grad).abs().max() <= 2e-6


def test_train_classifier_loop_matches_oracle_loop(tmp_path):
    """f3: the supervised loop itself (sun_meta_training/train_classifier.py:139-160: model.train(); CE; AdamW step; cosine schedule per epoch)
    replayed in plain fp32 torch - the oracle's train-mode Visformer under autograd, torch.optim.AdamW, the same permutations and batches -
    must give the same per-epoch training loss and the same head weights as the HIP driver in `parity` mode."""
    from fewshot_vit_amd import datasets, models, synthetic, train_classifier
    from fewshot_vit_amd.utils.schedulers import CosineLRScheduler
    from oracle import visformer_oracle as vo
    margs = dict(encoder='visformer_micro_80', encoder_args=dict(drop_path_rate=0.0, numerics='parity'),
                 classifier='linear-classifier', classifier_args=dict(n_classes=8))
    torch.manual_seed(1234)
    m0 = models.make('classifier', **margs)
    esd = synthetic.synthetic_checkpoint_sd({'encoder.' + k: tuple(v.shape) for k, v in m0.encoder.state_dict().items()}, calib='visformer_micro_80')
    m0.encoder.load_state_dict({k[len('encoder.'):]: v for k, v in esd.items()})
    ck_path = os.path.join(str(tmp_path), 'init.pth')
    torch.save({'model': 'classifier', 'model_args': margs, 'model_sd': m0.state_dict()}, ck_path)
    dargs = dict(split='train', n_classes=8, n_per_class=20, noise=1.0, seed=1)
    config = dict(train_dataset='synthetic-episodes', train_dataset_args=dargs, model='classifier', model_args=margs, load=ck_path,
                  batch_size=16, train_batches=2, max_epoch=2, optimizer='adamw', seed=5,
                  optimizer_args=dict(lr=5e-4, weight_decay=0.05, warmup_lr=1e-6, warmup=1))
    trlog = train_classifier.main(config, name='par', device=torch.device('cuda', 0), log=lambda *_: None, save_root=str(tmp_path))
    final = torch.load(os.path.join(str(tmp_path), 'par', 'epoch-last.pth'), map_location='cpu')['model_sd']

    # ---- the same loop on the CPU oracle
    cfg = vo.VisformerCfg()
    sd = {k: v.detach().cpu().clone() for k, v in m0.state_dict().items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))}
    opt = torch.optim.AdamW(list(params.values()), betas=(0.9, 0.999), eps=1e-8, lr=5e-4 * 16 / 512, weight_decay=0.05)
    sched = CosineLRScheduler(opt, warmup_lr_init=1e-6, t_initial=2, cycle_decay=0.1, warmup_t=1)
    ds = datasets.make('synthetic-episodes', **dargs)
    gen = torch.Generator().manual_seed(5)
    ref_tl = []
    for epoch in range(1, 3):
        losses = []
        perm = torch.randperm(len(ds), generator=gen)
        for bi in range(2):
            idx = perm[bi * 16:(bi + 1) * 16]
            data, label = train_classifier._gather(ds, idx, torch.device('cuda', 0))
            stats = {}
            feat = vo.visformer_forward(sd, data.cpu(), cfg, prefix='encoder.', mode='train', stats_out=stats)
            logits = torch.nn.functional.linear(feat, sd['classifier.linear.weight'], sd['classifier.linear.bias'])
            loss = torch.nn.functional.cross_entropy(logits, label.cpu())
            opt.zero_grad()
            loss.backward()
            opt.step()
            with torch.no_grad():
                for k, v in stats.items():
                    sd['encoder.' + k].copy_(v)
            losses.append(float(loss))
        sched.step(epoch - 1)
        ref_tl.append(float(np.mean(losses)))
    print('train_classifier loop: HIP per-epoch loss', trlog['tl'], 'oracle loop', ref_tl)
    assert np.allclose(trlog['tl'], ref_tl, rtol=2e-4, atol=2e-5)
    for k in ('classifier.linear.weight', 'classifier.linear.bias', 'encoder.norm.bn.weight', 'encoder.stage3.2.mlp.conv3.weight'):
        a, b = final[k], sd[k].detach()
        assert float((a - b).abs().max()) <= 2e-3 * max(1e-3, float(b.abs().max())), k
    for k in ('encoder.norm.bn.running_mean', 'encoder.stem.bn1.running_var'):
        assert torch.allclose(final[k], sd[k], rtol=2e-3, atol=1e-4), k


def test_train_classifier_epoch_ex_and_nn_classifier(tmp_path):
    """f3 leftovers: `epoch_ex` = one extra epoch after max_epoch (sun_train_teacher/train_classifier.py:141-148) and the
    `nn-classifier` head (test_phase/models/classifier.py:38-55) - cosine logits against learnable prototypes with a learnable
    temperature - trained through the same driver."""
    from fewshot_vit_amd import train_classifier
    config = dict(train_dataset='synthetic-episodes', train_dataset_args=dict(split='train', n_classes=6, n_per_class=12, noise=1.0, seed=1),
                  model='classifier', model_args=dict(encoder='visformer_micro_80', encoder_args=dict(drop_path_rate=0.0),
                                                      classifier='nn-classifier', classifier_args=dict(n_classes=6)),
                  synthetic_checkpoint='visformer_micro_80', batch_size=12, train_batches=2, max_epoch=2, epoch_ex=True, optimizer='adamw',
                  optimizer_args=dict(lr=5e-4, weight_decay=0.05, warmup_lr=1e-6, warmup=1))
    lines = []
    trlog = train_classifier.main(config, name='nn', device=torch.device('cuda', 0), log=lines.append, save_root=str(tmp_path))
    assert len(trlog['tl']) == 3 and np.isfinite(trlog['tl']).all()              # max_epoch + the extra epoch
    assert any(l.startswith('epoch 3,') for l in lines)
    ck = torch.load(os.path.join(str(tmp_path), 'nn', 'epoch-last.pth'), map_location='cpu')
    assert ck['model_sd']['classifier.proto'].shape == (6, 512) and 'classifier.temp' in ck['model_sd']


def test_nn_classifier_forward_backward_vs_torch():
    from fewshot_vit_amd import models
    g = torch.Generator().manual_seed(4)
    x = torch.randn(37, 96, generator=g)
    for metric in ('cos', 'dot'):
        m = models.make('nn-classifier', in_dim=96, n_classes=11, metric=metric).cuda()
        xg = x.clone().cuda().requires_grad_(True)
        w = torch.randn(37, 11, generator=g)
        (m(xg) * w.cuda()).sum().backward()
        xr = x.clone().requires_grad_(True)
        pr = m.proto.detach().cpu().clone().requires_grad_(True)
        if metric == 'cos':
            tr = m.temp.detach().cpu().clone().requires_grad_(True)
            ref = torch.nn.functional.normalize(xr, dim=-1) @ torch.nn.functional.normalize(pr, dim=-1).t() * tr
        else:
            ref = xr @ pr.t() * m.temp
        (ref * w).sum().backward()
        with torch.no_grad():
            assert (m(x.cuda()).cpu() - ref.detach()).abs().max() <= 1e-5 * max(1.0, float(ref.abs().max()))
        assert (xg.grad.cpu() - xr.grad).abs().max() <= 1e-5 * max(1.0, float(xr.grad.abs().max()))
        assert (m.proto.grad.cpu() - pr.grad).abs().max() <= 1e-5 * max(1.0, float(pr.grad.abs().max()))
        if metric == 'cos':
            assert abs(float(m.temp.grad) - float(tr.grad)) <= 1e-4 * max(1.0, abs(float(tr.grad)))
    with pytest.raises(NotImplementedError):
        models.make('nn-classifier', in_dim=8, n_classes=2, metric='sqr')


def test_distill_step_at_batch_512_is_the_mean_of_four_128_image_steps():
    """BASELINE configs[3] at its own size (sun_meta_training/offline.py:263-309 runs batch 512): with frozen BatchNorm and no DropPath every image is
    independent, so the gradient of the 512-image distillation loss (global CE + 0.5 x SoftTargetCrossEntropy on 512 x 25 x 65 token logits against the
    frozen teacher's soft labels) equals the mean of the four 128-image gradients to fp32 summation order.  Exact-fp32 mode; through the C-ABI trainer,
    fsvit_token_softlabel, fsvit_soft_target_ce and the HIP Linear heads at the shapes the reference trains at."""
    from fewshot_vit_amd import models, synthetic, utils
    from fewshot_vit_amd.models.classifier import SoftTargetCrossEntropy, generate_softlabel
    margs = dict(encoder='visformer_micro_80', encoder_args=dict(drop_path_rate=0.0, return_map=True, numerics='parity'), classifier='linear-classifier',
                 classifier_args=dict(n_classes=64))
    student, teacher = models.make('token-label', **margs).cuda(), models.make('token-label', **margs).cuda()
    for mdl in (student, teacher):
        enc_shapes = {k: tuple(v.shape) for k, v in mdl.encoder.state_dict().items()}
        esd = synthetic.synthetic_checkpoint_sd({'encoder.' + k: s for k, s in enc_shapes.items()}, calib='visformer_micro_80')
        mdl.encoder.load_state_dict({k[len('encoder.'):]: v for k, v in esd.items()})
    student.train()
    utils.freeze_bn(student)
    teacher.eval()
    crit = SoftTargetCrossEntropy()
    g = torch.Generator().manual_seed(77)
    x = torch.randn(512, 3, 80, 80, generator=g).cuda()
    label = torch.randint(0, 64, (512,), generator=g).cuda()

    def grads(xb, lb):
        for p in student.parameters():
            p.grad = None
        logits_token, logits, _ = student(xb)
        with torch.no_grad():
            lt_t, _, _ = teacher(xb, True)
            soft = generate_softlabel(lt_t, k=3, bp=10)
        loss = torch.nn.functional.cross_entropy(logits, lb) + 0.5 * crit(logits_token.permute(0, 2, 3, 1).reshape(-1, 65), soft)
        loss.backward()
        return float(loss), {k: p.grad.detach().clone() for k, p in student.named_parameters() if p.grad is not None}

    loss_full, full = grads(x, label)
    parts = [grads(x[i:i + 128], label[i:i + 128]) for i in range(0, 512, 128)]
    assert np.isfinite(loss_full) and loss_full == pytest.approx(np.mean([p[0] for p in parts]), rel=1e-5)
    worst = 0.0
    for k, v in full.items():
        mean = sum(p[1][k] for p in parts) / 4.0
        worst = max(worst, float((v - mean).norm() / (v.norm() + 1e-20)))
    print(f'[distill 512 = mean of 4 x 128] loss {loss_full:.5f}, worst gradient rel difference {worst:.3e} over {len(full)} tensors')
    assert len(full) >= 85 and worst <= 2e-5          # 85 encoder tensors + the two Linear heads' weights / biases
