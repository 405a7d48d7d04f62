#!/usr/bin/env python3
"""SUN meta-training (teacher -> student token-label distillation) driver with the reference's CLI / YAML surface
(sun_meta_training/offline.py:78-461): supervised training of `token-label` over a Visformer encoder - global cross-entropy on the
pooled feature + `token_label_weight`-free `0.5 *` SoftTargetCrossEntropy on the student's 25 token logits against soft labels generated
from the frozen teacher's token logits (top-k scatter + background tokens), AdamW(lr * batch_size / 512) with a cosine schedule and
linear warm-up stepped with (epoch - 1), few-shot `val` episodes scored by cosine prototypes on the pooled feature, checkpoints in the
reference's schema (`epoch-last.pth`, `epoch-N.pth`, `max-va.pth`).

MI355X-native: the encoder's train-mode forward / backward (both outputs of `return x, pooled`), the Linear heads, generate_softlabel,
the soft-target loss + gradient and the AdamW update are HIP kernels behind the C-ABI; cross-entropy on [B, 64] logits stays in the
caller as in the reference.  Multi-GPU = one process per GPU (torchrun): every rank takes its slice of each batch, BatchNorm
statistics stay per replica (as under the reference's nn.DataParallel, :222-225) and the one exchange per step is the all-reduce (mean) of
the flattened gradients over RCCL.  Not restated: tensorboard, dataset visualisation, the strong / weak augmentation pair (the teacher
sees the same image as the student unless the dataset returns three items), `epoch_ex`.

  python -m fewshot_vit_amd.offline --config few-shot-vit_amd/configs/offline_synthetic.yaml
"""
import argparse
import os

import numpy as np
import torch
import torch.nn.functional as F
import yaml

from . import datasets, models, parallel, utils
from .datasets.samplers import CategoriesSampler
from .models.classifier import FsvitAdamW, SoftTargetCrossEntropy, generate_softlabel
from .utils import few_shot as fs
from .utils.schedulers import CosineLRScheduler


def _gather(dataset, idx, device):
    items = [dataset[int(i)] for i in idx]
    data = torch.stack([it[0] for it in items]).to(device, non_blocking=True)
    weak = torch.stack([it[1] for it in items]).to(device, non_blocking=True) if len(items[0]) == 3 else data
    label = torch.tensor([int(it[-1]) for it in items], device=device)
    return data, weak, label


def few_shot_eval(model, dataset, sampler, n_way, n_shot, n_query, ep_per_batch, device):
    """offline.py:311-336: prototypes = mean of the support images' pooled features, cosine logits * 10."""
    out = model.encoder.out_dim
    la, aa = utils.Averager(), utils.Averager()
    for idx in sampler:
        data = torch.stack([dataset[int(i)][0] for i in idx]).to(device)
        x_shot, x_query = fs.split_shot_query(data, n_way, n_shot, n_query, ep_per_batch=ep_per_batch)
        label = fs.make_nk_label(n_way, n_query, ep_per_batch=ep_per_batch).to(device)
        with torch.no_grad():
            b, n, k = x_shot.shape[:3]
            _, _, q_tok = model(x_query.reshape(-1, *x_query.shape[-3:]))
            _, _, s_tok = model(x_shot.reshape(-1, *x_shot.shape[-3:]))
            q_tok = q_tok.view(b, -1, out)
            s_tok = s_tok.view(b, n, k, out).mean(dim=2)
            logits = utils.compute_logits(q_tok, s_tok, metric='cos', temp=10.0).view(-1, n_way)
            la.add(float(F.cross_entropy(logits, label)))
            aa.add(utils.compute_acc(logits, label))
    return la.item(), aa.item()


def distill_step(model, teacher, optimizer, criterion_tl, data, weak_data, label, tl_soft_k=3, bp=10, world=1):
    """One SUN meta-training step (offline.py:283-303): student forward (token logits with the extra background class, global logits), global
    cross-entropy, frozen teacher forward on the weak view -> generate_softlabel, 0.5 x SoftTargetCrossEntropy on the student's token logits,
    backward, (gradient all-reduce), AdamW.  Returns (loss, acc) as device tensors / float - the caller decides when to synchronise."""
    logits_token, logits, _ = model(data)                                            # :283
    cls_loss = F.cross_entropy(logits, label)
    acc = (logits.argmax(dim=1) == label).float().mean()                             # (utils.compute_acc without its host synchronisation)
    with torch.no_grad():                                                            # :296-298
        logits_token_t, _, _ = teacher(weak_data, True)
        soft_label = generate_softlabel(logits_token_t, k=tl_soft_k, bp=bp)
    c = logits_token_t.shape[1]
    logits_flatten = logits_token.permute(0, 2, 3, 1).reshape(-1, c + 1)            # :293
    token_loss = criterion_tl(logits_flatten, soft_label)
    loss = cls_loss + 0.5 * token_loss                                               # :300
    optimizer.zero_grad()
    loss.backward()
    if world > 1:
        parallel.allreduce_mean_grads(list(model.parameters()))
    optimizer.step()
    return loss.detach(), acc


def main(config, name=None, tag=None, rank=0, world=1, device=None, log=None, save_root='./save'):
    device = device or torch.device('cuda', 0)
    svname = name or 'classifier_{}_{}'.format(config['train_dataset'], config['model_args']['encoder'])
    if tag is not None:
        svname += '_' + tag
    save_path = os.path.join(save_root, svname)
    if rank == 0:
        utils.ensure_path(save_path, remove=False)
        utils.set_log_path(save_path)
        yaml.dump(config, open(os.path.join(save_path, 'config.yaml'), 'w'))
    log = log or utils.log
    bp = 10 if config.get('bg_token_num') is None else int(config['bg_token_num'])          # :96
    n_way, n_shot, n_query = config['n_way'], config['n_shot'], config['n_query']
    ep_per_batch = config.get('ep_per_batch') or 1
    batch_size = config['batch_size']
    if batch_size % world:
        raise ValueError(f'batch_size={batch_size} must divide over {world} ranks')

    train_dataset = datasets.make(config['train_dataset'], **config['train_dataset_args'])
    val_dataset = datasets.make(config['val_dataset'], **config['val_dataset_args'])
    val_sampler = CategoriesSampler(val_dataset.label, config.get('eval_batches', 200), n_way, n_shot + n_query, ep_per_batch=ep_per_batch)
    if rank == 0:
        log('train dataset: {} (x{}), {}'.format(tuple(train_dataset[0][0].shape), len(train_dataset), train_dataset.n_classes))

    def build():
        margs = dict(config['model_args'])
        margs['encoder_args'] = dict(margs.get('encoder_args') or {}, return_map=True)
        return models.make(config['model'], **margs).to(device)

    model, teacher = build(), build()                                                       # :212-219
    if config.get('load'):
        sv = torch.load(config['load'], map_location='cpu')
        teacher.load_state_dict(sv['model_sd'])
    if config.get('synthetic_checkpoint'):                                                   # offline stand-in for a published teacher
        from . import synthetic
        for mdl, salt in ((model, 'student.'), (teacher, 'teacher.')):
            enc_shapes = {k: tuple(v.shape) for k, v in mdl.encoder.state_dict().items()}
            esd = synthetic.synthetic_checkpoint_sd({'encoder.' + k: s for k, s in enc_shapes.items()}, calib=config['synthetic_checkpoint'])
            mdl.encoder.load_state_dict({k[len('encoder.'):]: v for k, v in esd.items()})
    teacher.eval()
    if rank == 0:
        log('num params: {}'.format(utils.compute_n_params(model)))

    oa = config['optimizer_args']
    lr = float(oa['lr']) * (batch_size / 512)                                                # :232
    optimizer = FsvitAdamW(model.parameters(), betas=(0.9, 0.999), eps=1e-8, lr=lr, weight_decay=float(oa['weight_decay']))
    lr_scheduler = CosineLRScheduler(optimizer, warmup_lr_init=float(oa['warmup_lr']), t_initial=config['max_epoch'], cycle_decay=0.1,
                                     warmup_t=int(oa['warmup']))
    max_epoch, save_epoch = config['max_epoch'], config.get('save_epoch')
    tl_soft_k = config['tl_soft_k'] if config.get('tl_soft_k') is not None else 3
    criterion_tl = SoftTargetCrossEntropy()
    max_va = 0.
    timer_used, timer_epoch = utils.Timer(), utils.Timer()
    trlog = {k: [] for k in ('tl', 'ta', 'vl', 'va')}
    n_local = batch_size // world
    gen = torch.Generator().manual_seed(config.get('seed', 0))

    for epoch in range(1, max_epoch + 1):
        timer_epoch.s()
        aves = {k: utils.Averager() for k in trlog}
        model.train()
        perm = torch.randperm(len(train_dataset), generator=gen)                            # DataLoader(shuffle=True): same stream on every rank
        n_batches = config.get('train_batches') or (len(train_dataset) // batch_size)
        for bi in range(n_batches):
            idx = perm[bi * batch_size:(bi + 1) * batch_size][rank * n_local:(rank + 1) * n_local]
            data, weak_data, label = _gather(train_dataset, idx, device)
            loss, acc = distill_step(model, teacher, optimizer, criterion_tl, data, weak_data, label, tl_soft_k, bp, world)
            aves['tl'].add(float(loss))
            aves['ta'].add(float(acc))

        model.eval()
        np.random.seed(0)
        vl, va = few_shot_eval(model, val_dataset, val_sampler, n_way, n_shot, n_query, ep_per_batch, device)
        aves['vl'].add(vl)
        aves['va'].add(va)
        lr_scheduler.step(epoch - 1)                                                         # :373
        for k, v in aves.items():
            aves[k] = v.item()
            trlog[k].append(aves[k])
        if rank == 0:
            log('epoch {}, train {:.4f}|{:.4f}, val {:.4f}|{:.4f}, {} {}/{}'.format(
                epoch, aves['tl'], aves['ta'], aves['vl'], aves['va'], utils.time_str(timer_epoch.t()), utils.time_str(timer_used.t()),
                utils.time_str(timer_used.t() / epoch * max_epoch)))
            training = {'epoch': epoch, 'optimizer': config.get('optimizer'), 'optimizer_args': config['optimizer_args'],
                        'optimizer_sd': optimizer.state_dict()}
            margs = dict(config['model_args'])
            save_obj = {'file': __file__, 'config': config, 'model': config['model'], 'model_args': margs, 'model_sd': model.state_dict(),
                        'training': training}
            torch.save(save_obj, os.path.join(save_path, 'epoch-last.pth'))
            if (save_epoch is not None) and epoch % save_epoch == 0:
                torch.save(save_obj, os.path.join(save_path, 'epoch-{}.pth'.format(epoch)))
            if aves['va'] > max_va:
                max_va = aves['va']
                torch.save(save_obj, os.path.join(save_path, 'max-va.pth'))
    return trlog


def cli():
    parser = argparse.ArgumentParser()
    parser.add_argument('--config')
    parser.add_argument('--name', default=None)
    parser.add_argument('--tag', default=None)
    parser.add_argument('--gpu', default=None, help='kept for CLI compatibility; use torchrun for multi-GPU')
    parser.add_argument('--save-root', default='./save')
    args = parser.parse_args()
    config = yaml.load(open(args.config, 'r'), Loader=yaml.FullLoader)
    rank, world, local = parallel.init_from_env()
    torch.cuda.set_device(local)
    main(config, args.name, args.tag, rank, world, torch.device('cuda', local), save_root=args.save_root)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    cli()
