#!/usr/bin/env python3
"""Supervised pre-training driver with the reference's CLI / YAML surface (sun_meta_training/train_classifier.py:24-286): `classifier`
(encoder + `linear-classifier` on the pooled feature) trained with cross-entropy, AdamW(lr * batch_size / 512) and a cosine schedule
with linear warm-up stepped with (epoch - 1) (:121-123,:195-196); per epoch a supervised pass over `val_dataset` (:165-176) and, every
`eval_fs_epoch` epochs, 5-way 1- and 5-shot episodes through `meta-baseline` sharing the encoder (:109-111,:178-191); checkpoints in the
reference's schema.  The teacher of the distillation phase (`offline.py`, key `load`) is a checkpoint of this driver.

MI355X-native: encoder forward / backward on the HIP trainer, Linear head, AdamW update and the few-shot evaluation on the HIP engine;
multi-GPU = one process per GPU with one gradient all-reduce per step and rank-sharded few-shot episodes.  Not restated: tensorboard,
dataset visualisation, train-time augmentation.

  python -m fewshot_vit_amd.train_classifier --config few-shot-vit_amd/configs/train_classifier_synthetic.yaml
"""
import argparse
import os

import numpy as np
import torch
import torch.nn.functional as F
import yaml

from . import datasets, models, parallel, utils
from .datasets.samplers import CategoriesSampler
from .models.classifier import FsvitAdamW
from .utils import few_shot as fs
from .utils.schedulers import CosineLRScheduler


def _gather(dataset, idx, device):
    items = [dataset[int(i)] for i in idx]
    return torch.stack([it[0] for it in items]).to(device, non_blocking=True), torch.tensor([int(it[-1]) for it in items], device=device)


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
    batch_size = config['batch_size']
    if batch_size % world:
        raise ValueError(f'batch_size={batch_size} must divide over {world} ranks')
    n_local = batch_size // world

    train_dataset = datasets.make(config['train_dataset'], **config['train_dataset_args'])
    val_dataset = datasets.make(config['val_dataset'], **config['val_dataset_args']) if config.get('val_dataset') else None
    fs_dataset, fs_samplers, n_shots = None, [], [1, 5]
    if config.get('fs_dataset'):                                                             # :73-97
        fs_dataset = datasets.make(config['fs_dataset'], **config['fs_dataset_args'])
        ef_epoch = config.get('eval_fs_epoch') or 5
        for n_shot in n_shots:
            fs_samplers.append(CategoriesSampler(fs_dataset.label, config.get('fs_batches', 200), 5, n_shot + 15, ep_per_batch=4, rank=rank,
                                                 world_size=world, shard=parallel.sampler_shard(world)))
    if rank == 0:
        log('train dataset: {} (x{}), {}'.format(tuple(train_dataset[0][0].shape), len(train_dataset), train_dataset.n_classes))

    if config.get('load'):
        model = models.load(torch.load(config['load'], map_location='cpu')).to(device)
    else:
        model = models.make(config['model'], **config['model_args']).to(device)
    if config.get('synthetic_checkpoint'):
        from . import synthetic
        enc_shapes = {'encoder.' + k: tuple(v.shape) for k, v in model.encoder.state_dict().items()}
        esd = synthetic.synthetic_checkpoint_sd(enc_shapes, calib=config['synthetic_checkpoint'])
        model.encoder.load_state_dict({k[len('encoder.'):]: v for k, v in esd.items()})
    fs_model = None
    if fs_dataset is not None:                                                               # :109-111
        fs_model = models.make('meta-baseline', encoder=None)
        fs_model.encoder = model.encoder
    if rank == 0:
        log('num params: {}'.format(utils.compute_n_params(model)))

    oa = config['optimizer_args']
    lr = float(oa['lr']) * (batch_size / 512)                                                # :121
    optimizer = FsvitAdamW(model.parameters(), betas=(0.9, 0.999), eps=1e-8, lr=lr, weight_decay=float(oa['weight_decay']))
    lr_scheduler = CosineLRScheduler(optimizer, warmup_lr_init=float(oa['warmup_lr']), t_initial=config['max_epoch'], cycle_decay=0.1,
                                     warmup_t=int(oa['warmup']))                             # :123 (`decay_rate` is timm's older name of cycle_decay)
    max_epoch, save_epoch = config['max_epoch'], config.get('save_epoch')
    max_va = 0.
    timer_used, timer_epoch = utils.Timer(), utils.Timer()
    keys = ['tl', 'ta', 'vl', 'va'] + (['fsa-' + str(n) for n in n_shots] if fs_dataset is not None else [])
    trlog = {k: [] for k in keys}
    gen = torch.Generator().manual_seed(config.get('seed', 0))

    for epoch in range(1, max_epoch + 1 + 1):
        if epoch == max_epoch + 1:
            # `epoch_ex` (sun_train_teacher/train_classifier.py:141-148): ONE extra epoch over the training set under its default
            # (un-augmented) transform.  The datasets here only have that transform, so the switch itself is a no-op.
            if not config.get('epoch_ex'):
                break
            if hasattr(train_dataset, 'default_transform'):
                train_dataset.transform = train_dataset.default_transform
        timer_epoch.s()
        aves = {k: utils.Averager() for k in keys}
        model.train()
        perm = torch.randperm(len(train_dataset), generator=gen)
        n_batches = config.get('train_batches') or (len(train_dataset) // batch_size)
        for bi in range(n_batches):
            idx = perm[bi * batch_size:(bi + 1) * batch_size][rank * n_local:(rank + 1) * n_local]
            data, label = _gather(train_dataset, idx, device)
            logits = model(data)                                                             # :150-156
            loss = F.cross_entropy(logits, label)
            acc = utils.compute_acc(logits, label)
            optimizer.zero_grad()
            loss.backward()
            parallel.allreduce_mean_grads(list(model.parameters()))
            optimizer.step()
            aves['tl'].add(float(loss))
            aves['ta'].add(acc)

        model.eval()
        if val_dataset is not None:                                                          # :165-176
            n_val = config.get('val_batches') or ((len(val_dataset) + batch_size - 1) // batch_size)
            for bi in range(n_val):
                idx = torch.arange(bi * batch_size, min((bi + 1) * batch_size, len(val_dataset)))
                if len(idx) == 0:
                    break
                data, label = _gather(val_dataset, idx, device)
                with torch.no_grad():
                    logits = model(data)
                aves['vl'].add(float(F.cross_entropy(logits, label)), len(idx))
                aves['va'].add(utils.compute_acc(logits, label), len(idx))
        if fs_model is not None and (epoch % ef_epoch == 0 or epoch == max_epoch):
            fs_model.eval()
            for i, n_shot in enumerate(n_shots):                                             # :178-191
                np.random.seed(0)
                sums = torch.zeros(2, dtype=torch.float64, device=device)
                for idx in fs_samplers[i]:
                    data = torch.stack([fs_dataset[int(j)][0] for j in idx]).to(device)
                    x_shot, x_query = fs.split_shot_query(data, 5, n_shot, 15, ep_per_batch=4)
                    label = fs.make_nk_label(5, 15, ep_per_batch=4).to(device)
                    with torch.no_grad():
                        logits = fs_model(x_shot, x_query).view(-1, 5)
                    sums += torch.stack([(logits.argmax(1) == label).double().mean(), torch.ones((), dtype=torch.float64, device=device)])
                if world > 1:
                    torch.distributed.all_reduce(sums)
                if float(sums[1]) > 0:
                    aves['fsa-' + str(n_shot)].add(float(sums[0] / sums[1]), float(sums[1]))

        lr_scheduler.step(epoch - 1)                                                         # :195-196
        for k, v in aves.items():
            aves[k] = v.item()
            trlog[k].append(aves[k])
        if rank == 0:
            s = 'epoch {}, train {:.4f}|{:.4f}'.format(epoch, aves['tl'], aves['ta'])
            if val_dataset is not None:
                s += ', val {:.4f}|{:.4f}'.format(aves['vl'], aves['va'])
            if fs_model is not None and (epoch % ef_epoch == 0 or epoch == max_epoch):
                s += ', fs' + ''.join(' {}: {:.4f}'.format(n, aves['fsa-' + str(n)]) for n in n_shots)
            log(s + ', {} {}/{}'.format(utils.time_str(timer_epoch.t()), utils.time_str(timer_used.t()), utils.time_str(timer_used.t() / epoch * max_epoch)))
            training = {'epoch': epoch, 'optimizer': config.get('optimizer'), 'optimizer_args': config['optimizer_args'],
                        'optimizer_sd': optimizer.state_dict()}
            save_obj = {'file': __file__, 'config': config, 'model': config['model'], 'model_args': config['model_args'],
                        'model_sd': model.state_dict(), 'training': training}
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
