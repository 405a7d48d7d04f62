#!/usr/bin/env python3
"""Meta-tuning driver with the reference's CLI / YAML surface (meta_tuning_sun_m/train_meta.py:28-270):
episodic training of `meta-baseline` over a Visformer encoder - per batch `ep_per_batch` episodes of
n_train_way x (n_train_shot + n_train_query) images, cross-entropy on the cosine-prototype logits, SGD(momentum 0.9,
weight decay) with MultiStepLR, then `tval` / `val` episodes in eval mode, checkpoints `epoch-last.pth`, `epoch-N.pth`,
`max-va.pth` in the reference's schema (so `models.load` of either code base reads them).

MI355X-native differences:
  * forward, backward and the optimizer update run on the HIP trainer (fsvit_visformer_train_forward / _backward,
    fsvit_proto_head(_backward), fsvit_sgd_step); the eval phases run on the packed eval engine;
  * multi-GPU = one process per GPU (torchrun) instead of nn.DataParallel: every rank replays the same sampler stream
    and keeps its slice of the batch's EPISODE axis (what DataParallel scatters, train_meta.py:131-132), BatchNorm
    statistics stay per replica as they do under DataParallel, and the one exchange per step is an all-reduce (mean) of
    the flattened gradients over RCCL (parallel.allreduce_mean_grads);
  * tensorboard / dataset visualisation are not restated (not on the path).

  python -m fewshot_vit_amd.train_meta --config few-shot-vit_amd/configs/train_meta_synthetic.yaml
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m fewshot_vit_amd.train_meta --config ...
"""
import argparse
import os

import numpy as np
import torch
import torch.nn.functional as F
import yaml

from . import datasets, models, parallel, utils
from .datasets.samplers import CategoriesSampler
from .utils import few_shot as fs


def fix_random_seeds(seed=12345):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)


def _batch(dataset, idx, device):
    if hasattr(dataset, 'gather'):                       # device-resident dataset: gather + transform on the GPU
        return dataset.gather(idx)
    return torch.stack([dataset[int(i)][0] for i in idx]).to(device, non_blocking=True)


def build_model(config):
    """`load` / `load_encoder` handling of train_meta.py:118-127 (+ the offline `synthetic_checkpoint` stand-in)."""
    if config.get('load'):
        return models.load(torch.load(config['load'], map_location='cpu'))
    model = models.make(config['model'], **config['model_args'])
    if config.get('load_encoder'):
        encoder = models.load(torch.load(config['load_encoder'], map_location='cpu')).encoder
        model.encoder.load_state_dict(encoder.state_dict())
    if config.get('synthetic_checkpoint'):
        from . import synthetic
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict(synthetic.synthetic_checkpoint_sd(shapes, calib=config['synthetic_checkpoint']))
    return model


_FUSED_CE = True       # head + cross entropy + accuracy in one launch (False: logits -> F.cross_entropy / compute_acc through ATen, the reference's three lines)


def train_step(model, optimizer, x_shot, x_query, label, n_way, world=1, bucket=None):
    """train_meta.py:161-174: forward, CE, zero_grad, backward, (gradient all-reduce), step.  Returns (loss, acc).
    `bucket`: the model's parallel.GradBucket (world > 1): the gradients already sit in its flat buffer, the exchange is one in-place all-reduce."""
    if _FUSED_CE and hasattr(model, 'forward_loss') and label.is_cuda:
        loss, acc, _ = model.forward_loss(x_shot, x_query, label)       # head + F.cross_entropy + compute_acc in one launch (fsvit_proto_head_ce)
    else:
        logits = model(x_shot, x_query).view(-1, n_way)
        loss = F.cross_entropy(logits, label)
        acc = utils.compute_acc(logits, label, reduction='none').mean()
    optimizer.zero_grad()
    loss.backward()
    if world > 1:
        if bucket is not None:
            bucket.allreduce_mean()
        else:
            parallel.allreduce_mean_grads(model.parameters())
    optimizer.step()
    return loss.item(), acc.item()          # the step's only host synchronisation: after the optimizer launches, not between forward and backward


def main(config, name=None, tag=None, rank=0, world=1, device=None, log=None, save_root='./save', warmup=False):
    log = log or utils.log
    fix_random_seeds(12345)
    svname = name
    if svname is None:
        svname = 'meta_{}-{}shot'.format(config['train_dataset'], config['n_shot'])
        svname += '_' + config['model'] + '-' + config['model_args']['encoder']
    if tag is not None:
        svname += '_' + tag
    save_path = os.path.join(save_root, svname)
    if rank == 0:
        utils.ensure_path(save_path, remove=False)
        utils.set_log_path(save_path)
        yaml.dump(config, open(os.path.join(save_path, 'config.yaml'), 'w'))
    device = device or torch.device('cuda', torch.cuda.current_device())

    n_way, n_shot, n_query = config['n_way'], config['n_shot'], config['n_query']
    n_train_query = config['n_train_query'] if config.get('n_train_query') is not None else n_query
    n_train_way = config['n_train_way'] if config.get('n_train_way') is not None else n_way
    n_train_shot = config['n_train_shot'] if config.get('n_train_shot') is not None else n_shot
    ep_per_batch = config['ep_per_batch'] if config.get('ep_per_batch') is not None else 1
    if ep_per_batch % world:
        raise ValueError(f'ep_per_batch={ep_per_batch} must divide over {world} ranks (episode axis is what is sharded)')
    ep_local = ep_per_batch // world

    train_dataset = datasets.make(config['train_dataset'], **config['train_dataset_args'])
    train_sampler = CategoriesSampler(train_dataset.label, config['train_batches'], n_train_way, n_train_shot + n_train_query,
                                      ep_per_batch=ep_per_batch)
    evals = []
    for nm, key in (('tval', 'tval_dataset'), ('val', 'val_dataset')):
        if config.get(key):
            ds = datasets.make(config[key], **config[key + '_args'])
            evals.append((nm, ds, CategoriesSampler(ds.label, config.get('eval_batches', 500 if warmup else 200), n_way, n_shot + n_query, ep_per_batch=4,
                                                    rank=rank, world_size=world, shard=parallel.sampler_shard(world))))
    if rank == 0:
        log('train dataset: {} (x{}), {}'.format(tuple(train_dataset[0][0].shape), len(train_dataset), train_dataset.n_classes))

    model = build_model(config).to(device)
    if rank == 0:
        log('num params: {}'.format(utils.compute_n_params(model)))
    if warmup:    # train_meta_warmup.py:140-141: SGD(momentum 0.9) + epoch-indexed multi-step schedule with a 3-epoch linear warm-up
        from .utils.schedulers import MultiStepLRScheduler
        oa = config['optimizer_args']
        optimizer = utils.FsvitSGD(model.parameters(), float(oa['lr']), momentum=0.9, weight_decay=float(oa['weight_decay']))
        lr_scheduler = MultiStepLRScheduler(optimizer, oa['milestones'], decay_rate=0.5, warmup_lr_init=1e-5, warmup_t=3)
    else:
        optimizer, lr_scheduler = utils.make_optimizer(model.parameters(), config['optimizer'], **config['optimizer_args'])

    bucket = parallel.GradBucket(model) if world > 1 else None      # gradients are written straight into the all-reduce buffer
    max_epoch, save_epoch = config['max_epoch'], config.get('save_epoch')
    max_va = 0.
    timer_used, timer_epoch = utils.Timer(), utils.Timer()
    aves_keys = ['tl', 'ta', 'tvl', 'tva', 'vl', 'va']
    trlog = {k: [] for k in aves_keys}

    for epoch in range(1, max_epoch + 1):
        timer_epoch.s()
        aves = {k: utils.Averager() for k in aves_keys}
        model.train()
        if config.get('freeze_bn'):          # train_meta.py:156-157
            utils.freeze_bn(model)
        np.random.seed(epoch)
        for idx in train_sampler:                               # every rank replays the same stream ...
            idx = parallel.shard_episode_axis(idx, ep_per_batch, rank, world)                            # ... and keeps its episodes
            data = _batch(train_dataset, idx, device)
            x_shot, x_query = fs.split_shot_query(data, n_train_way, n_train_shot, n_train_query, ep_per_batch=ep_local)
            label = fs.make_nk_label(n_train_way, n_train_query, ep_per_batch=ep_local).to(device)
            loss, acc = train_step(model, optimizer, x_shot, x_query, label, n_train_way, world, bucket)
            aves['tl'].add(loss)
            aves['ta'].add(acc)

        model.eval()
        _sig = -1
        for nm, ds, sampler in evals:
            name_l, name_a = ('tvl', 'tva') if nm == 'tval' else ('vl', 'va')
            np.random.seed(0)
            sums = torch.zeros(3, dtype=torch.float64, device=device)     # sum loss, sum acc, n batches
            for idx in sampler:
                data = _batch(ds, idx, device)
                x_shot, x_query = fs.split_shot_query(data, n_way, n_shot, n_query, ep_per_batch=4)
                label = fs.make_nk_label(n_way, n_query, ep_per_batch=4).to(device)
                with torch.no_grad():
                    logits = model(x_shot, x_query).view(-1, n_way)
                    sums += torch.stack([F.cross_entropy(logits, label).double(), (logits.argmax(1) == label).double().mean(),
                                         torch.ones((), dtype=torch.float64, device=device)])
                _sig = int(ds.label[int(idx[-1])])
            if world > 1:
                torch.distributed.all_reduce(sums)
            if float(sums[2]) > 0:
                aves[name_l].add(float(sums[0] / sums[2]), float(sums[2]))
                aves[name_a].add(float(sums[1] / sums[2]), float(sums[2]))

        if world > 1:                                           # train averages: mean over ranks (equal shard sizes)
            t = torch.tensor([aves['tl'].item(), aves['ta'].item()], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t)
            t /= world
            aves['tl'], aves['ta'] = utils.Averager(), utils.Averager()
            aves['tl'].add(float(t[0]))
            aves['ta'].add(float(t[1]))
        if lr_scheduler is not None:
            if warmup:
                lr_scheduler.step(epoch - 1)                    # train_meta_warmup.py:217
            else:
                lr_scheduler.step()
        for k, v in aves.items():
            aves[k] = v.item()
            trlog[k].append(aves[k])
        if rank == 0:
            t_epoch = utils.time_str(timer_epoch.t())
            t_used = utils.time_str(timer_used.t())
            t_estimate = utils.time_str(timer_used.t() / epoch * max_epoch)
            log('epoch {}, train {:.4f}|{:.4f}, tval {:.4f}|{:.4f}, val {:.4f}|{:.4f}, {} {}/{} (@{})'.format(
                epoch, aves['tl'], aves['ta'], aves['tvl'], aves['tva'], aves['vl'], aves['va'], t_epoch, t_used, t_estimate, _sig))
            training = {'epoch': epoch, 'optimizer': config['optimizer'], 'optimizer_args': config['optimizer_args'],
                        'optimizer_sd': optimizer.state_dict()}
            save_obj = {'file': __file__, 'config': config, 'model': config['model'], 'model_args': config['model_args'],
                        'model_sd': model.state_dict(), 'training': training}
            torch.save(save_obj, os.path.join(save_path, 'epoch-last.pth'))
            torch.save(trlog, os.path.join(save_path, 'trlog.pth'))
            if (save_epoch is not None) and epoch % save_epoch == 0:
                torch.save(save_obj, os.path.join(save_path, 'epoch-{}.pth'.format(epoch)))
            if aves['va'] > max_va:
                max_va = aves['va']
                torch.save(save_obj, os.path.join(save_path, 'max-va.pth'))
    return trlog


def cli(warmup=False):
    parser = argparse.ArgumentParser()
    parser.add_argument('--config')
    parser.add_argument('--name', default=None)
    parser.add_argument('--tag', default=None)
    parser.add_argument('--gpu', default=None, help='kept for CLI compatibility; use torchrun for multi-GPU')
    parser.add_argument('--save-root', default='./save')
    args = parser.parse_args()
    config = yaml.load(open(args.config, 'r'), Loader=yaml.FullLoader)
    if args.gpu is not None and ',' not in args.gpu:
        utils.set_gpu(args.gpu)
    rank, world, local = parallel.init_from_env()
    torch.cuda.set_device(local)
    main(config, args.name, args.tag, rank, world, torch.device('cuda', local), save_root=args.save_root, warmup=warmup)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    cli()
