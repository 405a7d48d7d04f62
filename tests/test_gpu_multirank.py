"""Every world > 1 code path of the two drivers on the ONE GPU a test box has (ADVICE r03 autograd.py:43):

  * `test_few_shot.evaluate(world=2)` with the real HIP engine - two ranks (gloo; both on cuda:0, RCCL refuses duplicate devices) shard the sampler
    stream rank::world and exchange the per-batch statistics once; acc / CI / va_lst must be bit-equal to the single-process run
    (reference: nn.DataParallel on the episode axis, test_phase/test_few_shot.py:65-66);
  * `train_meta.main(world=2)` - each rank takes its slice of the batch's episode axis, the HIP trainer writes its gradients straight into
    `parallel.GradBucket`'s flat buffer (the `_grad_sink` branch of autograd.VisformerTrainFn), ONE all-reduce, `FsvitSGD` on the bucket
    views; with frozen BatchNorm and no DropPath the parameters after the steps equal the world = 1 run to fp32 summation order
    (meta_tuning_sun_m/train_meta.py:128-129,161-174);
  * single process: gradients with the sink armed are bit-equal to gradients without it, for the Visformer and the ViT trainer, and a
    second backward before the optimizer step ACCUMULATES (the sink must not overwrite).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _train_config(numerics='bf16'):
    return dict(train_dataset='synthetic-episodes', train_dataset_args=dict(split='train', n_classes=12, n_per_class=30, noise=1.0, seed=1),
                tval_dataset='synthetic-episodes', tval_dataset_args=dict(split='test', n_classes=6, n_per_class=30, noise=1.0, seed=0),
                val_dataset='synthetic-episodes', val_dataset_args=dict(split='val', n_classes=6, n_per_class=30, noise=1.0, seed=2),
                model='meta-baseline', model_args=dict(encoder='visformer_micro_80', encoder_args=dict(drop_path_rate=0.0, numerics=numerics)),
                synthetic_checkpoint='visformer_micro_80', n_train_way=5, n_train_shot=5, n_train_query=5, n_way=5, n_shot=5, n_query=15,
                train_batches=3, eval_batches=3, ep_per_batch=4, max_epoch=1, freeze_bn=True, optimizer='sgd',
                optimizer_args=dict(lr=0.01, weight_decay=5e-4), save_epoch=1)


CHILD = r'''
import json, os, sys
repo, rank, port, mode, out = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
sys.path.insert(0, repo)
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=port, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK='0')
import torch, yaml
import torch.distributed as dist
from fewshot_vit_amd import parallel
torch.cuda.set_device(0)
r, world, _ = parallel.init_from_env(backend='gloo')          # two ranks, ONE device: gloo (RCCL refuses duplicate GPUs)
assert (r, world) == (rank, 2)
dev = torch.device('cuda', 0)
if mode == 'eval':
    from fewshot_vit_amd import test_few_shot
    cfg = yaml.safe_load(open(os.path.join(repo, 'few-shot-vit_amd', 'configs', 'test_synthetic.yaml')))
    res = test_few_shot.evaluate(cfg, shot=1, n_batch=9, launch_batches=4, numerics='bf16', rank=rank, world=2, device=dev, log=lambda s: None)
    if rank == 0:
        json.dump({k: res[k] for k in ('acc', 'ci', 'loss', 'n', 'va_lst')}, open(out, 'w'))
else:
    from fewshot_vit_amd import train_meta
    config = json.load(open(out + '.config.json'))
    sinks = []
    orig = parallel.GradBucket.allreduce_mean
    def spy(self):                                  # the gradients must already sit in the bucket when the exchange starts: count foreign .grad tensors
        sinks.append(sum(1 for p, v in zip(self.params, self.views) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()))
        return orig(self)
    parallel.GradBucket.allreduce_mean = spy
    trlog = train_meta.main(config, name='w2', rank=rank, world=2, device=dev, log=lambda s: None, save_root=out + '.save')
    if rank == 0:
        json.dump({'tl': trlog['tl'], 'ta': trlog['ta'], 'va': trlog['va'], 'foreign_grads_per_step': sinks}, open(out, 'w'))
dist.barrier()
dist.destroy_process_group()
print('rank-%d-ok' % rank)
'''


def _run_two_ranks(mode, out):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(k, None)
    port = str(_free_port())
    procs = [subprocess.Popen([sys.executable, '-c', CHILD, REPO, str(r), port, mode, out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=env, cwd=REPO) for r in range(2)]
    for r, p in enumerate(procs):
        try:
            so, se = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0 and 'rank-%d-ok' % r in so, (so[-1000:], se[-3000:])


def test_evaluate_world2_on_one_gpu_equals_single_process(tmp_path):
    from fewshot_vit_amd import test_few_shot
    cfg = yaml.safe_load(open(os.path.join(REPO, 'few-shot-vit_amd', 'configs', 'test_synthetic.yaml')))
    one = test_few_shot.evaluate(cfg, shot=1, n_batch=9, launch_batches=4, numerics='bf16', log=lambda s: None)
    out = str(tmp_path / 'eval.json')
    _run_two_ranks('eval', out)
    two = json.load(open(out))
    assert two['n'] == 9 and two['va_lst'] == one['va_lst']                 # stream order restored by the one all-gather: bit-equal per batch
    assert two['acc'] == one['acc'] and two['ci'] == one['ci'] and two['loss'] == pytest.approx(one['loss'], abs=1e-12)


@pytest.mark.parametrize('numerics,tol', [('parity', 1.5e-3), ('bf16', 5e-2)])
def test_train_meta_world2_on_one_gpu_matches_world1(tmp_path, numerics, tol):
    """3 SUN-M steps of 4 episodes (world 1) vs 2 ranks x 2 episodes: same parameters to fp32 summation order in the exact-fp32 mode (measured 4e-4 of the update on the stem
    weights, whose gradients pass the max-pool / LeakyReLU decisions at rounding-level ties; 1e-5 elsewhere).  In `bf16`
    the first step agrees to summation order too, but a weight that differs in its last fp32 bit can round to the other bf16 neighbour in the
    next step's pack (measured 1.5e-2 of the update after three steps).  Drives GradBucket + the trainer's gradient sink + FsvitSGD's pointer
    table on bucket views through the HIP trainer."""
    from fewshot_vit_amd import train_meta
    config = _train_config(numerics)
    train_meta.main(config, name='w1', device=torch.device('cuda', 0), log=lambda s: None, save_root=str(tmp_path / 'one'))
    out = str(tmp_path / 'train.json')
    json.dump(config, open(out + '.config.json', 'w'))
    _run_two_ranks('train', out)
    two = json.load(open(out))
    # every encoder gradient was written into the bucket by the trainer itself; the head's `temp` arrives through autograd (1 foreign tensor)
    assert len(two['foreign_grads_per_step']) == 3 and max(two['foreign_grads_per_step']) <= 1, two['foreign_grads_per_step']
    a = torch.load(os.path.join(str(tmp_path / 'one'), 'w1', 'epoch-last.pth'), map_location='cpu')
    b = torch.load(os.path.join(out + '.save', 'w2', 'epoch-last.pth'), map_location='cpu')
    from fewshot_vit_amd import models, synthetic
    m0 = models.make(config['model'], **config['model_args'])
    init = synthetic.synthetic_checkpoint_sd({k: tuple(v.shape) for k, v in m0.state_dict().items()}, calib='visformer_micro_80')
    worst, moved, rel = 0.0, 0, {}
    for k, va in a['model_sd'].items():
        vb = b['model_sd'][k]
        if not va.dtype.is_floating_point:
            assert torch.equal(va, vb), k
            continue
        step = (va - init[k]).float().norm().item()
        if k.endswith(('running_mean', 'running_var')):
            assert torch.equal(va, init[k]) and torch.equal(vb, init[k]), k        # freeze_bn
            continue
        assert step > 0, k                                                       # every parameter was updated in both runs
        moved += 1
        rel[k] = (va - vb).float().norm().item() / step
        worst = max(worst, rel[k])
    top = sorted(rel.items(), key=lambda kv: -kv[1])[:3]
    print(f'[world 2 vs world 1, {numerics}] {moved} tensors, worst |dparam| / |update| = {worst:.3e}; top: ' + ', '.join(f'{k} {v:.2e}' for k, v in top))
    assert worst <= tol
    la, lb = torch.load(os.path.join(str(tmp_path / 'one'), 'w1', 'trlog.pth')), two
    assert lb['tl'][0] == pytest.approx(la['tl'][0], rel=1e-3) and lb['va'][0] == pytest.approx(la['va'][0], abs=0.02)
    ma, mb = a['training']['optimizer_sd']['state'], b['training']['optimizer_sd']['state']
    assert len(ma) == len(mb) > 0                                                # same set of momentum buffers: no parameter skipped or added


def _one_backward(model, xs, xq, label, way, seed):
    torch.manual_seed(seed)                                                      # the DropPath draws of this forward
    logits = model(xs, xq).view(-1, way)
    torch.nn.functional.cross_entropy(logits, label).backward()


@pytest.mark.parametrize('encoder,args,img', [('visformer_micro_80', dict(drop_path_rate=0.5), 80), ('deit_nano_patch6_84', dict(drop_path_rate=0.1), 84)])
def test_gradient_sink_equals_plain_autograd_and_accumulates(encoder, args, img):
    from fewshot_vit_amd import models, parallel, synthetic, utils
    from fewshot_vit_amd.utils import few_shot as fs
    way, shot, query, E = 5, 1, 3, 2
    x = synthetic.synthetic_episodes(5, E, way, shot, query, img=img).cuda()
    xs, xq = fs.split_shot_query(x, way, shot, query, E)
    label = fs.make_nk_label(way, query, E).cuda()

    def fresh():
        m = models.make('meta-baseline', encoder=encoder, encoder_args=dict(args, numerics='bf16'))
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict(synthetic.synthetic_checkpoint_sd(shapes, calib='visformer_micro_80' if encoder.startswith('visformer') else None))
        return m.cuda().train()

    plain = fresh()
    _one_backward(plain, xs, xq, label, way, 7)
    ref = {k: p.grad.clone() for k, p in plain.named_parameters()}
    _one_backward(plain, xs, xq, label, way, 7)                                  # second backward, no zero_grad: autograd accumulates
    ref2 = {k: p.grad.clone() for k, p in plain.named_parameters()}

    m = fresh()
    bucket = parallel.GradBucket(m)
    opt = utils.FsvitSGD(m.parameters(), 0.01, momentum=0.9, weight_decay=5e-4)
    opt.zero_grad()                                                              # .grad = None: the state train_step leaves before backward
    _one_backward(m, xs, xq, label, way, 7)
    enc = [k for k, _ in m.named_parameters() if k != 'temp']
    views = {id(p): v for p, v in zip(bucket.params, bucket.views)}
    assert all(p.grad.data_ptr() == views[id(p)].data_ptr() for k, p in m.named_parameters() if k in enc)      # the trainer wrote into the bucket
    for k, p in m.named_parameters():
        assert torch.equal(p.grad, ref[k]), k                                    # bit-equal to the path without a sink
    _one_backward(m, xs, xq, label, way, 7)                                      # .grad is populated: the sink must not overwrite
    for k, p in m.named_parameters():
        assert torch.allclose(p.grad, ref2[k], rtol=1e-6, atol=1e-12), k
        assert p.grad.float().norm() > 1.5 * ref[k].float().norm() or ref[k].float().norm() == 0, k
    bucket.allreduce_mean()                                                      # world 1: collects only
    opt.step()
    # ... and the optimizer step on bucket views equals the step on plain gradients
    opt_p = utils.FsvitSGD(plain.parameters(), 0.01, momentum=0.9, weight_decay=5e-4)
    opt_p.step()
    for (k, p), (_, q) in zip(m.named_parameters(), plain.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-6, atol=1e-9), k
