#!/usr/bin/env python3
"""Benchmark of the fsvit hot path: BASELINE.json configs[1] — Visformer-S (`visformer_micro_80`)
5-way 5-shot episodic eval (15 queries/class), bf16 MFMA, synthetic 80x80 episodes resident in HBM.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" = one pass of `MetaBaseline.forward` (encoder + cosine head, one C-ABI call) over a batch of
`--episodes` episodes per GPU.  Episodes are independent, so ranks shard them with no data-path
collective (weak scaling); the only exchange is ONE all-reduce of the accuracy statistics at the end
of the timed region (RCCL over xGMI), as in the north star.  Rank 0 prints one JSON line.

Extra legs (rank 0, N=1 only unless disabled):
  roofline      per-launch HIP-event timing inside the timed region (engine profile mode) -> the
                dominant kernel's algorithmic TFLOP/s against the 2.5 PFLOP/s dense bf16 MFMA peak.
  cpu_baseline  the oracle (oracle/visformer_oracle.py, a port of the reference's CPU path) timed on
                the host cores over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic forward FLOPs per image (2*MAC, SURVEY.md 8d / BASELINE.md 2) and input size per encoder
MODELS = {'visformer_micro_80': (2.0306e9, 80), 'deit_small_patch16_224': (9.197e9, 224), 'deit_micro_patch6_84': (4.716e9, 84)}
HEAD_FLOP_PER_EPISODE = 0.38e6
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}    # dense peaks, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--episodes', type=int, default=64, help='episodes per GPU per step (ep_per_batch)')
    ap.add_argument('--shot', type=int, default=5)
    ap.add_argument('--model', default='visformer_micro_80', choices=sorted(MODELS), help='encoder (default = BASELINE configs[1])')
    ap.add_argument('--numerics', default='bf16', choices=['bf16', 'parity'])
    ap.add_argument('--chunk', type=int, default=int(os.environ.get('FSVIT_CHUNK', 6400)), help='images per encoder chunk')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--cpu-episodes', type=int, default=16)
    ap.add_argument('--layers', action='store_true', help='print the per-layer timing table to stderr')
    ap.add_argument('--mode', default='eval', choices=['eval', 'train'],
                    help="eval = BASELINE configs[1] (the headline metric); train = configs[2], one SUN-M meta-tuning step "
                         "(train_meta_mini_visformer_5shot.yaml geometry: 8 episodes x 10-way (5 shot + 5 query) = 800 images)")
    ap.add_argument('--train-episodes', type=int, default=8, help='train mode: episodes per GPU per step (ep_per_batch)')
    return ap.parse_args()


def device_episodes(seed, n_ep, way, shot, query, dev, img=80):
    """Class-structured synthetic episodes generated directly in HBM (x = mu_c + 1.5*eps, so accuracy
    is neither chance nor saturated); layout = what fs.split_shot_query returns."""
    g = torch.Generator(device=dev).manual_seed(seed)
    per = shot + query
    mu = torch.randn(n_ep, way, 1, 3, img, img, device=dev, generator=g)
    x = mu + 1.5 * torch.randn(n_ep, way, per, 3, img, img, device=dev, generator=g)
    x_shot = x[:, :, :shot].contiguous()
    x_query = x[:, :, shot:].contiguous().view(n_ep, way * query, 3, img, img)
    return x_shot, x_query


def cpu_baseline(sd, shot, n_ep, model='visformer_micro_80'):
    """Oracle timed at ep_per_batch=1 (the reference's test setting, test_few_shot.py:47-48)."""
    from fewshot_vit_amd import synthetic
    from fewshot_vit_amd.utils import few_shot as fs
    from oracle import visformer_oracle as vo
    cores = torch.get_num_threads()
    img = MODELS[model][1]
    x = synthetic.synthetic_episodes(12345, 1, 5, shot, 15, img=img)
    xs, xq = fs.split_shot_query(x, 5, shot, 15, 1)
    if model == 'visformer_micro_80':
        cfg = vo.VisformerCfg()
        run = lambda: vo.meta_baseline_forward(sd, xs, xq, cfg)
    else:
        from oracle import deit_oracle as do
        cfg = do.FACTORIES[model]

        def run():
            with torch.no_grad():
                f = do.deit_forward(sd, torch.cat([xs.reshape(-1, 3, img, img), xq.reshape(-1, 3, img, img)]), cfg, prefix='encoder.')
            n = xs.shape[1] * xs.shape[2]
            return vo.meta_baseline_head(f[:n].reshape(1, 5, shot, -1), f[n:].reshape(1, 75, -1), temp=10.0)
    for _ in range(2):
        run()
    t0 = time.perf_counter()
    for _ in range(n_ep):
        run()
    dt = time.perf_counter() - t0
    return {'value': n_ep / dt, 'unit': 'episodes/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n_ep} episodes 5-way {shot}-shot (100 images each at 5-shot), ep_per_batch=1, fp32 torch CPU '
                      f'oracle, 2 warm-up episodes, {dt:.1f} s'}


def cpu_train_baseline(sd, n_ep=1):
    """Oracle training step (train-mode forward + torch.autograd backward) on the host cores: 1 episode of the step's 8."""
    from fewshot_vit_amd import synthetic
    from fewshot_vit_amd.utils import few_shot as fs
    from oracle import visformer_oracle as vo
    cfg = vo.VisformerCfg()
    x = synthetic.synthetic_episodes(12345, n_ep, 10, 5, 5)
    xs, xq = fs.split_shot_query(x, 10, 5, 5, n_ep)
    label = fs.make_nk_label(10, 5, n_ep)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))}
    full = {k: v.clone() for k, v in sd.items()}
    full.update(params)

    def run():
        for p in params.values():
            p.grad = None
        logits = vo.meta_baseline_forward(full, xs, xq, cfg, mode='train', drop_path_rate=0.5).view(-1, 10)
        torch.nn.functional.cross_entropy(logits, label).backward()
    run()
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        run()
    dt = time.perf_counter() - t0
    return {'value': reps * n_ep / dt, 'unit': 'episodes/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{reps} x forward+backward of {n_ep} episode(s) 10-way 5-shot 5-query (100 images each), fp32 torch CPU oracle '
                      f'with autograd, no optimizer step, 1 warm-up, {dt:.1f} s'}


def train_main(args, rank, world, dev):
    """BASELINE configs[2]: one meta-tuning step = model.train() forward, CE, backward, (grad all-reduce), SGD step."""
    from fewshot_vit_amd import models, synthetic, utils, parallel
    from fewshot_vit_amd.utils import few_shot as fs
    model = models.make('meta-baseline', encoder='visformer_micro_80', encoder_args={'numerics': args.numerics, 'drop_path_rate': 0.5})
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synthetic.synthetic_checkpoint_sd(shapes, calib='visformer_micro_80')
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    opt, _ = utils.make_optimizer(model.parameters(), 'sgd', lr=0.001, weight_decay=5e-4)
    way, shot, query, E = 10, 5, 5, args.train_episodes
    x_shot, x_query = device_episodes(999 + rank, E, way, shot, query, dev, 80)
    label = fs.make_nk_label(way, query, E).to(dev)

    def step():
        logits = model(x_shot, x_query).view(-1, way)
        loss = torch.nn.functional.cross_entropy(logits, label)
        opt.zero_grad()
        loss.backward()
        if world > 1:
            parallel.allreduce_mean_grads(model.parameters())
        opt.step()
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        imgs = way * (shot + query)
        eps = world * E * args.steps / elapsed
        flops_ep = 3.0 * MODELS['visformer_micro_80'][0] * imgs          # forward + dgrad + wgrad
        peak = MFMA_PEAK_TFLOPS['bf16' if args.numerics == 'bf16' else 'f32']
        out = {'metric': 'train_episodes_per_sec_10way_5shot_visformer_s', 'value': eps, 'unit': 'episodes/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16' if args.numerics == 'bf16' else 'f32', 'data': 'synthetic',
               'config': {'workload': 'BASELINE configs[2]: SUN-M train_meta.py meta-tuning step, Visformer-S (visformer_micro_80, '
                                      'drop_path 0.5), ep_per_batch episodes of 10-way 5-shot 5-query 80x80 (train_meta_mini_visformer_5shot.yaml), '
                                      'forward + CE + backward + SGD(0.9, wd 5e-4), episodes resident in HBM',
                          'episodes_per_step_per_gpu': E, 'images_per_episode': imgs,
                          'parallelism': 'episode axis sharded x%d, one all-reduce of the flattened gradients per step' % world},
               'whole_path_tflops': eps * flops_ep / 1e12, 'whole_path_mfma_frac': eps * flops_ep / 1e12 / peak,
               'final_loss': float(loss)}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_train_baseline(sd)
        print(json.dumps(out))


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)')
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit('bench.py needs an MI355X: the fsvit hot path has no CPU fallback')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    if args.mode == 'train':
        train_main(args, rank, world, dev)
        if world > 1:
            dist.destroy_process_group()
        return

    from fewshot_vit_amd import models, synthetic
    os.environ['FSVIT_CHUNK'] = str(args.chunk)
    flop_per_image, img = MODELS[args.model]
    model = models.make('meta-baseline', encoder=args.model, encoder_args={'numerics': args.numerics})
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    # procedural weights by key name (+ the shipped BN calibration for the Visformer)
    sd = synthetic.synthetic_checkpoint_sd(shapes, calib=args.model if args.model == 'visformer_micro_80' else None)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    engine = model.encoder.engine()

    way, query, E = 5, 15, args.episodes
    x_shot, x_query = device_episodes(12345 + rank, E, way, args.shot, query, dev, img)
    temp = float(model.temp.detach())

    def step():
        return engine.meta_baseline_forward(x_shot, x_query, temp, 'cos', want_stats=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    profile = (not args.no_roofline) and rank == 0
    accs = []
    barrier()
    if profile:
        engine.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, acc, _ = step()
        accs.append(acc)
    # the one exchange of the path: all-reduce of (sum acc, sum acc^2, n) -> mean accuracy +- CI
    acc_all = torch.stack(accs).double().flatten()
    stats = torch.stack([acc_all.sum(), (acc_all * acc_all).sum(), torch.tensor(float(acc_all.numel()), device=dev, dtype=torch.float64)])
    if world > 1:
        dist.all_reduce(stats)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    recs = engine.profile_end() if profile else None

    n = float(stats[2].item())
    mean = float(stats[0].item()) / n
    var = max(0.0, (float(stats[1].item()) - n * mean * mean) / max(1.0, n - 1.0))
    import scipy.stats
    ci = (var / n) ** 0.5 * float(scipy.stats.t.ppf(0.975, max(1.0, n - 1.0)))

    if rank == 0:
        total_eps = world * E * args.steps
        eps = total_eps / elapsed
        imgs = way * (args.shot + query)
        flops_ep = flop_per_image * imgs + HEAD_FLOP_PER_EPISODE
        out = {
            'metric': 'episodes_per_sec_5way_%dshot_%s' % (args.shot, 'visformer_s' if args.model == 'visformer_micro_80' else args.model), 'value': eps, 'unit': 'episodes/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if args.numerics == 'bf16' else 'f32', 'data': 'synthetic',
            'config': {'workload': ('BASELINE configs[1]: Visformer-S (visformer_micro_80) miniImageNet-shaped 5-way %d-shot '
                                    'episodic eval, 15 query/class, 80x80 fp32 NCHW episodes resident in HBM, procedural weights + '
                                    'calibrated BN' % args.shot) if args.model == 'visformer_micro_80' else
                                   ('BASELINE configs[4] shape: %s, %dx%d, 5-way %d-shot episodic eval, 15 query/class, synthetic episodes '
                                    'resident in HBM, procedural weights' % (args.model, img, img, args.shot)),
                       'episodes_per_step_per_gpu': E, 'images_per_episode': imgs, 'encoder_chunk_images': args.chunk,
                       'parallelism': 'episode-parallel x%d, one all-reduce of accuracy stats' % world},
            'whole_path_tflops': eps * flops_ep / 1e12,
            'whole_path_mfma_frac': eps * flops_ep / 1e12 / MFMA_PEAK_TFLOPS['bf16' if args.numerics == 'bf16' else 'f32'],
            'accuracy': {'mean': mean, 'ci95': ci, 'episodes': int(n)},
        }
        if recs:
            bykern = {}
            for r in recs:
                k = bykern.setdefault(r['kernel'], {'ms': 0.0, 'flops': 0.0, 'launches': 0})
                k['ms'] += r['ms']
                k['flops'] += r['flops']
                k['launches'] += r['launches']
            dom = max(bykern, key=lambda k: bykern[k]['ms'])
            d = bykern[dom]
            peak = MFMA_PEAK_TFLOPS['bf16' if args.numerics == 'bf16' else 'f32']
            achieved = d['flops'] / (d['ms'] * 1e-3) / 1e12
            traffic = None          # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/)
            import glob
            import re
            tfs = sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*_hbm_traffic.json')),      # newest committed PMC passes (r01_v10 after r01_v9)
                         key=lambda p: [int(x) if x.isdigit() else x for x in re.split(r'(\d+)', os.path.basename(p))])
            if tfs and args.model == 'visformer_micro_80' and args.numerics == 'bf16' and E == 64 and args.chunk == 6400 and args.mode == 'eval':
                with open(tfs[-1]) as f:
                    tj = json.load(f)
                cand = [k for k in tj if k == dom] or [k for k in tj if k.split('<')[0] == dom.split('<')[0]]
                if cand:       # several template instantiations behind one kernel name (conv2 / conv3 of the stem): launch-weighted mean
                    n = sum(tj[k].get('launches', 1) for k in cand)
                    traffic = sum(tj[k].get('launches', 1) * tj[k].get('hbm_bytes_per_launch', 0.0) for k in cand) / n
            out['roofline'] = {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                               'traffic': traffic, 'kernel': dom, 'launches': d['launches'],
                               'avg_launch_us': 1e3 * d['ms'] / d['launches'],
                               'avg_launch_gflop': d['flops'] / d['launches'] / 1e9,
                               'share_of_gpu_time': d['ms'] / sum(k['ms'] for k in bykern.values())}
            out['kernels'] = {k: {'ms_per_step': v['ms'] / args.steps, 'tflops': (v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['ms'] > 0 else 0.0,
                                  'launches_per_step': v['launches'] / args.steps} for k, v in bykern.items()}
            if args.layers:
                tot = sum(r['ms'] for r in recs)
                for r in sorted(recs, key=lambda r: -r['ms']):
                    tf = r['flops'] / (r['ms'] * 1e-3) / 1e12 if r['ms'] > 0 else 0.0
                    print(f"  {r['layer']:<22} {r['kernel']:<40} {r['ms'] / args.steps:8.3f} ms/step {100 * r['ms'] / tot:5.1f}%  {tf:7.1f} TF/s",
                          file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(sd, args.shot, args.cpu_episodes, args.model)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
