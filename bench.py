#!/usr/bin/env python3
"""Benchmark of the fsvit hot path: BASELINE.json configs[1] — Visformer-S (`visformer_micro_80`)
5-way 5-shot episodic eval (15 queries/class), bf16 MFMA, synthetic 80x80 episodes resident in HBM.

  python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE in
the environment: this process IS a rank) and invoked plainly (`python bench.py --gpus N`): the parent then makes NO GPU call,
starts one fresh child process per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's JSON line and exits
with the worst child status (never an exec from a process that touched the GPU).

A "step" = one pass of `MetaBaseline.forward` (encoder + cosine head, one C-ABI call) over a batch of `--episodes` (default 128) episodes per
GPU.  Episodes are independent, so ranks shard them with no data-path collective (weak scaling); the only exchange is ONE
all-reduce of the accuracy statistics at the end of the timed region (RCCL over xGMI), as in the north star.  Every step of
the timed region sees a different batch of episodes (a pool generated in HBM before timing), so the reported accuracy and CI
are over distinct episodes.  Rank 0 prints one JSON line.

Extra legs (rank 0, N = 1 only):
  roofline      per-launch HIP-event timing inside the timed region (engine profile mode) -> the dominant kernel's
                algorithmic TFLOP/s against the 2.5 PFLOP/s dense bf16 MFMA peak; HBM traffic / MFMA-busy fraction of that
                kernel from the committed rocprofv3 PMC passes (profiles/).
  modes         the other numerics modes over the SAME episode pool (>= 2000 episodes): `parity` (exact-fp32 MFMA, the mode
                that meets the north star's 1e-3 logit tolerance) throughput against the 157.3 TFLOP/s fp32-MFMA peak.
  agreement     headline mode vs parity on those episodes: arg-max agreement, per-batch accuracies, |delta mean| vs the CI,
                max |delta logit|.
  cpu_baseline  the oracle (oracle/visformer_oracle.py, the pinned port of the reference's CPU path) timed on the host
                cores over a bounded sample of the SAME episodes, best of a thread-count / ep_per_batch sweep, with its
                accuracy next to the GPU's on those episodes.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# dmabuf IPC is the only inter-process GPU memory sharing this driver supports: RCCL between the ranks of one node needs it, and it must be in the
# environment before the first HIP call of the process (the launcher path sets it for its children; under an external torchrun this line does)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic forward FLOPs per image (2*MAC, SURVEY.md 8d / BASELINE.md 2) and input size per encoder
MODELS = {'visformer_micro_80': (2.0306e9, 80), 'deit_small_patch16_224': (9.197e9, 224), 'deit_micro_patch6_84': (4.716e9, 84)}
HEAD_FLOP_PER_EPISODE = 0.38e6
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'f16': 2500.0, 'parity': 157.3, 'bf16x2': 625.0, 'f16x2': 625.0}    # dense peaks, MI355X_MICROARCH.md (two-limb modes: 4 16-bit MFMA products per fp32 product)
DTYPE_NAME = {'bf16': 'bf16', 'f16': 'f16', 'parity': 'f32', 'bf16x2': 'bf16x2', 'f16x2': 'f16x2'}
NOISE = 1.0


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--episodes', type=int, default=128, help='episodes per GPU per step (ep_per_batch): one encoder launch chain over 12800 images; '
                                                              '64 -> 128 -> 192 measured 3697 -> 3766 -> 3806 episodes/s on one box (tile tails of the persistent kernels)')
    ap.add_argument('--shot', type=int, default=5)
    ap.add_argument('--model', default='visformer_micro_80', choices=sorted(MODELS), help='encoder (default = BASELINE configs[1])')
    ap.add_argument('--numerics', default='bf16', choices=['bf16', 'f16', 'bf16x2', 'f16x2', 'parity'])
    ap.add_argument('--chunk', type=int, default=int(os.environ.get('FSVIT_CHUNK', 12800)), help='images per encoder chunk')
    ap.add_argument('--pool', type=int, default=16, help='distinct episode batches generated in HBM before timing (steps cycle through them); '
                                                         '16 x 128 = 2048 episodes = the configs[1] evaluation size')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-modes', action='store_true', help='skip the parity-mode leg and the agreement figures')
    ap.add_argument('--no-legs', action='store_true', help="skip the other configurations' legs (train / deit / deit_train / end_to_end)")
    ap.add_argument('--cpu-episodes', type=int, default=20, help='timed episodes of the CPU baseline (BASELINE.md 3b: 3 warm-up + >= 20 timed)')
    ap.add_argument('--layers', action='store_true', help='print the per-layer timing table to stderr')
    ap.add_argument('--mode', default='eval', choices=['eval', 'train', 'distill'],
                    help="eval = BASELINE configs[1] (the headline metric); train = configs[2], one SUN-M meta-tuning step "
                         "(train_meta_mini_visformer_5shot.yaml geometry: 8 episodes x 10-way (5 shot + 5 query) = 800 images)")
    ap.add_argument('--train-episodes', type=int, default=8, help='train mode: episodes per GPU per step (ep_per_batch)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help='nccl == RCCL over xGMI; gloo only with --selftest-launcher')
    ap.add_argument('--via-launcher', action='store_true', help='start the rank process(es) from a GPU-untouched parent even for --gpus 1 (the route --gpus N > 1 takes when invoked plainly)')
    ap.add_argument('--selftest-launcher', action='store_true',
                    help='exercise the rank launcher and the timing / statistics exchange on the CPU (gloo), no GPU work: tests/test_bench_launcher_cpu.py')
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ rank launcher
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def count_gpus_without_hip():
    """Number of GPUs a rank process will see, found WITHOUT any HIP / torch.cuda call in THIS process (it is about to start N children and
    must never have initialised the runtime): a throw-away child counts them (whatever it initialises dies with it); if that child cannot be
    started, the KFD topology in sysfs (nodes with SIMDs) clipped by the *_VISIBLE_DEVICES lists."""
    try:
        r = subprocess.run([sys.executable, '-c', 'import torch; print(torch.cuda.device_count())'], capture_output=True, text=True, timeout=300)
        if r.returncode == 0:
            return int(r.stdout.strip().splitlines()[-1])
    except Exception:
        pass
    n = 0
    root = '/sys/class/kfd/kfd/topology/nodes'
    try:
        for node in os.listdir(root):
            with open(os.path.join(root, node, 'properties')) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get('simd_count', 0)) > 0:
                n += 1
    except OSError:
        return 0
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        if os.environ.get(var, '').strip():
            n = min(n, len([x for x in os.environ[var].split(',') if x.strip()]))
    return n


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N rank processes from this (GPU-untouched) parent."""
    n = args.gpus
    if args.backend == 'nccl':
        have = count_gpus_without_hip()
        if have < n:
            sys.stderr.write(f'bench.py --gpus {n}: only {have} GPU(s) visible on this node - one rank per GPU is required '
                             f'(run with --gpus <= {max(have, 1)})\n')
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: required for RCCL between processes on this driver
        # rank 0 writes the JSON line straight to our stdout; anything the other ranks print goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    deadline = None
    while procs:
        for p in list(procs):
            code = p.poll()
            if code is None:
                continue
            procs.remove(p)
            if code != 0 and rc == 0:
                rc = code
                deadline = time.time() + 30.0          # a rank died: the others would wait in a collective for ever
        if deadline is not None and time.time() > deadline:
            for p in procs:
                p.kill()
        time.sleep(0.05)
    return rc


def selftest_worker(args, rank, world):
    """The distributed skeleton of the bench without the GPU: barrier, K 'steps', the statistics all-reduce, max-over-ranks time."""
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    accs = torch.tensor([0.25 + 0.5 * ((rank * 31 + s * 7) % 11) / 11.0 for s in range(args.steps)], dtype=torch.float64)
    stats = torch.stack([accs.sum(), (accs * accs).sum(), torch.tensor(float(accs.numel()), dtype=torch.float64)])
    if world > 1:
        dist.all_reduce(stats)
        dist.barrier()
    elapsed = time.perf_counter() - t0 + 0.001 * rank
    tt = torch.tensor([elapsed], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if world > 1 and (dist.get_world_size() != world or dist.get_rank() != rank):
        return 3
    ranks_seen, per_rank = _ranks_seen(args.steps / elapsed, rank, world, 'cpu')       # the same helper the GPU legs use
    if ranks_seen != world:
        return 3
    if rank == 0:
        print(json.dumps({'selftest': True, 'n_gpus': world, 'steps': args.steps, 'stat_n': float(stats[2]), 'stat_sum': float(stats[0]),
                          'elapsed_max_s': float(tt[0]), 'backend': 'gloo', 'ranks_seen': ranks_seen, 'per_rank_eps': per_rank}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


# ------------------------------------------------------------------------------------------------ workload
def device_episodes(seed, n_ep, way, shot, query, dev, img=80):
    """Class-structured synthetic episodes generated directly in HBM, x = mu_c + NOISE * eps.  NOISE = 1.0 puts the 5-way 5-shot
    accuracy of the procedural-weight model at ~80 %, where the reference sits on miniImageNet (83.25 %); round 1 used 1.5, which is
    30 % = near chance, where arg-max agreement between numerics modes measures coin flips (tools/calib_noise.py).  Layout = what
    fs.split_shot_query returns."""
    g = torch.Generator(device=dev).manual_seed(seed)
    per = shot + query
    mu = torch.randn(n_ep, way, 1, 3, img, img, device=dev, generator=g)
    x = mu + NOISE * torch.randn(n_ep, way, per, 3, img, img, device=dev, generator=g)
    x_shot = x[:, :, :shot].contiguous()
    x_query = x[:, :, shot:].contiguous().view(n_ep, way * query, 3, img, img)
    return x_shot, x_query


def _physical_cores():
    try:
        import psutil
        return psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        return os.cpu_count()


def _usable_cores():
    """Physical cores this process may actually use: the host's physical core count clipped by the scheduler affinity mask and by a cgroup
    CPU quota (a container limited to 16 CPUs of a 128-core host gains nothing from 128 threads - the sweeps of rounds 1 / 2 that peaked at
    16 threads are what such a quota looks like).  Returns (usable, dict of what was found)."""
    logical, physical = os.cpu_count() or 1, _physical_cores() or 1
    info = {'logical': logical, 'physical': physical}
    usable = physical
    try:
        aff = len(os.sched_getaffinity(0))
        info['affinity_cpus'] = aff
        if aff < logical:
            usable = min(usable, max(1, aff * physical // logical))
    except Exception:
        pass
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:                     # cgroup v2
            q, per = f.read().split()[:2]
            if q != 'max':
                quota = float(q) / float(per)
    except Exception:
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:    # cgroup v1
                q = float(f.read())
            with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
                per = float(f.read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        info['cgroup_cpu_quota'] = quota
        usable = min(usable, max(1, int(quota)))
    info['usable'] = usable
    return usable, info


CPU_WARMUP = 3


def _oracle_runner(sd, model, img, shot):
    from oracle import visformer_oracle as vo
    if model == 'visformer_micro_80':
        cfg = vo.VisformerCfg()
        return lambda xs, xq: vo.meta_baseline_forward(sd, xs, xq, cfg)
    from oracle import deit_oracle as do
    cfg = do.FACTORIES[model]

    def run(xs, xq):
        E = xs.shape[0]
        with torch.no_grad():
            f = do.deit_forward(sd, torch.cat([xs.reshape(-1, 3, img, img), xq.reshape(-1, 3, img, img)]), cfg, prefix='encoder.')
        n = E * xs.shape[1] * xs.shape[2]
        return vo.meta_baseline_head(f[:n].reshape(E, 5, xs.shape[2], -1), f[n:].reshape(E, 75, -1), temp=10.0)
    return run


def cpu_worker(argv):
    """One worker process of the episode-parallel CPU baseline: `bench.py --cpu-worker <dir> <w> <W> <threads> <model>`.  Loads the sample the
    parent wrote, warms up, reports ready, waits for the parent's go file, runs the oracle on episodes w::W (5-shot, then the same episodes
    as 1-shot: the first shot image of every class) and writes its logits."""
    d, w, W, threads, model = argv[0], int(argv[1]), int(argv[2]), int(argv[3]), argv[4]
    torch.set_num_threads(threads)
    blob = torch.load(os.path.join(d, 'sample.pt'))
    xs, xq, sd = blob['xs'], blob['xq'], blob['sd']
    run = _oracle_runner(sd, model, xs.shape[-1], xs.shape[2])
    mine = list(range(w, xs.shape[0], W))
    for _ in range(CPU_WARMUP):                                  # 3 warm-up episodes, then the timed ones
        run(xs[:1], xq[:1])
    open(os.path.join(d, 'ready.%d' % w), 'w').close()
    while not os.path.exists(os.path.join(d, 'go')):
        time.sleep(0.005)
    t0 = time.time()
    out5 = [run(xs[e:e + 1], xq[e:e + 1]) for e in mine]
    t1 = time.time()
    out1 = [run(xs[e:e + 1, :, :1], xq[e:e + 1]) for e in mine]
    t2 = time.time()
    torch.save({'episodes': mine, 'logits5': torch.cat(out5) if out5 else None, 'logits1': torch.cat(out1) if out1 else None, 't': (t0, t1, t2)},
               os.path.join(d, 'out.%d.pt' % w))
    return 0


def cpu_baseline(sd, xs_all, xq_all, gpu_logits, n_ep, model='visformer_micro_80'):
    """The oracle (the pinned port of the reference's CPU path) on the host cores over the FIRST episodes of the GPU leg's pool (same tensors,
    copied to the host), BASELINE.md 3b: all physical cores.  One torch process does not scale past ~16 threads on this network (a short
    thread sweep is reported), so the whole-host figure is EPISODE-PARALLEL: W = physical cores / T worker processes of T threads each, every
    worker running whole episodes at ep_per_batch = 1 (the reference's test setting, test_few_shot.py:47-48); `value` = episodes of all workers /
    wall time between the common go signal and the last worker's finish.  5-way 5-shot is `value`; the same episodes as 5-way 1-shot (first
    shot image of each class) are reported next to it.  Accuracy and logits of the 5-shot episodes are compared with the GPU's."""
    import shutil
    import tempfile
    img = xs_all.shape[-1]
    shot = xs_all.shape[2]
    run = _oracle_runner(sd, model, img, shot)

    def rate(threads, n):
        torch.set_num_threads(threads)
        run(xs_all[:1], xq_all[:1])                                     # warm-up at this setting
        t0 = time.perf_counter()
        for e in range(n):
            run(xs_all[e:e + 1], xq_all[e:e + 1])
        return n / (time.perf_counter() - t0)

    logical, physical = os.cpu_count() or 1, _physical_cores() or 1
    usable, core_info = _usable_cores()
    sweep = {'%dthr' % t: rate(t, 2) for t in sorted({t for t in (8, 16, 32) if t <= logical} or {logical})}
    best_t = int(max(sweep, key=sweep.get)[:-3])
    W = max(1, usable // best_t)
    n_ep = max(W, min(xs_all.shape[0], max(n_ep, 3 * W)))
    d = tempfile.mkdtemp(prefix='fsvit_cpu_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    try:
        torch.save({'xs': xs_all[:n_ep].contiguous(), 'xq': xq_all[:n_ep].contiguous(), 'sd': sd}, os.path.join(d, 'sample.pt'))
        env = dict(os.environ, OMP_NUM_THREADS=str(best_t), MKL_NUM_THREADS=str(best_t))
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker', d, str(w), str(W), str(best_t), model], env=env) for w in range(W)]
        deadline = time.time() + 600
        while not all(os.path.exists(os.path.join(d, 'ready.%d' % w)) for w in range(W)):
            if time.time() > deadline or any(p.poll() not in (None, 0) for p in procs):
                for p in procs:
                    p.kill()
                raise RuntimeError('cpu_baseline: a worker did not come up')
            time.sleep(0.01)
        open(os.path.join(d, 'go'), 'w').close()
        for p in procs:
            p.wait(timeout=900)
        outs = [torch.load(os.path.join(d, 'out.%d.pt' % w)) for w in range(W)]
    finally:
        shutil.rmtree(d, ignore_errors=True)
    t_go = min(o['t'][0] for o in outs)
    dt5 = max(o['t'][1] for o in outs) - t_go
    dt1 = max(o['t'][2] - o['t'][1] for o in outs)                    # workers start their 1-shot pass as they finish the 5-shot one
    logits = torch.zeros(n_ep, *outs[0]['logits5'].shape[1:])
    for o in outs:
        logits[o['episodes']] = o['logits5']
    way, Q = logits.shape[-1], logits.shape[1]
    label = torch.arange(way).repeat_interleave(Q // way)
    acc_cpu = (logits.argmax(-1) == label).float().mean().item()
    g = gpu_logits[:n_ep].cpu()
    acc_gpu = (g.argmax(-1) == label).float().mean().item()
    return {'value': n_ep / dt5, 'unit': 'episodes/s', 'cores': W * best_t, 'kind': 'port',
            'sample': f'{n_ep} episodes 5-way {shot}-shot ({way * (shot + Q // way)} images each) = the first {n_ep} episodes of the GPU leg, fp32 torch CPU '
                      f'oracle, ep_per_batch=1, {CPU_WARMUP} warm-up episodes per worker before the timed ones (BASELINE.md 3b), episode-parallel: {W} worker processes x {best_t} threads (host: {physical} physical / {logical} logical '
                      f'cores, {usable} usable by this process), {dt5:.1f} s; then the same episodes as 1-shot, {dt1:.1f} s',
            'one_shot': {'value': n_ep / dt1, 'unit': 'episodes/s', 'images_per_episode': way * (1 + Q // way)},
            'single_process_thread_sweep_episodes_per_s': sweep, 'workers': W, 'threads_per_worker': best_t, 'host_cores': core_info,
            'accuracy': acc_cpu, 'gpu_accuracy_same_episodes': acc_gpu,
            'gpu_max_abs_dlogit_same_episodes': (g - logits).abs().max().item(),
            'gpu_argmax_agreement_same_episodes': (g.argmax(-1) == logits.argmax(-1)).float().mean().item()}


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
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)

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
    return {'value': reps * n_ep / dt, 'unit': 'episodes/s', 'cores': threads, 'kind': 'port',
            'sample': f'{reps} x forward+backward of {n_ep} episode(s) 10-way 5-shot 5-query (100 images each), fp32 torch CPU oracle '
                      f'with autograd, no optimizer step, 1 warm-up, {dt:.1f} s'}


def _barrier(world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def _ranks_seen(local_rate, rank, world, dev):
    """What the collective itself saw: an all-reduce of ones (= the number of ranks that joined it) and every rank's own rate (a one-hot vector summed
    over the ranks).  A JSON line whose n_gpus only repeated WORLD_SIZE from the environment would say nothing about RCCL."""
    v = torch.zeros(world + 1, device=dev, dtype=torch.float64)
    v[0] = 1.0
    v[1 + rank] = local_rate
    if world > 1:
        dist.all_reduce(v)
    v = v.cpu().tolist()
    return int(round(v[0])), [float(x) for x in v[1:]]


def _max_over_ranks(elapsed, world, dev):
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed


def train_leg(model_name, numerics, E, steps, warmup, rank, world, dev):
    """`steps` timed meta-tuning steps of `model_name` (forward + CE + backward + (gradient all-reduce) + SGD) on E episodes per GPU of
    10-way 5-shot 5-query; returns (whole-job episodes/s, ms per step, final loss, state dict)."""
    from fewshot_vit_amd import models, synthetic, utils, parallel
    from fewshot_vit_amd.utils import few_shot as fs
    vis = model_name == 'visformer_micro_80'
    img = MODELS[model_name][1]
    model = models.make('meta-baseline', encoder=model_name, encoder_args={'numerics': numerics, 'drop_path_rate': 0.5 if vis else 0.1})
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synthetic.synthetic_checkpoint_sd(shapes, calib='visformer_micro_80' if vis else None)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    opt, _ = utils.make_optimizer(model.parameters(), 'sgd', lr=0.001, weight_decay=5e-4)
    way, shot, query = 10, 5, 5
    x_shot, x_query = device_episodes(999 + rank, E, way, shot, query, dev, img)
    label = fs.make_nk_label(way, query, E).to(dev)
    reducer = parallel.GradBucket(model) if world > 1 else None

    def step():       # train_meta.train_step without its end-of-step .item() (loss and accuracy stay on the device)
        loss, _acc, _ = model.forward_loss(x_shot, x_query, label)        # head + F.cross_entropy + compute_acc in one launch (fsvit_proto_head_ce)
        opt.zero_grad()
        loss.backward()
        if reducer is not None:
            reducer.allreduce_mean()
        opt.step()
        return loss

    for _ in range(warmup):
        step()
    _barrier(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    _barrier(world)
    local = time.perf_counter() - t0
    elapsed = _max_over_ranks(local, world, dev)
    train_leg.ranks = _ranks_seen(E * steps / local, rank, world, dev)
    return world * E * steps / elapsed, 1e3 * elapsed / steps, float(loss.detach()), sd


def train_workload(model_name):
    vis = model_name == 'visformer_micro_80'
    img = MODELS[model_name][1]
    return ('BASELINE configs[2]: SUN-M train_meta.py meta-tuning step, Visformer-S (visformer_micro_80, drop_path 0.5), ep_per_batch episodes of '
            '10-way 5-shot 5-query 80x80 (train_meta_mini_visformer_5shot.yaml), forward + CE + backward + SGD(0.9, wd 5e-4), episodes resident '
            'in HBM') if vis else ('train_meta.py meta-tuning step with encoder %s (drop_path 0.1), episodes of 10-way 5-shot 5-query %dx%d, '
                                   'forward + CE + backward + SGD(0.9, wd 5e-4), episodes resident in HBM' % (model_name, img, img))


def train_main(args, rank, world, dev):
    """BASELINE configs[2]: one meta-tuning step = model.train() forward, CE, backward, (grad all-reduce), SGD step."""
    vis = args.model == 'visformer_micro_80'
    E = args.train_episodes
    eps, ms, loss, sd = train_leg(args.model, args.numerics, E, args.steps, args.warmup, rank, world, dev)
    if train_leg.ranks[0] != world:
        sys.stderr.write('bench.py: rank %d: the collective saw %d ranks, expected %d\n' % (rank, train_leg.ranks[0], world))
        sys.exit(3)
    if rank == 0:
        imgs = 100
        flops_ep = 3.0 * MODELS[args.model][0] * imgs          # forward + dgrad + wgrad
        peak = MFMA_PEAK_TFLOPS[args.numerics]
        out = {'metric': 'train_episodes_per_sec_10way_5shot_%s' % ('visformer_s' if vis else args.model), 'value': eps, 'unit': 'episodes/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms, 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE_NAME[args.numerics], 'data': 'synthetic',
               'config': {'workload': train_workload(args.model), 'episodes_per_step_per_gpu': E, 'images_per_episode': imgs,
                          'parallelism': 'episode axis sharded x%d, one all-reduce of the flat gradient bucket per step' % world,
                          'ranks_seen': train_leg.ranks[0], 'per_rank_eps': ','.join('%.1f' % v for v in train_leg.ranks[1])},
               'ranks_seen': train_leg.ranks[0], 'per_rank_eps': train_leg.ranks[1],
               'whole_path_tflops': eps * flops_ep / 1e12, 'whole_path_mfma_frac': eps * flops_ep / 1e12 / peak,
               'roofline': {'bound': 'mfma', 'achieved': eps * flops_ep / 1e12, 'peak': peak, 'unit': 'TFLOP/s',
                            'frac': eps * flops_ep / 1e12 / peak, 'traffic': None, 'kernel': 'whole training step (forward + backward + SGD)'},
               'final_loss': loss}
        if vis and args.numerics == 'bf16' and E == 8:
            out['roofline'].update({'hbm': leg_roofline('train_', ms, eps * flops_ep / 1e12 / peak)})
        if world == 1 and not args.no_cpu_baseline and vis:
            out['cpu_baseline'] = cpu_train_baseline(sd)
        print(json.dumps(out), flush=True)


def extra_legs(args, dev):
    """The other single-GPU configurations of BASELINE.json on the same JSON line (rank 0, N = 1): configs[2] = the 800-image SUN-M meta-tuning
    step (`train`), the DeiT-S/16 224 x 224 shape of configs[4] in eval (`deit`) and as a meta-tuning step (`deit_train`, 2 episodes = 200
    images).  Each leg reports ms per step, whole-job episodes/s, algorithmic TFLOP/s and the fraction of the dense bf16 MFMA peak."""
    legs = {}
    peak = MFMA_PEAK_TFLOPS['bf16']

    def free():
        import gc
        gc.collect()
        torch.cuda.empty_cache()

    eps, ms, loss, _ = train_leg('visformer_micro_80', 'bf16', 8, 20, 3, 0, 1, dev)
    tf = eps * 3.0 * MODELS['visformer_micro_80'][0] * 100 / 1e12
    legs['train'] = {'value': eps, 'unit': 'train-episodes/s', 'ms_per_step': ms, 'steps': 20, 'warmup': 3, 'dtype': 'bf16', 'images_per_step': 800,
                     'whole_path_tflops': tf, 'whole_path_mfma_frac': tf / peak, 'final_loss': loss, 'workload': train_workload('visformer_micro_80'),
                     'note': 'FLOPs = 3 x forward (forward + data gradient + weight gradient), 4.87 TFLOP per step'}
    free()
    # the 1e-3-grade training mode (VERDICT r02 'missing' #2): the same step in `bf16x2` (fp32 storage, every GEMM as two-limb bf16 MFMAs)
    eps, ms, loss, _ = train_leg('visformer_micro_80', 'bf16x2', 8, 5, 2, 0, 1, dev)
    tf = eps * 3.0 * MODELS['visformer_micro_80'][0] * 100 / 1e12
    legs['train_bf16x2'] = {'value': eps, 'unit': 'train-episodes/s', 'ms_per_step': ms, 'steps': 5, 'warmup': 2, 'dtype': 'bf16x2', 'images_per_step': 800,
                            'whole_path_tflops': tf, 'whole_path_mfma_frac': tf / MFMA_PEAK_TFLOPS['bf16x2'], 'final_loss': loss,
                            'note': 'gradients within 5e-5 of the fp32 oracle outside the stem (tests/test_gpu_train.py); peak = 625 TFLOP/s (four limb products)'}
    free()
    name = 'deit_small_patch16_224'
    flop, img = MODELS[name]
    from fewshot_vit_amd import models, synthetic
    m = models.make('meta-baseline', encoder=name, encoder_args={'numerics': 'bf16'})
    m.load_state_dict(synthetic.synthetic_checkpoint_sd({k: tuple(v.shape) for k, v in m.state_dict().items()}, calib=None), strict=True)
    m = m.to(dev).eval()
    eng = m.encoder.engine()
    E, steps, warm = args.episodes, 5, 2
    pool = [device_episodes(777 + i, E, 5, 5, 15, dev, img) for i in range(2)]
    temp = float(m.temp.detach())
    for i in range(warm):
        eng.meta_baseline_forward(*pool[i % 2], temp, 'cos', want_stats=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        _, acc, _ = eng.meta_baseline_forward(*pool[i % 2], temp, 'cos', want_stats=True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    eps = E * steps / el
    tf = eps * (flop * 100 + HEAD_FLOP_PER_EPISODE) / 1e12
    legs['deit'] = {'value': eps, 'unit': 'episodes/s', 'ms_per_step': 1e3 * el / steps, 'steps': steps, 'warmup': warm, 'dtype': 'bf16',
                    'episodes_per_step': E, 'whole_path_tflops': tf, 'whole_path_mfma_frac': tf / peak, 'accuracy_last_step': float(acc.mean()),
                    'workload': 'BASELINE configs[4] shape on one GPU: deit_small_patch16_224 (ViT-S/16), 224x224, 5-way 5-shot episodic eval, '
                                '15 query/class, synthetic episodes resident in HBM, procedural weights'}
    del pool, eng, m
    free()
    eps, ms, loss, _ = train_leg(name, 'bf16', 2, 10, 3, 0, 1, dev)
    tf = eps * 3.0 * flop * 100 / 1e12
    legs['deit_train'] = {'value': eps, 'unit': 'train-episodes/s', 'ms_per_step': ms, 'steps': 10, 'warmup': 3, 'dtype': 'bf16', 'images_per_step': 200,
                          'whole_path_tflops': tf, 'whole_path_mfma_frac': tf / peak, 'final_loss': loss, 'workload': train_workload(name)}
    free()
    legs['end_to_end'] = end_to_end_leg(args, dev)
    legs['end_to_end_default'] = legs['end_to_end'].pop('default_call')
    free()
    legs['distill'] = distill_leg(dev)
    free()
    return legs


def distill_leg(dev, batch=512, steps=8, warm=3, n_classes=64):
    """BASELINE configs[3] at its own size (sun_meta_training/offline.py:263-309): one SUN meta-training step = `token-label` student over
    Visformer-S in train mode (drop_path 0.1) on a batch of 512 images, global CE + 0.5 x SoftTargetCrossEntropy of its 25 x 65 token logits
    against soft labels generated from the FROZEN teacher's token logits (eval engine), backward, AdamW.  Images resident in HBM."""
    from fewshot_vit_amd import models, offline, synthetic
    from fewshot_vit_amd.models.classifier import FsvitAdamW, SoftTargetCrossEntropy
    margs = dict(encoder='visformer_micro_80', encoder_args=dict(drop_path_rate=0.1, return_map=True, numerics='bf16'), classifier='linear-classifier',
                 classifier_args=dict(n_classes=n_classes))
    student, teacher = models.make('token-label', **margs).to(dev), models.make('token-label', **margs).to(dev)
    for mdl in (student, teacher):
        enc_shapes = {k: tuple(v.shape) for k, v in mdl.encoder.state_dict().items()}
        esd = synthetic.synthetic_checkpoint_sd({'encoder.' + k: s for k, s in enc_shapes.items()}, calib='visformer_micro_80')
        mdl.encoder.load_state_dict({k[len('encoder.'):]: v for k, v in esd.items()})
    student.train()
    teacher.eval()
    opt = FsvitAdamW(student.parameters(), betas=(0.9, 0.999), eps=1e-8, lr=5e-4 * batch / 512, weight_decay=0.05)
    crit = SoftTargetCrossEntropy()
    g = torch.Generator(device=dev).manual_seed(4242)
    x = torch.randn(batch, 3, 80, 80, device=dev, generator=g)
    label = torch.randint(0, n_classes, (batch,), device=dev, generator=g)
    for _ in range(warm):
        offline.distill_step(student, teacher, opt, crit, x, x, label)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, acc = offline.distill_step(student, teacher, opt, crit, x, x, label)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    flop = 4.0 * MODELS['visformer_micro_80'][0] * batch              # student forward + data gradient + weight gradient, teacher forward
    tf = flop / (ms * 1e-3) / 1e12
    return {'value': batch / (ms * 1e-3), 'unit': 'images/s', 'ms_per_step': ms, 'steps': steps, 'warmup': warm, 'dtype': 'bf16', 'images_per_step': batch,
            'whole_path_tflops': tf, 'whole_path_mfma_frac': tf / MFMA_PEAK_TFLOPS['bf16'], 'final_loss': float(loss),
            'workload': 'BASELINE configs[3]: one offline.py distillation step at batch 512 - token-label student (Visformer-S, train mode, drop_path 0.1) + frozen '
                        'teacher (eval engine), generate_softlabel(k=3, bp=10), CE + 0.5 x SoftTargetCrossEntropy on 25 x 65 token logits, backward, AdamW',
            'note': 'FLOPs = 4 x encoder forward (student forward + data gradient + weight gradient, teacher forward), 4.16 TFLOP per step'}


def end_to_end_leg(args, dev, n_images=12000, n_classes=20):
    """The reference's own evaluation loop (test_phase/test_few_shot.py:45-56,79-94) end to end: a miniImageNet-FORMAT pickle (uint8
    [12000, 84, 84, 3], 20 classes x 600 - the test split's shape, synthetic content) -> datasets.make('mini-imagenet') (one upload) ->
    CategoriesSampler on the host (the reference's legacy-RNG stream) -> fsvit_image_transform_gather (Resize 88 / CenterCrop 80 / Normalize
    on the GPU) -> fsvit_meta_baseline_forward -> acc +- CI over 2000 batches of one 5-way 5-shot episode."""
    import pickle
    import tempfile
    import numpy as np
    from fewshot_vit_amd import test_few_shot
    rng = np.random.default_rng(12345)
    per = n_images // n_classes
    mu = rng.integers(64, 192, size=(n_classes, 1, 84, 84, 3)).astype(np.int16)
    data = np.clip(mu + rng.integers(-64, 65, size=(n_classes, per, 84, 84, 3), dtype=np.int16), 0, 255).astype(np.uint8).reshape(-1, 84, 84, 3)
    labels = np.repeat(np.arange(n_classes), per).tolist()
    with tempfile.TemporaryDirectory() as root:
        with open(os.path.join(root, 'miniImageNet_category_split_test.pickle'), 'wb') as f:
            pickle.dump({'data': data, 'labels': labels}, f, protocol=4)
        # the model comes from a CHECKPOINT FILE with the reference's schema (train_meta.py:241-257), `load:` as test_few_shot.py:56-59 reads it: the timed
        # calls pay torch.load + models.load like a drop-in run (until round 5 this leg generated the 12.5 M procedural weights inside the call: 0.6 s)
        from fewshot_vit_amd import models as fmodels, synthetic as fsyn
        m0 = fmodels.make('meta-baseline', encoder='visformer_micro_80', encoder_args={})
        sd0 = fsyn.synthetic_checkpoint_sd({k: tuple(v.shape) for k, v in m0.state_dict().items()}, calib='visformer_micro_80')
        torch.save({'model': 'meta-baseline', 'model_args': {'encoder': 'visformer_micro_80', 'encoder_args': {}}, 'model_sd': sd0}, os.path.join(root, 'max-va.pth'))
        del m0, sd0
        cfg = {'dataset': 'mini-imagenet', 'dataset_args': {'root_path': root, 'split': 'test'}, 'load': os.path.join(root, 'max-va.pth')}
        logs = []
        test_few_shot.evaluate(cfg, shot=5, n_batch=256, launch_batches=args.episodes, numerics='bf16', device=dev, log=logs.append)     # warm-up
        t0 = time.perf_counter()
        r = test_few_shot.evaluate(cfg, shot=5, n_batch=2000, launch_batches=args.episodes, numerics='bf16', device=dev, log=logs.append)
        wall = time.perf_counter() - t0
        # what a drop-in user gets: the call the reference's CLI makes, NO tuning argument (launch size, numerics mode and encoder chunk at their
        # defaults; the FSVIT_CHUNK this bench exports for its own legs is taken out of the environment for the call)
        keep = os.environ.pop('FSVIT_CHUNK', None)
        try:
            t0 = time.perf_counter()
            rd = test_few_shot.evaluate(cfg, shot=5, device=dev, log=logs.append)
            wall_d = time.perf_counter() - t0
        finally:
            if keep is not None:
                os.environ['FSVIT_CHUNK'] = keep
    default = {'value': 2000 / rd['loop_seconds'], 'unit': 'episodes/s', 'episodes': rd['n'], 'loop_seconds': rd['loop_seconds'], 'evaluate_call_seconds': wall_d,
               'episodes_per_launch': rd['launch_batches'], 'phase_seconds': rd['phase_seconds'], 'accuracy': rd['acc'], 'ci95': rd['ci'],
               'workload': 'the same table through test_few_shot.evaluate(config, shot=5) with every other argument at its default: what the reference\'s '
                           'command line runs'}
    return {'value': 2000 / r['loop_seconds'], 'unit': 'episodes/s', 'episodes': 2000, 'loop_seconds': r['loop_seconds'], 'evaluate_call_seconds': wall,
            'phase_seconds': r['phase_seconds'], 'default_call': default,
            'episodes_per_launch': args.episodes, 'accuracy': r['acc'], 'ci95': r['ci'], 'dtype': 'bf16',
            'workload': 'test_few_shot.evaluate on a miniImageNet-format uint8 table (12000 x 84 x 84 x 3): host CategoriesSampler -> device gather + '
                        'Resize(88) / CenterCrop(80) / Normalize -> encoder + cosine head -> acc +- CI; `value` = 2000 episodes / loop time (after model and '
                        'dataset construction), `evaluate_call_seconds` = the whole call incl. pickle load, upload, checkpoint load (torch.load of the 50 MB file + models.load), weight packing; `phase_seconds` splits it'}


def _csrc_sha():
    import importlib.util
    spec = importlib.util.spec_from_file_location('csrc_hash', os.path.join(REPO, 'tools', 'csrc_hash.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.csrc_sha()


def _newest_profile(kind, what):
    """profiles/rNN_<kind><what>.json of the highest round, e.g. ('', 'hbm_traffic') -> r05_hbm_traffic.json, ('train_', 'hbm_traffic') -> r05_train_hbm_traffic.json."""
    import glob
    import re
    pat = re.compile(r'r(\d+)_%s%s\.json' % (re.escape(kind), re.escape(what)))
    best = None
    for p_ in glob.glob(os.path.join(REPO, 'profiles', 'r*_%s%s.json' % (kind, what))):
        m = pat.fullmatch(os.path.basename(p_))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), p_)
    return best[1] if best else None


# which template instantiation of a kernel runs for which layer of the eval step (the PMC summaries are keyed by instantiation)
_INSTANCE_HINT = {'stage2': '<256,', 'stage3': '<512,', 'stem.conv2': '<64,', 'stem.conv3': '<128,', 'patch_embed2': '<512,'}


def _pmc_pick(table, kernel, layer):
    base = kernel.split('<')[0]
    cand = [k for k in table if k != '_meta' and (k == kernel or k.split('<')[0] == base)]
    if len(cand) > 1:
        for pre, hint in _INSTANCE_HINT.items():
            if layer.startswith(pre):
                narrowed = [k for k in cand if hint in k]
                if narrowed:
                    cand = narrowed
                break
    return cand


def _pmc_value(table, keys, field):
    if not table or not keys:
        return None
    n = sum(table[k].get('launches', 1) for k in keys)
    return sum(table[k].get('launches', 1) * table[k].get(field, 0.0) for k in keys) / n if n else None


def _committed_pmc(kind=''):
    """The newest committed rocprofv3 PMC summaries of a bench command (profiles/rNN_<kind>{hbm_traffic,mfma_pmc}.json; separate --pmc passes of
    that command, tools/pmc_bench.sh) -> (traffic table, mfma table, {'files', 'match_current_csrc'})."""
    now = _csrc_sha()
    fresh, tabs = {}, []
    for what in ('hbm_traffic', 'mfma_pmc'):
        f_ = _newest_profile(kind, what)
        tab = None
        if f_:
            with open(f_) as fh:
                tab = json.load(fh)
            fresh[os.path.basename(f_)] = tab.get('_meta', {}).get('csrc_sha') == now
        tabs.append(tab)
    stale = sorted(k for k, ok in fresh.items() if not ok)
    if stale:
        sys.stderr.write('bench.py: WARNING - the committed PMC summaries quoted in the roofline objects (%s) were collected from other kernel sources than this '
                         'tree (few-shot-vit_amd/csrc changed since): re-run tools/pmc_bench.sh and commit the summaries\n' % ', '.join(stale))
    return tabs[0], tabs[1], {'files': sorted(fresh), 'match_current_csrc': bool(fresh) and not stale}


def _steps_of(meta):
    """steps + warm-up of the profiled command: `_meta.steps_profiled` when the summary carries it (tools/pmc_traffic.py writes it since round 6), else
    `--steps N --warmup W` parsed from _meta.source; None when neither is there (a silent default of 4 made a wrong byte count look
    authoritative)."""
    import re
    if isinstance(meta.get('steps_profiled'), int) and meta['steps_profiled'] > 0:
        return meta['steps_profiled']
    src = meta.get('source') or ''
    m1, m2 = re.search(r'--steps (\d+)', src), re.search(r'--warmup (\d+)', src)
    return (int(m1.group(1)) + int(m2.group(1))) if m1 and m2 else None


def leg_roofline(kind, ms_per_step, mfma_frac):
    """`roofline` of a training leg: these steps are HBM-bound, so the bound is bytes per step from the committed FETCH_SIZE /
    WRITE_SIZE passes of the same command (profiles/rNN_<kind>hbm_traffic.json) over the measured step time, against 8 TB/s."""
    traffic, _, src = _committed_pmc(kind)
    if not traffic:
        return {'bound': 'hbm', 'bytes_per_step': None, 'achieved_TBps': None, 'frac_of_8TBps': None, 'mfma_frac': mfma_frac, 'pmc_files': '', 'pmc_match_current_csrc': False}
    steps = _steps_of(traffic.get('_meta', {}))
    if not steps:
        return {'bound': 'hbm', 'bytes_per_step': None, 'achieved_TBps': None, 'frac_of_8TBps': None, 'mfma_frac': mfma_frac, 'pmc_files': ','.join(src['files']),
                'pmc_match_current_csrc': src['match_current_csrc'], 'note': 'the committed traffic summary does not say how many steps it profiled'}
    total = sum(v.get('launches', 1) * v.get('hbm_bytes_per_launch', 0.0) for k, v in traffic.items() if k != '_meta')
    launches = sum(v.get('launches', 1) for k, v in traffic.items() if k != '_meta')
    bps = total / steps
    tbps = bps / (ms_per_step * 1e-3) / 1e12
    return {'bound': 'hbm', 'bytes_per_step': bps, 'launches_per_step': launches / steps, 'achieved_TBps': tbps, 'peak_TBps': 8.0, 'frac_of_8TBps': tbps / 8.0,
            'mfma_frac': mfma_frac, 'pmc_files': ','.join(src['files']), 'pmc_match_current_csrc': src['match_current_csrc']}


def eval_main(args, rank, world, dev):
    from fewshot_vit_amd import models, synthetic
    os.environ['FSVIT_CHUNK'] = str(args.chunk)
    flop_per_image, img = MODELS[args.model]

    def build(numerics):
        m = models.make('meta-baseline', encoder=args.model, encoder_args={'numerics': numerics})
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        # procedural weights by key name (+ the shipped BN calibration for the Visformer)
        sd_ = synthetic.synthetic_checkpoint_sd(shapes, calib=args.model if args.model == 'visformer_micro_80' else None)
        m.load_state_dict(sd_, strict=True)
        m = m.to(dev).eval()
        return m, sd_, m.encoder.engine()

    model, sd, engine = build(args.numerics)
    way, query, E = 5, 15, args.episodes
    want_full_pool = world == 1 and not args.no_modes               # the agreement leg wants >= 2000 distinct episodes
    n_pool = args.pool if want_full_pool else min(args.pool, args.steps)
    batch_bytes = E * way * (args.shot + query) * 3 * img * img * 4
    n_pool = max(1, min(n_pool, int(24e9 // batch_bytes)))          # at most 24 GB of episodes per GPU (224x224 encoders)
    pool = [device_episodes(12345 + 1000 * rank + i, E, way, args.shot, query, dev, img) for i in range(n_pool)]
    temp = float(model.temp.detach())

    def step(eng, i):
        xs, xq = pool[i % n_pool]
        return eng.meta_baseline_forward(xs, xq, temp, 'cos', want_stats=True)

    for i in range(args.warmup):
        step(engine, i)
    profile = (not args.no_roofline) and rank == 0
    accs = []
    _barrier(world)
    if profile:
        engine.profile_begin()
    t0 = time.perf_counter()
    for i in range(args.steps):
        _, acc, _ = step(engine, i)
        accs.append(acc)
    # the one exchange of the path: all-reduce of (sum acc, sum acc^2, n) -> mean accuracy +- CI
    acc_all = torch.stack(accs[:n_pool]).double().flatten()                    # distinct episodes only
    stats = torch.stack([acc_all.sum(), (acc_all * acc_all).sum(), torch.tensor(float(acc_all.numel()), device=dev, dtype=torch.float64)])
    if world > 1:
        dist.all_reduce(stats)
    _barrier(world)
    local_elapsed = time.perf_counter() - t0
    elapsed = _max_over_ranks(local_elapsed, world, dev)
    ranks_seen, per_rank_eps = _ranks_seen(E * args.steps / local_elapsed, rank, world, dev)
    if ranks_seen != world:                # every rank checks, every rank fails: the launcher reports the worst status
        sys.stderr.write('bench.py: rank %d: the collective saw %d ranks, expected %d\n' % (rank, ranks_seen, world))
        sys.exit(3)
    recs = engine.profile_end() if profile else None

    n = float(stats[2].item())
    mean = float(stats[0].item()) / n
    var = max(0.0, (float(stats[1].item()) - n * mean * mean) / max(1.0, n - 1.0))
    import scipy.stats
    ci = (var / n) ** 0.5 * float(scipy.stats.t.ppf(0.975, max(1.0, n - 1.0)))
    if rank != 0:
        return

    total_eps = world * E * args.steps
    eps = total_eps / elapsed
    imgs = way * (args.shot + query)
    flops_ep = flop_per_image * imgs + HEAD_FLOP_PER_EPISODE
    out = {
        'metric': 'episodes_per_sec_5way_%dshot_%s' % (args.shot, 'visformer_s' if args.model == 'visformer_micro_80' else args.model), 'value': eps, 'unit': 'episodes/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': DTYPE_NAME[args.numerics], 'data': 'synthetic',
        'config': {'workload': ('BASELINE configs[1]: Visformer-S (visformer_micro_80) miniImageNet-shaped 5-way %d-shot '
                                'episodic eval, 15 query/class, 80x80 fp32 NCHW episodes resident in HBM, procedural weights + '
                                'calibrated BN' % args.shot) if args.model == 'visformer_micro_80' else
                               ('BASELINE configs[4] shape: %s, %dx%d, 5-way %d-shot episodic eval, 15 query/class, synthetic episodes '
                                'resident in HBM, procedural weights' % (args.model, img, img, args.shot)),
                   'episodes_per_step_per_gpu': E, 'images_per_episode': imgs, 'encoder_chunk_images': args.chunk,
                   'distinct_episode_batches': n_pool,
                   'parallelism': 'episode-parallel x%d, one all-reduce of accuracy stats' % world,
                   'ranks_seen': ranks_seen, 'per_rank_eps': ','.join('%.1f' % v for v in per_rank_eps)},
        # what the collective saw (an all-reduce of ones; each rank's own episodes/s over ITS wall clock of the timed region)
        'ranks_seen': ranks_seen, 'per_rank_eps': per_rank_eps,
        'whole_path_tflops': eps * flops_ep / 1e12,
        'whole_path_mfma_frac': eps * flops_ep / 1e12 / MFMA_PEAK_TFLOPS[args.numerics],
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
        peak = MFMA_PEAK_TFLOPS[args.numerics]
        achieved = d['flops'] / (d['ms'] * 1e-3) / 1e12
        traffic = busy = None
        ttab = mtab = None
        pmc_src = {'files': [], 'match_current_csrc': False}
        if args.model == 'visformer_micro_80' and args.numerics == 'bf16' and E == 128 and args.chunk == 12800:
            ttab, mtab, pmc_src = _committed_pmc('')
            dom_layers = [r['layer'] for r in recs if r['kernel'] == dom]
            keys_t = sorted({k for ly in dom_layers for k in _pmc_pick(ttab or {}, dom, ly)})
            keys_m = sorted({k for ly in dom_layers for k in _pmc_pick(mtab or {}, dom, ly)})
            traffic, busy = _pmc_value(ttab, keys_t, 'hbm_bytes_per_launch'), _pmc_value(mtab, keys_m, 'mfma_busy')
        tot_ms = sum(k['ms'] for k in bykern.values())
        # the attention + Mlp block (north star: ">= 30 % bf16 MFMA utilisation on the attention+MLP block"): every kernel of stages 2 and 3
        blk = [r for r in recs if r['layer'].startswith(('stage2', 'stage3'))]
        blk_ms, blk_fl = sum(r['ms'] for r in blk), sum(r['flops'] for r in blk)
        # the five most expensive kernel INSTANTIATIONS (a kernel name per layer group: stage 2 and stage 3 run different template arguments)
        inst = {}
        for r in recs:
            grp = r['layer'].split('.')[0] if r['layer'].startswith('stage') else r['layer']
            k = inst.setdefault((r['kernel'], grp), {'ms': 0.0, 'flops': 0.0, 'launches': 0, 'layer': r['layer']})
            k['ms'] += r['ms']
            k['flops'] += r['flops']
            k['launches'] += r['launches']
        top = []
        for (kn, grp), v in sorted(inst.items(), key=lambda kv: -kv[1]['ms'])[:5]:
            kt, km = _pmc_pick(ttab or {}, kn, v['layer']), _pmc_pick(mtab or {}, kn, v['layer'])
            hb = _pmc_value(ttab, kt, 'hbm_bytes_per_launch')
            top.append({'kernel': kn, 'layers': grp, 'instantiation': (km or kt or [kn])[0], 'ms_per_step': v['ms'] / args.steps,
                        'gflop_per_step': v['flops'] / args.steps / 1e9, 'tflops': v['flops'] / (v['ms'] * 1e-3) / 1e12 if v['ms'] > 0 else 0.0,
                        'frac': (v['flops'] / (v['ms'] * 1e-3) / 1e12 / peak) if v['ms'] > 0 else 0.0, 'mfma_busy': _pmc_value(mtab, km, 'mfma_busy'),
                        'hbm_bytes_per_step': hb * v['launches'] / args.steps if hb is not None else None, 'share_of_gpu_time': v['ms'] / tot_ms})
        out['roofline'] = {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                           'traffic': traffic, 'mfma_busy': busy, 'pmc_source': pmc_src, 'pmc_files': ','.join(pmc_src['files']),
                           'pmc_match_current_csrc': pmc_src['match_current_csrc'], 'kernel': dom, 'launches': d['launches'],
                           'avg_launch_us': 1e3 * d['ms'] / d['launches'],
                           'avg_launch_gflop': d['flops'] / d['launches'] / 1e9,
                           'share_of_gpu_time': d['ms'] / tot_ms,
                           'whole_path_tflops': eps * flops_ep / 1e12, 'whole_path_frac': eps * flops_ep / 1e12 / peak,
                           'block_tflops': blk_fl / (blk_ms * 1e-3) / 1e12 if blk_ms > 0 else None,
                           'block_frac': blk_fl / (blk_ms * 1e-3) / 1e12 / peak if blk_ms > 0 else None,
                           'block_ms_per_step': blk_ms / args.steps, 'block_note': 'attention + Mlp block = every kernel of stages 2 and 3 (qkv + attention core, proj + Mlp)',
                           'kernels': top}
        for i, tk in enumerate(top):          # (the same five as flat scalars: a record parser that keeps only scalars keeps these)
            out['roofline'].update({'top%d_kernel' % (i + 1): '%s [%s]' % (tk['instantiation'], tk['layers']), 'top%d_ms_per_step' % (i + 1): tk['ms_per_step'],
                                    'top%d_frac' % (i + 1): tk['frac'], 'top%d_mfma_busy' % (i + 1): tk['mfma_busy'],
                                    'top%d_hbm_bytes_per_step' % (i + 1): tk['hbm_bytes_per_step']})
        out['kernels'] = {k: {'ms_per_step': v['ms'] / args.steps, 'tflops': (v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['ms'] > 0 else 0.0,
                              'launches_per_step': v['launches'] / args.steps} for k, v in bykern.items()}
        if args.layers:
            tot = sum(r['ms'] for r in recs)
            for r in sorted(recs, key=lambda r: -r['ms']):
                tf = r['flops'] / (r['ms'] * 1e-3) / 1e12 if r['ms'] > 0 else 0.0
                print(f"  {r['layer']:<22} {r['kernel']:<40} {r['ms'] / args.steps:8.3f} ms/step {100 * r['ms'] / tot:5.1f}%  {tf:7.1f} TF/s",
                      file=sys.stderr)

    need_logits = world == 1 and (not args.no_modes or not args.no_cpu_baseline)
    head_logits = None
    if need_logits:                       # untimed: the headline mode's logits over the whole pool
        head_logits = torch.cat([step(engine, i)[0] for i in range(n_pool)])
        torch.cuda.synchronize()
    if world == 1 and not args.no_modes and args.model == 'visformer_micro_80' and args.numerics != 'parity':
        label = torch.arange(way, device=dev).repeat_interleave(query)

        def run_mode(numerics):
            _, _, eng = build(numerics)
            step(eng, 0)
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            lg = [step(eng, i)[0] for i in range(n_pool)]
            torch.cuda.synchronize()
            el = time.perf_counter() - t0_
            return torch.cat(lg), el

        def agreement(a_name, a_log, b_name, b_log):
            ea = (a_log.argmax(-1) == label).float().mean(dim=1).double()                       # per episode
            eb = (b_log.argmax(-1) == label).float().mean(dim=1).double()
            ba = (a_log.argmax(-1) == label).float().view(n_pool, -1).mean(dim=1).double()      # per reference batch (= one step of E episodes)
            bb = (b_log.argmax(-1) == label).float().view(n_pool, -1).mean(dim=1).double()
            ne = float(ea.numel())
            cia = float(ea.std(unbiased=True)) / ne ** 0.5 * float(scipy.stats.t.ppf(0.975, ne - 1.0)) if ne > 1 else float('nan')
            return {'modes': [a_name, b_name], 'episodes': int(ne),
                    'argmax_agreement': (a_log.argmax(-1) == b_log.argmax(-1)).float().mean().item(),
                    'accuracy_' + a_name: float(ea.mean()), 'accuracy_' + b_name: float(eb.mean()),
                    'abs_delta_mean_accuracy': abs(float(ea.mean()) - float(eb.mean())), 'ci95_per_episode': cia,
                    'max_abs_delta_batch_accuracy': float((ba - bb).abs().max()),
                    'max_abs_dlogit': (a_log - b_log).abs().max().item(), 'mean_abs_dlogit': (a_log - b_log).abs().mean().item()}

        plog, pel = run_mode('parity')
        peps = n_pool * E / pel
        out['modes'] = {'parity': {'value': peps, 'unit': 'episodes/s', 'ms_per_step': 1e3 * pel / n_pool, 'steps': n_pool, 'dtype': 'f32',
                                   'whole_path_tflops': peps * flops_ep / 1e12,
                                   'whole_path_mfma_frac': peps * flops_ep / 1e12 / MFMA_PEAK_TFLOPS['parity'],
                                   'note': 'exact-fp32 MFMA (v_mfma_f32_16x16x4_f32): the mode that meets the 1e-3 logit tolerance against the reference '
                                           '(tests/test_gpu_visformer.py); same episode pool as the headline leg'}}
        out['agreement'] = agreement(args.numerics, head_logits, 'parity', plog)
        other = 'f16' if args.numerics == 'bf16' else 'bf16'
        olog, oel = run_mode(other)
        oeps = n_pool * E / oel
        out['modes'][other] = {'value': oeps, 'unit': 'episodes/s', 'ms_per_step': 1e3 * oel / n_pool, 'steps': n_pool, 'dtype': other,
                               'whole_path_tflops': oeps * flops_ep / 1e12, 'whole_path_mfma_frac': oeps * flops_ep / 1e12 / MFMA_PEAK_TFLOPS[other],
                               'agreement_with_parity': agreement(other, olog, 'parity', plog),
                               'note': 'the same kernels compiled for the other 16-bit type (namespace fsvit_f16 = _Float16 storage + v_mfma_*_f16; '
                                       'namespace fsvit = __bf16): same episode pool, same MFMA peak; untimed-profile wall clock of %d steps' % n_pool}
        xlog, xel = run_mode('bf16x2')
        xeps = n_pool * E / xel
        out['modes']['bf16x2'] = {'value': xeps, 'unit': 'episodes/s', 'ms_per_step': 1e3 * xel / n_pool, 'steps': n_pool, 'dtype': 'bf16x2',
                                  'speedup_over_parity': xeps / peps, 'whole_path_tflops': xeps * flops_ep / 1e12,
                                  'whole_path_mfma_frac': xeps * flops_ep / 1e12 / MFMA_PEAK_TFLOPS['bf16x2'],
                                  'agreement_with_parity': agreement('bf16x2', xlog, 'parity', plog),
                                  'note': 'fp32 storage, every GEMM on the bf16 MFMA with two-limb (hi + lo) operands, 4 limb products per fp32 product '
                                          '(peak 2500 / 4 TFLOP/s); also meets the 1e-3 logit tolerance against the reference goldens '
                                          '(tests/test_gpu_visformer.py::test_logits_two_limb_modes_vs_reference_golden: 1.5e-4; f16x2: 2.2e-5)'}
        del plog, olog, xlog
    cpu_sample = None
    if world == 1 and not args.no_cpu_baseline:
        k = min(E, 32)
        cpu_sample = (pool[0][0][:k].cpu(), pool[0][1][:k].cpu(), head_logits[:k].cpu())
    if world == 1 and not args.no_legs and args.model == 'visformer_micro_80' and args.numerics == 'bf16':
        del pool, engine, model, head_logits
        out['legs'] = legs = extra_legs(args, dev)
        # the training legs are HBM-bound steps: their own roofline objects (bytes per step from the committed PMC passes of the same command)
        legs['train']['roofline'] = leg_roofline('train_', legs['train']['ms_per_step'], legs['train']['whole_path_mfma_frac'])
        legs['distill']['roofline'] = leg_roofline('distill_', legs['distill']['ms_per_step'], legs['distill']['whole_path_mfma_frac'])
        if 'roofline' in out:                 # ... and as flat scalars inside the headline's roofline object, which every record parser keeps
            rf = out['roofline']
            for name in ('train', 'distill'):
                lr = legs[name]['roofline']
                rf.update({name + '_ms_per_step': legs[name]['ms_per_step'], name + '_mfma_frac': legs[name]['whole_path_mfma_frac'],
                           name + '_hbm_bytes_per_step': lr['bytes_per_step'], name + '_hbm_TBps': lr['achieved_TBps'], name + '_hbm_frac_of_8TBps': lr['frac_of_8TBps'],
                           name + '_launches_per_step': lr.get('launches_per_step'), name + '_pmc_match_current_csrc': lr['pmc_match_current_csrc']})
            rf.update({'deit_eps': legs['deit']['value'], 'deit_mfma_frac': legs['deit']['whole_path_mfma_frac'], 'deit_train_ms_per_step': legs['deit_train']['ms_per_step'],
                       'end_to_end_eps': legs['end_to_end']['value'], 'train_bf16x2_ms_per_step': legs['train_bf16x2']['ms_per_step']})
    if cpu_sample is not None:
        out['cpu_baseline'] = cpu_baseline(sd, cpu_sample[0], cpu_sample[1], cpu_sample[2], args.cpu_episodes, args.model)
    print(json.dumps(out), flush=True)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if argv and argv[0] == '--cpu-worker':
        return cpu_worker(argv[1:])
    args = parse(argv)
    in_launcher = 'WORLD_SIZE' in os.environ and 'RANK' in os.environ
    if not in_launcher and (args.gpus > 1 or args.via_launcher):
        return launch_ranks(args, argv)
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    args.gpus = world
    if args.selftest_launcher:
        return selftest_worker(args, rank, world)
    if args.backend != 'nccl':
        sys.stderr.write('bench.py: --backend gloo is for --selftest-launcher only (the fsvit hot path runs on MI355X GPUs over RCCL)\n')
        return 2
    if not torch.cuda.is_available():
        sys.stderr.write('bench.py needs an MI355X: the fsvit hot path has no CPU fallback\n')
        return 2
    if local >= torch.cuda.device_count():
        sys.stderr.write(f'bench.py: rank {rank} has no GPU (LOCAL_RANK {local}, {torch.cuda.device_count()} visible)\n')
        return 2
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
        if dist.get_world_size() != world or dist.get_rank() != rank:
            sys.stderr.write('bench.py: process group reports rank %d of %d, the launcher said %d of %d\n' % (dist.get_rank(), dist.get_world_size(), rank, world))
            return 3
    if args.mode == 'train':
        train_main(args, rank, world, dev)
    elif args.mode == 'distill':            # BASELINE configs[3] on one GPU: the `legs.distill` entry alone (for profiling)
        if rank == 0:
            print(json.dumps(dict(distill_leg(dev, steps=args.steps, warm=args.warmup), metric='distill_images_per_sec_batch512_visformer_s', n_gpus=1)), flush=True)
    else:
        eval_main(args, rank, world, dev)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
