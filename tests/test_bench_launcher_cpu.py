"""`python bench.py --gpus N` invoked plainly: the parent starts one process per rank itself.  Here the rank
skeleton (barrier, statistics all-reduce, max-over-ranks time, rank 0 prints ONE JSON line) runs over gloo on the CPU; asking for
more GPUs than the node has must fail cleanly with a message instead of hanging or crashing."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, 'bench.py')


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    return env


def test_plain_invocation_launches_ranks_gloo():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '5', '--backend', 'gloo', '--selftest-launcher'],
                       capture_output=True, text=True, timeout=300, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                                   # rank 0's line only
    out = json.loads(lines[0])
    assert out['selftest'] and out['n_gpus'] == 2 and out['stat_n'] == 10.0       # 5 steps x 2 ranks went through the all-reduce
    assert out['ranks_seen'] == 2 and len(out['per_rank_eps']) == 2 and all(v > 0 for v in out['per_rank_eps'])     # what the collective saw, not WORLD_SIZE
    expect = sum(0.25 + 0.5 * ((rank * 31 + s * 7) % 11) / 11.0 for rank in range(2) for s in range(5))
    assert abs(out['stat_sum'] - expect) < 1e-12


def test_under_external_launcher_env_is_a_rank():
    """With RANK / WORLD_SIZE already set (torch.distributed.run) bench.py is a rank, not a launcher."""
    env = _clean_env()
    env.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29512')
    r = subprocess.run([sys.executable, BENCH, '--gpus', '1', '--steps', '3', '--backend', 'gloo', '--selftest-launcher'],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])['n_gpus'] == 1


def test_more_gpus_than_visible_fails_cleanly():
    import torch
    have = torch.cuda.device_count()
    r = subprocess.run([sys.executable, BENCH, '--gpus', str(have + 2), '--steps', '1'], capture_output=True, text=True, timeout=300, env=_clean_env())
    assert r.returncode == 2
    assert 'GPU(s) visible' in r.stderr and '{' not in r.stdout
