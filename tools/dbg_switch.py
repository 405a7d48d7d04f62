import os, subprocess, sys, torch
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import importlib.util
spec = importlib.util.spec_from_file_location('t', '/root/repo/tests/test_gpu_unfused_paths.py'); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
def run(env, tag):
    out = f'/tmp/{tag}.pt'
    r = subprocess.run([sys.executable, '-c', t.TRAIN_CHILD, out], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env), cwd='/root/repo')
    assert r.returncode == 0, r.stderr[-2000:]
    return torch.load(out)
base = run({}, 'base')
for env in ({'FSVIT_STAGE1_BLOCK_FUSED': '0'}, {'FSVIT_STAGE1_TRAIN_FUSED': '0'}, {'FSVIT_BN_PRODUCER_STATS': '0'}):
    o = run(env, 'o')
    d = sorted(((float((o[k] - v).norm() / (v.norm() + 1e-12)), k) for k, v in base.items() if float(v.norm()) > 1e-5), reverse=True)[:6]
    print(env, [(f'{a:.2e}', k) for a, k in d])
