#!/usr/bin/env python3
"""Prototype (CPU, rounding-point oracle): calibration-free bias correction of the 16-bit weight rounding.

    b'[n] = b[n] - sum_k (round16(W')[n][k] - W'[n][k]) * E[a_k]

E[a_k] comes from the checkpoint alone: BatchNorm running means where a BatchNorm sees the operand (exact), otherwise a Gaussian
prior for the activation's mean refined by ridge least squares against the running mean of the next BatchNorm downstream.
Compares against the true means measured on data."""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from fewshot_vit_amd import synthetic  # noqa: E402
from fewshot_vit_amd.utils import few_shot as fs  # noqa: E402
from oracle import visformer_emul as ve  # noqa: E402
from oracle import visformer_oracle as vo  # noqa: E402

# Gauss-Hermite nodes for E[f(N(mu, sigma^2))]
_GH_X, _GH_W = np.polynomial.hermite_e.hermegauss(64)
_GH_W = _GH_W / _GH_W.sum()


def gauss_mean(f, mu, sig):
    z = mu[:, None] + sig[:, None] * _GH_X[None, :]
    v = f(z)
    m = (v * _GH_W).sum(1)
    m2 = (v * v * _GH_W).sum(1)
    return m, np.maximum(m2 - m * m, 0.0)


def gelu(z):
    return 0.5 * z * (1.0 + np.vectorize(math.erf)(z / math.sqrt(2.0)))


def lrelu(z):
    return np.where(z > 0, z, 0.1 * z)


def ridge(A, r, prior, lam_rel=1e-3):
    """argmin |A m - r|^2 + lam |m - prior|^2"""
    A = np.asarray(A, np.float64)
    d = r - A @ prior
    n, k = A.shape
    lam = lam_rel * np.trace(A @ A.T) / n
    if n <= k:
        return prior + A.T @ np.linalg.solve(A @ A.T + lam * np.eye(n), d)
    return prior + np.linalg.solve(A.T @ A + lam * np.eye(k), A.T @ d)


def tap_fraction(H, stride, k=3, pad=1):
    """fraction of output positions for which tap (ky, kx) lies inside the (H x H) input"""
    Ho = (H + 2 * pad - k) // stride + 1
    f1 = np.zeros(k)
    for t in range(k):
        pos = np.arange(Ho) * stride - pad + t
        f1[t] = ((pos >= 0) & (pos < H)).mean()
    return f1[:, None] * f1[None, :]


def estimate_means(sd, cfg, lam=1e-3, use_ls=True):
    """-> dict layer -> mean of the layer's GEMM operand, shaped like the conv weight's [Cin(,kh,kw)]"""
    g = lambda k: sd[k].double().numpy()
    eps = cfg.bn_eps
    M = {}

    def bn(p):
        s = g(p + '.weight') / np.sqrt(g(p + '.running_var') + eps)
        return s, g(p + '.bias'), g(p + '.running_mean')          # scale, beta, running mean

    img = cfg.img_size
    # ---- stem
    f_s2 = tap_fraction(img, 2)
    w1, wd = g('stem.conv1.weight'), g('stem.downsample.0.weight')
    A = np.concatenate([(w1 * f_s2).sum((2, 3)), (wd * f_s2).sum((2, 3))], 0)
    r = np.concatenate([bn('stem.bn1')[2], bn('stem.downsample.1')[2]])
    m_img = ridge(A, r, np.zeros(3), lam) if use_ls else np.zeros(3)
    M['stem.conv1'] = M['stem.downsample'] = m_img
    f_s1 = tap_fraction(img // 2, 1)
    s1, b1, _ = bn('stem.bn1')
    pri, _ = gauss_mean(lrelu, b1, np.abs(g('stem.bn1.weight')))
    w2 = g('stem.conv2.weight')
    M['stem.conv2'] = ridge((w2 * f_s1).sum((2, 3)), bn('stem.bn2')[2], pri, lam) if use_ls else pri
    pri, _ = gauss_mean(lrelu, bn('stem.bn2')[1], np.abs(g('stem.bn2.weight')))
    w3 = g('stem.conv3.weight')
    M['stem.conv3'] = ridge((w3 * f_s1).sum((2, 3)), bn('stem.bn3')[2], pri, lam) if use_ls else pri

    # mean of the residual stream leaving a stage: solved from the following PatchEmbed's BatchNorm
    def pe_input_mean(s, prior):
        p = f'patch_embed{s}.'
        w = g(p + 'proj.weight')
        r = bn(p + 'norm.bn')[2] - g(p + 'proj.bias')
        return ridge(w.sum((2, 3)), r, prior, lam) if use_ls else prior

    # ---- stage 1
    n1 = cfg.depth[0]
    rm = [bn(f'stage1.{i}.norm2.bn')[2] for i in range(n1)]
    h2_pri = []
    for i in range(n1):
        p = f'stage1.{i}.'
        s, beta, mean = bn(p + 'norm2.bn')
        M[p + 'mlp.conv1'] = mean
        W1 = g(p + 'mlp.conv1.weight')[:, :, 0, 0]
        gam = np.abs(g(p + 'norm2.bn.weight'))
        mu1, sg1 = W1 @ beta, np.sqrt((W1 ** 2) @ (gam ** 2))
        m_h1, v_h1 = gauss_mean(gelu, mu1, sg1)
        M[p + 'mlp.conv2'] = m_h1
        W2 = g(p + 'mlp.conv2.weight')                         # [hid][Cg][3][3]
        Cg = W2.shape[1]
        f20 = tap_fraction(img // 4, 1)
        hid = W2.shape[0]
        mu2 = np.zeros(hid); var2 = np.zeros(hid)
        for n in range(hid):
            grp = n // (hid // cfg.group)
            mm = m_h1[grp * Cg:(grp + 1) * Cg]; vv = v_h1[grp * Cg:(grp + 1) * Cg]
            mu2[n] = ((W2[n] * f20).sum((1, 2)) * mm).sum()
            var2[n] = (((W2[n] ** 2) * f20).sum((1, 2)) * vv).sum()
        m_h2, _ = gauss_mean(gelu, mu2, np.sqrt(var2))
        h2_pri.append(m_h2)
    x1_out = pe_input_mean(2, rm[-1] + g(f'stage1.{n1 - 1}.mlp.conv3.weight')[:, :, 0, 0] @ h2_pri[-1])
    for i in range(n1):
        p = f'stage1.{i}.'
        W3 = g(p + 'mlp.conv3.weight')[:, :, 0, 0]
        nxt = rm[i + 1] if i + 1 < n1 else x1_out
        M[p + 'mlp.conv3'] = ridge(W3, nxt - rm[i], h2_pri[i], lam) if use_ls else h2_pri[i]
    M['patch_embed2.proj'] = x1_out

    # ---- stages 2, 3
    heads = cfg.num_heads
    final_rm = bn('norm.bn')[2]
    x_out = {}
    for s in (3, 2):
        nb = cfg.depth[s - 1]
        pri_h, pri_ctx, r1, r2 = [], [], [], []
        for i in range(nb):
            p = f'stage{s}.{i}.'
            s1_, beta1, mean1 = bn(p + 'norm1.bn')
            s2_, beta2, mean2 = bn(p + 'norm2.bn')
            r1.append(mean1); r2.append(mean2)
            M[p + 'attn.qkv'] = mean1
            M[p + 'mlp.conv1'] = mean2
            Wq = g(p + 'attn.qkv.weight')[:, :, 0, 0]
            nv = Wq.shape[0] // 3
            pri_ctx.append(Wq[2 * nv:] @ beta1)
            W1 = g(p + 'mlp.conv1.weight')[:, :, 0, 0]
            gam = np.abs(g(p + 'norm2.bn.weight'))
            mh, _ = gauss_mean(gelu, W1 @ beta2, np.sqrt((W1 ** 2) @ (gam ** 2)))
            pri_h.append(mh)
        if s == 3:
            out = final_rm
        else:
            W3 = g(f'stage2.{nb - 1}.mlp.conv3.weight')[:, :, 0, 0]
            out = pe_input_mean(3, r2[-1] + W3 @ pri_h[-1])
            M['patch_embed3.proj'] = out
        for i in range(nb):
            p = f'stage{s}.{i}.'
            Wp = g(p + 'attn.proj.weight')[:, :, 0, 0]
            M[p + 'attn.proj'] = ridge(Wp, r2[i] - r1[i], pri_ctx[i], lam) if use_ls else pri_ctx[i]
            W3 = g(p + 'mlp.conv3.weight')[:, :, 0, 0]
            nxt = r1[i + 1] if i + 1 < nb else out
            M[p + 'mlp.conv3'] = ridge(W3, nxt - r2[i], pri_h[i], lam) if use_ls else pri_h[i]
    return M


def true_means(sd, x, cfg):
    """operand means measured on data with the fp32 oracle's arithmetic (emulator with every rounding site off)"""
    rec = {}
    ve.SKIP = set(['input', 'w_stem', 'act_stem', 'w_s1', 'act_s1', 'w_pe', 'w_attn', 'qkv', 'P', 'ctx', 'w_mlp', 'act_mlp', 'xop', 'w'])
    ve.RECORD = rec
    sdp = {k[len('encoder.'):]: v for k, v in sd.items() if k.startswith('encoder.')}
    with torch.no_grad():
        ve.visformer_forward_emul(sdp, x, cfg, residual='fp32')
    ve.RECORD = None
    ve.SKIP = set()
    return rec


def main():
    cfg = vo.VisformerCfg()
    shapes = vo.state_dict_shapes(cfg, prefix='encoder.')
    shapes['temp'] = ()
    sd = synthetic.synthetic_checkpoint_sd(shapes)
    z = np.load(os.path.join(REPO, 'tests', 'golden', 'full_visformer_micro_80.npz'))
    x = synthetic.synthetic_episodes(11, 1, 5, 5, 15)
    xs, xq = fs.split_shot_query(x, 5, 5, 15, 1)
    ref = torch.from_numpy(z['logits_5shot'])
    sdp = {k[len('encoder.'):]: v for k, v in sd.items() if k.startswith('encoder.')}

    def run(means, storage=torch.bfloat16):
        ve.MEANS = means
        lg = ve.meta_baseline_forward_emul(sd, xs, xq, cfg, residual='bf16', storage=storage)
        ve.MEANS = None
        return '%.3e max  %.3e mean' % ((lg - ref).abs().max().item(), (lg - ref).abs().mean().item())

    print('no correction                 ', run(None))
    # true means from a different batch of data (calibration episode)
    xc = synthetic.synthetic_episodes(5, 1, 5, 5, 15)
    tm = true_means(sd, xc, cfg)
    print('true means (calibration batch)', run(tm))
    for lam in (1e-3, 1e-2):
        em = estimate_means(sdp, cfg, lam=lam)
        print('estimated, ridge lam %.0e    ' % lam, run(em))
    em0 = estimate_means(sdp, cfg, use_ls=False)
    print('estimated, priors only        ', run(em0))
    em = estimate_means(sdp, cfg, lam=1e-3)
    for k in sorted(tm):
        t = tm[k]
        t = t if t.ndim == 1 else t
        e = em.get(k)
        if e is None:
            print('%-28s no estimate' % k); continue
        e = np.asarray(e)
        tt = t.reshape(len(e), -1).mean(1) if t.size != e.size else t.reshape(-1)
        print('%-28s |true| %.3e  |est-true| %.3e  |prior-true| %.3e' % (k, np.linalg.norm(tt), np.linalg.norm(e - tt),
                                                                       np.linalg.norm(np.asarray(em0[k]) - tt)))
    print('f16 no correction             ', run(None, torch.float16))
    print('f16 estimated                 ', run(em, torch.float16))


if __name__ == '__main__':
    main()
