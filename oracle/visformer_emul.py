"""ORACLE (test infrastructure, NOT product code): the Visformer eval path with the HIP kernels' ROUNDING POINTS.

`visformer_oracle.py` is the fp32 restatement of the reference (pinned to goldens).  This file re-runs the same
arithmetic the way the bf16 throughput mode of the HIP engine does it, so that the end-to-end bf16 test can compare the
kernels with an expectation that differs from them only by accumulation order (tolerance ~1e-2 of the logit scale
instead of the 0.25 "anything goes" bound of round 1), and so that the effect of a storage decision (residual stream in
bf16 / bf16 hi+lo / fp32) on the logits can be measured on the CPU before a kernel is touched:

  * eval BatchNorm is FOLDED in fp64 into the neighbouring conv (post-conv BN -> row scale + bias; pre-norm BN before a
    1x1 conv -> column scale + bias[n] = sum_c W[n][c] t[c]), then the weights are rounded to bf16 (csrc/engine.hip
    `build`, restating test_phase/models/visformer.py:118-124,202-239,259-263);
  * every conv / GEMM multiplies bf16 operands and accumulates in fp32; bias, activation, residual adds, softmax, pooling
    are fp32; GELU is the engine's `gelu_sig` (csrc/fsvit_common.h), max |d| 2.6e-5 against erf - except in the stage-1 block of the bf16
    mode, whose GELUs are a table look-up of the exact erf form on the bf16-rounded pre-activation (`gelu_s1`, round 6);
  * activations that a kernel STORES are rounded to bf16 where the engine stores them: im2col patches, stem c1 / c2,
    the two hidden maps of a stage-1 block, q / k / v, the un-normalised softmax numerators P, ctx, the Mlp hidden map,
    and the block input x1 = x + proj(ctx) as the Mlp's GEMM operand;
  * the residual stream x1 / x2 / x3 (`residual=`): 'bf16' = stored as one bf16 value (round 1), 'hilo' = stored as
    bf16 hi + bf16 lo (hi is what every consumer GEMM reads, hi + lo is what residual adds and the pooling read),
    'fp32' = never rounded (GEMM operands still are).

Only `tests/` and tools may import this.  Reference lines restated: visformer.py:127-163 (Mlp), :166-194 (Attention),
:202-239 (ConvBlock), :266-288 (PatchEmbed), :424-462 (forward); the algebra is visformer_oracle.py's.
"""
from typing import Dict

import torch
import torch.nn.functional as F

from .visformer_oracle import VisformerCfg, meta_baseline_head


SKIP = set()       # analysis only (tools/emul_ablation.py): rounding sites left in fp32
STORAGE = torch.bfloat16     # the engine's 16-bit storage / MFMA operand type: torch.bfloat16 ('bf16' mode) or torch.float16 ('f16' mode)


def bf(t: torch.Tensor, site: str = '') -> torch.Tensor:
    """Round to the 16-bit storage type (nearest even) and back: the value a store keeps / an MFMA operand carries."""
    if site and site in SKIP:
        return t
    return t.to(STORAGE).to(torch.float32)


def gelu_sig(x: torch.Tensor) -> torch.Tensor:
    """csrc/fsvit_common.h gelu_sig: x * sigmoid(x (c0 + c1 x^2 + c2 x^4)), coefficients carry -log2(e)."""
    u = torch.clamp(x * x, max=64.0)
    p = 1.0153755e-3 * u - 1.0678257e-1
    p = p * u - 2.3011138
    return x / (1.0 + torch.exp2(x * p))


S1_GELU_TABLE = True     # round 6: the bf16 kernels with GELU micro-stages (csrc/stage1_w4.hip, csrc/mlp_rows.hip) evaluate them as a table look-up on the bf16-ROUNDED pre-activation


def gelu_s1(x: torch.Tensor, site: str = 'act_s1') -> torch.Tensor:
    """The GELU of the stage-1 block and of the stage-2 / 3 Mlps as the engine computes it.  bf16 storage (csrc/fsvit_common.h gelu_tab: stage1_w4.hip,
    mlp_rows.hip): the pre-activation is rounded to bf16, its magnitude clamped to the table's range [2^-10, 32) and the exact erf form
    (visformer.py:146-163's nn.GELU) looked up; fp16 storage (and the geometries that run the general kernels) keep gelu_sig on the fp32 pre-activation."""
    if not S1_GELU_TABLE or STORAGE != torch.bfloat16 or site in SKIP:
        return gelu_sig(x)
    xb = x.to(torch.bfloat16).to(torch.float32)
    a = xb.abs().clamp(min=2.0 ** -10, max=32.0 * (1.0 - 2.0 ** -8))
    xb = torch.where(torch.signbit(xb), -a, a)
    return F.gelu(xb.double()).float()


class _Stream:
    """The residual stream under one of the three storage models."""

    def __init__(self, kind: str):
        assert kind in ('bf16', 'hilo', 'fp32')
        self.kind = kind

    def store(self, x):
        """-> (operand, full): what consumer GEMMs read, what residual adds / pooling read."""
        if self.kind == 'bf16':
            h = bf(x)
            return h, h
        if self.kind == 'hilo':
            h = bf(x)
            return bf(x, 'xop'), h + bf(x - h)
        return bf(x, 'xop'), x


def _fold_post(w, sd, bn, eps):
    s = sd[bn + '.weight'].double() / torch.sqrt(sd[bn + '.running_var'].double() + eps)
    t = sd[bn + '.bias'].double() - sd[bn + '.running_mean'].double() * s
    return (w.double() * s.view(-1, 1, 1, 1)), t


def _fold_pre(w, sd, bn, eps):
    s = sd[bn + '.weight'].double() / torch.sqrt(sd[bn + '.running_var'].double() + eps)
    t = sd[bn + '.bias'].double() - sd[bn + '.running_mean'].double() * s
    w2 = w.double()[:, :, 0, 0]
    return (w2 * s.view(1, -1)).view(*w.shape), (w2 @ t)


def _w(wd, site='w'):
    return bf(wd.float(), site)


WROUND = True      # the engine's weight-rounding bias correction (csrc/engine.hip `WRound`): on in the 16-bit modes
MEANS = None       # analysis only (tools/wround_proto.py): operand means that replace operand_means()
RECORD = None      # analysis only: dict that receives the measured per-channel operand means


def _tapfrac(H, stride, k, pad):
    """[k][k] fraction of the output positions whose tap (ky, kx) lies inside the H x H input (zero padding otherwise)"""
    Ho = (H + 2 * pad - k) // stride + 1
    f1 = torch.zeros(k, dtype=torch.float64)
    for t in range(k):
        pos = torch.arange(Ho) * stride - pad + t
        f1[t] = ((pos >= 0) & (pos < H)).double().mean()
    return f1.view(k, 1) * f1.view(1, k)


def _gauss_moments(f, mu, sig):
    """E[f(y)], Var[f(y)] for y ~ N(mu, sig^2) per channel: trapezoid rule on z in [-8, 8], 257 nodes (engine.hip gauss_moments)"""
    z = torch.linspace(-8.0, 8.0, 257, dtype=torch.float64)
    w = torch.exp(-0.5 * z * z)
    w = w / w.sum()
    v = f(mu.view(-1, 1) + sig.view(-1, 1) * z.view(1, -1))
    m = (v * w).sum(1)
    return m, ((v * v * w).sum(1) - m * m).clamp(min=0.0)


def _gelu64(z):
    return 0.5 * z * (1.0 + torch.erf(z * 0.7071067811865476))


def _lrelu_mean(mu, sig):
    """E[LeakyReLU_0.1(N(mu, sig^2))] = mu (0.1 + 0.9 Phi(mu/sig)) + 0.9 sig phi(mu/sig)"""
    t = mu / sig
    Phi = 0.5 * (1.0 + torch.erf(t * 0.7071067811865476))
    phi = torch.exp(-0.5 * t * t) * 0.3989422804014327
    return mu * (0.1 + 0.9 * Phi) + 0.9 * sig * phi


def operand_means(sd, cfg):
    """E[a_c] of every GEMM operand, from the checkpoint alone (restates csrc/engine.hip `WRound`): the running mean of the BatchNorm
    that sees the operand (pre-norm 1x1 convs: exact), 0 for the normalised image, and elsewhere the mean of the activation of a
    Gaussian pre-activation whose moments follow from the preceding BatchNorm's (beta, gamma) under channel independence."""
    g = lambda k: sd[k].double()
    eps = cfg.bn_eps
    M = {}
    M['stem.conv1'] = M['stem.downsample'] = torch.zeros(3, dtype=torch.float64)
    M['stem.conv2'] = _lrelu_mean(g('stem.bn1.bias'), g('stem.bn1.weight').abs())
    M['stem.conv3'] = _lrelu_mean(g('stem.bn2.bias'), g('stem.bn2.weight').abs())
    H1 = cfg.img_size // 4
    f = _tapfrac(H1, 1, 3, 1)
    x_out = None
    for i in range(cfg.depth[0]):
        p = f'stage1.{i}.'
        beta, gam, rm = g(p + 'norm2.bn.bias'), g(p + 'norm2.bn.weight').abs(), g(p + 'norm2.bn.running_mean')
        M[p + 'mlp.conv1'] = rm
        W1 = g(p + 'mlp.conv1.weight')[:, :, 0, 0]
        m1, v1 = _gauss_moments(_gelu64, W1 @ beta, torch.sqrt((W1 * W1) @ (gam * gam)))
        W2 = g(p + 'mlp.conv2.weight')
        hid, Cg = W2.shape[0], W2.shape[1]
        grp = torch.arange(hid) // (hid // cfg.group)
        mg, vg = m1.view(cfg.group, Cg)[grp], v1.view(cfg.group, Cg)[grp]
        mu2 = ((W2 * f).sum(dim=(2, 3)) * mg).sum(1)
        var2 = ((W2 * W2 * f).sum(dim=(2, 3)) * vg).sum(1)
        m2, _ = _gauss_moments(_gelu64, mu2, torch.sqrt(var2))
        M[p + 'mlp.conv3'] = m2
        x_out = rm + g(p + 'mlp.conv3.weight')[:, :, 0, 0] @ m2
    for s in (2, 3):
        M[f'patch_embed{s}.proj'] = x_out
        for i in range(cfg.depth[s - 1]):
            p = f'stage{s}.{i}.'
            M[p + 'attn.qkv'] = g(p + 'norm1.bn.running_mean')
            Wq = g(p + 'attn.qkv.weight')[:, :, 0, 0]
            M[p + 'attn.proj'] = Wq[2 * (Wq.shape[0] // 3):] @ g(p + 'norm1.bn.bias')        # E[ctx] ~ E[v] = W_v beta_1
            rm2 = g(p + 'norm2.bn.running_mean')
            M[p + 'mlp.conv1'] = rm2
            W1 = g(p + 'mlp.conv1.weight')[:, :, 0, 0]
            gam = g(p + 'norm2.bn.weight').abs()
            mh, _ = _gauss_moments(_gelu64, W1 @ g(p + 'norm2.bn.bias'), torch.sqrt((W1 * W1) @ (gam * gam)))
            M[p + 'mlp.conv3'] = mh
            x_out = rm2 + g(p + 'mlp.conv3.weight')[:, :, 0, 0] @ mh
    return M


def _corr(means, name, x, wd, site, stride=1, pad=0, groups=1):
    """fp64 [N]: the bias delta that cancels the MEAN effect of rounding the folded weights `wd` ([N][Cg][k][k]):
    -sum_k (round16(w) - w)[n][k] f_tap E[a_c]; zeros when the correction is off or the site is not rounded."""
    if RECORD is not None:
        RECORD[name] = x.double().mean(dim=(0, 2, 3)).numpy()
    w = wd.double()
    N, Cg, k = w.shape[0], w.shape[1], w.shape[-1]
    if means is None or site in SKIP or name not in means:
        return torch.zeros(N, dtype=torch.float64)
    m = torch.as_tensor(means[name], dtype=torch.float64)
    dw = w.float().to(STORAGE).double() - w                        # rounding error of each weight
    f = _tapfrac(x.shape[-1], stride, k, pad) if k > 1 and pad > 0 else torch.ones(k, k, dtype=torch.float64)
    mg = m.view(groups, Cg)[torch.arange(N) // (N // groups)]      # [N][Cg]: the channels output n reads
    return -((dw * f.view(1, 1, k, k)).sum(dim=(2, 3)) * mg).sum(dim=1)


def _b(v):
    return v.float().view(1, -1, 1, 1)


def visformer_forward_emul(sd: Dict[str, torch.Tensor], x: torch.Tensor, cfg: VisformerCfg, prefix: str = '',
                           residual: str = 'hilo', taps: dict = None) -> torch.Tensor:
    """[B,3,img,img] fp32 -> pooled [B, out_dim] with the bf16 engine's rounding points."""
    if prefix:
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    eps = cfg.bn_eps
    st = _Stream(residual)
    heads = cfg.num_heads
    means = MEANS if MEANS is not None else (operand_means(sd, cfg) if WROUND else None)

    def tap(name, t):
        if taps is not None:
            taps[name] = t

    # ---- stem (visformer.py:219-239): patches bf16; conv1 -> c1 bf16; conv2 -> c2 bf16; conv3 + downsample + LeakyReLU + pool + pos1
    xb = bf(x, 'input')
    w1, t1 = _fold_post(sd['stem.conv1.weight'], sd, 'stem.bn1', eps)
    c1 = bf(F.leaky_relu(F.conv2d(xb, _w(w1, 'w_stem'), stride=2, padding=1) + _b(t1 + _corr(means, 'stem.conv1', xb, w1, 'w_stem', 2, 1)), 0.1), 'act_stem')
    w2, t2 = _fold_post(sd['stem.conv2.weight'], sd, 'stem.bn2', eps)
    c2 = bf(F.leaky_relu(F.conv2d(c1, _w(w2, 'w_stem'), padding=1) + _b(t2 + _corr(means, 'stem.conv2', c1, w2, 'w_stem', 1, 1)), 0.1), 'act_stem')
    w3, t3 = _fold_post(sd['stem.conv3.weight'], sd, 'stem.bn3', eps)
    wd, td = _fold_post(sd['stem.downsample.0.weight'], sd, 'stem.downsample.1', eps)
    out = F.conv2d(c2, _w(w3, 'w_stem'), padding=1) + F.conv2d(xb, _w(wd, 'w_stem'), stride=2, padding=1) \
        + _b(t3 + td + _corr(means, 'stem.conv3', c2, w3, 'w_stem', 1, 1) + _corr(means, 'stem.downsample', xb, wd, 'w_stem', 2, 1))
    out = F.max_pool2d(F.leaky_relu(out, 0.1), 2) + sd['pos_embed1']
    xop, xfull = st.store(out)
    tap('stem', xfull)

    # Corrections of the bias-free convs that feed the residual stream (conv3 / proj of a block) are per-channel constants of the
    # stream: the engine does not add them but carries them as `cst` into the bias of every later consumer (W' cst) and into the pooled
    # feature.  The grouped 3x3 conv of a stage-1 Mlp has no bias and a GELU behind it: not corrected.
    cst = torch.zeros(xfull.shape[1], dtype=torch.float64)
    # ---- stage 1 (Block :259-263 with Mlp :152-163, spatial conv)
    for i in range(cfg.depth[0]):
        p = f'stage1.{i}.'
        wa, ba = _fold_pre(sd[p + 'mlp.conv1.weight'], sd, p + 'norm2.bn', eps)
        ba = ba + _corr(means, p + 'mlp.conv1', xop, wa, 'w_s1') + wa[:, :, 0, 0] @ cst
        h1 = bf(gelu_s1(F.conv2d(xop, _w(wa, 'w_s1')) + _b(ba)), 'act_s1')
        h2 = bf(gelu_s1(F.conv2d(h1, bf(sd[p + 'mlp.conv2.weight'], 'w_s1'), padding=1, groups=cfg.group)), 'act_s1')
        cst = cst + _corr(means, p + 'mlp.conv3', h2, sd[p + 'mlp.conv3.weight'], 'w_s1')
        y = xfull + F.conv2d(h2, bf(sd[p + 'mlp.conv3.weight'], 'w_s1'))
        xop, xfull = st.store(y)
        tap(p[:-1], xfull)

    # ---- stages 2, 3
    for s in (2, 3):
        pe = f'patch_embed{s}.'
        wp, tp = _fold_post(sd[pe + 'proj.weight'], sd, pe + 'norm.bn', eps)
        sbn = sd[pe + 'norm.bn.weight'].double() / torch.sqrt(sd[pe + 'norm.bn.running_var'].double() + eps)
        bias = sbn * sd[pe + 'proj.bias'].double() + tp + _corr(means, pe + 'proj', xop, wp, 'w_pe', 2, 0) + wp.sum(dim=(2, 3)) @ cst
        cst = torch.zeros(wp.shape[0], dtype=torch.float64)
        y = F.conv2d(xop, _w(wp, 'w_pe'), stride=2) + _b(bias) + sd[f'pos_embed{s}']
        xop, xfull = st.store(y)
        tap(pe[:-1], xfull)
        B, C, H, W = xfull.shape
        S = H * W
        for i in range(cfg.depth[s - 1]):
            p = f'stage{s}.{i}.'
            wq, bq = _fold_pre(sd[p + 'attn.qkv.weight'], sd, p + 'norm1.bn', eps)
            hd = wq.shape[0] // (3 * heads)
            bq = bq + _corr(means, p + 'attn.qkv', xop, wq, 'w_attn') + wq[:, :, 0, 0] @ cst
            qkv = bf(F.conv2d(xop, _w(wq, 'w_attn')) + _b(bq), 'qkv')
            qkv = qkv.reshape(B, 3, heads, hd, S).permute(1, 0, 2, 4, 3)
            q, k, v = qkv[0], qkv[1], qkv[2]
            sc = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
            e = torch.exp(sc - sc.amax(dim=-1, keepdim=True))
            o = (bf(e, 'P') @ v) / e.sum(dim=-1, keepdim=True)
            ctx = bf(o.permute(0, 1, 3, 2).reshape(B, heads * hd, H, W), 'ctx')
            cst = cst + _corr(means, p + 'attn.proj', ctx, sd[p + 'attn.proj.weight'], 'w_attn')
            x1 = xfull + F.conv2d(ctx, bf(sd[p + 'attn.proj.weight'], 'w_attn'))
            x1op, x1full = st.store(x1)
            wf, bfc = _fold_pre(sd[p + 'mlp.conv1.weight'], sd, p + 'norm2.bn', eps)
            bfc = bfc + _corr(means, p + 'mlp.conv1', x1op, wf, 'w_mlp') + wf[:, :, 0, 0] @ cst
            hid = bf(gelu_s1(F.conv2d(x1op, _w(wf, 'w_mlp')) + _b(bfc), 'act_mlp'), 'act_mlp')
            cst = cst + _corr(means, p + 'mlp.conv3', hid, sd[p + 'mlp.conv3.weight'], 'w_mlp')
            y = x1full + F.conv2d(hid, bf(sd[p + 'mlp.conv3.weight'], 'w_mlp'))
            xop, xfull = st.store(y)
            tap(p[:-1], xfull)

    sN = sd['norm.bn.weight'].double() / torch.sqrt(sd['norm.bn.running_var'].double() + eps)
    tN = sd['norm.bn.bias'].double() - sd['norm.bn.running_mean'].double() * sN
    return xfull.mean(dim=(2, 3)) * sN.float() + (tN + sN * cst).float()


def meta_baseline_forward_emul(sd, x_shot, x_query, cfg: VisformerCfg, method='cos', residual='hilo', storage=None):
    """MetaBaseline.forward (meta_baseline.py:24-47) on the emulated encoder; the head is fp32 in the engine too.
    storage: torch.bfloat16 / torch.float16 for this call (default: the module-level STORAGE)."""
    global STORAGE
    if storage is not None:
        keep, STORAGE = STORAGE, storage
        try:
            return meta_baseline_forward_emul(sd, x_shot, x_query, cfg, method=method, residual=residual)
        finally:
            STORAGE = keep
    img_shape = x_shot.shape[-3:]
    xs, xq = x_shot.reshape(-1, *img_shape), x_query.reshape(-1, *img_shape)
    with torch.no_grad():
        tot = visformer_forward_emul(sd, torch.cat([xs, xq], dim=0), cfg, prefix='encoder.', residual=residual)
    fs = tot[:len(xs)].reshape(*x_shot.shape[:-3], -1)
    fq = tot[-len(xq):].reshape(*x_query.shape[:-3], -1)
    return meta_baseline_head(fs, fq, method=method, temp=float(sd['temp']) if 'temp' in sd else 10.0)
