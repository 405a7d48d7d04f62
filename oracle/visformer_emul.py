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
    are fp32; GELU is the engine's `gelu_sig` (csrc/fsvit_common.h), max |d| 2.6e-5 against erf;
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


def visformer_forward_emul(sd: Dict[str, torch.Tensor], x: torch.Tensor, cfg: VisformerCfg, prefix: str = '',
                           residual: str = 'hilo', taps: dict = None) -> torch.Tensor:
    """[B,3,img,img] fp32 -> pooled [B, out_dim] with the bf16 engine's rounding points."""
    if prefix:
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    eps = cfg.bn_eps
    st = _Stream(residual)
    heads = cfg.num_heads

    def tap(name, t):
        if taps is not None:
            taps[name] = t

    # ---- stem (visformer.py:219-239): patches bf16; conv1 -> c1 bf16; conv2 -> c2 bf16; conv3 + downsample + LeakyReLU + pool + pos1
    xb = bf(x, 'input')
    w1, t1 = _fold_post(sd['stem.conv1.weight'], sd, 'stem.bn1', eps)
    c1 = bf(F.leaky_relu(F.conv2d(xb, _w(w1, 'w_stem'), stride=2, padding=1) + t1.float().view(1, -1, 1, 1), 0.1), 'act_stem')
    w2, t2 = _fold_post(sd['stem.conv2.weight'], sd, 'stem.bn2', eps)
    c2 = bf(F.leaky_relu(F.conv2d(c1, _w(w2, 'w_stem'), padding=1) + t2.float().view(1, -1, 1, 1), 0.1), 'act_stem')
    w3, t3 = _fold_post(sd['stem.conv3.weight'], sd, 'stem.bn3', eps)
    wd, td = _fold_post(sd['stem.downsample.0.weight'], sd, 'stem.downsample.1', eps)
    out = F.conv2d(c2, _w(w3, 'w_stem'), padding=1) + F.conv2d(xb, _w(wd, 'w_stem'), stride=2, padding=1) + (t3 + td).float().view(1, -1, 1, 1)
    out = F.max_pool2d(F.leaky_relu(out, 0.1), 2) + sd['pos_embed1']
    xop, xfull = st.store(out)
    tap('stem', xfull)

    # ---- stage 1 (Block :259-263 with Mlp :152-163, spatial conv)
    for i in range(cfg.depth[0]):
        p = f'stage1.{i}.'
        wa, ba = _fold_pre(sd[p + 'mlp.conv1.weight'], sd, p + 'norm2.bn', eps)
        h1 = bf(gelu_sig(F.conv2d(xop, _w(wa, 'w_s1')) + ba.float().view(1, -1, 1, 1)), 'act_s1')
        h2 = bf(gelu_sig(F.conv2d(h1, bf(sd[p + 'mlp.conv2.weight'], 'w_s1'), padding=1, groups=cfg.group)), 'act_s1')
        y = xfull + F.conv2d(h2, bf(sd[p + 'mlp.conv3.weight'], 'w_s1'))
        xop, xfull = st.store(y)
        tap(p[:-1], xfull)

    # ---- stages 2, 3
    for s in (2, 3):
        pe = f'patch_embed{s}.'
        wp, tp = _fold_post(sd[pe + 'proj.weight'], sd, pe + 'norm.bn', eps)
        sbn = sd[pe + 'norm.bn.weight'].double() / torch.sqrt(sd[pe + 'norm.bn.running_var'].double() + eps)
        bias = (sbn * sd[pe + 'proj.bias'].double() + tp).float()
        y = F.conv2d(xop, _w(wp, 'w_pe'), stride=2) + bias.view(1, -1, 1, 1) + sd[f'pos_embed{s}']
        xop, xfull = st.store(y)
        tap(pe[:-1], xfull)
        B, C, H, W = xfull.shape
        S = H * W
        for i in range(cfg.depth[s - 1]):
            p = f'stage{s}.{i}.'
            wq, bq = _fold_pre(sd[p + 'attn.qkv.weight'], sd, p + 'norm1.bn', eps)
            hd = wq.shape[0] // (3 * heads)
            qkv = bf(F.conv2d(xop, _w(wq, 'w_attn')) + bq.float().view(1, -1, 1, 1), 'qkv')
            qkv = qkv.reshape(B, 3, heads, hd, S).permute(1, 0, 2, 4, 3)
            q, k, v = qkv[0], qkv[1], qkv[2]
            sc = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
            e = torch.exp(sc - sc.amax(dim=-1, keepdim=True))
            o = (bf(e, 'P') @ v) / e.sum(dim=-1, keepdim=True)
            ctx = bf(o.permute(0, 1, 3, 2).reshape(B, heads * hd, H, W), 'ctx')
            x1 = xfull + F.conv2d(ctx, bf(sd[p + 'attn.proj.weight'], 'w_attn'))
            x1op, x1full = st.store(x1)
            wf, bfc = _fold_pre(sd[p + 'mlp.conv1.weight'], sd, p + 'norm2.bn', eps)
            hid = bf(gelu_sig(F.conv2d(x1op, _w(wf, 'w_mlp')) + bfc.float().view(1, -1, 1, 1)), 'act_mlp')
            y = x1full + F.conv2d(hid, bf(sd[p + 'mlp.conv3.weight'], 'w_mlp'))
            xop, xfull = st.store(y)
            tap(p[:-1], xfull)

    sN = sd['norm.bn.weight'].double() / torch.sqrt(sd['norm.bn.running_var'].double() + eps)
    tN = sd['norm.bn.bias'].double() - sd['norm.bn.running_mean'].double() * sN
    return xfull.mean(dim=(2, 3)) * sN.float() + tN.float()


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
