"""ORACLE (test infrastructure, NOT product code).

CPU fp32 restatement of the reference's Visformer encoder + MetaBaseline head, written
as plain functional torch over a state-dict (no nn.Module, no reference imports).  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function here against
golden vectors produced by importing the reference itself in the build container
(`tests/golden/make_golden.py`, recipe of SURVEY.md Appendix B): per-module intermediates
of a tiny Visformer, full-size `visformer_micro_80` logits / pooled features, and the
BN-calibration running statistics.

Each function cites the reference lines (relative to /root/reference/) it restates.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F


@dataclass
class VisformerCfg:
    """Constructor arguments of the reference `Visformer` that matter on the hot path
    (test_phase/models/visformer.py:292-295).  Defaults = `visformer_micro_80`
    (visformer.py:482-487)."""
    img_size: int = 80
    init_channels: int = 64
    embed_dim: int = 256
    depth: tuple = (4, 2, 3)
    num_heads: int = 6
    mlp_ratio: float = 4.0
    group: int = 8
    bn_eps: float = 1e-5       # visformer.py:121 (BatchNorm wrapper) and nn.BatchNorm2d default (:199)

    @property
    def out_dim(self):         # visformer.py:298
        return self.embed_dim * 2


def state_dict_shapes(cfg: VisformerCfg, prefix: str = '') -> Dict[str, tuple]:
    """Key names and shapes of `Visformer(...).state_dict()` for attn_stage='011',
    spatial_conv='100', norm_layer=BatchNorm (derived from
    visformer.py:202-217 (stem), :127-150 (Mlp), :166-178 (Attention), :266-281
    (PatchEmbed), :338-392 (stage lists))."""
    D = cfg.embed_dim
    C0, C1, C2, C3 = cfg.init_channels, D // 2, D, D * 2
    H1 = cfg.img_size // 4
    sh: Dict[str, tuple] = {}

    def bn(name, c):
        sh[name + '.weight'] = (c,)
        sh[name + '.bias'] = (c,)
        sh[name + '.running_mean'] = (c,)
        sh[name + '.running_var'] = (c,)
        sh[name + '.num_batches_tracked'] = ()

    sh['pos_embed1'] = (1, C1, H1, H1)
    sh['pos_embed2'] = (1, C2, H1 // 2, H1 // 2)
    sh['pos_embed3'] = (1, C3, H1 // 4, H1 // 4)
    sh['stem.conv1.weight'] = (C0, 3, 3, 3)
    bn('stem.bn1', C0)
    sh['stem.conv2.weight'] = (C1, C0, 3, 3)
    bn('stem.bn2', C1)
    sh['stem.conv3.weight'] = (C1, C1, 3, 3)
    bn('stem.bn3', C1)
    sh['stem.downsample.0.weight'] = (C1, 3, 3, 3)
    bn('stem.downsample.1', C1)
    for i in range(cfg.depth[0]):
        p = f'stage1.{i}.'
        hid = C1 * 2                                   # visformer.py:139-140 (group >= 2)
        bn(p + 'norm2.bn', C1)
        sh[p + 'mlp.conv1.weight'] = (hid, C1, 1, 1)
        sh[p + 'mlp.conv2.weight'] = (hid, hid // cfg.group, 3, 3)
        sh[p + 'mlp.conv3.weight'] = (C1, hid, 1, 1)
    for s, C, pe_in in ((2, C2, C1), (3, C3, C2)):
        sh[f'patch_embed{s}.proj.weight'] = (C, pe_in, 2, 2)
        sh[f'patch_embed{s}.proj.bias'] = (C,)
        bn(f'patch_embed{s}.norm.bn', C)
        hd = round(C // cfg.num_heads * 1.0)           # visformer.py:172
        for i in range(cfg.depth[s - 1]):
            p = f'stage{s}.{i}.'
            bn(p + 'norm1.bn', C)
            sh[p + 'attn.qkv.weight'] = (hd * cfg.num_heads * 3, C, 1, 1)
            sh[p + 'attn.proj.weight'] = (C, hd * cfg.num_heads, 1, 1)
            bn(p + 'norm2.bn', C)
            hid = int(C * cfg.mlp_ratio)               # visformer.py:256
            sh[p + 'mlp.conv1.weight'] = (hid, C, 1, 1)
            sh[p + 'mlp.conv3.weight'] = (C, hid, 1, 1)
    bn('norm.bn', C3)
    return {prefix + k: v for k, v in sh.items()}


class _BN:
    """Eval-mode / train-mode BatchNorm2d restatement (nn.BatchNorm2d semantics used at
    visformer.py:118-124 and :199).  In `calibrate` mode it behaves like train() with
    momentum=None on a freshly reset module after ONE batch: running_mean = batch mean,
    running_var = unbiased batch variance (torch cumulative-average rule with
    num_batches_tracked = 1), and normalises with the biased batch variance."""

    def __init__(self, sd, eps, mode, stats_out):
        self.sd, self.eps, self.mode, self.stats_out = sd, eps, mode, stats_out

    def __call__(self, x, name):
        w, b = self.sd[name + '.weight'], self.sd[name + '.bias']
        if self.mode == 'eval':
            return F.batch_norm(x, self.sd[name + '.running_mean'], self.sd[name + '.running_var'],
                                w, b, False, 0.0, self.eps)
        mean = x.mean(dim=(0, 2, 3))
        var_b = x.var(dim=(0, 2, 3), unbiased=False)
        if self.stats_out is not None:
            n = x.numel() // x.shape[1]
            if self.mode == 'train':      # nn.BatchNorm2d(momentum=0.1) running-stat update (visformer.py:121)
                m = 0.1
                self.stats_out[name + '.running_mean'] = ((1 - m) * self.sd[name + '.running_mean'] + m * mean).detach()
                self.stats_out[name + '.running_var'] = ((1 - m) * self.sd[name + '.running_var'] + m * var_b * (n / (n - 1))).detach()
            else:
                self.stats_out[name + '.running_mean'] = mean.detach().clone()
                self.stats_out[name + '.running_var'] = (var_b * (n / (n - 1))).detach()
        return F.batch_norm(x, None, None, w, b, True, 0.0, self.eps)


def attention(sd, x, p, num_heads, taps=None):
    """visformer.py:180-194.  qkv 1x1 conv (no bias) -> channel split 'b (x y z) h w ->
    x b y (h w) z' (channel index = x*heads*hd + y*hd + z) -> softmax((q k^T) * hd^-0.5) v
    -> 'b y (h w) z -> b (y z) h w' -> proj 1x1 conv (no bias)."""
    B, C, H, W = x.shape
    wq = sd[p + 'qkv.weight']
    hd = wq.shape[0] // (3 * num_heads)
    scale = hd ** -0.5                                  # visformer.py:174
    qkv = F.conv2d(x, wq)                               # [B, 3*heads*hd, H, W]
    qkv = qkv.reshape(B, 3, num_heads, hd, H * W).permute(1, 0, 2, 4, 3)   # x b y (hw) z
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * scale
    attn = attn.softmax(dim=-1)
    o = attn @ v                                        # [B, heads, HW, hd]
    o = o.permute(0, 1, 3, 2).reshape(B, num_heads * hd, H, W)              # b (y z) h w
    if taps is not None:
        taps[p + 'ctx'] = o
    return F.conv2d(o, sd[p + 'proj.weight'])


def mlp(sd, x, p, group):
    """visformer.py:152-163.  conv1 1x1 -> GELU(erf) [-> conv2 3x3 groups -> GELU] -> conv3 1x1.
    `spatial_conv` is inferred from the presence of conv2 in the state dict."""
    x = F.gelu(F.conv2d(x, sd[p + 'conv1.weight']))
    if (p + 'conv2.weight') in sd:
        x = F.gelu(F.conv2d(x, sd[p + 'conv2.weight'], padding=1, groups=group))
    return F.conv2d(x, sd[p + 'conv3.weight'])


def stem(sd, x, bn, p='stem.'):
    """ConvBlock, visformer.py:219-239."""
    out = F.leaky_relu(bn(F.conv2d(x, sd[p + 'conv1.weight'], stride=2, padding=1), p + 'bn1'), 0.1)
    out = F.leaky_relu(bn(F.conv2d(out, sd[p + 'conv2.weight'], padding=1), p + 'bn2'), 0.1)
    out = bn(F.conv2d(out, sd[p + 'conv3.weight'], padding=1), p + 'bn3')
    identity = bn(F.conv2d(x, sd[p + 'downsample.0.weight'], stride=2, padding=1), p + 'downsample.1')
    out = F.leaky_relu(out + identity, 0.1)
    return F.max_pool2d(out, 2)


def drop_path(x, rate, mode, masks_in=None, masks_out=None):
    """visformer.py:89-96."""
    if rate == 0.0 or mode != 'train':
        return x
    keep = 1.0 - rate
    if masks_in is not None:
        m = masks_in.pop(0).to(x.dtype).reshape(-1, 1, 1, 1)
    else:
        m = (keep + torch.rand((x.shape[0], 1, 1, 1), dtype=x.dtype)).floor_()
    if masks_out is not None:
        masks_out.append(m.reshape(-1).clone())
    return x.div(keep) * m


def visformer_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, cfg: VisformerCfg,
                      prefix: str = '', mode: str = 'eval',
                      stats_out: Optional[dict] = None, taps: Optional[dict] = None,
                      return_map: bool = False, drop_path_rate: float = 0.0,
                      droppath_masks: Optional[list] = None, masks_out: Optional[list] = None, freeze_bn: bool = False):
    """Visformer.forward, visformer.py:424-462 (eval: DropPath is identity, all Dropout p=0).

    sd      state dict (reference key names, optionally under `prefix`)
    x       [B,3,img,img] float32
    mode    'eval' (running stats), 'calibrate' (batch stats; fills stats_out with the batch statistics) or
            'train' (batch stats, stats_out receives the momentum-0.1 running-stat update, DropPath active)
    taps    optional dict receiving named intermediates (NCHW) for golden checks
    drop_path_rate / droppath_masks   train mode only: per-block rates linspace(0, rate, depth) (visformer.py:312);
            masks are drawn from the global torch RNG in the reference's call order (floor(keep + U[0,1)), :92-96)
            unless a list of [B] 0/1 tensors is supplied (one per DropPath call, in forward order).
    """
    if prefix:
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    assert x.shape[-1] == cfg.img_size and x.shape[-2] == cfg.img_size, \
        f"Input image size ({x.shape[-2]}*{x.shape[-1]}) does not match model ({cfg.img_size}*{cfg.img_size})."
    # freeze_bn: utils.freeze_bn inside a training step (train_meta.py:156-157) - BatchNorm2d modules in eval(), everything else in train mode
    bn = _BN(sd, cfg.bn_eps, 'eval' if (mode == 'eval' or freeze_bn) else mode, stats_out)

    def tap(name, t):
        if taps is not None:
            taps[name] = t

    dpr = torch.linspace(0, drop_path_rate, sum(cfg.depth)).tolist()  # :312
    masks_in = list(droppath_masks) if droppath_masks is not None else None
    blk = 0
    x = stem(sd, x, bn)                                              # :425-426
    tap('stem', x)
    x = x + sd['pos_embed1']                                         # :431
    for i in range(cfg.depth[0]):                                    # :433-434 ; Block.forward :259-263
        p = f'stage1.{i}.'
        x = x + drop_path(mlp(sd, bn(x, p + 'norm2.bn'), p + 'mlp.', cfg.group), dpr[blk], mode, masks_in, masks_out)
        blk += 1
        tap(p[:-1], x)
    for s in (2, 3):
        pe = f'patch_embed{s}.'
        x = F.conv2d(x, sd[pe + 'proj.weight'], sd[pe + 'proj.bias'], stride=2)   # :285
        x = bn(x, pe + 'norm.bn')                                    # :286-287
        tap(pe[:-1], x)
        x = x + sd[f'pos_embed{s}']                                  # :440 / :449
        for i in range(cfg.depth[s - 1]):
            p = f'stage{s}.{i}.'
            a = attention(sd, bn(x, p + 'norm1.bn'), p + 'attn.', cfg.num_heads, taps)
            tap(p + 'attn', a)
            x = x + drop_path(a, dpr[blk], mode, masks_in, masks_out)                                  # :261
            x = x + drop_path(mlp(sd, bn(x, p + 'norm2.bn'), p + 'mlp.', cfg.group), dpr[blk], mode, masks_in, masks_out)   # :262
            blk += 1
            tap(p[:-1], x)
    x = bn(x, 'norm.bn')                                             # :455
    tap('norm', x)
    pooled = x.mean(dim=(2, 3))                                      # AdaptiveAvgPool2d(1) :457 + view :462
    tap('pooled', pooled)
    if return_map:                                                   # sun_meta_training variant (:464)
        return x, pooled
    return pooled


def calibrate_bn(sd, x, cfg, prefix=''):
    """BN calibration of SURVEY.md 8c: one batch-statistics forward; returns a copy of `sd`
    with every running_mean / running_var replaced by that batch's statistics."""
    stats: dict = {}
    with torch.no_grad():
        visformer_forward(sd, x, cfg, prefix=prefix, mode='calibrate', stats_out=stats)
    out = dict(sd)
    for k, v in stats.items():
        out[prefix + k] = v
    return out


def compute_logits(feat, proto, metric='dot', temp=1.0):
    """utils/__init__.py:78-101 (2-D and 3-D forms)."""
    assert feat.dim() == proto.dim()
    if metric == 'cos':
        feat, proto, metric = F.normalize(feat, dim=-1), F.normalize(proto, dim=-1), 'dot'
    if metric == 'dot':
        logits = feat @ proto.transpose(-1, -2)
    elif metric == 'sqr':
        logits = -(feat.unsqueeze(-2) - proto.unsqueeze(-3)).pow(2).sum(dim=-1)
    else:
        raise ValueError(metric)
    return logits * temp


def meta_baseline_head(feat_shot, feat_query, method='cos', temp=10.0):
    """MetaBaseline.forward after the encoder call, meta_baseline.py:33-47.
    feat_shot [E,way,shot,D], feat_query [E,Q,D] -> logits [E,Q,way]."""
    proto = feat_shot.mean(dim=-2)
    if method == 'cos':
        proto = F.normalize(proto, dim=-1)          # eps 1e-12
        feat_query = F.normalize(feat_query, dim=-1)
        metric = 'dot'
    elif method == 'sqr':
        metric = 'sqr'
    else:
        raise ValueError(method)
    return compute_logits(feat_query, proto, metric=metric, temp=temp)


def meta_baseline_forward(sd, x_shot, x_query, cfg: VisformerCfg, method='cos', mode='eval', **fwd_kwargs):
    """MetaBaseline.forward, meta_baseline.py:24-47.  `sd` holds 'temp' and 'encoder.*'.
    mode='train' keeps the autograd graph (reference gradients = torch.autograd of this restatement)."""
    shot_shape, query_shape = x_shot.shape[:-3], x_query.shape[:-3]
    img_shape = x_shot.shape[-3:]
    xs = x_shot.reshape(-1, *img_shape)
    xq = x_query.reshape(-1, *img_shape)
    with torch.set_grad_enabled(mode == 'train'):
        tot = visformer_forward(sd, torch.cat([xs, xq], dim=0), cfg, prefix='encoder.', mode=mode, **fwd_kwargs)
        fs, fq = tot[:len(xs)], tot[-len(xq):]
        fs = fs.reshape(*shot_shape, -1)
        fq = fq.reshape(*query_shape, -1)
        temp = sd['temp'] if 'temp' in sd else 10.0
        if mode != 'train':
            temp = float(temp)
        return meta_baseline_head(fs, fq, method=method, temp=temp)
