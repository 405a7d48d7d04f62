"""Visformer encoder with the reference's constructor / state-dict surface
(test_phase/models/visformer.py:291-487) and an MI355X-native forward.

The nn.Module tree below exists to own parameters under the reference's key names
(`stem.conv1.weight`, `stage1.N.norm2.bn.*`, `stage2.N.attn.qkv.weight`,
`patch_embed2.proj.{weight,bias}`, `pos_embed{1,2,3}`, ...), so published checkpoints load with
`load_state_dict(strict=True)`.  No arithmetic happens in these modules: `forward` hands the
whole encoder to the packed HIP engine (`engine.VisformerEngine`, libfsvit.so).
"""
import math

import torch
import torch.nn as nn

from .models import register


class _BatchNorm(nn.Module):
    """Key-compatible holder for the reference's BatchNorm wrapper (`.bn`), visformer.py:118-124."""

    def __init__(self, dim):
        super().__init__()
        self.bn = nn.BatchNorm2d(dim, eps=1e-5, momentum=0.1, track_running_stats=True)


class _Mlp(nn.Module):
    """conv1 1x1 -> GELU [-> conv2 3x3 grouped -> GELU] -> conv3 1x1 (visformer.py:127-163)."""

    def __init__(self, in_features, hidden_features, group, spatial_conv):
        super().__init__()
        if spatial_conv:
            hidden_features = in_features * 5 // 6 if group < 2 else in_features * 2
        self.conv1 = nn.Conv2d(in_features, hidden_features, 1, bias=False)
        if spatial_conv:
            self.conv2 = nn.Conv2d(hidden_features, hidden_features, 3, padding=1, groups=group, bias=False)
        self.conv3 = nn.Conv2d(hidden_features, in_features, 1, bias=False)


class _Attention(nn.Module):
    """qkv / proj 1x1 convs without bias (visformer.py:166-178)."""

    def __init__(self, dim, num_heads, head_dim_ratio):
        super().__init__()
        self.head_dim = round(dim // num_heads * head_dim_ratio)
        self.qkv = nn.Conv2d(dim, self.head_dim * num_heads * 3, 1, bias=False)
        self.proj = nn.Conv2d(self.head_dim * num_heads, dim, 1, bias=False)


class _Block(nn.Module):
    def __init__(self, dim, num_heads, head_dim_ratio, mlp_ratio, group, attn_disabled, spatial_conv, drop_path):
        super().__init__()
        self.drop_path_rate = drop_path
        if not attn_disabled:
            self.norm1 = _BatchNorm(dim)
            self.attn = _Attention(dim, num_heads, head_dim_ratio)
        self.norm2 = _BatchNorm(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio), group, spatial_conv)


class _Stem(nn.Module):
    """ConvBlock parameters (visformer.py:202-217)."""

    def __init__(self, inplanes, hidden_planes, planes):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, hidden_planes, 3, stride=2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(hidden_planes)
        self.conv2 = nn.Conv2d(hidden_planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes)
        self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 3, stride=2, padding=1, bias=False),
                                        nn.BatchNorm2d(planes))


class _PatchEmbed(nn.Module):
    def __init__(self, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=2, stride=2)
        self.norm = _BatchNorm(embed_dim)


class Visformer(nn.Module):
    """Supported family: stem + BatchNorm + attn_stage='011' + spatial_conv='100' (every shipped
    Visformer factory; visformer.py:466-487).  Other combinations raise at construction."""

    def __init__(self, img_size=80, init_channels=64, embed_dim=256, depth=(4, 2, 3), num_heads=6, mlp_ratio=4.,
                 group=8, drop_path_rate=0., attn_stage='011', spatial_conv='100', numerics=None, return_map=False, **unused):
        super().__init__()
        if attn_stage != '011' or spatial_conv != '100' or init_channels is None:
            raise NotImplementedError('fsvit builds the stem + attn_stage=011 + spatial_conv=100 Visformer family')
        if isinstance(depth, int):
            d1 = d3 = depth // 3
            depth = (d1, depth - d1 - d3, d3)          # visformer.py:308-309
        self.cfg = dict(img_size=img_size, init_channels=init_channels, embed_dim=embed_dim, depth=tuple(depth),
                        num_heads=num_heads, mlp_ratio=mlp_ratio, group=group)
        self.numerics = numerics
        self.return_map = bool(return_map)           # distillation phase: forward returns (map [B,C,h,w], pooled) (sun_meta_training/models/visformer.py:464)
        self.img_size = img_size
        self.embed_dim = self.num_features = embed_dim
        self.out_dim = embed_dim * 2                   # visformer.py:298
        dpr = torch.linspace(0, drop_path_rate, sum(depth)).tolist()       # :312 (train-time DropPath rates)
        s = img_size // 4
        self.stem = _Stem(3, init_channels, embed_dim // 2)
        self.pos_embed1 = nn.Parameter(torch.zeros(1, embed_dim // 2, s, s))
        self.stage1 = nn.ModuleList([_Block(embed_dim // 2, num_heads, 0.5, mlp_ratio, group, True, True, dpr[i])
                                     for i in range(depth[0])])
        self.patch_embed2 = _PatchEmbed(embed_dim // 2, embed_dim)
        self.pos_embed2 = nn.Parameter(torch.zeros(1, embed_dim, s // 2, s // 2))
        self.stage2 = nn.ModuleList([_Block(embed_dim, num_heads, 1.0, mlp_ratio, group, False, False, dpr[depth[0] + i])
                                     for i in range(depth[1])])
        self.patch_embed3 = _PatchEmbed(embed_dim, embed_dim * 2)
        self.pos_embed3 = nn.Parameter(torch.zeros(1, embed_dim * 2, s // 4, s // 4))
        self.stage3 = nn.ModuleList([_Block(embed_dim * 2, num_heads, 1.0, mlp_ratio, group, False, False,
                                            dpr[depth[0] + depth[1] + i]) for i in range(depth[2])])
        self.norm = _BatchNorm(embed_dim * 2)
        self._init_weights()
        self.drop_path_rate = float(drop_path_rate)
        self._engine = None
        self._engine_key = None
        self._trainer = None

    def _init_weights(self):
        """conv_init=True initialisation of the shipped factories (visformer.py:395-422)."""
        for p in (self.pos_embed1, self.pos_embed2, self.pos_embed3):
            nn.init.trunc_normal_(p, std=0.02)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0.)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0.)

    # ------------------------------------------------------------------ engine management
    def _fingerprint(self):
        from ..engine import weights_fingerprint
        return weights_fingerprint(self)

    def engine(self):
        """Packed HIP engine for the current weights (re-packed when any tensor changed)."""
        from ..engine import VisformerEngine
        dev = self.pos_embed1.device
        if dev.type != 'cuda':
            raise RuntimeError('fsvit: the encoder lives on %s; the HIP engine needs an MI355X (no CPU fallback)' % dev)
        key = (self._fingerprint(), self.numerics, str(dev))
        if self._engine is None or self._engine_key != key:
            self._engine = VisformerEngine(self.cfg, self.state_dict(), numerics=self.numerics, device=dev)
            self._engine_key = key
        return self._engine

    def trainer(self):
        from ..engine import VisformerTrainer
        dev = self.pos_embed1.device
        if dev.type != 'cuda':
            raise RuntimeError('fsvit: the encoder lives on %s; the HIP trainer needs an MI355X (no CPU fallback)' % dev)
        if self._trainer is None or self._trainer.device != dev:
            self._trainer = VisformerTrainer(self.cfg, numerics=self.numerics, device=dev)
        return self._trainer

    def draw_droppath_masks(self, n_img, device):
        """The Bernoulli draws of every DropPath call with a non-zero rate, in call order:
        floor(keep_prob + rand(B)) (visformer.py:93-95)."""
        n = self.trainer().n_droppath_calls(self.drop_path_rate)
        if n == 0:
            return None
        depth = self.cfg['depth']
        key = (str(device), self.drop_path_rate)
        if getattr(self, '_keep_key', None) != key:          # keep-probabilities per DropPath call, uploaded once (an upload per step is a stream synchronisation)
            rates = torch.linspace(0, self.drop_path_rate, sum(depth)).tolist()
            keep = [1.0 - r for b, r in enumerate(rates) if r > 0 for _ in range(1 if b < depth[0] else 2)]
            self._keep_dev = torch.tensor(keep, dtype=torch.float32).to(device).unsqueeze(1)
            self._keep_key = key
        # floor(keep_prob + rand(B)) of every call from ONE generator launch: the draws are i.i.d. uniforms either way, and the device generator's stream
        # is not the reference's (a CUDA Philox stream cannot be reproduced here); one launch + two in-place passes instead of 16 + 3 (70 us per step)
        return torch.rand(n, n_img, device=device).add_(self._keep_dev).floor_()

    def forward(self, x, droppath_masks=None):
        """[B,3,img,img] fp32 -> [B,out_dim] pooled features (visformer.py:424-462).
        eval: packed engine (BN folded).  train: batch-statistics BN (running stats updated in place), DropPath, and a
        backward through the HIP trainer; `droppath_masks` overrides the random draws (tests)."""
        assert x.shape[-2] == self.img_size and x.shape[-1] == self.img_size, \
            f"Input image size ({x.shape[-2]}*{x.shape[-1]}) does not match model ({self.img_size}*{self.img_size})."
        hw = self.img_size // 16
        if not self.training:
            feat = self.engine().forward(x)
            if not self.return_map:
                return feat
            tok = self.engine().last_tokens(x.shape[0], hw * hw)
            return tok.permute(0, 2, 1).reshape(x.shape[0], self.out_dim, hw, hw), feat
        bn_modes = {m.training for m in self.modules() if isinstance(m, nn.BatchNorm2d)}
        if len(bn_modes) > 1:
            raise NotImplementedError('fsvit: BatchNorm layers must be all in train mode or all frozen (utils.freeze_bn) inside a training step')
        frozen = bn_modes == {False}                    # utils.freeze_bn (train_meta.py:156-157): running statistics normalise, nothing is updated
        self.trainer().set_freeze_bn(frozen)
        from ..autograd import VisformerTrainFn
        named = [(k, p) for k, p in self.named_parameters()]
        names = tuple(k for k, _ in named)
        buffers = {k: b for k, b in self.named_buffers() if not k.endswith('num_batches_tracked')}
        masks = droppath_masks if droppath_masks is not None else self.draw_droppath_masks(x.shape[0], x.device)
        self.trainer().grad_sink = getattr(self, '_grad_sink', None)       # parallel.GradBucket: gradients land in the flat all-reduce buffer
        if self.return_map:
            from ..autograd import VisformerTrainMapFn
            tok, feat = VisformerTrainMapFn.apply(x, self.trainer(), names, buffers, self.drop_path_rate, masks, hw * hw, *[p for _, p in named])
        else:
            feat = VisformerTrainFn.apply(x, self.trainer(), names, buffers, self.drop_path_rate, masks, *[p for _, p in named])
        if not frozen:      # nn.BatchNorm2d counts its train-mode forwards (visformer.py:53-64 via torch): one multi-tensor add for the 21 counters
            torch._foreach_add_([b for k, b in self.named_buffers() if k.endswith('num_batches_tracked')], 1)
        if self.return_map:
            return tok.permute(0, 2, 1).reshape(x.shape[0], self.out_dim, hw, hw), feat
        return feat


@register('visformer_micro_80')
def visformer_small_80(**kwargs):
    """'Visformer-S' of the paper = registry name visformer_micro_80 (visformer.py:482-487)."""
    return Visformer(img_size=80, init_channels=64, embed_dim=256, depth=[4, 2, 3], num_heads=6, mlp_ratio=4., group=8,
                     attn_stage='011', spatial_conv='100', **kwargs)
