"""ORACLE (test infrastructure, NOT product code).

CPU fp32 functional restatement of the reference's DeiT / VisionTransformer eval forward
(test_phase/models/deit.py:14-218) over a state dict.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this.

Parity status: PINNED by tests/test_oracle_golden.py::test_deit_* against vectors captured from the
imported reference (tests/golden/make_golden.py; timm symbols stubbed as in SURVEY.md Appendix B —
none of them touches eval-mode arithmetic).
"""
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F


@dataclass
class DeitCfg:
    """VisionTransformer constructor arguments (deit.py:142-144); defaults = deit_small_patch16_224 (:245-250)."""
    img_size: int = 224
    patch_size: int = 16
    embed_dim: int = 384
    depth: int = 12
    num_heads: int = 6
    mlp_ratio: float = 4.0
    ln_eps: float = 1e-6          # partial(nn.LayerNorm, eps=1e-6) in every factory

    @property
    def out_dim(self):            # deit.py:147
        return self.embed_dim

    @property
    def num_patches(self):
        return (self.img_size // self.patch_size) ** 2


FACTORIES = {                     # deit.py:220-357
    'deit_tiny_patch16_224': DeitCfg(224, 16, 192, 12, 3),
    'deit_small_patch16_224': DeitCfg(224, 16, 384, 12, 6),
    'deit_base_patch16_224': DeitCfg(224, 16, 768, 12, 12),
    'deit_nano_patch16_224': DeitCfg(224, 16, 224, 12, 4),
    'deit_nano_patch6_84': DeitCfg(84, 6, 224, 12, 4),
    'deit_micro_patch6_84': DeitCfg(84, 6, 272, 12, 4),
}


def state_dict_shapes(cfg: DeitCfg, prefix: str = '') -> Dict[str, tuple]:
    """SURVEY.md Appendix A (DeiT part)."""
    D, hid, p = cfg.embed_dim, int(cfg.embed_dim * cfg.mlp_ratio), cfg.patch_size
    sh = {'cls_token': (1, 1, D), 'pos_embed': (1, cfg.num_patches + 1, D),
          'patch_embed.proj.weight': (D, 3, p, p), 'patch_embed.proj.bias': (D,),
          'norm.weight': (D,), 'norm.bias': (D,)}
    for i in range(cfg.depth):
        b = f'blocks.{i}.'
        sh.update({b + 'norm1.weight': (D,), b + 'norm1.bias': (D,), b + 'norm2.weight': (D,), b + 'norm2.bias': (D,),
                   b + 'attn.qkv.weight': (3 * D, D), b + 'attn.qkv.bias': (3 * D,),
                   b + 'attn.proj.weight': (D, D), b + 'attn.proj.bias': (D,),
                   b + 'mlp.fc1.weight': (hid, D), b + 'mlp.fc1.bias': (hid,),
                   b + 'mlp.fc2.weight': (D, hid), b + 'mlp.fc2.bias': (D,)})
    return {prefix + k: v for k, v in sh.items()}


def attention(sd, x, p, num_heads):
    """deit.py:46-58."""
    B, N, C = x.shape
    hd = C // num_heads
    qkv = F.linear(x, sd[p + 'qkv.weight'], sd[p + 'qkv.bias']).reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(x, sd[p + 'proj.weight'], sd[p + 'proj.bias'])


def deit_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, cfg: DeitCfg, prefix: str = '', taps: Optional[dict] = None,
                 drop_path_rate: float = 0.0, droppath_masks: Optional[list] = None):
    """VisionTransformer.forward_features, deit.py:196-213 (eval: dropout / DropPath are identity).
    Train mode = drop_path_rate > 0 with `droppath_masks`: per block i with rate r_i = linspace(0, rate, depth)[i] > 0 (deit.py:161) two [B]
    masks of 0 / 1 (attention branch, then Mlp), the branch is scaled by mask / (1 - r_i) (timm DropPath, deit.py:70,76-77); dropout rates are 0
    in every factory.  Under torch.enable_grad() the autograd of this function is the reference gradient."""
    if prefix:
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    assert x.shape[-2] == cfg.img_size and x.shape[-1] == cfg.img_size, \
        f"Input image size ({x.shape[-2]}*{x.shape[-1]}) doesn't match model ({cfg.img_size}*{cfg.img_size})."
    B, D = x.shape[0], cfg.embed_dim
    x = F.conv2d(x, sd['patch_embed.proj.weight'], sd['patch_embed.proj.bias'], stride=cfg.patch_size)   # :99
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat((sd['cls_token'].expand(B, -1, -1), x), dim=1) + sd['pos_embed']                        # :200-202
    if taps is not None:
        taps['embed'] = x
    rates = torch.linspace(0, drop_path_rate, cfg.depth).tolist()
    mi = 0

    def dp(branch, r):
        nonlocal mi
        if r == 0.0 or droppath_masks is None:
            return branch
        m = droppath_masks[mi].to(branch.dtype).view(-1, 1, 1) / (1.0 - r)
        mi += 1
        return branch * m

    for i in range(cfg.depth):                                                                            # Block.forward :75-78
        b = f'blocks.{i}.'
        x = x + dp(attention(sd, F.layer_norm(x, (D,), sd[b + 'norm1.weight'], sd[b + 'norm1.bias'], cfg.ln_eps), b + 'attn.', cfg.num_heads), rates[i])
        h = F.linear(F.layer_norm(x, (D,), sd[b + 'norm2.weight'], sd[b + 'norm2.bias'], cfg.ln_eps), sd[b + 'mlp.fc1.weight'], sd[b + 'mlp.fc1.bias'])
        x = x + dp(F.linear(F.gelu(h), sd[b + 'mlp.fc2.weight'], sd[b + 'mlp.fc2.bias']), rates[i])
        if taps is not None:
            taps[f'blocks.{i}'] = x
    x = F.layer_norm(x, (D,), sd['norm.weight'], sd['norm.bias'], cfg.ln_eps)                             # :212
    return x[:, 0]                                                                                        # :213
