"""DeiT / VisionTransformer encoders with the reference's constructor / state-dict surface
(test_phase/models/deit.py:139-357) and an MI355X-native forward (engine.VitEngine, libfsvit.so).

As for the Visformer, the nn.Module tree only owns parameters under the reference's key names
(`cls_token`, `pos_embed`, `patch_embed.proj.*`, `blocks.N.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}.*`,
`norm.*` - SURVEY.md Appendix A); no arithmetic happens here: eval runs on engine.VitEngine, model.train() on engine.VitTrainer
(fsvit_vit_train_forward / _backward: LayerNorm / attention / Mlp with saved activations, DropPath, every parameter gradient)."""
import torch
import torch.nn as nn

from .models import register


class _Attention(nn.Module):
    def __init__(self, dim, qkv_bias):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _Block(nn.Module):
    def __init__(self, dim, mlp_ratio, qkv_bias, eps, drop_path):
        super().__init__()
        self.drop_path_rate = drop_path
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = _Attention(dim, qkv_bias)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


class _PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, embed_dim):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(3, embed_dim, kernel_size=patch_size, stride=patch_size)


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4., qkv_bias=True,
                 drop_path_rate=0., ln_eps=1e-6, numerics=None, **unused):
        super().__init__()
        if not qkv_bias:
            raise NotImplementedError('fsvit builds the qkv_bias=True DeiT factories')
        self.cfg = dict(img_size=img_size, patch_size=patch_size, embed_dim=embed_dim, depth=depth, num_heads=num_heads,
                        mlp_ratio=mlp_ratio, ln_eps=ln_eps)
        self.numerics = numerics
        self.img_size = img_size
        self.num_features = self.out_dim = self.embed_dim = embed_dim       # deit.py:147
        self.patch_embed = _PatchEmbed(img_size, patch_size, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        dpr = torch.linspace(0, drop_path_rate, depth).tolist()
        self.blocks = nn.ModuleList([_Block(embed_dim, mlp_ratio, qkv_bias, ln_eps, dpr[i]) for i in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=ln_eps)
        nn.init.trunc_normal_(self.pos_embed, std=.02)                       # deit.py:176-189
        nn.init.trunc_normal_(self.cls_token, std=.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        self.drop_path_rate = float(drop_path_rate)
        self._engine = None
        self._engine_key = None
        self._trainer = None

    def engine(self):
        from ..engine import VitEngine
        dev = self.pos_embed.device
        if dev.type != 'cuda':
            raise RuntimeError('fsvit: the encoder lives on %s; the HIP engine needs an MI355X (no CPU fallback)' % dev)
        from ..engine import weights_fingerprint
        key = (weights_fingerprint(self), self.numerics, str(dev))
        if self._engine is None or self._engine_key != key:
            self._engine = VitEngine(self.cfg, self.state_dict(), numerics=self.numerics, device=dev)
            self._engine_key = key
        return self._engine

    def trainer(self):
        from ..engine import VitTrainer
        dev = self.pos_embed.device
        if dev.type != 'cuda':
            raise RuntimeError('fsvit: the encoder lives on %s; the HIP trainer needs an MI355X (no CPU fallback)' % dev)
        if self._trainer is None or self._trainer.device != dev:
            self._trainer = VitTrainer(self.cfg, numerics=self.numerics, device=dev)
        return self._trainer

    def draw_droppath_masks(self, n_img, device):
        """timm DropPath (deit.py:70,76-77): floor(keep_prob + U[0,1)) per sample, drawn per block in forward order (attention branch, then Mlp)."""
        masks = []
        for r in torch.linspace(0, self.drop_path_rate, self.cfg['depth']).tolist():
            if r > 0:
                for _ in range(2):
                    masks.append(torch.floor((1.0 - r) + torch.rand(n_img, device=device)))
        return torch.stack(masks) if masks else None

    def forward(self, x, droppath_masks=None):
        """[B,3,img,img] fp32 -> [B,embed_dim] = norm(tokens)[:, 0] (deit.py:196-213).  eval: packed engine; train: LayerNorm / attention / Mlp with
        saved activations, DropPath, and a backward through the HIP trainer (`droppath_masks` overrides the random draws)."""
        assert x.shape[-2] == self.img_size and x.shape[-1] == self.img_size, \
            f"Input image size ({x.shape[-2]}*{x.shape[-1]}) doesn't match model ({self.img_size}*{self.img_size})."
        if not self.training:
            return self.engine().forward(x)
        from ..autograd import VisformerTrainFn
        named = [(k, p) for k, p in self.named_parameters()]
        names = tuple(k for k, _ in named)
        masks = droppath_masks if droppath_masks is not None else self.draw_droppath_masks(x.shape[0], x.device)
        self.trainer().grad_sink = getattr(self, '_grad_sink', None)       # parallel.GradBucket: gradients land in the flat all-reduce buffer
        return VisformerTrainFn.apply(x, self.trainer(), names, {}, self.drop_path_rate, masks, *[p for _, p in named])


def _factory(name, **fixed):
    @register(name)
    def make(pretrained=False, **kwargs):
        if pretrained:
            raise NotImplementedError('pretrained DeiT weights are loaded through load_state_dict (no network here)')
        return VisionTransformer(**fixed, **kwargs)
    make.__name__ = name
    return make


deit_tiny_patch16_224 = _factory('deit_tiny_patch16_224', patch_size=16, embed_dim=192, depth=12, num_heads=3)
deit_small_patch16_224 = _factory('deit_small_patch16_224', patch_size=16, embed_dim=384, depth=12, num_heads=6)
deit_base_patch16_224 = _factory('deit_base_patch16_224', patch_size=16, embed_dim=768, depth=12, num_heads=12)
deit_nano_patch16_224 = _factory('deit_nano_patch16_224', patch_size=16, embed_dim=224, depth=12, num_heads=4)
deit_nano_patch6_84 = _factory('deit_nano_patch6_84', img_size=84, patch_size=6, embed_dim=224, depth=12, num_heads=4)
deit_micro_patch6_84 = _factory('deit_micro_patch6_84', img_size=84, patch_size=6, embed_dim=272, depth=12, num_heads=4)
