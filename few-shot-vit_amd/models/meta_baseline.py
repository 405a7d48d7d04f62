"""'meta-baseline': cosine-prototype few-shot classifier over an encoder
(surface of test_phase/models/meta_baseline.py:10-47)."""
import torch
import torch.nn as nn

from .models import make as _make
from .models import register


@register('meta-baseline')
class MetaBaseline(nn.Module):

    def __init__(self, encoder, encoder_args={}, method='cos', temp=10., temp_learnable=True):
        super().__init__()
        self.encoder = _make(encoder, **encoder_args)
        self.method = method
        if temp_learnable:
            self.temp = nn.Parameter(torch.tensor(temp))
        else:
            self.temp = temp

    def forward(self, x_shot, x_query):
        """x_shot [E,way,shot,C,H,W], x_query [E,way*query,C,H,W] -> logits [E,way*query,way].
        Dim 0 stays the episode axis.  Eval mode runs encoder + head through the HIP engine in one
        C-ABI call (fsvit_meta_baseline_forward)."""
        if self.method not in ('cos', 'sqr'):
            raise ValueError(self.method)
        if x_shot.dim() != 6 or x_query.dim() != 5:
            raise ValueError('expected x_shot [E,way,shot,C,H,W] and x_query [E,Q,C,H,W]')
        if self.training:
            return self._forward_train(x_shot, x_query)
        return self._forward_eval(x_shot, x_query)

    def _forward_train(self, x_shot, x_query, label=None):
        """The meta-tuning step's forward (train_meta.py:167): one encoder pass over shot + query images of all
        episodes (so BatchNorm sees the whole batch, meta_baseline.py:31), then the differentiable head ('cos' or 'sqr')."""
        if not hasattr(self.encoder, 'trainer'):
            raise NotImplementedError('fsvit: the training path is built for the Visformer and ViT / DeiT encoders')
        from ..autograd import ProtoHeadFn
        E, way, shot = x_shot.shape[:3]
        Q = x_query.shape[1]
        img = x_shot.shape[-3:]
        x_tot = self.encoder(torch.cat([x_shot.reshape(-1, *img), x_query.reshape(-1, *img)], dim=0))
        if isinstance(x_tot, tuple):                       # distillation-phase encoder: `_, x_tot = self.encoder(...)` (sun_meta_training/models/meta_baseline.py:31)
            x_tot = x_tot[1]
        n_shot = E * way * shot
        f_shot = x_tot[:n_shot].view(E, way, shot, -1)
        f_query = x_tot[n_shot:].view(E, Q, -1)
        temp = self.temp if isinstance(self.temp, torch.Tensor) else torch.tensor(float(self.temp))      # (a host scalar: passed by value)
        if label is not None:
            from ..autograd import ProtoHeadCEFn
            return ProtoHeadCEFn.apply(f_shot, f_query, temp, label, self.method)
        return ProtoHeadFn.apply(f_shot, f_query, temp, self.method)

    def forward_loss(self, x_shot, x_query, label):
        """Training only: (loss, acc, logits) of train_meta.py:167-169 - `logits = model(x_shot, x_query).view(-1, n_way)`, `F.cross_entropy(logits, label)`,
        `utils.compute_acc(logits, label)` - with head, loss and accuracy in one launch (fsvit_proto_head_ce).  loss / acc are 0-d device tensors."""
        if not self.training:
            raise RuntimeError('forward_loss is the training-step entry; eval mode returns logits from forward()')
        if x_shot.dim() != 6 or x_query.dim() != 5:
            raise ValueError('expected x_shot [E,way,shot,C,H,W] and x_query [E,Q,C,H,W]')
        if self.method not in ('cos', 'sqr'):
            raise ValueError(self.method)
        return self._forward_train(x_shot, x_query, label)

    def _forward_eval(self, x_shot, x_query):
        engine = self.encoder.engine()
        temp = float(self.temp.detach()) if isinstance(self.temp, torch.Tensor) else float(self.temp)
        return engine.meta_baseline_forward(x_shot, x_query, temp, self.method)
