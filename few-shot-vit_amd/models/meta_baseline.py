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
        if self.training:
            raise NotImplementedError(
                'fsvit: the meta-training (backward) path is not built yet; call model.eval() '
                '(train-mode BatchNorm statistics and gradients are scheduled after the eval path)')
        if x_shot.dim() != 6 or x_query.dim() != 5:
            raise ValueError('expected x_shot [E,way,shot,C,H,W] and x_query [E,Q,C,H,W]')
        engine = self.encoder.engine()
        temp = float(self.temp.detach()) if isinstance(self.temp, torch.Tensor) else float(self.temp)
        return engine.meta_baseline_forward(x_shot, x_query, temp, self.method)
