"""`linear-classifier` and the `token-label` distillation model of the SUN meta-training phase
(sun_meta_training/models/classifier.py:27-34, models/token_label.py:36-60), same registry names, constructor arguments and
state-dict keys (`classifier.linear.{weight,bias}`, `classifier_local.linear.{weight,bias}`, `encoder.*`).

The Linear layers run on the HIP kernels behind `fsvit_linear_forward / _backward` (autograd.LinearFn); CPU tensors raise
(no fallback).  `TokenLabelOffline.forward` needs an encoder that returns `(feature map [B, D, H, W], pooled [B, D])` as the
reference's sun_meta_training Visformer does (models/visformer.py:464): any module with that contract can be passed as
`encoder=<module>`; wiring the HIP trainer's pre-pool map out through `visformer_micro_80` is the remaining step of SURVEY 8f.2
(DESIGN.md 7)."""
import math

import torch
import torch.nn as nn

from .models import make, register
from ..autograd import LinearFn, SoftTargetCEFn
from ..engine import ops


@register('linear-classifier')
class LinearClassifier(nn.Module):
    """classifier.py:27-34.  Parameters are initialised as nn.Linear does (kaiming_uniform(a=sqrt 5) / uniform bias)."""

    def __init__(self, in_dim, n_classes):
        super().__init__()
        self.linear = nn.Linear(in_dim, n_classes)

    def forward(self, x):
        if x.device.type != 'cuda':
            raise RuntimeError('fsvit: LinearClassifier needs cuda tensors (no CPU fallback)')
        return LinearFn.apply(x, self.linear.weight, self.linear.bias)


@register('nn-classifier')
class NNClassifier(nn.Module):
    """test_phase/models/classifier.py:38-55: logits = utils.compute_logits(x, proto, metric, temp) with learnable prototypes
    [n_classes, in_dim] (and a learnable temperature, initial 10, for the default 'cos' metric).  'cos' = row normalisation of both
    operands (fsvit_row_normalize) + the HIP Linear kernel; 'dot' = the Linear kernel alone; differentiable through both."""

    def __init__(self, in_dim, n_classes, metric='cos', temp=None):
        super().__init__()
        if metric not in ('cos', 'dot'):
            raise NotImplementedError("fsvit: nn-classifier is built for metric 'cos' (the reference default) and 'dot'")
        self.proto = nn.Parameter(torch.empty(n_classes, in_dim))
        nn.init.kaiming_uniform_(self.proto, a=math.sqrt(5))
        if temp is None:
            temp = nn.Parameter(torch.tensor(10.)) if metric == 'cos' else 1.0
        self.metric = metric
        self.temp = temp

    def forward(self, x):
        if x.device.type != 'cuda':
            raise RuntimeError('fsvit: NNClassifier needs cuda tensors (no CPU fallback)')
        from ..autograd import RowNormalizeFn
        p = self.proto
        if self.metric == 'cos':
            x, p = RowNormalizeFn.apply(x), RowNormalizeFn.apply(p)
        return LinearFn.apply(x, p, None) * self.temp


@register('classifier')
class Classifier(nn.Module):
    """sun_meta_training/models/classifier.py:11-24: encoder (returning `(map, pooled)`) + classifier on the pooled feature."""

    def __init__(self, encoder, encoder_args, classifier, classifier_args):
        super().__init__()
        if isinstance(encoder, nn.Module):
            self.encoder = encoder
        else:
            self.encoder = make(encoder, **dict(encoder_args or {}, return_map=True))
        classifier_args = dict(classifier_args)
        classifier_args['in_dim'] = self.encoder.out_dim
        self.classifier = make(classifier, **classifier_args)

    def forward(self, x):
        _, x1 = self.encoder(x)
        return self.classifier(x1)


@register('token-label')
class TokenLabelOffline(nn.Module):
    """token_label.py:36-60: a global classifier on the pooled feature and on the teacher's tokens, a local classifier with one
    extra (background) class on the student's tokens."""

    def __init__(self, encoder, encoder_args, classifier, classifier_args):
        super().__init__()
        self.encoder = encoder if isinstance(encoder, nn.Module) else make(encoder, **encoder_args)
        classifier_args = dict(classifier_args)
        classifier_args['in_dim'] = self.encoder.out_dim
        local_args = {'in_dim': self.encoder.out_dim, 'n_classes': int(classifier_args['n_classes'] + 1)}
        self.classifier = make(classifier, **classifier_args)
        self.classifier_local = make(classifier, **local_args)

    def forward(self, x, is_teacher=False):
        fmap, x1 = self.encoder(x)
        x_reshape = fmap.permute(0, 2, 3, 1)                                     # [B, H, W, D]
        y_reshape = self.classifier(x_reshape) if is_teacher else self.classifier_local(x_reshape)
        y_token = y_reshape.permute(0, 3, 1, 2)                                  # [B, C(+1), H, W] as the reference returns it
        y = self.classifier(x1)
        return y_token, y, x1


def generate_softlabel(logits, smoothing=0.1, k=3, bp=10):
    """offline.py:57-76 on the device: teacher token logits [B, C, H, W] (the reference's layout) -> soft labels [B*H*W, C+1]."""
    B, Cc = logits.shape[:2]
    lt = logits.permute(0, 2, 3, 1).reshape(B, -1, Cc)
    return ops.token_softlabel(lt, k=k, bp=bp, smoothing=smoothing)


class SoftTargetCrossEntropy(nn.Module):
    """offline.py:34-45."""

    def forward(self, x, target):
        return SoftTargetCEFn.apply(x, target)


class FsvitAdamW(torch.optim.Optimizer):
    """AdamW(betas, eps, lr, weight_decay) as the distillation phase builds it (offline.py:232-233; timm's AdamW == decoupled weight
    decay), the update done by the HIP kernel behind fsvit_adamw_step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            b1, b2 = group['betas']
            batches = {}                     # update number -> tensors at that number (one launch each: all parameters of a run share it)
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st['step'] += 1
                data = p.data if p.data.dim() > 0 else p.data.view(1)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if data.is_contiguous() and data.dtype == torch.float32:
                    b = batches.setdefault(st['step'], ([], [], [], []))
                    b[0].append(data); b[1].append(g); b[2].append(st['exp_avg']); b[3].append(st['exp_avg_sq'])
                else:
                    ops.adamw_step(data, g.view(-1), st['exp_avg'].view(-1), st['exp_avg_sq'].view(-1), group['lr'], b1, b2,
                                   group['eps'], group['weight_decay'], st['step'])
            for step_no, (ps, gs, ms, vs) in batches.items():
                ops.adamw_step_multi(ps, gs, ms, vs, group['lr'], b1, b2, group['eps'], group['weight_decay'], step_no)
