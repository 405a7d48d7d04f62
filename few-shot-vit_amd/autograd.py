"""torch.autograd bridges of the meta-tuning step (meta_tuning_sun_m/train_meta.py:161-177).

`loss.backward()` reaches the HIP trainer through two Functions: the train-mode encoder
(fsvit_visformer_train_forward / _backward) and the cosine prototype head
(fsvit_proto_head / fsvit_proto_head_backward).  Cross-entropy stays in the caller, as in the
reference loop.  Both Functions only move device pointers; there is no torch arithmetic fallback.
"""
import torch

from .engine import ops


def _check_generation(ctx):
    """The trainer keeps ONE set of saved activations: a second train-mode forward on the same encoder before this backward has
    overwritten them (separate shot / query passes, several micro-batches per backward) - fail loudly instead of returning gradients
    of the wrong batch."""
    if ctx.trainer.generation != ctx.generation:
        raise RuntimeError('fsvit: the encoder ran another train-mode forward before this backward; its saved activations are gone '
                           '(one forward per backward: encode cat([shot, query]) in one call as MetaBaseline.forward does)')


class VisformerTrainFn(torch.autograd.Function):
    """feat = encoder(x) in train mode.  `params` are passed as inputs so autograd routes their gradients.  With a gradient sink
    (`trainer.grad_sink`: name -> (parameter, view into parallel.GradBucket's flat buffer)) the trainer writes every gradient straight into
    the bucket, `.grad` is pointed at the view and autograd is handed None for those inputs.  That overwrites, so it is only taken when
    every `.grad` is None (the state `optimizer.zero_grad()` leaves: train_meta.py:168-170 zero_grad / backward / step); with a gradient
    already in place (a second backward before the step, a second train-mode encoder call in one graph, zero_grad(set_to_none=False)) the
    gradients are returned to autograd, which ACCUMULATES them - into the bucket views when `.grad` is one - exactly as without a sink."""

    @staticmethod
    def forward(ctx, x, trainer, names, buffers, drop_path_rate, masks, *params):
        tensors = dict(zip(names, params))
        tensors.update(buffers)
        feat = trainer.forward(tensors, x, drop_path_rate, masks)
        ctx.trainer, ctx.names, ctx.buffers, ctx.generation = trainer, names, buffers, trainer.generation
        ctx.sink = getattr(trainer, 'grad_sink', None)
        ctx.save_for_backward(*params)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        _check_generation(ctx)
        params = ctx.saved_tensors
        tensors = dict(zip(ctx.names, params))
        sink = ctx.sink
        if sink is not None and all(k in sink and sink[k][1].shape == tensors[k].shape and sink[k][0].grad is None for k in ctx.names):
            grads = {k: sink[k][1] for k in ctx.names}
            tensors.update(ctx.buffers)
            ctx.trainer.backward(tensors, grads, dfeat)
            for k in ctx.names:
                sink[k][0].grad = sink[k][1]
            return (None,) * (6 + len(ctx.names))
        grads = {k: torch.empty_like(v) for k, v in tensors.items()}
        tensors.update(ctx.buffers)
        ctx.trainer.backward(tensors, grads, dfeat)
        return (None, None, None, None, None, None) + tuple(grads[k] for k in ctx.names)


class VisformerTrainMapFn(torch.autograd.Function):
    """(tokens, feat) = encoder(x) in train mode for the distillation phase, whose encoder returns the post-norm map next to the
    pooled feature (sun_meta_training/models/visformer.py:464); tokens [B, T, D] token-major."""

    @staticmethod
    def forward(ctx, x, trainer, names, buffers, drop_path_rate, masks, n_tok, *params):
        tensors = dict(zip(names, params))
        tensors.update(buffers)
        feat = trainer.forward(tensors, x, drop_path_rate, masks)
        tokens = trainer.tokens(x.shape[0], n_tok)
        ctx.trainer, ctx.names, ctx.buffers, ctx.generation = trainer, names, buffers, trainer.generation
        ctx.save_for_backward(*params)
        return tokens, feat

    @staticmethod
    def backward(ctx, dtokens, dfeat):
        _check_generation(ctx)
        params = ctx.saved_tensors
        tensors = dict(zip(ctx.names, params))
        grads = {k: torch.empty_like(v) for k, v in tensors.items()}
        tensors.update(ctx.buffers)
        if dfeat is None:
            dfeat = torch.zeros(dtokens.shape[0], dtokens.shape[2], device=dtokens.device)
        ctx.trainer.backward(tensors, grads, dfeat, dtokens)
        return (None,) * 7 + tuple(grads[k] for k in ctx.names)


class ProtoHeadFn(torch.autograd.Function):
    """logits = temp * cos(query, mean_shot) or -temp * |query - mean_shot|^2 (meta_baseline.py:33-47, methods 'cos' / 'sqr')."""

    @staticmethod
    def forward(ctx, feat_shot, feat_query, temp, method='cos'):
        feat_shot, feat_query = feat_shot.contiguous(), feat_query.contiguous()
        logits, _, _ = ops.proto_head(feat_shot, feat_query, temp if temp.is_cuda else float(temp), method)
        ctx.save_for_backward(feat_shot, feat_query, temp)
        ctx.method = method
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        feat_shot, feat_query, temp = ctx.saved_tensors
        ds, dq, dt = ops.proto_head_backward(feat_shot, feat_query, dlogits, temp if temp.is_cuda else float(temp), ctx.method)
        return ds, dq, dt.reshape(temp.shape), None


class ProtoHeadCEFn(torch.autograd.Function):
    """Head + mean cross entropy + accuracy of the meta-tuning step (train_meta.py:167-169) as ONE launch each way: returns (loss, acc, logits); only
    the loss carries a gradient.  The forward already forms dlogits; the backward multiplies with the upstream gradient inside the head's kernel."""

    @staticmethod
    def forward(ctx, feat_shot, feat_query, temp, label, method='cos'):
        feat_shot, feat_query = feat_shot.contiguous().float(), feat_query.contiguous().float()
        logits, dlogits, stats = ops.proto_head_ce(feat_shot, feat_query, temp if temp.is_cuda else float(temp), label, method)
        ctx.save_for_backward(feat_shot, feat_query, temp, dlogits)
        ctx.method = method
        loss, acc = stats[0], stats[1]
        ctx.mark_non_differentiable(acc, logits)
        return loss, acc, logits

    @staticmethod
    def backward(ctx, dloss, _dacc, _dlogits):
        feat_shot, feat_query, temp, dlogits = ctx.saved_tensors
        ds, dq, dt = ops.proto_head_ce_backward(feat_shot, feat_query, dlogits, dloss.contiguous().float(), temp if temp.is_cuda else float(temp), ctx.method)
        return ds, dq, dt.reshape(temp.shape), None, None


class LinearFn(torch.autograd.Function):
    """y = x W^T + b on the HIP linear kernels (classifier.py:27-34); x [..., K] fp32."""

    @staticmethod
    def forward(ctx, x, w, b):
        x2 = x.reshape(-1, x.shape[-1]).contiguous().float()
        ctx.save_for_backward(x2, w)
        ctx.xshape, ctx.has_b = x.shape, b is not None
        return ops.linear(x2, w, b).view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous().float()
        dx, dw, db = ops.linear_backward(dy2, x2, w, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_b)
        return (dx.view(ctx.xshape) if dx is not None else None), dw, (db if ctx.has_b else None)


class RowNormalizeFn(torch.autograd.Function):
    """F.normalize(x, dim=-1) for [R, D] fp32 rows on the HIP kernels (utils.compute_logits metric 'cos')."""

    @staticmethod
    def forward(ctx, x):
        y, inv = ops.row_normalize(x)
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        return ops.row_normalize_backward(y, inv, dy)


class SoftTargetCEFn(torch.autograd.Function):
    """SoftTargetCrossEntropy (offline.py:34-45): mean over rows of sum(-target * log_softmax(logits)); the gradient comes out of
    the same kernel launch as the loss."""

    @staticmethod
    def forward(ctx, logits, target):
        R = logits.shape[0]
        if target.shape[0] != R:                                   # offline.py:41-43
            target = target.repeat(R // target.shape[0], 1)
        row, dz = ops.soft_target_ce(logits, target, grad_scale=1.0 / R)
        ctx.save_for_backward(dz)
        return row.mean()

    @staticmethod
    def backward(ctx, dloss):
        (dz,) = ctx.saved_tensors
        return dz * dloss, None
