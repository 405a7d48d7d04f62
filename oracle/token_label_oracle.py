"""TEST INFRASTRUCTURE ONLY - CPU restatement of the distillation head of the reference's SUN meta-training phase
(sun_meta_training/offline.py, models/token_label.py); never imported by the product.  PARITY PINNED:
tests/test_token_label_cpu.py checks every function against tests/golden/token_label.npz, which
tests/golden/make_token_label_golden.py produced by executing the reference's own definitions.

Layouts here are token-major (what the HIP kernels use): token logits [B, T, C] with T = H*W tokens in row-major (h, w) order -
the reference's [B, C, H, W] permuted by (0, 2, 3, 1), exactly the view its own code flattens (offline.py:69, :293).
"""
import numpy as np


def linear(x, w, b):
    """classifier.py:27-34 LinearClassifier: x [..., K] @ w[N, K]^T + b."""
    return x @ w.T + b


def token_label_forward(fmap, pooled, sd, is_teacher=False):
    """models/token_label.py:48-60 TokenLabelOffline.forward after the encoder: fmap [B, D, H, W], pooled [B, D] ->
    (y_token [B, T, C or C+1], y [B, C], pooled): the student uses `classifier_local` (C + 1 outputs: background class), the
    teacher the global `classifier`, on every token; both use `classifier` on the pooled feature."""
    B, D = fmap.shape[:2]
    tok = fmap.reshape(B, D, -1).transpose(0, 2, 1)                              # :49 x.permute(0, 2, 3, 1)
    head = 'classifier' if is_teacher else 'classifier_local'                    # :50-53
    y_token = linear(tok, sd[head + '.linear.weight'], sd[head + '.linear.bias'])
    y = linear(pooled, sd['classifier.linear.weight'], sd['classifier.linear.bias'])   # :57
    return y_token, y, pooled


BG_COLUMN = 1      # offline.py:61,71 (see generate_softlabel)


def generate_softlabel(logits_t, smoothing=0.1, k=3, bp=10):
    """offline.py:57-76.  logits_t [B, T, C] teacher token logits -> soft label [B*T, C+1]:
    * per token the top-k classes get `on_value`, everything else `off_value` (:70-73), on an extra background column C;
    * the `bp` tokens of an image with the SMALLEST max-logit are background tokens (:60-65: top (T - bp) of the per-token max are
      kept): their row is `on_value` at ONE column only (:71,:74-75).  That column is index 1, not the extra background column C:
      the reference fills `bg_map` with `c` AFTER rebinding `b, c, h, w = logits_max.size()` (:61), where c = 1.  Reproduced as is
      (BG_COLUMN) - parity is with the reference's behaviour, and the golden vectors pin it."""
    B, T, C = logits_t.shape
    off_value = smoothing / C                                                    # :58-59 (n_classes = teacher classes)
    on_value = 1.0 - smoothing + off_value
    tmax = logits_t.max(axis=2)                                                  # :60
    order = np.argsort(-tmax, axis=1, kind='stable')                             # :63 topk(T - bp) over tokens
    pos = np.zeros((B, T), dtype=bool)
    for b in range(B):
        pos[b, order[b, :T - bp]] = True
    soft = np.full((B * T, C + 1), off_value, dtype=np.float32)
    flat = logits_t.reshape(B * T, C)
    top = np.argsort(-flat, axis=1, kind='stable')[:, :k]                        # :70
    for r in range(B * T):
        if pos.reshape(-1)[r]:
            soft[r, top[r]] = on_value
        else:
            soft[r, BG_COLUMN] = on_value
    return soft


def soft_target_cross_entropy(logits, target):
    """offline.py:34-45: mean over rows of sum(-target * log_softmax(logits)); returns (loss, dloss/dlogits).
    The target rows are NOT normalised (a positive row sums to 65 off + k (on - off)), so the gradient is softmax * sum(t) - t."""
    z = logits.astype(np.float64)
    t = target.astype(np.float64)
    m = z.max(axis=1, keepdims=True)
    lse = m + np.log(np.exp(z - m).sum(axis=1, keepdims=True))
    logp = z - lse
    loss = (-(t * logp).sum(axis=1)).mean()
    grad = (np.exp(logp) * t.sum(axis=1, keepdims=True) - t) / z.shape[0]
    return loss, grad.astype(np.float32)


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """timm.optim.AdamW as constructed at offline.py:233 (= torch.optim.AdamW, decoupled decay; timm is absent - the update below is
    the published algorithm): returns the new (p, m, v) after update number `step` (1-based)."""
    p = p * (1.0 - lr * weight_decay)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    denom = np.sqrt(v) / np.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v
