"""Python face of the C-ABI: device-pointer plumbing only (torch owns memory and streams).

`VisformerEngine` wraps one packed `fsvit_visformer` handle; `ops` exposes the operator-level
entry points used by the parity tests.  Every call goes through libfsvit.so — see `_lib.py`.
"""
import ctypes as C
import os
import sys
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib

DTYPES = {'f32': _lib.F32, 'parity': _lib.F32, 'fp32': _lib.F32, 'bf16': _lib.BF16, 'f16': _lib.F16, 'fp16': _lib.F16,
          'bf16x2': _lib.BF16X2, 'f16x2': _lib.F16X2}
TORCH_DTYPE = {_lib.F32: torch.float32, _lib.BF16: torch.bfloat16, _lib.F16: torch.float16, _lib.BF16X2: torch.float32, _lib.F16X2: torch.float32}
EVAL_ONLY = (_lib.F16, _lib.F16X2)        # 'bf16x2' trains too (fp32 storage, every forward / dgrad / wgrad GEMM on two-limb bf16 MFMAs)


# Packed eval engines cache BN-folded copies of the weights.  The HIP optimizers and the HIP trainer write parameters and running
# statistics through raw device pointers, which torch's per-tensor `_version` never sees - so every such write bumps a generation
# PER WRITTEN TENSOR (keyed by its data pointer) and the encoders' engine fingerprints include the generations of their own tensors
# only: a frozen teacher keeps its packed engine while a student trains next to it (offline.py), a stale engine is still impossible.
_tensor_generation = {}


def bump_weight_generation(tensors) -> None:
    """`tensors`: the torch tensors a HIP kernel has just written behind torch's back."""
    for t in tensors:
        if t is not None:
            p = t.data_ptr()
            _tensor_generation[p] = _tensor_generation.get(p, 0) + 1


def weight_generation(t) -> int:
    return _tensor_generation.get(t.data_ptr(), 0)


def weights_fingerprint(module) -> tuple:
    """(data_ptr, torch version, HIP-side generation) of every parameter and buffer of `module`."""
    return tuple((t.data_ptr(), t._version, _tensor_generation.get(t.data_ptr(), 0)) for t in list(module.parameters()) + list(module.buffers()))


def default_numerics() -> str:
    """'bf16' (throughput mode) unless FSVIT_NUMERICS selects another one: 'f16' (fp16 storage + MFMA: same kernels, 0.93 x the rate, 7 x smaller
    logit deviation than bf16, eval only), 'bf16x2' / 'f16x2' (fp32 storage, every GEMM on the 16-bit MFMA with two-limb operands: meets the
    1e-3 logit tolerance at several times the fp32-MFMA rate; 'bf16x2' also trains, 'f16x2' is eval only) or 'parity' / 'f32' (exact-fp32 MFMA)."""
    return os.environ.get('FSVIT_NUMERICS', 'bf16')


# What a drop-in user gets from each mode, measured against the reference's CPU path (tests/test_gpu_visformer.py gates every line
# against the reference goldens, bench.py `modes` / `agreement` re-measure the agreement on 2048 episodes): max |dlogit| on the golden 5-shot
# episode, arg-max agreement with `parity` on the bench episodes, throughput relative to bf16.
NUMERICS_NOTE = {
    'bf16': 'bf16 storage + MFMA: logits within ~5e-2 of the reference (golden 5-shot 4.9e-2, 1-shot 5.0e-2), 98.8 % arg-max agreement; the throughput mode - NOT the '
            '1e-3-grade mode: FSVIT_NUMERICS=f16 (0.93 x the rate, 7 x tighter), bf16x2 (1e-3-grade, 0.23 x) or parity (exact fp32, 0.11 x)',
    'f16': 'fp16 storage + MFMA: logits within ~8e-3 of the reference (golden 5-shot 7.2e-3), 99.85 % arg-max agreement, 0.93 x the bf16 rate (its GELUs stay on the VALU); eval only',
    'bf16x2': 'fp32 storage, two-limb bf16 MFMA: logits within 1.5e-4 of the reference (meets the 1e-3 tolerance), 0.24 x the bf16 rate; also trains',
    'f16x2': 'fp32 storage, two-limb fp16 MFMA: logits within 2.2e-5 of the reference (meets the 1e-3 tolerance), 0.24 x the bf16 rate; eval only',
    'parity': 'exact-fp32 MFMA: logits within 1e-3 of the reference (the parity mode), 0.12 x the bf16 rate',
    'f32': 'exact-fp32 MFMA: logits within 1e-3 of the reference (the parity mode), 0.12 x the bf16 rate',
}
_numerics_logged = set()


def log_numerics_once(numerics: str, what: str) -> None:
    """One line per process and (mode, engine kind): which numerics mode an engine was built with and what that means for the logits - so that the
    5e-2 throughput mode is never handed to a drop-in user silently.  FSVIT_QUIET=1 silences it."""
    key = (numerics, what)
    if key in _numerics_logged or os.environ.get('FSVIT_QUIET') == '1':
        return
    _numerics_logged.add(key)
    src = 'FSVIT_NUMERICS' if os.environ.get('FSVIT_NUMERICS') == numerics else 'default / encoder_args'
    sys.stderr.write('fsvit: %s built in numerics mode %r (%s) - %s\n' % (what, numerics, src, NUMERICS_NOTE.get(numerics, '')))


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('fsvit kernels run on an MI355X only: tensor is on %s (no CPU fallback)' % t.device)


class _EncoderEngine:
    """One packed eval-mode encoder on one GPU (shared plumbing of the Visformer and ViT handles)."""

    _fn = {}          # C entry points: create / destroy / out_dim / workspace_bytes / forward

    def _make_cfg(self, cfg: dict):
        raise NotImplementedError

    def __init__(self, cfg: dict, state_dict: Dict[str, torch.Tensor], numerics: str = None, device=None,
                 chunk_images: int = None):
        self.lib = _lib.load()
        numerics = numerics or default_numerics()
        if numerics not in DTYPES:
            raise ValueError(f'unknown numerics mode {numerics!r} (bf16 | f16 | bf16x2 | f16x2 | parity)')
        self.dtype = DTYPES[numerics]
        self.numerics = numerics
        self.device = torch.device(device if device is not None else 'cuda')
        if self.device.type != 'cuda':
            raise RuntimeError(f'{type(self).__name__} needs a GPU device (no CPU fallback)')
        log_numerics_once(numerics, type(self).__name__)
        self.chunk_images = int(chunk_images or os.environ.get('FSVIT_CHUNK', self._default_chunk))
        self._chunk_given = bool(chunk_images or os.environ.get('FSVIT_CHUNK'))
        c = self._make_cfg(cfg)
        self.img_size = cfg['img_size']
        keep, arr = [], (_lib.Tensor * len(state_dict))()
        n = 0
        for k, v in state_dict.items():
            if k.endswith('num_batches_tracked'):
                continue
            a = np.ascontiguousarray(v.detach().to('cpu', torch.float32).numpy())
            if self.dtype in (_lib.F16, _lib.F16X2) and a.size and not np.all(np.abs(a) < 65504.0):
                # fp16 limbs top out at 65504 (hi = inf, lo = NaN beyond) and go subnormal below 6e-5: the 'f16' / 'f16x2' modes assume
                # BatchNorm / LayerNorm-scaled networks (weights O(1)); anything else belongs in 'bf16x2' (8-bit exponent, no range limit)
                raise ValueError(f"fsvit: {k} holds values outside the fp16 range; use numerics='bf16x2' (or 'parity') for this checkpoint")
            keep.append(a)
            arr[n].name = k.encode()
            arr[n].data = a.ctypes.data_as(C.POINTER(C.c_float))
            arr[n].ndim = a.ndim
            for i, sdim in enumerate(a.shape):
                arr[n].shape[i] = sdim
            n += 1
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(getattr(self.lib, self._fn['create'])(C.byref(c), arr, n, self.dtype, C.byref(h)))
        self.h = h
        self.out_dim = getattr(self.lib, self._fn['out_dim'])(h)
        self._ws = None
        self._taps = {}

    def __del__(self):
        h, self.h = getattr(self, 'h', None), None
        if h:
            try:
                getattr(self.lib, self._fn['destroy'])(h)
            except Exception:       # interpreter shutdown
                pass

    def workspace(self, n_img: int) -> torch.Tensor:
        chunk = max(1, min(n_img, self.chunk_images))
        ws_bytes = getattr(self.lib, self._fn['workspace_bytes'])
        need = ws_bytes(self.h, chunk)
        if self._ws is None or self._ws.numel() < need:
            if not self._chunk_given:
                # the default chunk (the benched 12 800 images) shrinks on a device that cannot spare half of its free memory for the workspace
                free = torch.cuda.mem_get_info(self.device)[0] + (self._ws.numel() if self._ws is not None else 0)
                while need > 0.5 * free and chunk > 100:
                    chunk //= 2
                    need = ws_bytes(self.h, chunk)
                    self.chunk_images = chunk                # (the C side takes the largest chunk that fits the workspace it is handed)
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def set_tap(self, name: str, shape) -> torch.Tensor:
        t = torch.zeros(shape, dtype=TORCH_DTYPE[self.dtype], device=self.device)
        _lib.check(self.lib.fsvit_encoder_set_tap(self.h, name.encode(), _ptr(t), t.numel() * t.element_size()))
        self._taps[name] = t
        return t

    def profile_begin(self):
        _lib.check(self.lib.fsvit_encoder_profile_begin(self.h))

    def profile_end(self):
        """-> list of dict(layer, kernel, launches, flops, ms): HIP-event time of every launch since
        profile_begin, summed per (layer, kernel template instantiation)."""
        recs = (_lib.ProfRec * 256)()
        n = C.c_int(0)
        _lib.check(self.lib.fsvit_encoder_profile_end(self.h, recs, 256, C.byref(n)))
        return [dict(layer=recs[i].layer.decode(), kernel=self.lib.fsvit_kernel_name(recs[i].kernel_id, self.dtype).decode(),
                     launches=recs[i].launches, flops=recs[i].flops, ms=recs[i].ms) for i in range(n.value)]

    def forward(self, x: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        """x [B,3,H,W] fp32 cuda -> features [B,out_dim] fp32."""
        _require_cuda(x)
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError('expected [B,3,H,W] input')
        x = x.contiguous().float()
        B = x.shape[0]
        if out is None:
            out = torch.empty(B, self.out_dim, dtype=torch.float32, device=x.device)
        if B == 0:
            return out
        ws = self.workspace(B)
        with torch.cuda.device(x.device):
            _lib.check(getattr(self.lib, self._fn['forward'])(self.h, _ptr(x), B, x.shape[2], x.shape[3], _ptr(out), _ptr(ws),
                                                             ws.numel(), _stream_ptr(x.device)))
        return out

    def last_tokens(self, n_img: int, tokens_per_image: int) -> torch.Tensor:
        """Post-norm token map [n_img, T, out_dim] fp32 of the images of the immediately preceding `forward` call (Visformer only; the
        call must have fitted one chunk) - the `x` of the distillation encoder's `return x, pooled`."""
        out = torch.empty(n_img, tokens_per_image, self.out_dim, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fsvit_visformer_last_tokens(self.h, _ptr(self._ws), self._ws.numel(), n_img, _ptr(out), _stream_ptr(self.device)))
        return out

    def meta_baseline_forward(self, x_shot, x_query, temp: float, method: str = 'cos', want_stats=False):
        """x_shot [E,way,shot,3,H,W], x_query [E,Q,3,H,W] -> logits [E,Q,way] (+ per-episode acc, loss)."""
        _require_cuda(x_shot, x_query)
        E, way, shot = x_shot.shape[:3]
        Q = x_query.shape[1]
        x_shot = x_shot.contiguous().float()
        x_query = x_query.contiguous().float()
        dev = x_shot.device
        feat = torch.empty(E * way * shot + E * Q, self.out_dim, dtype=torch.float32, device=dev)
        logits = torch.empty(E, Q, way, dtype=torch.float32, device=dev)
        acc = torch.empty(E, dtype=torch.float32, device=dev)
        loss = torch.empty(E, dtype=torch.float32, device=dev)
        ws = self.workspace(E * way * shot + E * Q)      # shots + queries go through the encoder in one pass when they fit one chunk
        m = {'cos': _lib.HEAD_COS, 'sqr': _lib.HEAD_SQR, 'dot': _lib.HEAD_DOT}[method]
        with torch.cuda.device(dev):
            _lib.check(self.lib.fsvit_meta_baseline_forward(
                self.h, _ptr(x_shot), _ptr(x_query), E, way, shot, Q, x_shot.shape[-2], x_shot.shape[-1], float(temp), m,
                _ptr(logits), _ptr(acc), _ptr(loss), _ptr(feat), _ptr(ws), ws.numel(), _stream_ptr(dev)))
        if want_stats:
            return logits, acc, loss
        return logits


class VisformerEngine(_EncoderEngine):
    """cfg: dict(img_size, init_channels, embed_dim, depth, num_heads, mlp_ratio, group[, bn_eps]);
    state_dict: encoder-relative reference keys."""
    _fn = dict(create='fsvit_visformer_create', destroy='fsvit_visformer_destroy', out_dim='fsvit_visformer_out_dim',
               workspace_bytes='fsvit_visformer_workspace_bytes', forward='fsvit_visformer_forward')
    _default_chunk = 12800      # the benched launch size (round 6: was 3200 - a drop-in user ran the persistent kernels at a quarter of their launch size)

    def _make_cfg(self, cfg):
        c = _lib.VisformerCfg()
        c.img_size, c.init_channels, c.embed_dim = cfg['img_size'], cfg['init_channels'], cfg['embed_dim']
        for i in range(3):
            c.depth[i] = cfg['depth'][i]
        c.num_heads, c.mlp_ratio, c.group = cfg['num_heads'], cfg.get('mlp_ratio', 4.0), cfg.get('group', 8)
        c.bn_eps = cfg.get('bn_eps', 1e-5)
        return c


class VisformerTrainer:
    """Train-mode Visformer (meta-tuning step, train_meta.py:161-177): forward with batch-statistics BN + DropPath
    and saved activations, backward to all parameter gradients.  Parameters are read from (and running stats are
    updated in) the caller's own fp32 cuda tensors every call - nothing is packed ahead of time."""

    def __init__(self, cfg: dict, numerics: str = None, device=None):
        self.lib = _lib.load()
        numerics = numerics or default_numerics()
        if numerics not in DTYPES:
            raise ValueError(f'unknown numerics mode {numerics!r} (bf16 | parity)')
        if DTYPES[numerics] in EVAL_ONLY:
            raise NotImplementedError(f"fsvit: the {numerics!r} numerics mode is an eval mode; train in 'bf16', 'bf16x2' or 'parity'")
        self.dtype = DTYPES[numerics]
        log_numerics_once(numerics, type(self).__name__)
        self.device = torch.device(device if device is not None else 'cuda')
        if self.device.type != 'cuda':
            raise RuntimeError('VisformerTrainer needs a GPU device (no CPU fallback)')
        self.cfg = dict(cfg)
        self.out_dim = cfg['embed_dim'] * 2
        c = VisformerEngine._make_cfg(None, cfg)
        h = C.c_void_p()
        _lib.check(self.lib.fsvit_visformer_trainer_create(C.byref(c), self.dtype, C.byref(h)))
        self.h = h
        self._ws = None
        self._keep = None
        self.generation = 0        # bumped by every forward: the ONE set of saved activations belongs to the latest forward only

    def __del__(self):
        h, self.h = getattr(self, 'h', None), None
        if h:
            try:
                self.lib.fsvit_visformer_trainer_destroy(h)
            except Exception:
                pass

    def set_freeze_bn(self, on: bool):
        """BatchNorm layers in eval mode inside the step (utils.freeze_bn): the running statistics normalise and are not updated; gamma / beta
        still receive gradients, dz = gamma * invstd * dy."""
        _lib.check(self.lib.fsvit_visformer_trainer_set_freeze_bn(self.h, int(bool(on))))

    def n_droppath_calls(self, rate: float) -> int:
        d = self.cfg['depth']
        depth = sum(d)
        n = 0
        for b in range(depth):
            if depth > 1 and rate * b / (depth - 1) > 0:
                n += 1 if b < d[0] else 2
        return n

    @staticmethod
    def _table(tensors: Dict[str, torch.Tensor], grads: Optional[Dict[str, torch.Tensor]]):
        arr = (_lib.Param * len(tensors))()
        keep = []
        for i, (k, v) in enumerate(tensors.items()):
            if v.dtype != torch.float32 or not v.is_contiguous() or not v.is_cuda:
                raise ValueError(f'{k}: training tensors must be contiguous fp32 cuda tensors')
            kb = k.encode()
            keep.append(kb)
            arr[i].name = kb
            arr[i].data = v.data_ptr()
            g = None if grads is None else grads.get(k)
            arr[i].grad = g.data_ptr() if g is not None else None
            arr[i].numel = v.numel()
        return arr, keep

    def forward(self, tensors: Dict[str, torch.Tensor], x: torch.Tensor, drop_path_rate: float = 0.0,
                masks: Optional[torch.Tensor] = None) -> torch.Tensor:
        """tensors: encoder-relative name -> fp32 cuda tensor (parameters AND running stats);
        x [B,3,H,W]; masks [n_droppath_calls, B] of 0/1.  Returns feat [B,out_dim]; activations stay in the workspace."""
        _require_cuda(x)
        x = x.contiguous().float()
        B = x.shape[0]
        arr, keep = self._table(tensors, None)
        need = self.lib.fsvit_visformer_trainer_workspace_bytes(self.h, arr, len(tensors), B, float(drop_path_rate))
        if need == 0:
            _lib.check(_lib.ERR_KEY if 'missing' in self.lib.fsvit_last_error().decode() else _lib.ERR_ARG)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        feat = torch.empty(B, self.out_dim, dtype=torch.float32, device=x.device)
        if masks is not None:
            masks = masks.contiguous().float()
        with torch.cuda.device(x.device):
            _lib.check(self.lib.fsvit_visformer_train_forward(self.h, arr, len(tensors), _ptr(x), B, x.shape[2], x.shape[3],
                                                              float(drop_path_rate), _ptr(masks), _ptr(feat), _ptr(self._ws),
                                                              self._ws.numel(), _stream_ptr(x.device)))
        self._keep = (x, masks)
        self.generation += 1
        # BatchNorm running statistics were updated in place
        bump_weight_generation(v for k, v in tensors.items() if k.endswith(('running_mean', 'running_var')))
        return feat

    def tokens(self, B: int, tokens_per_image: int) -> torch.Tensor:
        """Post-norm token map [B, T, out_dim] fp32 of the last forward (sun_meta_training/models/visformer.py:464)."""
        out = torch.empty(B, tokens_per_image, self.out_dim, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fsvit_visformer_train_tokens(self.h, _ptr(out), _stream_ptr(self.device)))
        return out

    def backward(self, tensors: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], dfeat: torch.Tensor, dtokens: torch.Tensor = None):
        """Overwrites grads[name] (same shapes as tensors[name]) from dfeat [B,out_dim] (+ dtokens [B,T,out_dim], the gradient of the
        token map handed out by `tokens`)."""
        _require_cuda(dfeat)
        if self._keep is None:
            raise RuntimeError('fsvit: backward without a pending train-mode forward (the saved activations were already consumed)')
        dfeat = dfeat.contiguous().float()
        arr, keep = self._table(tensors, grads)
        if dtokens is not None:
            dtokens = dtokens.contiguous().float()
            _lib.check(self.lib.fsvit_visformer_train_set_token_grad(self.h, _ptr(dtokens)))
        with torch.cuda.device(dfeat.device):
            _lib.check(self.lib.fsvit_visformer_train_backward(self.h, arr, len(tensors), _ptr(dfeat), _stream_ptr(dfeat.device)))
        self._keep = None


class VitTrainer:
    """Train-mode ViT / DeiT (deit.py:61-78, 139-218): forward with LayerNorm row statistics + DropPath and saved activations, backward to all
    parameter gradients - the ViT counterpart of VisformerTrainer, same calling convention."""

    def __init__(self, cfg: dict, numerics: str = None, device=None):
        self.lib = _lib.load()
        numerics = numerics or default_numerics()
        if numerics not in DTYPES:
            raise ValueError(f'unknown numerics mode {numerics!r} (bf16 | parity)')
        if DTYPES[numerics] in EVAL_ONLY:
            raise NotImplementedError(f"fsvit: the {numerics!r} numerics mode is an eval mode; train in 'bf16', 'bf16x2' or 'parity'")
        self.dtype = DTYPES[numerics]
        log_numerics_once(numerics, type(self).__name__)
        self.device = torch.device(device if device is not None else 'cuda')
        if self.device.type != 'cuda':
            raise RuntimeError('VitTrainer needs a GPU device (no CPU fallback)')
        self.cfg = dict(cfg)
        self.out_dim = cfg['embed_dim']
        c = VitEngine._make_cfg(None, cfg)
        h = C.c_void_p()
        _lib.check(self.lib.fsvit_vit_trainer_create(C.byref(c), self.dtype, C.byref(h)))
        self.h = h
        self._ws = None
        self._keep = None
        self.generation = 0

    def __del__(self):
        h, self.h = getattr(self, 'h', None), None
        if h:
            try:
                self.lib.fsvit_vit_trainer_destroy(h)
            except Exception:
                pass

    def n_droppath_calls(self, rate: float) -> int:
        return int(self.lib.fsvit_vit_trainer_droppath_calls(self.h, float(rate)))

    def forward(self, tensors: Dict[str, torch.Tensor], x: torch.Tensor, drop_path_rate: float = 0.0, masks: Optional[torch.Tensor] = None) -> torch.Tensor:
        _require_cuda(x)
        x = x.contiguous().float()
        B = x.shape[0]
        arr, keep = VisformerTrainer._table(tensors, None)
        need = self.lib.fsvit_vit_trainer_workspace_bytes(self.h, arr, len(tensors), B, float(drop_path_rate))
        if need == 0:
            _lib.check(_lib.ERR_KEY if 'missing' in self.lib.fsvit_last_error().decode() else _lib.ERR_ARG)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        feat = torch.empty(B, self.out_dim, dtype=torch.float32, device=x.device)
        if masks is not None:
            masks = masks.contiguous().float()
        with torch.cuda.device(x.device):
            _lib.check(self.lib.fsvit_vit_train_forward(self.h, arr, len(tensors), _ptr(x), B, x.shape[2], x.shape[3], float(drop_path_rate), _ptr(masks),
                                                        _ptr(feat), _ptr(self._ws), self._ws.numel(), _stream_ptr(x.device)))
        self._keep = (x, masks)
        self.generation += 1
        return feat

    def backward(self, tensors: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], dfeat: torch.Tensor, dtokens=None):
        _require_cuda(dfeat)
        if self._keep is None:
            raise RuntimeError('fsvit: backward without a pending train-mode forward (the saved activations were already consumed)')
        if dtokens is not None:
            raise NotImplementedError('fsvit: the ViT trainer returns the cls feature only')
        dfeat = dfeat.contiguous().float()
        arr, keep = VisformerTrainer._table(tensors, grads)
        with torch.cuda.device(dfeat.device):
            _lib.check(self.lib.fsvit_vit_train_backward(self.h, arr, len(tensors), _ptr(dfeat), _stream_ptr(dfeat.device)))
        self._keep = None


class VitEngine(_EncoderEngine):
    """cfg: dict(img_size, patch_size, embed_dim, depth, num_heads[, mlp_ratio, ln_eps]) (deit.py:142-144)."""
    _fn = dict(create='fsvit_vit_create', destroy='fsvit_vit_destroy', out_dim='fsvit_vit_out_dim',
               workspace_bytes='fsvit_vit_workspace_bytes', forward='fsvit_vit_forward')
    _default_chunk = 12800      # the benched launch size (round 6: was 400); shrinks to half of the free device memory in workspace()

    def _make_cfg(self, cfg):
        c = _lib.VitCfg()
        c.img_size, c.patch_size, c.embed_dim, c.depth = cfg['img_size'], cfg['patch_size'], cfg['embed_dim'], cfg['depth']
        c.num_heads, c.mlp_ratio, c.ln_eps = cfg['num_heads'], cfg.get('mlp_ratio', 4.0), cfg.get('ln_eps', 1e-6)
        return c


class ops:
    """Operator-level entry points (storage dtype follows the input tensors: fp32 or bf16)."""
    _sgd_tables = {}

    @staticmethod
    def _dt(t):
        if t.dtype == torch.float32:
            return _lib.F32
        if t.dtype == torch.bfloat16:
            return _lib.BF16
        if t.dtype == torch.float16:
            return _lib.F16
        raise TypeError(t.dtype)

    @staticmethod
    def x2_limbs(w: torch.Tensor, numerics: str) -> torch.Tensor:
        """fp32 weights -> the two-limb words of the 'bf16x2' / 'f16x2' GEMM (upper half hi = w rounded to the 16-bit type, lower half
        lo = (w - hi) rounded to it), as a float32-typed tensor of the same shape (fsvit.h: FSVIT_BF16X2 / FSVIT_F16X2)."""
        t16 = torch.bfloat16 if numerics == 'bf16x2' else torch.float16
        w = w.float()
        hi = w.to(t16)
        lo = (w - hi.float()).to(t16)
        if not (torch.isfinite(hi.float()).all() and torch.isfinite(lo.float()).all()):
            raise ValueError("x2_limbs: values outside the 16-bit type's range (fp16 limbs: |w| < 65504); use 'bf16x2'")
        words = (hi.view(torch.int16).to(torch.int32) << 16) | (lo.view(torch.int16).to(torch.int32) & 0xffff)
        return words.view(torch.float32)

    @staticmethod
    def conv_gemm(x, w, bias, res, pos, B, H, W, Cin, KH, KW, stride, pad, N, groups, act, res_first, x_cstride=None, numerics=None):
        """x NHWC [B,H,W,x_cstride]; w [groups][N][Kw] packed; returns y NHWC [B,OH,OW,groups*N].  numerics 'bf16x2' / 'f16x2': x fp32,
        w = ops.x2_limbs(packed fp32 weights)."""
        _require_cuda(x, w)
        lib = _lib.load()
        OH = (H + 2 * pad - KH) // stride + 1
        OW = (W + 2 * pad - KW) // stride + 1
        x_cstride = x_cstride or x.shape[-1]
        y = torch.empty(B, OH, OW, groups * N, dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.fsvit_conv_gemm(_ptr(x), _ptr(w), _ptr(bias), _ptr(res), _ptr(pos), _ptr(y), B, H, W, Cin, x_cstride,
                                           KH, KW, stride, pad, N, groups * N, w.shape[-1], groups, act, int(res_first),
                                           DTYPES[numerics] if numerics else ops._dt(x), _stream_ptr(x.device)))
        return y

    @staticmethod
    def gconv3x3(x, w_packed):
        """x NHWC [B,H,W,256] (bf16 / fp16), w_packed [256][Kw] -> the grouped 3x3 convolution (8 groups of 32 -> 32) of the training step."""
        _require_cuda(x, w_packed)
        B, H, W, _ = x.shape
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().fsvit_gconv3x3(_ptr(x), _ptr(w_packed), w_packed.shape[-1], _ptr(y), B, H, W, ops._dt(x), _stream_ptr(x.device)))
        return y

    @staticmethod
    def conv_stem_tail(x, w, bias, pos, x2, K2):
        """Fused stem tail: x NHWC [B,H,W,Cin], w [N][Kw] (conv3 taps | one tail K slice), x2 [B*H*W][x2_cstride] im2col rows,
        pos [(H/2)*(W/2)][N] fp32 -> y NHWC [B,H/2,W/2,N]."""
        _require_cuda(x, w, x2, pos)
        lib = _lib.load()
        B, H, W, Cin = x.shape
        N = w.shape[0]
        y = torch.empty(B, H // 2, W // 2, N, dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.fsvit_conv_stem_tail(_ptr(x), _ptr(w), _ptr(bias), _ptr(pos), _ptr(x2), x2.shape[-1], int(K2), _ptr(y), B, H, W, Cin, N,
                                                w.shape[-1], ops._dt(x), _stream_ptr(x.device)))
        return y

    @staticmethod
    def stage1_block(x, w1, b1, w2, w3):
        """x NHWC [B,20,20,128] bf16; packed w1 [256][128], w2 [8][32][320], w3 [128][256] bf16; b1 [256] fp32."""
        _require_cuda(x, w1, w2, w3, b1)
        lib = _lib.load()
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(lib.fsvit_stage1_block(_ptr(x), _ptr(y), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(w3), x.shape[0], _stream_ptr(x.device)))
        return y

    @staticmethod
    def stage1_block_hw(x, w1, b1, w2, w3):
        """x NHWC [B,H,W,128] (bf16 / fp16), H = W in 4 .. 20: the ring kernel of the stage-1 block (fsvit_stage1_block_hw)."""
        _require_cuda(x, w1, w2, w3, b1)
        lib = _lib.load()
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(lib.fsvit_stage1_block_hw(_ptr(x), _ptr(y), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(w3), x.shape[0], x.shape[1], x.shape[2], ops._dt(x),
                                                 _stream_ptr(x.device)))
        return y

    # ---- distillation head (sun_meta_training/offline.py): fp32, token-major rows
    @staticmethod
    def linear(x, w, b=None):
        """x [M, K] fp32, w [N, K], b [N] or None -> x w^T + b (classifier.py:27-34)."""
        _require_cuda(x, w)
        x, w = x.contiguous().float(), w.contiguous().float()
        y = torch.empty(x.shape[0], w.shape[0], dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().fsvit_linear_forward(_ptr(x), _ptr(w), _ptr(None if b is None else b.contiguous().float()), _ptr(y),
                                                        x.shape[0], w.shape[0], x.shape[1], _stream_ptr(x.device)))
        return y

    @staticmethod
    def linear_backward(dy, x, w, need_dx=True, need_dw=True, need_db=True):
        _require_cuda(dy, x, w)
        dy, x, w = dy.contiguous().float(), x.contiguous().float(), w.contiguous().float()
        dx = torch.empty_like(x) if need_dx else None
        dw = torch.empty_like(w) if need_dw else None
        db = torch.empty(w.shape[0], dtype=torch.float32, device=x.device) if (need_db and need_dw) else None
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().fsvit_linear_backward(_ptr(dy), _ptr(x), _ptr(w), _ptr(dx), 0, _ptr(dw), _ptr(db), x.shape[0], w.shape[0], x.shape[1],
                                                         _stream_ptr(x.device)))
        return dx, dw, db

    @staticmethod
    def token_softlabel(teacher_logits, k=3, bp=10, smoothing=0.1):
        """teacher token logits [B, T, C] -> soft labels [B*T, C+1] (offline.py:57-76)."""
        _require_cuda(teacher_logits)
        lt = teacher_logits.contiguous().float()
        B, T, Cc = lt.shape
        soft = torch.empty(B * T, Cc + 1, dtype=torch.float32, device=lt.device)
        with torch.cuda.device(lt.device):
            _lib.check(_lib.load().fsvit_token_softlabel(_ptr(lt), _ptr(soft), B, T, Cc, int(k), int(bp), float(smoothing), _stream_ptr(lt.device)))
        return soft

    @staticmethod
    def soft_target_ce(logits, target, grad_scale=None):
        """-> (row_loss [R], dlogits [R, C] = grad_scale * d(sum of row losses)/dlogits, or None)."""
        _require_cuda(logits, target)
        z, t = logits.contiguous().float(), target.contiguous().float()
        R, Cc = z.shape
        row = torch.empty(R, dtype=torch.float32, device=z.device)
        dz = torch.empty_like(z) if grad_scale is not None else None
        with torch.cuda.device(z.device):
            _lib.check(_lib.load().fsvit_soft_target_ce(_ptr(z), _ptr(t), _ptr(row), _ptr(dz), R, Cc, float(grad_scale or 0.0), _stream_ptr(z.device)))
        return row, dz

    @staticmethod
    def row_normalize(x):
        """F.normalize(x, dim=-1) of a [R, D] fp32 matrix -> (y, inv_norm [R])."""
        _require_cuda(x)
        x = x.contiguous().float()
        y, inv = torch.empty_like(x), torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().fsvit_row_normalize(_ptr(x), _ptr(y), _ptr(inv), x.shape[0], x.shape[1], _stream_ptr(x.device)))
        return y, inv

    @staticmethod
    def row_normalize_backward(y, inv, dy):
        _require_cuda(y, inv, dy)
        dy = dy.contiguous().float()
        dx = torch.empty_like(y)
        with torch.cuda.device(y.device):
            _lib.check(_lib.load().fsvit_row_normalize_backward(_ptr(y), _ptr(inv), _ptr(dy), _ptr(dx), y.shape[0], y.shape[1], _stream_ptr(y.device)))
        return dx

    @staticmethod
    def adamw_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step):
        _require_cuda(p, g, m, v)
        bump_weight_generation((p,))
        with torch.cuda.device(p.device):
            _lib.check(_lib.load().fsvit_adamw_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
                                                    float(weight_decay), int(step), _stream_ptr(p.device)))

    _adamw_tables = {}

    @staticmethod
    def adamw_step_multi(params, grads, ms, vs, lr, beta1, beta2, eps, weight_decay, step):
        """One launch for a list of fp32 tensors at the same update number (fsvit_adamw_step_multi); the pointer table is cached per device and re-uploaded
        only when a pointer changed (as ops.sgd_step_multi)."""
        if not params:
            return
        _require_cuda(*params)
        lib = _lib.load()
        bump_weight_generation(params)
        dev = params[0].device
        rows = tuple((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()) for p, g, m, v in zip(params, grads, ms, vs))
        key = (dev, len(rows))
        cached = ops._adamw_tables.get(key)
        if cached is None or cached[0] != rows:
            table = torch.tensor(rows, dtype=torch.int64).pin_memory()
            with torch.cuda.device(dev):
                cached = (rows, table.to(dev, non_blocking=True), table)
            ops._adamw_tables[key] = cached
        with torch.cuda.device(dev):
            _lib.check(lib.fsvit_adamw_step_multi(_ptr(cached[1]), len(rows), max(r[4] for r in rows), float(lr), float(beta1), float(beta2), float(eps),
                                                  float(weight_decay), int(step), _stream_ptr(dev)))

    @staticmethod
    def proj_mlp_rows(x, ctx, wp, w1, b1, w2, b2=None):
        """x [M][C], ctx [M][KC] bf16, wp [C][KpW]: x1 = x + ctx wp^T; y = x1 + W2 GELU(W1 x1 + b1) + b2  ((C, KC) = (256, 288) | (512, 576))."""
        _require_cuda(x, ctx, wp, w1, w2)
        lib = _lib.load()
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(lib.fsvit_proj_mlp_rows(_ptr(x), _ptr(y), _ptr(ctx), _ptr(wp), wp.shape[-1], ctx.shape[1], _ptr(w1), w1.shape[-1], _ptr(b1),
                                               _ptr(w2), w2.shape[-1], _ptr(b2), x.shape[0], x.shape[1], w1.shape[0], _stream_ptr(x.device)))
        return y

    @staticmethod
    def ln_linear_rows(x, w, b, eps=1e-6):
        """y = b + LN(x) w^T on bf16 rows (x [M][384], w [N][>=384], N % 32 == 0): the DeiT block's norm1 + qkv (deit.py:40-47,:69), LN without
        affine (gamma / beta folded into w / b by the caller).  x [M][512]: the same row-wise kernel without the LayerNorm (Visformer stage-3
        qkv, visformer.py:175), b may be None."""
        _require_cuda(x, w)
        y = torch.empty(x.shape[0], w.shape[0], device=x.device, dtype=x.dtype)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().fsvit_ln_linear_rows(_ptr(x), _ptr(y), _ptr(w), w.shape[-1], _ptr(b), x.shape[0], x.shape[1], w.shape[0], float(eps),
                                                        _stream_ptr(x.device)))
        return y

    @staticmethod
    def patch_embed2x2(x, w, bias, pos):
        """Visformer PatchEmbed (conv k2 s2 with folded BN, + pos_embed; visformer.py:266-288) on the row-wise kernel: x NHWC bf16 [B, H, H, Ci]
        (4 Ci = 512), w [N, 4 Ci] in (ky, kx, c) order, bias [N] or None, pos fp32 [(H/2)^2, N] -> [B (H/2)^2, N]."""
        _require_cuda(x, w, pos)
        B, H, _, Ci = x.shape
        y = torch.empty(B * (H // 2) * (H // 2), w.shape[0], device=x.device, dtype=x.dtype)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().fsvit_patch_embed2x2(_ptr(x), _ptr(y), _ptr(w), w.shape[-1], _ptr(bias), _ptr(pos), B, H, Ci, w.shape[0],
                                                        _stream_ptr(x.device)))
        return y

    @staticmethod
    def vit_block_tail(x, ctx, wp, bp, w1, b1, w2, b2, eps=1e-6):
        """DeiT block tail (deit.py:69-72) on bf16 rows, (C, KC, hidden) = (384, 384, 1536): x1 = x + bp + ctx wp^T;
        y = x1 + b2 + W2 GELU(W1 LN(x1) + b1), LN without affine (norm2's gamma / beta folded into w1 / b1 by the caller)."""
        _require_cuda(x, ctx, wp, bp, w1, b1, w2, b2)
        lib = _lib.load()
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(lib.fsvit_vit_block_tail(_ptr(x), _ptr(y), _ptr(ctx), _ptr(wp), wp.shape[-1], ctx.shape[1], _ptr(bp), _ptr(w1), w1.shape[-1],
                                                _ptr(b1), _ptr(w2), w2.shape[-1], _ptr(b2), x.shape[0], x.shape[1], w1.shape[0], float(eps),
                                                _stream_ptr(x.device)))
        return y

    @staticmethod
    def mlp_rows(x, w1, b1, w2, b2=None):
        """x [M][C] bf16, C = 256 or 512; w1 [4C][K1w], w2 [C][K2w] packed K-major bf16; b1 [4C], b2 [C] fp32 or None.  y = x + W2 GELU(W1 x + b1) + b2."""
        _require_cuda(x, w1, w2)
        lib = _lib.load()
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(lib.fsvit_mlp_rows(_ptr(x), _ptr(y), _ptr(w1), w1.shape[-1], _ptr(b1), _ptr(w2), w2.shape[-1], _ptr(b2),
                                          x.shape[0], x.shape[1], w1.shape[0], _stream_ptr(x.device)))
        return y

    @staticmethod
    def attention(qkv, B, S, heads, hdp, scale):
        _require_cuda(qkv)
        lib = _lib.load()
        ctx = torch.empty(B * S, heads * hdp, dtype=qkv.dtype, device=qkv.device)
        with torch.cuda.device(qkv.device):
            _lib.check(lib.fsvit_attention(_ptr(qkv), _ptr(ctx), B, S, heads, hdp, float(scale), ops._dt(qkv), _stream_ptr(qkv.device)))
        return ctx

    @staticmethod
    def qkv_attention(x, wqkv, bias, B, S, heads, hdp, scale):
        """Fused qkv conv + attention core (qkv_attn.hip): x [B*S, C] bf16, wqkv [3*heads*hdp, kw] bf16 -> ctx [B*S, heads*hdp]."""
        _require_cuda(x)
        lib = _lib.load()
        ctx = torch.empty(B * S, heads * hdp, dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.fsvit_qkv_attention(_ptr(x), _ptr(wqkv), wqkv.shape[-1], _ptr(bias), _ptr(ctx), B, S, x.shape[1], heads, hdp,
                                               float(scale), _stream_ptr(x.device)))
        return ctx

    @staticmethod
    def vit_ln_qkv_attention(x, wqkv, bias, B, S, heads, hdp, scale, eps=1e-6):
        """norm1 + qkv Linear + attention core of a ViT block in one launch (mlp_rows.hip vit_attn_rows): x [B*S, 384] bf16, wqkv [3*heads*64, kw]
        bf16 (LayerNorm affine folded in by the caller) -> ctx [B*S, heads*64]."""
        _require_cuda(x, wqkv)
        ctx = torch.empty(B * S, heads * hdp, dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().fsvit_vit_ln_qkv_attention(_ptr(x), _ptr(wqkv), wqkv.shape[-1], _ptr(bias), _ptr(ctx), B, S, x.shape[1], heads, hdp,
                                                              float(eps), float(scale), _stream_ptr(x.device)))
        return ctx

    @staticmethod
    def im2col27(x, dtype):
        _require_cuda(x)
        lib = _lib.load()
        B, _, H, W = x.shape
        out = torch.empty(B * (H // 2) * (W // 2), 32, dtype=dtype, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.fsvit_im2col27(_ptr(x.contiguous()), _ptr(out), B, H, W, ops._dt(out), _stream_ptr(x.device)))
        return out

    @staticmethod
    def stem_conv1(x, w, bias):
        """im2col + stem conv1 + LeakyReLU in one pass (stem.hip): x [B,3,80,80] fp32, w [64, kw] bf16 -> (patches [B*1600,32], c1 [B*1600,64])."""
        _require_cuda(x, w)
        lib = _lib.load()
        B, _, H, W = x.shape
        patches = torch.empty(B * (H // 2) * (W // 2), 32, dtype=w.dtype, device=x.device)
        c1 = torch.empty(B * (H // 2) * (W // 2), 64, dtype=w.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.fsvit_stem_conv1(_ptr(x.contiguous()), _ptr(w), w.shape[-1], _ptr(bias), _ptr(patches), _ptr(c1), B, H, W, _stream_ptr(x.device)))
        return patches, c1

    @staticmethod
    def maxpool2_pos(x, pos):
        _require_cuda(x)
        lib = _lib.load()
        B, H, W, Cc = x.shape
        out = torch.empty(B, H // 2, W // 2, Cc, dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.fsvit_maxpool2_pos(_ptr(x), _ptr(pos), _ptr(out), B, H // 2, W // 2, Cc, ops._dt(x), _stream_ptr(x.device)))
        return out

    @staticmethod
    def pool_affine(x, scale, shift):
        _require_cuda(x)
        lib = _lib.load()
        B, HW, Cc = x.shape
        out = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.fsvit_pool_affine(_ptr(x), _ptr(scale), _ptr(shift), _ptr(out), B, HW, Cc, ops._dt(x), _stream_ptr(x.device)))
        return out

    @staticmethod
    def attention_backward(qkv, dctx, B, S, heads, hd, hdp, scale):
        _require_cuda(qkv, dctx)
        lib = _lib.load()
        dqkv = torch.empty_like(qkv)
        with torch.cuda.device(qkv.device):
            _lib.check(lib.fsvit_attention_backward(_ptr(qkv), _ptr(dctx), _ptr(dqkv), B, S, heads, hd, hdp, float(scale),
                                                    ops._dt(qkv), _stream_ptr(qkv.device)))
        return dqkv

    @staticmethod
    def proto_head_backward(feat_shot, feat_query, dlogits, temp, method='cos'):
        """-> dfeat_shot [E,way,shot,D], dfeat_query [E,Q,D], dtemp (scalar tensor); method 'cos' or 'sqr'."""
        if method not in ('cos', 'sqr'):
            raise ValueError(method)
        _require_cuda(feat_shot, feat_query, dlogits)
        lib = _lib.load()
        E, way, shot, D = feat_shot.shape
        Q = feat_query.shape[1]
        dev = feat_shot.device
        ds, dq = torch.empty_like(feat_shot), torch.empty_like(feat_query)
        dt = torch.empty(E, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            if isinstance(temp, torch.Tensor) and temp.is_cuda:
                _lib.check(lib.fsvit_proto_head_backward_devtemp(_ptr(feat_shot), _ptr(feat_query), _ptr(dlogits.contiguous().float()), E, way, shot, Q, D,
                                                                 _ptr(temp.detach().float().contiguous()), _lib.HEAD_COS if method == 'cos' else _lib.HEAD_SQR,
                                                                 _ptr(ds), _ptr(dq), _ptr(dt), _stream_ptr(dev)))
            else:
                fn = lib.fsvit_proto_head_backward if method == 'cos' else lib.fsvit_proto_head_backward_sqr
                _lib.check(fn(_ptr(feat_shot), _ptr(feat_query), _ptr(dlogits.contiguous().float()), E, way, shot, Q, D, float(temp), _ptr(ds), _ptr(dq), _ptr(dt),
                              _stream_ptr(dev)))
        return ds, dq, dt.sum()

    @staticmethod
    def conv1x1_wgrad(x, dz, limbs='bf16'):
        """x [M,C], dz [M,N] (bf16 / fp16, or fp32 = two-limb arithmetic with `limbs` 'bf16' / 'f16') -> dW [N,C] fp32 (fsvit_conv1x1_wgrad)."""
        _require_cuda(x, dz)
        lib = _lib.load()
        assert x.dtype == dz.dtype and x.dtype in (torch.bfloat16, torch.float16, torch.float32) and x.shape[0] == dz.shape[0]
        x, dz = x.contiguous(), dz.contiguous()
        dw = torch.empty(dz.shape[1], x.shape[1], dtype=torch.float32, device=x.device)
        dt = {torch.bfloat16: _lib.BF16, torch.float16: _lib.F16, torch.float32: _lib.BF16X2 if limbs == 'bf16' else _lib.F16X2}[x.dtype]
        with torch.cuda.device(x.device):
            _lib.check(lib.fsvit_conv1x1_wgrad(_ptr(x), _ptr(dz), _ptr(dw), x.shape[0], dz.shape[1], x.shape[1], dt, _stream_ptr(x.device)))
        return dw

    @staticmethod
    def conv3x3_wgrad(x_nhwc, dz, O, Ig, groups, limbs='bf16'):
        """x [B,H,W,groups*Ig], dz [B,H,W,O] (bf16 / fp16, or fp32 = two-limb arithmetic with `limbs` 'bf16' / 'f16') -> dW [O,Ig,3,3] fp32 of a
        3x3 / stride 1 / pad 1 convolution (fsvit_conv3x3_wgrad)."""
        _require_cuda(x_nhwc, dz)
        lib = _lib.load()
        assert x_nhwc.dtype == dz.dtype and x_nhwc.dtype in (torch.bfloat16, torch.float16, torch.float32)
        B, H, W, C = x_nhwc.shape
        assert C == groups * Ig and dz.shape == (B, H, W, O)
        x_nhwc, dz = x_nhwc.contiguous(), dz.contiguous()
        dw = torch.empty(O, Ig, 3, 3, dtype=torch.float32, device=x_nhwc.device)
        with torch.cuda.device(x_nhwc.device):
            dt = {torch.bfloat16: _lib.BF16, torch.float16: _lib.F16, torch.float32: _lib.BF16X2 if limbs == 'bf16' else _lib.F16X2}[x_nhwc.dtype]
            _lib.check(lib.fsvit_conv3x3_wgrad(_ptr(x_nhwc), _ptr(dz), _ptr(dw), B, H, W, O, Ig, groups, dt, _stream_ptr(x_nhwc.device)))
        return dw

    @staticmethod
    def sgd_step_multi(params, grads, bufs, lr, momentum, weight_decay, first_step):
        """One launch for a list of fp32 tensors (fsvit_sgd_step_multi): the pointer table is built on the host and copied with the stream."""
        if not params:
            return
        _require_cuda(*params)
        lib = _lib.load()
        bump_weight_generation(params)
        dev = params[0].device
        rows = tuple((p.data_ptr(), g.data_ptr(), b.data_ptr(), p.numel()) for p, g, b in zip(params, grads, bufs))
        cached = ops._sgd_tables.get(dev)
        if cached is None or cached[0] != rows:
            # (pinned allocation + upload only when a pointer changed: with parallel.GradBucket or zero_grad(set_to_none=False) the gradients keep
            # their addresses and a step uploads nothing; hipHostMalloc per step stalled the host behind the whole backward)
            table = torch.tensor(rows, dtype=torch.int64).pin_memory()
            with torch.cuda.device(dev):
                cached = (rows, table.to(dev, non_blocking=True), table)
            ops._sgd_tables[dev] = cached
        with torch.cuda.device(dev):
            _lib.check(lib.fsvit_sgd_step_multi(_ptr(cached[1]), len(params), max(r[3] for r in rows), float(lr), float(momentum),
                                                float(weight_decay), int(bool(first_step)), _stream_ptr(dev)))
        return cached[1]

    @staticmethod
    def sgd_step(param, grad, buf, lr, momentum, weight_decay, first_step):
        _require_cuda(param, grad, buf)
        lib = _lib.load()
        bump_weight_generation((param,))
        with torch.cuda.device(param.device):
            _lib.check(lib.fsvit_sgd_step(_ptr(param), _ptr(grad), _ptr(buf), param.numel(), float(lr), float(momentum),
                                          float(weight_decay), int(bool(first_step)), _stream_ptr(param.device)))

    _tickets = {}

    @staticmethod
    def _ticket(dev):
        """Two zeroed device words per (device, STREAM): forward / backward of the fused CE head count their workgroups on them and leave them at
        zero.  Launches on one stream are ordered; two models stepping on different streams each get their own pair (one pair per device
        let concurrent launches race on the counter)."""
        key = (dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream)
        if key not in ops._tickets:
            ops._tickets[key] = torch.zeros(2, dtype=torch.int32, device=dev)
        return ops._tickets[key]

    @staticmethod
    def proto_head_ce(feat_shot, feat_query, temp, label=None, method='cos'):
        """Head + F.cross_entropy + compute_acc of the meta-tuning step in one launch (fsvit_proto_head_ce): -> logits [E,Q,way], dlogits [E,Q,way]
        (of the mean CE), stats [2 + 2 E] = {loss, acc, acc per episode .., loss per episode ..}.  label: int64 [E*Q] on the device (None: make_nk_label)."""
        _require_cuda(feat_shot, feat_query)
        lib = _lib.load()
        E, way, shot, D = feat_shot.shape
        Q = feat_query.shape[1]
        dev = feat_shot.device
        logits = torch.empty(E, Q, way, dtype=torch.float32, device=dev)
        dlogits = torch.empty(E, Q, way, dtype=torch.float32, device=dev)
        stats = torch.full((2 + 2 * E,), float('nan'), dtype=torch.float32, device=dev)      # (a skipped final reduction must not read as a loss)
        m = {'cos': _lib.HEAD_COS, 'sqr': _lib.HEAD_SQR, 'dot': _lib.HEAD_DOT}[method]
        if label is not None:
            if label.dtype != torch.int64 or not label.is_cuda or label.numel() != E * Q:
                raise ValueError('label: int64 [E*Q] on the device')
            label = label.contiguous()
        dev_temp = isinstance(temp, torch.Tensor) and temp.is_cuda
        tk = ops._ticket(dev)
        with torch.cuda.device(dev):
            _lib.check(lib.fsvit_proto_head_ce(_ptr(feat_shot), _ptr(feat_query), _ptr(label) if label is not None else None, E, way, shot, Q, D,
                                               0.0 if dev_temp else float(temp), _ptr(temp.detach()) if dev_temp else None, m, _ptr(logits), _ptr(dlogits),
                                               _ptr(stats[2:2 + E]), _ptr(stats[2 + E:]), _ptr(stats), tk.data_ptr(), _stream_ptr(dev)))
        return logits, dlogits, stats

    @staticmethod
    def proto_head_ce_backward(feat_shot, feat_query, dlogits, dloss, temp, method='cos'):
        """-> dfeat_shot, dfeat_query, dtemp (0-d tensor): fsvit_proto_head_ce_backward with the upstream gradient `dloss` read on the device."""
        if method not in ('cos', 'sqr'):
            raise ValueError(method)
        _require_cuda(feat_shot, feat_query, dlogits)
        lib = _lib.load()
        E, way, shot, D = feat_shot.shape
        Q = feat_query.shape[1]
        dev = feat_shot.device
        ds, dq = torch.empty_like(feat_shot), torch.empty_like(feat_query)
        dt = torch.empty(E + 1, dtype=torch.float32, device=dev)
        dev_temp = isinstance(temp, torch.Tensor) and temp.is_cuda
        if dloss is not None and not (dloss.is_cuda and dloss.dtype == torch.float32 and dloss.numel() == 1):
            raise ValueError('dloss: one fp32 value on the device')
        tk = ops._ticket(dev)
        with torch.cuda.device(dev):
            _lib.check(lib.fsvit_proto_head_ce_backward(_ptr(feat_shot), _ptr(feat_query), _ptr(dlogits), _ptr(dloss) if dloss is not None else None, E, way,
                                                        shot, Q, D, 0.0 if dev_temp else float(temp), _ptr(temp.detach()) if dev_temp else None,
                                                        _lib.HEAD_COS if method == 'cos' else _lib.HEAD_SQR, _ptr(ds), _ptr(dq), _ptr(dt),
                                                        tk.data_ptr() + 4, _stream_ptr(dev)))
        return ds, dq, dt[E]

    @staticmethod
    def proto_head(feat_shot, feat_query, temp, method='cos'):
        """feat_shot [E,way,shot,D], feat_query [E,Q,D] fp32 -> logits [E,Q,way], acc [E], loss [E]."""
        _require_cuda(feat_shot, feat_query)
        lib = _lib.load()
        E, way, shot, D = feat_shot.shape
        Q = feat_query.shape[1]
        dev = feat_shot.device
        logits = torch.empty(E, Q, way, dtype=torch.float32, device=dev)
        acc = torch.empty(E, dtype=torch.float32, device=dev)
        loss = torch.empty(E, dtype=torch.float32, device=dev)
        m = {'cos': _lib.HEAD_COS, 'sqr': _lib.HEAD_SQR, 'dot': _lib.HEAD_DOT}[method]
        with torch.cuda.device(dev):
            if isinstance(temp, torch.Tensor) and temp.is_cuda:      # the learnable temperature stays on the device: no .item() synchronisation per step
                _lib.check(lib.fsvit_proto_head_devtemp(_ptr(feat_shot.contiguous().float()), _ptr(feat_query.contiguous().float()), E, way,
                                                        shot, Q, D, _ptr(temp.detach().float().contiguous()), m, _ptr(logits), _ptr(acc), _ptr(loss),
                                                        _stream_ptr(dev)))
            else:
                _lib.check(lib.fsvit_proto_head(_ptr(feat_shot.contiguous().float()), _ptr(feat_query.contiguous().float()), E, way,
                                                shot, Q, D, float(temp), m, _ptr(logits), _ptr(acc), _ptr(loss), _stream_ptr(dev)))
        return logits, acc, loss
