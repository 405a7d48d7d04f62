"""Host side of the device-resident eval transform: Pillow's BILINEAR resampling coefficient tables
(third-party arithmetic the reference reaches through torchvision.transforms.Resize on PIL images,
test_phase/datasets/mini_imagenet.py:50-51): for each output coordinate the first input index, the tap count and the
22-bit fixed-point taps, computed in double precision as Pillow's src/libImaging/Resample.c `precompute_coeffs` +
`normalize_coeffs_8bpc` do.  The HIP kernel behind fsvit_image_transform_gather applies them (two 8-bit passes)."""
import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2
IMAGENET_MEAN = (0.485, 0.456, 0.406)        # mini_imagenet.py:43-44
IMAGENET_STD = (0.229, 0.224, 0.225)


def pil_bilinear_tables(in_size: int, out_size: int):
    """-> xmin [out] int32, count [out] int32, coef [out, ksize] int32 (zero padded)."""
    scale = float(in_size) / float(out_size)
    filterscale = max(scale, 1.0)
    support = filterscale                                    # bilinear support 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    inv = 1.0 / filterscale
    xmin = np.zeros(out_size, np.int32)
    cnt = np.zeros(out_size, np.int32)
    coef = np.zeros((out_size, ksize), np.int32)
    for o in range(out_size):
        center = (o + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        taps = []
        total = 0.0
        for x in range(lo, hi):
            a = abs((x - center + 0.5) * inv)               # same association as Resample.c: (x + xmin - center + 0.5) * ss
            w = 1.0 - a if a < 1.0 else 0.0
            taps.append(w)
            total += w
        for j, w in enumerate(taps):
            if total != 0.0:
                w = w / total
            v = w * float(1 << PRECISION_BITS)
            coef[o, j] = int(v - 0.5) if w < 0 else int(v + 0.5)
        xmin[o], cnt[o] = lo, hi - lo
    return xmin, cnt, coef


class DeviceTransform:
    """Resize((resize_h, resize_w)) -> CenterCrop(crop) -> ToTensor -> Normalize over uint8 images [N,H,W,3] resident on the GPU."""

    def __init__(self, in_hw, resize, crop, device, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        import ctypes as C
        self.H, self.W = in_hw
        self.RH, self.RW = (resize, resize) if isinstance(resize, int) else resize
        self.crop = crop
        self.y0 = int(round((self.RH - crop) / 2.0))          # torchvision F.center_crop
        self.x0 = int(round((self.RW - crop) / 2.0))
        th = pil_bilinear_tables(self.W, self.RW)
        tv = pil_bilinear_tables(self.H, self.RH)
        self.kh, self.kv = th[2].shape[1], tv[2].shape[1]
        self.tab_h = [torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in th]
        self.tab_v = [torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in tv]
        self.mean = (C.c_float * 3)(*mean)
        self.std = (C.c_float * 3)(*std)

    def __call__(self, images: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
        from .. import _lib
        from ..engine import _ptr, _require_cuda, _stream_ptr
        _require_cuda(images)
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[-1] != 3 or not images.is_contiguous():
            raise ValueError('images must be a contiguous uint8 [N,H,W,3] tensor')
        index = index.to(images.device, torch.int64).contiguous()
        B = index.numel()
        out = torch.empty(B, 3, self.crop, self.crop, dtype=torch.float32, device=images.device)
        lib = _lib.load()
        with torch.cuda.device(images.device):
            _lib.check(lib.fsvit_image_transform_gather(
                _ptr(images), self.H, self.W, _ptr(index), B, _ptr(self.tab_h[0]), _ptr(self.tab_h[1]), _ptr(self.tab_h[2]), self.kh,
                _ptr(self.tab_v[0]), _ptr(self.tab_v[1]), _ptr(self.tab_v[2]), self.kv, self.y0, self.x0, self.crop, self.crop,
                self.mean, self.std, _ptr(out), _stream_ptr(images.device)))
        return out
