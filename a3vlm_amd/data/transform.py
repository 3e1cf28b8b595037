"""Image preprocessing of the reference (``data/transform.py:13-68``) without torchvision: PadToSquare with the
CLIP-mean fill -> PIL bicubic resize of the (square) image to ``size`` -> ToTensor (/255, CHW) -> Normalize with
the CLIP mean / std.  torchvision's ``Resize`` on a PIL image IS ``Image.resize(..., BICUBIC)``, so this follows
the same pixels; parity with the reference transform is unpinned only because torchvision is not installed here."""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch
from PIL import Image

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class PadToSquare:
    def __init__(self, background_color: Tuple[float, float, float]):
        self.bg_color = tuple(int(x * 255) for x in background_color)

    def __call__(self, img: Image.Image) -> Image.Image:
        w, h = img.size
        if w == h:
            return img
        side = max(w, h)
        out = Image.new(img.mode, (side, side), self.bg_color)
        out.paste(img, (0, (w - h) // 2) if w > h else ((h - w) // 2, 0))
        return out


def _resize_short_edge(img: Image.Image, size: int) -> Image.Image:
    w, h = img.size                      # torchvision.transforms.Resize(int): shorter edge -> size, aspect kept
    if w <= h:
        return img.resize((size, max(1, int(size * h / w))), Image.BICUBIC)
    return img.resize((max(1, int(size * w / h)), size), Image.BICUBIC)


def to_normalized_tensor(img: Image.Image) -> torch.Tensor:
    a = np.asarray(img.convert("RGB"), dtype=np.uint8)
    t = torch.from_numpy(a.copy()).permute(2, 0, 1).float().div_(255.0)
    mean = torch.tensor(CLIP_MEAN).view(3, 1, 1)
    std = torch.tensor(CLIP_STD).view(3, 1, 1)
    return (t - mean) / std


class T_padded_resize:
    def __init__(self, size: int = 224):
        self.size = size
        self.pad = PadToSquare(CLIP_MEAN)

    def __call__(self, img: Image.Image) -> torch.Tensor:
        return to_normalized_tensor(_resize_short_edge(self.pad(img), self.size))


class T_resized_center_crop:
    """Resize(shorter edge -> size, bicubic) + CenterCrop(size) (transform.py:48-56)."""

    def __init__(self, size: int = 224):
        self.size = size

    def __call__(self, img: Image.Image) -> torch.Tensor:
        img = _resize_short_edge(img, self.size)
        w, h = img.size
        left, top = int(round((w - self.size) / 2.0)), int(round((h - self.size) / 2.0))
        return to_normalized_tensor(img.crop((left, top, left + self.size, top + self.size)))


class T_random_resized_crop:
    """RandomResizedCrop(size, scale=(0.9, 1.0), ratio=(3/4, 4/3), bicubic) (transform.py:39-45) with torchvision's
    sampling procedure: 10 tries of area ~ U(scale)*A, log-ratio ~ U(log r0, log r1) drawn from torch's global RNG,
    then the centre-crop-to-ratio fallback."""

    def __init__(self, size: int = 224, scale=(0.9, 1.0), ratio=(0.75, 1.3333)):
        self.size, self.scale, self.ratio = size, scale, ratio

    def get_params(self, w: int, h: int):
        import math
        area = h * w
        log_ratio = torch.log(torch.tensor(self.ratio))
        for _ in range(10):
            target = area * torch.empty(1).uniform_(self.scale[0], self.scale[1]).item()
            ar = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
            cw, ch = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
            if 0 < cw <= w and 0 < ch <= h:
                top = torch.randint(0, h - ch + 1, size=(1,)).item()
                left = torch.randint(0, w - cw + 1, size=(1,)).item()
                return top, left, ch, cw
        in_ratio = float(w) / float(h)
        if in_ratio < min(self.ratio):
            cw, ch = w, int(round(w / min(self.ratio)))
        elif in_ratio > max(self.ratio):
            ch, cw = h, int(round(h * max(self.ratio)))
        else:
            cw, ch = w, h
        return (h - ch) // 2, (w - cw) // 2, ch, cw

    def __call__(self, img: Image.Image) -> torch.Tensor:
        top, left, ch, cw = self.get_params(*img.size)
        img = img.crop((left, top, left + cw, top + ch)).resize((self.size, self.size), Image.BICUBIC)
        return to_normalized_tensor(img)


def get_transform(transform_type: str, size: int = 224, on_device: bool = False):
    """``on_device`` (padded_resize only): the DataLoader workers only DECODE (``T_decode_rgb``: uint8 HWC); pad / bicubic resize /
    ToTensor / Normalize run on the GPU over the whole batch (``GpuPaddedResize.batch`` through ``DevicePreprocessLoader``) and give
    bit for bit the tensor ``T_padded_resize`` gives."""
    if on_device and transform_type == "padded_resize":
        return T_decode_rgb(size)
    if transform_type == "random_resized_crop":
        return T_random_resized_crop(size)
    if transform_type == "resized_center_crop":
        return T_resized_center_crop(size)
    if transform_type == "padded_resize":
        return T_padded_resize(size)
    raise ValueError("unknown transform type: transform_type")


# ------------------------------------------------------------------------------------------------------------------------
# The same padded-resize transform on the device (a3v_preprocess_image): Pillow's 8-bit two-pass bicubic resampling with
# host-computed fixed-point coefficient tables, the padded square never materialised, normalisation fused into the second pass.
# ------------------------------------------------------------------------------------------------------------------------
_PRECISION_BITS = 32 - 8 - 2          # Pillow Resample.c: 8-bit samples, 2 bits of headroom for the bicubic overshoot


def _bicubic(x: float, a: float = -0.5) -> float:
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pillow_bicubic_coeffs(in_size: int, out_size: int):
    """(coefficients int32 [out_size, ksize], bounds int32 [out_size, 2] = (first input index, tap count)) of one resampling pass
    from ``in_size`` to ``out_size`` samples, as Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` build them: filter
    support 2 (x the down-scaling factor), weights normalised in double precision, then rounded half away from zero to 2^22."""
    import math
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)                                   # Pillow accumulates in this order, in double
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << _PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


class T_decode_rgb:
    """Worker-side half of the on-device padded_resize: the decoded RGB pixels as uint8 [H, W, 3], nothing else."""
    deferred = "padded_resize"

    def __init__(self, size: int = 224):
        self.size = size

    def __call__(self, img: Image.Image) -> torch.Tensor:
        return torch.from_numpy(np.asarray(img.convert("RGB"), dtype=np.uint8).copy())


def _is_raw_image(x) -> bool:
    return torch.is_tensor(x) and x.dtype == torch.uint8 and x.dim() == 3 and x.shape[-1] == 3


def collate_raw_images(samples):
    """default_collate for everything but raw decoded images (uint8 HWC tensors of differing sizes), which stay a list per field."""
    from torch.utils.data import default_collate
    if not isinstance(samples[0], (tuple, list)):
        return default_collate(samples)
    fields = list(zip(*samples))
    return type(samples[0])([list(f) if _is_raw_image(f[0]) else default_collate(list(f)) for f in fields])


class DevicePreprocessLoader:
    """Wraps a DataLoader whose dataset decodes only (``T_decode_rgb``): every list of raw images in a batch becomes the normalised
    ``[B, 3, size, size]`` tensor on the device (one host-to-device copy + two launches per 16 images).  ``len`` and the sampler
    contract are the inner loader's."""

    def __init__(self, loader, size: int, device, dtype: torch.dtype = torch.float32):
        self.loader, self.pre = loader, GpuPaddedResize(size, device, dtype)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for batch in self.loader:
            if isinstance(batch, (tuple, list)):
                batch = type(batch)([self.pre.batch(f) if isinstance(f, list) and f and _is_raw_image(f[0]) else f for f in batch])
            yield batch


class GpuPaddedResize:
    """``T_padded_resize`` evaluated on the device: takes a decoded RGB image (PIL image, HWC uint8 array or tensor) and returns the
    normalised ``[3, size, size]`` tensor on ``device`` -- bit-identical to the PIL path (tests/test_gpu_preprocess.py).  The
    coefficient tables depend only on (padded side, size) and are cached on the device."""

    def __init__(self, size: int = 224, device="cuda", dtype: torch.dtype = torch.float32):
        self.size, self.device, self.dtype = size, torch.device(device), dtype
        self.fill = tuple(int(x * 255) for x in CLIP_MEAN)
        self._tables = {}

    def _table(self, side: int):
        if side not in self._tables:
            kk, bounds = pillow_bicubic_coeffs(side, self.size)
            self._tables[side] = (torch.from_numpy(kk).to(self.device), torch.from_numpy(bounds).to(self.device), kk.shape[1])
        return self._tables[side]

    def batch(self, images) -> torch.Tensor:
        """A list of decoded RGB images (uint8 [H, W, 3] tensors / arrays, any sizes) -> ``[n, 3, size, size]`` on the device: the
        pixels of all images cross PCIe in ONE copy of a pinned staging buffer, then a3v_preprocess_batch (two launches per 16)."""
        import ctypes
        from .. import lib as _l
        from ..ops import dt
        imgs = [torch.as_tensor(np.asarray(im.convert("RGB"), dtype=np.uint8)) if isinstance(im, Image.Image) else torch.as_tensor(im) for im in images]
        for t in imgs:
            if not _is_raw_image(t):
                raise TypeError("GpuPaddedResize.batch takes RGB images as uint8 [H, W, 3]")
        n = len(imgs)
        sizes = [int(t.numel()) for t in imgs]
        offs = [0]
        for sz in sizes:
            offs.append(offs[-1] + (sz + 255) // 256 * 256)
        stage = getattr(self, "_stage", None)
        if stage is None or stage.numel() < offs[-1]:
            stage = self._stage = torch.empty(max(offs[-1], 1 << 20), dtype=torch.uint8).pin_memory()
        ev = getattr(self, "_copied", None)
        if ev is not None:
            ev.synchronize()                       # the previous batch's copy has left the staging buffer
        for t, o, sz in zip(imgs, offs, sizes):
            stage[o:o + sz].copy_(t.reshape(-1))
        dbuf = getattr(self, "_dbuf", None)
        if dbuf is None or dbuf.numel() < offs[-1]:
            dbuf = self._dbuf = torch.empty(stage.numel(), dtype=torch.uint8, device=self.device)
        dbuf[:offs[-1]].copy_(stage[:offs[-1]], non_blocking=True)
        self._copied = torch.cuda.Event()
        self._copied.record()
        descs = (_l.ImageDesc * n)()
        max_side = 0
        for i, t in enumerate(imgs):
            H, W = int(t.shape[0]), int(t.shape[1])
            side = max(H, W)
            kk, bounds, ksize = self._table(side)
            px, py = ((side - W) // 2, 0) if H > W else (0, (side - H) // 2)
            d = descs[i]
            d.src, d.coeffs, d.bounds = dbuf.data_ptr() + offs[i], kk.data_ptr(), bounds.data_ptr()
            d.H, d.W, d.side, d.pad_x, d.pad_y, d.ksize = H, W, side, px, py, ksize
            max_side = max(max_side, side)
        tmp_stride = max_side * self.size * 3
        tmp = torch.empty(n * tmp_stride, dtype=torch.uint8, device=self.device)
        out = torch.empty(n, 3, self.size, self.size, dtype=self.dtype, device=self.device)
        fill = (ctypes.c_int * 3)(*self.fill)
        mean = (ctypes.c_float * 3)(*CLIP_MEAN)
        std = (ctypes.c_float * 3)(*CLIP_STD)
        rc = _l.load().a3v_preprocess_batch(ctypes.cast(descs, ctypes.c_void_p), n, ctypes.cast(fill, ctypes.c_void_p), self.size, tmp.data_ptr(),
                                            tmp_stride, out.data_ptr(), 3 * self.size * self.size, dt(out), ctypes.cast(mean, ctypes.c_void_p),
                                            ctypes.cast(std, ctypes.c_void_p), torch.cuda.current_stream().cuda_stream)
        _l.check(rc, "a3v_preprocess_batch")
        return out

    def __call__(self, img) -> torch.Tensor:
        import ctypes
        from .. import lib as _l
        from ..ops import dt
        if isinstance(img, Image.Image):
            img = np.asarray(img.convert("RGB"), dtype=np.uint8)
        src = torch.as_tensor(img)
        if src.dtype != torch.uint8 or src.dim() != 3 or src.shape[2] != 3:
            raise TypeError("GpuPaddedResize takes one RGB image as uint8 [H, W, 3]")
        src = src.to(self.device, non_blocking=True).contiguous()
        H, W = int(src.shape[0]), int(src.shape[1])
        side = max(H, W)
        px, py = ((side - W) // 2, 0) if H > W else (0, (side - H) // 2)      # PadToSquare: centred on the shorter axis
        kk, bounds, ksize = self._table(side)
        tmp = torch.empty(side, self.size, 3, dtype=torch.uint8, device=self.device)
        out = torch.empty(3, self.size, self.size, dtype=self.dtype, device=self.device)
        fill = (ctypes.c_int * 3)(*self.fill)
        mean = (ctypes.c_float * 3)(*CLIP_MEAN)
        std = (ctypes.c_float * 3)(*CLIP_STD)
        rc = _l.load().a3v_preprocess_image(src.data_ptr(), H, W, side, px, py, ctypes.cast(fill, ctypes.c_void_p), kk.data_ptr(), bounds.data_ptr(),
                                            ksize, kk.data_ptr(), bounds.data_ptr(), ksize, self.size, tmp.data_ptr(), out.data_ptr(), dt(out),
                                            ctypes.cast(mean, ctypes.c_void_p), ctypes.cast(std, ctypes.c_void_p),
                                            torch.cuda.current_stream().cuda_stream)
        _l.check(rc, "a3v_preprocess_image")
        return out
