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


def get_transform(transform_type: str, size: int = 224):
    if transform_type == "random_resized_crop":
        return T_random_resized_crop(size)
    if transform_type == "resized_center_crop":
        return T_resized_center_crop(size)
    if transform_type == "padded_resize":
        return T_padded_resize(size)
    raise ValueError("unknown transform type: transform_type")
