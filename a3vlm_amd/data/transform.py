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


def get_transform(transform_type: str, size: int = 224):
    if transform_type == "padded_resize":
        return T_padded_resize(size)
    raise ValueError(f"unsupported transform type: {transform_type} (padded_resize is what A3VLM trains and evaluates with)")
