"""CPU: the host restatement of Pillow's 8-bit bicubic resampling tables (``pillow_bicubic_coeffs``, what the device kernel
``a3v_preprocess_image`` consumes) applied in numpy reproduces ``PIL.Image.resize(BICUBIC)`` bit for bit -- down- and up-scaling,
odd sizes, and the padded demo render of tests/golden/demo."""
import os

import numpy as np
import pytest
from PIL import Image

from a3vlm_amd.data.transform import CLIP_MEAN, PadToSquare, pillow_bicubic_coeffs


def _resample(img: np.ndarray, out: int) -> np.ndarray:
    """Two passes (horizontal, then vertical) over an HWC uint8 square image with the integer arithmetic of the kernel."""
    side = img.shape[0]
    kk, bounds = pillow_bicubic_coeffs(side, out)
    tmp = np.zeros((side, out, 3), dtype=np.uint8)
    for x in range(out):
        lo, n = bounds[x]
        acc = (1 << 21) + np.tensordot(img[:, lo:lo + n, :].astype(np.int64), kk[x, :n].astype(np.int64), axes=([1], [0]))
        tmp[:, x, :] = np.clip(acc >> 22, 0, 255)
    res = np.zeros((out, out, 3), dtype=np.uint8)
    for y in range(out):
        lo, n = bounds[y]
        acc = (1 << 21) + np.tensordot(tmp[lo:lo + n].astype(np.int64), kk[y, :n].astype(np.int64), axes=([0], [0]))
        res[y] = np.clip(acc >> 22, 0, 255)
    return res


@pytest.mark.parametrize("side,out", [(336, 448), (336, 224), (300, 224), (97, 224), (662, 448), (224, 224), (50, 37)])
def test_integer_resampling_equals_pillow(side, out):
    rng = np.random.default_rng(side * 1000 + out)
    img = rng.integers(0, 256, (side, side, 3), dtype=np.uint8)
    img[: side // 3] = rng.integers(0, 256, (1, 1, 3), dtype=np.uint8)        # flat area + noise: exercises the clip at both ends
    want = np.asarray(Image.fromarray(img).resize((out, out), Image.BICUBIC))
    assert np.array_equal(_resample(img, out), want)


def test_padded_demo_render(golden_dir):
    img = Image.open(os.path.join(golden_dir, "demo", "render_336x300.png")).convert("RGB")
    sq = PadToSquare(CLIP_MEAN)(img)
    want = np.asarray(sq.resize((448, 448), Image.BICUBIC))
    assert np.array_equal(_resample(np.asarray(sq), 448), want)
