"""-m gpu: ``a3v_preprocess_image`` (pad-to-square with the CLIP-mean fill -> bicubic resize -> /255 -> normalise, on the device)
against the PIL transform of the input contract (``T_padded_resize`` = data/transform.py:59-68 on PIL images): bit-identical
fp32 tensors for landscape / portrait / square inputs, up- and down-scaling, the demo render, and a bf16 output."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from a3vlm_amd.data.transform import GpuPaddedResize, T_padded_resize

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,size", [(336, 300, 448), (300, 336, 448), (336, 336, 336), (662, 620, 448), (97, 150, 224), (640, 480, 224)])
def test_matches_pil_padded_resize(w, h, size):
    rng = np.random.default_rng(w * 7 + h)
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    a[: h // 4, : w // 2] = 255
    a[h // 2:, : w // 3] = 0
    img = Image.fromarray(a)
    want = T_padded_resize(size)(img)
    got = GpuPaddedResize(size, "cuda")(img)
    assert got.shape == (3, size, size) and got.dtype == torch.float32
    assert torch.equal(got.cpu(), want)
    # uint8 HWC tensors are accepted as well (a decoded batch that already lives on the device)
    assert torch.equal(GpuPaddedResize(size, "cuda")(torch.from_numpy(a).cuda()).cpu(), want)


def test_demo_render_and_bf16(golden_dir):
    img = Image.open(os.path.join(golden_dir, "demo", "render_336x300.png")).convert("RGB")
    want = T_padded_resize(448)(img)
    assert torch.equal(GpuPaddedResize(448, "cuda")(img).cpu(), want)
    gb = GpuPaddedResize(448, "cuda", torch.bfloat16)(img)
    assert gb.dtype == torch.bfloat16 and torch.equal(gb.cpu(), want.to(torch.bfloat16))
    with pytest.raises(TypeError):
        GpuPaddedResize(224, "cuda")(torch.zeros(3, 8, 8))
