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


def test_batch_of_mixed_sizes_equals_the_pil_transform():
    """a3v_preprocess_batch (what the entry points' loaders call: workers decode, the device does the rest): 21 images of five different
    sizes (two launch chunks, tables of four different padded sides) in one call, bit-identical per image to T_padded_resize; bf16 too;
    a second batch through the same object reuses the pinned staging buffer."""
    rng = np.random.default_rng(5)
    dims = [(336, 300), (300, 336), (336, 336), (97, 150), (640, 480)]
    imgs = [rng.integers(0, 256, (dims[i % 5][1], dims[i % 5][0], 3), dtype=np.uint8) for i in range(21)]
    pre = GpuPaddedResize(224, "cuda")
    got = pre.batch([torch.from_numpy(a) for a in imgs])
    assert got.shape == (21, 3, 224, 224) and got.dtype == torch.float32
    tf = T_padded_resize(224)
    for i, a in enumerate(imgs):
        assert torch.equal(got[i].cpu(), tf(Image.fromarray(a))), i
    got2 = pre.batch([Image.fromarray(a) for a in imgs[:3]])
    assert torch.equal(got2.cpu(), got[:3].cpu())
    gb = GpuPaddedResize(224, "cuda", torch.bfloat16).batch(imgs[:5])
    assert torch.equal(gb.cpu(), got[:5].cpu().to(torch.bfloat16))
    with pytest.raises(TypeError):
        pre.batch([torch.zeros(3, 8, 8)])


def test_device_preprocess_loader_feeds_the_same_batches_as_the_pil_loader(golden_dir):
    """DataLoader(decode-only dataset, collate_raw_images) wrapped in DevicePreprocessLoader == DataLoader(PIL-transform dataset):
    same tuples, images equal bit for bit (what main_finetune.py --preprocess gpu / cpu feed the trainer)."""
    from a3vlm_amd.data.transform import DevicePreprocessLoader, collate_raw_images, get_transform
    img = Image.open(os.path.join(golden_dir, "demo", "render_336x300.png")).convert("RGB")
    variants = [img, img.crop((0, 0, 200, 300)), img.resize((150, 90)), img.transpose(Image.FLIP_LEFT_RIGHT)]

    class DS(torch.utils.data.Dataset):
        def __init__(self, transform):
            self.transform = transform

        def __len__(self):
            return 8

        def __getitem__(self, i):
            return torch.full((5,), i), torch.full((5,), -i), torch.ones(5), self.transform(variants[i % 4])
    cpu = torch.utils.data.DataLoader(DS(get_transform("padded_resize", 112)), batch_size=4)
    raw = torch.utils.data.DataLoader(DS(get_transform("padded_resize", 112, on_device=True)), batch_size=4, collate_fn=collate_raw_images)
    gpu = DevicePreprocessLoader(raw, 112, "cuda")
    assert len(gpu) == len(cpu) == 2
    for a, b in zip(cpu, gpu):
        assert len(a) == len(b) == 4
        for x, y in zip(a[:3], b[:3]):
            assert torch.equal(x, y)
        assert b[3].is_cuda and torch.equal(a[3], b[3].cpu())
