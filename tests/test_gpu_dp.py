"""-m gpu: the DP bucket path on the device with RCCL (a one-rank group: the only multi-process layout a single-GPU box allows; the
two-rank arithmetic is covered on CPU over gloo in tests/test_dp_cpu.py).  What runs here is what runs at N = 8: per bucket, on the
reducer's side stream, scale+cast to the bf16 wire buffer -> all-reduce (AVG) -> widen back -> the clip's sums of squares, all queued
behind the bucket's producers without a host wait."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

from a3vlm_amd.dp import GradReducer, GradSquareSums, clip_grad_norm  # noqa: E402


class DeviceEngine:
    def __init__(self, sizes, device):
        self._flat = torch.zeros(sum(sizes), device=device)
        self._ranges, o = [], 0
        for i, s in enumerate(sizes):
            self._ranges.append((f"layer{i}", o, o + s))
            o += s
        self.on_layer_grads_ready = None

    def flat_grads(self):
        return self._flat

    def grad_ranges(self):
        return list(self._ranges)


@pytest.fixture(scope="module")
def rccl_single_rank():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29800 + os.getpid() % 100))
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    if created:
        dist.destroy_process_group()


@pytest.mark.parametrize("wire", [None, torch.bfloat16])
def test_bucket_path_on_the_device(rccl_single_rank, wire):
    dev = torch.device("cuda", 0)
    eng = DeviceEngine([1 << 20, 4096, 3 << 20, 0], dev)
    red = GradReducer(eng, rccl_single_rank, reduce_dtype=wire, reduce_single_rank=True)
    sq = GradSquareSums(eng, red)
    g = torch.Generator(device=dev).manual_seed(5)
    for step in range(2):
        local = torch.randn(eng.flat_grads().numel(), device=dev, generator=g)
        eng.flat_grads().copy_(local)
        for n, s, e in reversed(eng.grad_ranges()):
            eng.on_layer_grads_ready(n, s, e)
        red.finish()
        want = local if wire is None else local.to(wire).float()           # one rank: the average is the (wire-rounded) value itself
        assert torch.equal(eng.flat_grads(), want)
        params = [torch.nn.Parameter(torch.zeros(e - s, device=dev)) for _, s, e in eng.grad_ranges() if e > s]
        for p, (_, s, e) in zip(params, [r for r in eng.grad_ranges() if r[2] > r[1]]):
            p.grad = eng.flat_grads()[s:e]
        norm, coef = clip_grad_norm(params, 8.0, flat=eng.flat_grads(), defer=True, sumsq=sq)
        ref = torch.linalg.vector_norm(eng.flat_grads().double()).float()
        assert torch.allclose(norm, ref, rtol=1e-5), (float(norm), float(ref))
        assert torch.allclose(coef, torch.clamp(8.0 / (ref + 1e-6), max=1.0), rtol=1e-5)
