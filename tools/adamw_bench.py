"""AdamW step on 7B-sized fp32 state: torch.optim.AdamW(fused=True) vs a3vlm_amd.optim.FusedAdamW (with / without bf16 image)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from a3vlm_amd.optim import FusedAdamW  # noqa: E402
from tools.gemm_fp8_bench import t_us  # noqa: E402

DEV = "cuda:0"
shapes = [(4096, 4096)] * 8 + [(11008, 4096)] * 6 + [(4096, 11008)] * 3 + [(4096,)] * 8      # ~0.5 G parameters
ps = [torch.randn(*s, device=DEV).requires_grad_(True) for s in shapes]
for p in ps:
    p.grad = torch.randn_like(p) * 0.01
n = sum(p.numel() for p in ps)
o1 = torch.optim.AdamW(ps, lr=2e-5, betas=(0.9, 0.95), weight_decay=0.0, fused=True)
o2 = FusedAdamW(ps, lr=2e-5, betas=(0.9, 0.95), weight_decay=0.0)
imgs = {id(p): torch.empty(p.shape, dtype=torch.bfloat16, device=DEV) for p in ps}
o3 = FusedAdamW(ps, lr=2e-5, betas=(0.9, 0.95), weight_decay=0.0, image_of=lambda p: imgs[id(p)])
for name, o, b in (("torch fused", o1, 28), ("a3v_adamw", o2, 28), ("a3v_adamw + bf16 image", o3, 30)):
    t = t_us(o.step, n=5)
    print(f"{name:24s}: {t / 1e3:7.2f} ms for {n / 1e9:.2f} G params = {n * b / t / 1e6:5.2f} TB/s  (7B: {t / 1e3 * 6.74e9 / n:6.1f} ms)", flush=True)
