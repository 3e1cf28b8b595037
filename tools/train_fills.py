#!/usr/bin/env python3
"""Which torch-side kernels (fills, copies, casts) run inside one full fine-tune step, with their shapes: torch profiler over one
steady-state step of the bench's training leg."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from a3vlm_amd.train import TrainEngine
from a3vlm_amd.util import promote_trainable_params_to_fp32
from a3vlm_amd.optim import FusedAdamW

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
m, args = bench.build_model("7b", dev, 2048)
B, T = 8, 512
gen = torch.Generator(device=dev).manual_seed(100)
image = torch.randn(B, 3, 336, 336, device=dev, generator=gen).to(torch.bfloat16)
tokens = torch.randint(3, args.vocab_size, (B, T), device=dev, generator=gen)
tokens[:, 0] = 1
for n, p in m.named_parameters():
    p.requires_grad_(not n.startswith("clip."))
promote_trainable_params_to_fp32(m)
eng = TrainEngine(m)
params = [p for p in m.parameters() if p.requires_grad]
opt = FusedAdamW(params, lr=1e-5, betas=(0.9, 0.95), weight_decay=0.02, engine=eng)
step = bench._train_step_fn(eng, opt, None, params, tokens, tokens.clone(), image)
for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:25]:
    print(f"{e.key:28s} n={e.count:4d} dev_us={e.device_time_total:9.1f} shapes={str(e.input_shapes)[:110]}")
