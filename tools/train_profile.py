#!/usr/bin/env python3
"""Run a few fine-tuning steps of the bench workload in ONE mode (full | lora) -- for clean rocprofv3 kernel tables.
usage: train_profile.py full|lora [steps] [engine_attr=0|1 ...]   (e.g. tn_wgrad=0 for an A/B on one box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

mode = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for kv in sys.argv[3:]:
    from a3vlm_amd.train import TrainEngine
    name, val = kv.split("=")
    assert hasattr(TrainEngine, name), name
    setattr(TrainEngine, name, bool(int(val)))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
B, T = 8, 512
m, args = bench.build_model("7b", dev, 2048)
gen = torch.Generator(device=dev).manual_seed(100)
image = torch.randn(B, 3, 336, 336, device=dev, generator=gen)
tokens = torch.randint(3, args.vocab_size, (B, T), device=dev, generator=gen)
tokens[:, 0] = 1
if mode == "lora":
    sec, loss, mem, ntr = bench.lora_leg(m, args, B, T, image, tokens, steps, None, dev)
else:
    sec, loss, mem = bench.train_leg(m, args, B, T, image, tokens, steps, None, dev)
print(f"{mode}: {sec * 1e3:.1f} ms/step, loss {loss:.4f}, {mem:.1f} GiB")
