#!/usr/bin/env python3
"""A/B on one box: bench forward (bs 8, 512-token prompt, 7B) with a Transformer attribute toggled.
usage: ab_forward.py <attr> [reps]   e.g. ab_forward.py _fuse_qkv_rope
       ab_forward.py env:A3V_GEMM_FAST_EPI [reps]   toggles an environment switch the library reads per launch (1 / 0)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

attr = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
B, T = 8, 512
m, args = bench.build_model("7b", dev, 2048)
gen = torch.Generator(device=dev).manual_seed(100)
image = torch.randn(B, 3, 336, 336, device=dev, generator=gen).to(torch.bfloat16)
tokens = torch.randint(3, args.vocab_size, (B, T), device=dev, generator=gen)
tokens[:, 0] = 1


def run(n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        m.forward_inference(tokens, 0, image)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    for _ in range(reps):
        for val in (True, False):
            if attr.startswith("env:"):
                os.environ[attr[4:]] = "1" if val else "0"
                __import__("a3vlm_amd.lib", fromlist=["load"]).load().a3v_reload_env()   # the library caches its switches
            else:
                setattr(m, attr, val)
            run(2)
            print(f"{attr}={val}: {run(8):.2f} ms/step", flush=True)
