#!/usr/bin/env python3
"""A/B on one box: greedy decode model step (bs 8, 7B, context ~1100) with an environment switch of the library toggled per
round inside ONE process (the library caches its switches: every toggle goes through lib.env, which calls a3v_reload_env -- a bare
os.environ write is NOT seen and both legs silently run the same kernels).  usage: ab_decode.py ENV_NAME [rounds]   e.g. ab_decode.py A3V_GEMV_NT"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from a3vlm_amd import lib, ops  # noqa: E402

name = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
B, T = 8, 512
m, args = bench.build_model("7b", dev, 2048)
gen = torch.Generator(device=dev).manual_seed(100)
image = torch.randn(B, 3, 336, 336, device=dev, generator=gen).to(torch.bfloat16)
tokens = torch.randint(3, args.vocab_size, (B, T), device=dev, generator=gen)
tokens[:, 0] = 1
nt = torch.empty(B, dtype=torch.long, device=dev)
cur = torch.empty(B, 1, dtype=torch.long, device=dev)


def run(n):
    lg = m.forward_inference(tokens, 0, image)
    for i in range(2):
        ops.argmax(lg, nt); cur[:, 0] = nt; lg = m.forward_inference(cur, T + i, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        ops.argmax(lg, nt); cur[:, 0] = nt; lg = m.forward_inference(cur, T + 2 + i, None)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    for mode in (None, "fp8"):
        m.quantize_decode_weights(mode)
        for _ in range(rounds):
            for val in ("1", "0"):
                with lib.env(**{name: val}):
                    ms = run(32)
                print(f"weights={'bf16' if mode is None else mode} {name}={val}: {ms:.3f} ms/step  {B / ms * 1e3:.0f} tok/s", flush=True)
