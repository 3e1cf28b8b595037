#!/usr/bin/env python3
"""Is the greedy decode step launch-bound?  One model step of the bench's decode leg (7B, bs 8, context ~1100) captured in a HIP
graph at a fixed position and replayed, against the same step issued eagerly (one C call + the LM head / argmax launches).  A graph
replay has no host launch cost at all, so the difference is the most a device-resident `pos` + graph capture of the real loop could
buy.  usage: python tools/decode_graph_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from a3vlm_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
m, args = bench.build_model("7b", dev, 2048)
B, T = 8, 512
g = torch.Generator(device=dev).manual_seed(1)
img = torch.randn(B, 3, 336, 336, device=dev, generator=g).bfloat16()
tok = torch.randint(3, args.vocab_size, (B, T), device=dev, generator=g)
tok[:, 0] = 1
m.forward_inference(tok, 0, img)
pos = T
cur = torch.randint(3, args.vocab_size, (B, 1), device=dev, generator=g)
nt = torch.empty(B, dtype=torch.long, device=dev)


def step():
    lg = m.forward_inference(cur, pos, None)
    ops.argmax(lg, nt)


def timeit(fn, n=64):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


eager = timeit(step)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    step()
graph = timeit(gr.replay)
print(f"decode model step, bs {B}, context {T + m.image_words}: eager {eager * 1e3:.3f} ms ({B / eager:.0f} tok/s)   graph replay {graph * 1e3:.3f} ms "
      f"({B / graph:.0f} tok/s)   ratio {eager / graph:.3f}")
