#!/usr/bin/env python3
"""The 13B ZeRO-1 shard step of bench.py's m13b leg ALONE (rank 0 of DP-8 emulated on one GPU, micro-batch 8, no recompute), for
rocprofv3 --kernel-trace + tools/rocprof_summary.py --step-marker embed_assemble.   usage: m13b_zero1_profile.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
B, T = 8, 512
m, args = bench.build_model("13b", dev, 2048)
g = torch.Generator(device=dev).manual_seed(1)
img = torch.randn(B, 3, 336, 336, device=dev, generator=g).bfloat16()
tok = torch.randint(3, args.vocab_size, (B, T), device=dev, generator=g)
tok[:, 0] = 1
timer = bench.Timer(None, dev)
z = bench.zero1_leg(m, B, T, img, tok, steps, 1, timer, shard_of=8, recompute=False)
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in z.items() if k in ("sec", "loss", "hbm_gib")})
