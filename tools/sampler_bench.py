"""Time a3v_sample_top_p / the generate-step launch alone (B rows x V logits), HIP events.  usage (GPU box): PYTHONPATH=. python tools/sampler_bench.py"""
import torch
from a3vlm_amd import ops

DEV = "cuda"


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n * 1e3


for B, V, std in ((8, 32000, 0.3), (8, 32000, 3.0), (1, 32000, 0.3), (32, 32000, 0.3)):
    g = torch.Generator(device=DEV).manual_seed(0)
    logits = torch.randn(B, V, device=DEV, generator=g) * std
    u = torch.rand(B, device=DEV, generator=g)
    out = torch.empty(B, dtype=torch.long, device=DEV)
    t = timeit(lambda: ops.sample_top_p(logits, 0.1, 0.75, u, out))
    t2 = timeit(lambda: ops.argmax(logits, out))
    print(f"B={B} V={V} logit std {std}: sample_top_p {t:.1f} us   argmax {t2:.1f} us", flush=True)
