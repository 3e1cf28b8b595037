import sys, os, json
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from a3vlm_amd import ops
from a3vlm_amd.lib import load
dev = "cuda"
def setv(v, grp):
    os.environ["A3V_ATTN_PERSIST"] = v
    os.environ["A3V_ATTN_HEAD_GROUP"] = str(grp)
    load().a3v_reload_env()
for (B, S, H, hd) in [(8, 1091, 32, 128), (4, 2048, 32, 128), (8, 2048, 32, 128), (8, 1091, 40, 128)]:
    sp = (S + 63) // 64 * 64
    q = torch.randn(B, S, H, hd, device=dev, dtype=torch.bfloat16)
    k = torch.randn(B, H, sp, hd, device=dev, dtype=torch.bfloat16)
    vt = torch.randn(B, H, hd, sp, device=dev, dtype=torch.bfloat16)
    st = (S*H*hd, H*hd, hd, H*sp*hd, sp*hd, hd, H*hd*sp, hd*sp, sp, S*H*hd, H*hd, hd)
    o = torch.empty_like(q); lse = torch.empty(B, H, S, device=dev)
    variants = [(p, g) for p in ("0", "1") for g in (8, 16, 32)]
    times = {v: [] for v in variants}
    for r in range(5):
        for v in variants:
            setv(*v)
            f = lambda: ops.attention_lse(q, k, vt, o, lse, B, S, S, H, H, hd, st, True)
            f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 5 * 1e3)
    print(B, S, H, {f"p{v[0]}g{v[1]}": round(sorted(t)[2], 1) for v, t in times.items()}, flush=True)
