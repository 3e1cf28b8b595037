"""fp8 W8A8 GEMM (MX-scaled MFMA) vs the bf16 kernel on the 7B prefill shapes.  python tools/gemm_fp8_bench.py [tokens]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from a3vlm_amd import ops  # noqa: E402

DEV = "cuda:0"


def t_us(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 8728
    zero = len(sys.argv) > 2 and sys.argv[2] == "zero"      # all-zero operands: the same instruction stream at minimal toggle power
    shapes = [("qkv", 12288, 4096), ("wo", 4096, 4096), ("w13", 22016, 4096), ("w2", 4096, 11008)]
    if T == 8192:
        shapes = [("sq4k", 8192, 4096), ("sq12k", 8192, 12288)]
    for name, N, K in shapes:
        a = (torch.randn(T, K, device=DEV) * (0.0 if zero else 0.5)).bfloat16()
        w = (torch.randn(N, K, device=DEV) * (0.0 if zero else 0.02)).bfloat16()
        out = torch.empty(T, N, device=DEV, dtype=torch.bfloat16)
        aq = torch.empty(T, K, device=DEV, dtype=torch.uint8)
        sa = torch.empty(T, device=DEV)
        wq = torch.empty(N, K, device=DEV, dtype=torch.uint8)
        sw = torch.empty(N, device=DEV)
        ops.quantize_rows_fp8(w, wq, sw)
        tq = t_us(lambda: ops.quantize_rows_fp8(a, aq, sa))
        tb = t_us(lambda: ops.gemm_nt(a, w, out))
        tf = t_us(lambda: ops.gemm_nt_fp8(aq, sa, wq, sw, out))
        fl = 2.0 * T * N * K
        print(f"{name:4s} N={N:6d} K={K:6d} T={T}: bf16 {tb:8.1f} us ({fl / tb / 1e6:6.1f} TF)  fp8 {tf:8.1f} us ({fl / tf / 1e6:6.1f} TF)  "
              f"quantise A {tq:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
