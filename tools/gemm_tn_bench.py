"""Weight-gradient product on the 7B shapes: a3v_gemm_tn on the token-major operands vs two a3v_transpose + a3v_gemm_nt.
Run on the GPU box: python tools/gemm_tn_bench.py [tokens]"""
import sys

import torch

from a3vlm_amd import ops

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
    Tp = (T + 63) // 64 * 64
    for name, N, K in [("wo", 4096, 4096), ("qkv", 12288, 4096), ("w13", 22016, 4096), ("w2", 4096, 11008)]:
        dy = (torch.randn(T, N, device=DEV) * 0.1).bfloat16()
        x = (torch.randn(T, K, device=DEV) * 0.1).bfloat16()
        g = torch.zeros(N, K, device=DEV)
        dyt, xt = torch.zeros(N, Tp, device=DEV, dtype=torch.bfloat16), torch.zeros(K, Tp, device=DEV, dtype=torch.bfloat16)

        def old():
            ops.transpose(dy, dyt, T, N, Tp)
            ops.transpose(x, xt, T, K, Tp)
            ops.gemm_nt(dyt, xt, g, residual=g, epilogue=ops.EPI_RES_F32)

        def nt_only():
            ops.gemm_nt(dyt, xt, g, residual=g, epilogue=ops.EPI_RES_F32)

        def new():
            ops.gemm_tn(dy, x, g, residual=g, epilogue=ops.EPI_RES_F32)

        a, b, c = t_us(old), t_us(nt_only), t_us(new)
        fl = 2.0 * T * N * K
        print(f"{name:4s} N={N:6d} K={K:6d} T={T}: transposes+nt {a:8.1f} us  nt alone {b:8.1f} us ({fl / b / 1e6:6.1f} TF)  "
              f"tn {c:8.1f} us ({fl / c / 1e6:6.1f} TF)", flush=True)


if __name__ == "__main__":
    main()
