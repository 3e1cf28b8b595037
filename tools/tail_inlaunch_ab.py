"""Same-box A/B of the ring GEMM's rows beyond whole tile rounds: K-slice units inside the persistent launch (round 5, default)
against the separate split-K tail launch + reduce pass (A3V_GEMM_TAIL_INLAUNCH=0), and M = 8192 (no such rows) as the floor.
usage (GPU box): python tools/tail_inlaunch_ab.py"""
import torch
from a3vlm_amd import lib, ops

BF, DEV = torch.bfloat16, "cuda"


def timeit(fn, n=40):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    for (M, N, K) in [(8728, 4096, 4096), (8728, 4096, 11008), (8728, 4096, 12288), (8728, 4096, 22016), (4616, 4096, 1024)]:
        a = torch.randn(M, K, generator=g).to(BF).to(DEV)
        w = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(DEV)
        r = torch.randn(M, N, generator=g).to(BF).to(DEV)
        o = torch.empty(M, N, dtype=BF, device=DEV)
        row = []
        for rep in range(2):
            for inl in ("1", "0"):
                with lib.env(A3V_GEMM_TAIL_INLAUNCH=inl):
                    row.append(timeit(lambda: ops.gemm_nt(a, w, o, residual=r)))
        for s in (3, 4, 5):
            with lib.env(A3V_GEMM_TAIL_SLICES=str(s)):
                row.append(timeit(lambda: ops.gemm_nt(a, w, o, residual=r)))
        for s in (3, 4, 5):
            with lib.env(A3V_GEMM_TAIL_SLICES=str(s), A3V_GEMM_TAIL_WT="1"):
                row.append(timeit(lambda: ops.gemm_nt(a, w, o, residual=r)))
        a0 = a[:8192] if M > 8192 else a[:4096]
        o0, r0 = o[:a0.shape[0]], r[:a0.shape[0]]
        floor = timeit(lambda: ops.gemm_nt(a0, w, o0, residual=r0))
        fl = 2.0 * M * N * K
        print(f"({M}, {N}, {K}) in-launch {row[0]:.1f} / {row[2]:.1f} us  separate tail {row[1]:.1f} / {row[3]:.1f} us  "
              f"slices 3,4,5 | written through 3,4,5: {' '.join(f'{x:.1f}' for x in row[4:])}  whole rounds only (M = {a0.shape[0]}) {floor:.1f} us  "
              f"-> {fl / row[0] / 1e6:.0f} TF vs {fl / row[1] / 1e6:.0f} TF", flush=True)


if __name__ == "__main__":
    main()
