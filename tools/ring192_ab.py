"""Same-box A/B: the ring GEMM on 192 x 256 tiles (EPI_TILE_192PP) against the auto dispatch (256 x 256 ring + split-K tail where the tile
count leaves rows beyond whole rounds) and the all-ring 256 x 256 launch.  usage (GPU box): PYTHONPATH=. python tools/ring192_ab.py"""
import torch
from a3vlm_amd import lib, ops

BF, DEV = torch.bfloat16, "cuda"


def timeit(fn, n=30):
    for _ in range(6):
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
    shapes = [(8728, 4096, 4096), (8728, 4096, 11008), (8728, 4096, 12288), (8728, 4096, 22016), (8728, 12288, 4096), (8728, 22016, 4096),
              (8728, 11008, 4096), (8192, 4096, 4096), (4616, 4096, 1024), (4616, 3072, 1024), (4616, 1024, 4096), (8728, 5120, 5120), (8728, 5120, 13824)]
    for (M, N, K) in shapes:
        a = torch.randn(M, K, generator=g).to(BF).to(DEV)
        w = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(DEV)
        r = torch.randn(M, N, generator=g).to(BF).to(DEV)
        o = torch.empty(M, N, dtype=BF, device=DEV)
        row = {}
        for rep in range(2):
            for name, tile in (("auto", 0), ("t192", lib.EPI_TILE_192PP), ("ring256", lib.EPI_TILE_256PP)):
                row.setdefault(name, []).append(timeit(lambda: ops.gemm_nt(a, w, o, residual=r, epilogue=tile | ops.EPI_RESIDUAL)))
        fl = 2.0 * M * N * K
        best = {k: min(v) for k, v in row.items()}
        print(f"({M}, {N}, {K}) auto {best['auto']:.1f} us ({fl / best['auto'] / 1e6:.0f} TF)  192-row tiles {best['t192']:.1f} us ({fl / best['t192'] / 1e6:.0f} TF)  "
              f"all-ring 256 {best['ring256']:.1f} us ({fl / best['ring256'] / 1e6:.0f} TF)", flush=True)


if __name__ == "__main__":
    main()
