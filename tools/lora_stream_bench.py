"""EXPERIMENT driver (needs `make -C a3vlm_amd/csrc EXPERIMENTS=1`): the streaming forms of the rank-64 LoRA products
(csrc/a3v_skinny.hip: a3v_gemm_nt_skinny64 + a3v_skinny_reduce, a3v_gemm_tn_strip2) against the product kernels (a3v_gemm_nt_splitk,
a3v_gemm_tn_strip) at the 7B step's shapes: correctness against torch fp32 on the same bf16 operands, TB/s of the streamed operand,
and (ABL=1) the ablations A3V_SK_ABL = 1 no small-operand loads | 2 no X loads | 4 no LDS reads / MFMA."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from a3vlm_amd import lib as _lib  # noqa: E402
from a3vlm_amd import ops  # noqa: E402
from tools.gemm_fp8_bench import t_us  # noqa: E402

DEV, BF, T = "cuda:0", torch.bfloat16, int(os.environ.get("T", 8728))
Tp = (T + 63) // 64 * 64
so = ctypes.CDLL(_lib.LIB_PATH)
P, L, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
so.a3v_gemm_nt_skinny64.argtypes = [P, L, P, L, P, I, I, I, P]
so.a3v_skinny_reduce.argtypes = [P, I, I, P, L, P, L, P]
so.a3v_gemm_tn_strip2.argtypes = [P, L, P, L, P, I, I, I, I, P]
lib = _lib.load()


def st():
    return torch.cuda.current_stream().cuda_stream


def nt_new(x, a, out, out_t, sc, S):
    M, K = x.shape
    assert so.a3v_gemm_nt_skinny64(x.data_ptr(), x.stride(0), a.data_ptr(), a.stride(0), sc.data_ptr(), M, K, S, st()) == 0
    assert so.a3v_skinny_reduce(sc.data_ptr(), S, M, out.data_ptr(), out.stride(0), out_t.data_ptr(), out_t.stride(0), st()) == 0


def strip_new(tT, x, out, sc, S):
    Kt, N = x.shape
    assert so.a3v_gemm_tn_strip2(tT.data_ptr(), tT.stride(0), x.data_ptr(), x.stride(0), sc.data_ptr(), 64, N, Kt, S, st()) == 0
    assert lib.a3v_splitk_reduce(sc.data_ptr(), S, 64, N, out.data_ptr(), out.stride(0), ops.dt(out), 0, st()) == 0


def env(**kw):
    for k, v in kw.items():
        os.environ[k] = str(v)
    lib.a3v_reload_env()


for K in (4096, 11008, 12288, 22016):
    g = torch.Generator(device=DEV).manual_seed(K)
    x = (torch.randn(T, K + 64, device=DEV, generator=g) * 0.5).to(BF)[:, :K]        # strided like the step's K-extended buffers
    a = (torch.randn(64, K, device=DEV, generator=g) * 0.02).to(BF)
    mb = T * K * 2 / 1e6
    want = x.float() @ a.float().t()
    t_old = torch.empty(T, 64, device=DEV, dtype=BF)
    sc = torch.empty(64 * 64 * K, device=DEV, dtype=torch.float32)
    us_old = t_us(lambda: ops.gemm_nt_splitk(x, a, t_old, sc, 4))
    for S in (3, 4, 8):
        t_new = torch.zeros(T, 64, device=DEV, dtype=BF)
        tT = torch.full((64, Tp), 7.0, device=DEV, dtype=BF)
        nt_new(x, a, t_new, tT, sc, S)
        torch.cuda.synchronize()
        err = float((t_new.float() - want).abs().max() / want.abs().max())
        okT = bool(torch.equal(tT[:, :T], t_new.t())) and bool((tT[:, T:] == 0).all())
        us = t_us(lambda: nt_new(x, a, t_new, tT, sc, S))
        print(f"nt   [T,{K}] x [64,{K}]^T  S={S}: {us:6.1f} us ({mb / us:.2f} TB/s)  product kernel S=4 {us_old:6.1f} us ({mb / us_old:.2f})  rel err {err:.1e} "
              f"(product {float((t_old.float() - want).abs().max() / want.abs().max()):.1e})  transposed copy ok={okT}", flush=True)
    want2 = t_new.float().t() @ x.float()
    tiles = (K + 127) // 128
    S_old = max(1, min(64, (T + 63) // 64, -(-512 // tiles)))
    g_old = torch.empty(64, K, device=DEV, dtype=torch.float32)
    us_o = t_us(lambda: ops.gemm_tn_strip(t_new, x, g_old, sc, S_old))
    for wide in (0, 1):
        tl = (K + (255 if wide else 127)) // (256 if wide else 128)
        for S in sorted({max(1, min(64, (T + 63) // 64, -(-wb // tl))) for wb in (256, 384, 512, 768)}):
            for stg in (3, 4):
                env(A3V_STRIP2_STAGES=stg, A3V_STRIP2_WIDE=wide)
                g_new = torch.zeros(64, K, device=DEV, dtype=torch.float32)
                strip_new(tT, x, g_new, sc, S)
                torch.cuda.synchronize()
                err = float((g_new - want2).abs().max() / want2.abs().max())
                us = t_us(lambda: strip_new(tT, x, g_new, sc, S))
                print(f"   strip [64,{K}] over T  panel {512 if wide else 256} B  S={S:2d} stages={stg}: {us:6.1f} us ({mb / us:.2f} TB/s)  product kernel S={S_old} "
                      f"{us_o:6.1f} us ({mb / us_o:.2f})  rel err {err:.1e}", flush=True)
    if os.environ.get("ABL") == "1" and K in (4096, 22016):
        env(A3V_STRIP2_STAGES=4, A3V_STRIP2_WIDE=0)
        S2 = max(1, -(-384 // tiles))
        for abl in (0, 1, 2, 4, 5, 6, 7):
            env(A3V_SK_ABL=abl)
            us1 = t_us(lambda: so.a3v_gemm_nt_skinny64(x.data_ptr(), x.stride(0), a.data_ptr(), a.stride(0), sc.data_ptr(), T, K, 4, st()))
            us2 = t_us(lambda: so.a3v_gemm_tn_strip2(tT.data_ptr(), tT.stride(0), x.data_ptr(), x.stride(0), sc.data_ptr(), 64, K, T, S2, st()))
            print(f"   ablation {abl}: nt S=4 {us1:6.1f} us ({mb / us1:5.2f} TB/s)   strip S={S2} {us2:6.1f} us ({mb / us2:5.2f} TB/s)", flush=True)
        env(A3V_SK_ABL=0)
