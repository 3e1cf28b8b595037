"""ctypes loader for tools/ubench/liba3v_probe.so: measurement / self-check kernels that are NOT part of the product C-ABI
(built by `make -C a3vlm_amd/csrc`, i.e. by __graft_entry__.build()).
  probe_mfma_tflops(iters)   (constant operands, random operands) TFLOP/s of a bare v_mfma_f32_16x16x32_bf16 stream on every CU: what the
                             matrix pipe delivers under the chip's power limit (bench.py `roofline.mfma_pipe_measured`; context for the
                             roofline fractions, which stay priced against the nominal 2.5 PF/s)
  probe_wave_reduce(x)       lanes whose wave_sum / wave_max (csrc/a3v_common.h: v_permlane swaps + DPP) differ in any bit from the
                             __shfl_xor butterflies they replaced (must be 0)"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def load():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liba3v_probe.so")
        if not os.path.isfile(path):
            raise RuntimeError(f"{path} not found: run `make -C a3vlm_amd/csrc`")
        import torch  # noqa: F401  (torch's HIP runtime first, as a3vlm_amd.lib does)
        lib = ctypes.CDLL(path)
        lib.a3v_probe_mfma_tflops.restype = ctypes.c_int
        lib.a3v_probe_mfma_tflops.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.a3v_probe_wave_reduce.restype = ctypes.c_int
        lib.a3v_probe_wave_reduce.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib = lib
    return _lib


def probe_mfma_tflops(iters: int = 20000):
    import torch
    cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    scratch = torch.empty(8 * cus * 256, dtype=torch.float32, device="cuda")
    res = (ctypes.c_float * 2)()
    rc = load().a3v_probe_mfma_tflops(int(iters), scratch.data_ptr(), ctypes.cast(res, ctypes.c_void_p), torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError(f"a3v_probe_mfma_tflops failed: {rc}")
    return float(res[0]), float(res[1])


def probe_wave_reduce(x):
    """x: [n_waves, 64] fp32 device tensor -> number of mismatching lanes."""
    import torch
    bad = torch.zeros(1, dtype=torch.int32, device=x.device)
    rc = load().a3v_probe_wave_reduce(x.data_ptr(), x.shape[0], bad.data_ptr(), torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError(f"a3v_probe_wave_reduce failed: {rc}")
    torch.cuda.synchronize()
    return int(bad.item())
