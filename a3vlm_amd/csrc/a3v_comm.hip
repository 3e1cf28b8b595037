// The DP exchange step of the path behind the C-ABI (SURVEY 8(b) last row, 8(e)): one gradient bucket of the flat fp32 buffer
// averaged (or summed) over the ranks of an RCCL communicator, on the caller's stream -- what FSDP's reduce-scatter + all-gather of
// the reference amounts to under pure DP (main_finetune.py:241-263; reduce_dtype bf16 :251-255; util/misc.py:311-313 skips it on
// accumulation micro-steps, which is the CALLER's decision here).  The Python host drives RCCL through torch.distributed
// (a3vlm_amd/dp.py); this entry is for a host that owns an ncclComm_t itself.
//
// RCCL is not linked: ncclAllReduce is resolved at run time from the librccl the process already has (PyTorch-ROCm ships one), or
// from A3V_RCCL_LIB / the loader path.
#include "a3v_common.h"
#include <dlfcn.h>
#include <stdlib.h>

extern "C" int a3v_scale_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, float scale, void* stream);

namespace {
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
constexpr int NCCL_FLOAT32 = 7, NCCL_BFLOAT16 = 9, NCCL_SUM = 0, NCCL_AVG = 4;     // ncclDataType_t / ncclRedOp_t (nccl.h)

nccl_allreduce_fn resolve() {
  static nccl_allreduce_fn fn = nullptr;
  static bool tried = false;
  if (tried) return fn;
  tried = true;
  const char* names[4] = {getenv("A3V_RCCL_LIB"), "librccl.so", "librccl.so.1", nullptr};
  for (int pass = 0; pass < 2 && !fn; ++pass)              // pass 0: only a library that is already loaded (torch's), pass 1: load one
    for (int i = 0; i < 3 && !fn; ++i) {
      if (!names[i]) continue;
      void* h = dlopen(names[i], RTLD_NOW | (pass == 0 ? RTLD_NOLOAD : 0));
      if (h) fn = (nccl_allreduce_fn)dlsym(h, "ncclAllReduce");
    }
  return fn;
}
}  // namespace

extern "C" int a3v_rccl_available(void) { return resolve() != nullptr; }

extern "C" int a3v_grad_bucket_allreduce(void* comm, float* grad, int64_t n, void* wire_bf16, int average, void* stream) {
  if (!comm || !grad || n <= 0) return A3V_ERR_ARG;
  nccl_allreduce_fn ar = resolve();
  if (!ar) return A3V_ERR_ARG;                            // no RCCL in the process and none on the loader path
  const int op = average ? NCCL_AVG : NCCL_SUM;
  int rc;
  if (wire_bf16) {                                        // bf16 on the wire: one fused cast in, one widening cast out (dp.GradReducer)
    if ((rc = a3v_scale_cast(grad, A3V_F32, wire_bf16, A3V_BF16, n, 1.f, stream))) return rc;
    if (ar(wire_bf16, wire_bf16, (size_t)n, NCCL_BFLOAT16, op, comm, (hipStream_t)stream) != 0) return A3V_ERR_ARG;
    return a3v_scale_cast(wire_bf16, A3V_BF16, grad, A3V_F32, n, 1.f, stream);
  }
  return ar(grad, grad, (size_t)n, NCCL_FLOAT32, op, comm, (hipStream_t)stream) == 0 ? A3V_OK : A3V_ERR_ARG;
}
