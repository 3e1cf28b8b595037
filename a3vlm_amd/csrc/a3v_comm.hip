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

typedef int (*nccl_reducescatter_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*nccl_count_fn)(void*, int*);

void* rccl_handle() {
  static void* h = nullptr;
  static bool tried = false;
  if (tried) return h;
  tried = true;
  const char* names[4] = {getenv("A3V_RCCL_LIB"), "librccl.so", "librccl.so.1", nullptr};
  for (int pass = 0; pass < 2 && !h; ++pass)               // pass 0: only a library that is already loaded (torch's), pass 1: load one
    for (int i = 0; i < 3 && !h; ++i) {
      if (!names[i]) continue;
      void* c = dlopen(names[i], RTLD_NOW | (pass == 0 ? RTLD_NOLOAD : 0));
      if (c && dlsym(c, "ncclAllReduce")) h = c;
    }
  return h;
}
nccl_allreduce_fn resolve() {
  static nccl_allreduce_fn fn = rccl_handle() ? (nccl_allreduce_fn)dlsym(rccl_handle(), "ncclAllReduce") : nullptr;
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

// ZeRO-1 (main_finetune.py:241-263, FSDP SHARD_GRAD_OP: gradients and optimizer state sharded over DP; a3vlm_amd/zero1.py drives the same
// two collectives through torch.distributed): the gradient bucket's sharded span reduce-scattered so that rank r receives the average of
// elements [r n_per_rank, (r + 1) n_per_rank), and the updated bf16 parameters of every rank's slice gathered back into the flat buffer.
extern "C" int a3v_grad_bucket_reduce_scatter(void* comm, const float* grad, int64_t n_per_rank, float* shard, void* wire_bf16, void* wire_shard_bf16,
                                              int average, void* stream) {
  if (!comm || !grad || !shard || n_per_rank <= 0 || ((wire_bf16 == nullptr) != (wire_shard_bf16 == nullptr))) return A3V_ERR_ARG;
  void* h = rccl_handle();
  nccl_reducescatter_fn rs = h ? (nccl_reducescatter_fn)dlsym(h, "ncclReduceScatter") : nullptr;
  nccl_count_fn cnt = h ? (nccl_count_fn)dlsym(h, "ncclCommCount") : nullptr;
  int world = 0;
  if (!rs || !cnt || cnt(comm, &world) != 0 || world < 1) return A3V_ERR_ARG;
  const int op = average ? NCCL_AVG : NCCL_SUM;
  int rc;
  if (wire_bf16) {                                        // bf16 on the wire (FSDP's reduce_dtype): cast the whole span in, widen the rank's slice out
    if ((rc = a3v_scale_cast(grad, A3V_F32, wire_bf16, A3V_BF16, n_per_rank * world, 1.f, stream))) return rc;
    if (rs(wire_bf16, wire_shard_bf16, (size_t)n_per_rank, NCCL_BFLOAT16, op, comm, (hipStream_t)stream) != 0) return A3V_ERR_ARG;
    return a3v_scale_cast(wire_shard_bf16, A3V_BF16, shard, A3V_F32, n_per_rank, 1.f, stream);
  }
  return rs(grad, shard, (size_t)n_per_rank, NCCL_FLOAT32, op, comm, (hipStream_t)stream) == 0 ? A3V_OK : A3V_ERR_ARG;
}

extern "C" int a3v_param_shard_all_gather(void* comm, const void* shard_bf16, void* flat_bf16, int64_t n_per_rank, void* stream) {
  if (!comm || !shard_bf16 || !flat_bf16 || n_per_rank <= 0) return A3V_ERR_ARG;
  void* h = rccl_handle();
  nccl_allgather_fn ag = h ? (nccl_allgather_fn)dlsym(h, "ncclAllGather") : nullptr;
  if (!ag) return A3V_ERR_ARG;
  return ag(shard_bf16, flat_bf16, (size_t)n_per_rank, NCCL_BFLOAT16, comm, (hipStream_t)stream) == 0 ? A3V_OK : A3V_ERR_ARG;
}

// ranks of an RCCL communicator as the library sees them (ncclCommCount): what bench.py's rccl_ranks counts through torch.distributed
extern "C" int a3v_rccl_comm_count(void* comm) {
  void* h = rccl_handle();
  nccl_count_fn cnt = h ? (nccl_count_fn)dlsym(h, "ncclCommCount") : nullptr;
  int world = 0;
  if (!comm || !cnt || cnt(comm, &world) != 0) return -1;
  return world;
}
