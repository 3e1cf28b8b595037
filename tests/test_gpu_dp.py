"""-m gpu: the DP bucket path on the device with RCCL (a one-rank group: the only multi-process layout a single-GPU box allows; the
two-rank arithmetic is covered on CPU over gloo in tests/test_dp_cpu.py).  What runs here is what runs at N = 8: per bucket, on the
reducer's side stream, scale+cast to the bf16 wire buffer -> all-reduce (AVG) -> widen back -> the clip's sums of squares, all queued
behind the bucket's producers without a host wait."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

from a3vlm_amd.dp import GradReducer, GradSquareSums, clip_grad_norm  # noqa: E402


class DeviceEngine:
    def __init__(self, sizes, device):
        self._flat = torch.zeros(sum(sizes), device=device)
        self._ranges, o = [], 0
        for i, s in enumerate(sizes):
            self._ranges.append((f"layer{i}", o, o + s))
            o += s
        self.on_layer_grads_ready = None

    def flat_grads(self):
        return self._flat

    def grad_ranges(self):
        return list(self._ranges)


@pytest.fixture(scope="module")
def rccl_single_rank():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29800 + os.getpid() % 100))
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    if created:
        dist.destroy_process_group()


@pytest.mark.parametrize("wire", [None, torch.bfloat16])
def test_bucket_path_on_the_device(rccl_single_rank, wire):
    dev = torch.device("cuda", 0)
    eng = DeviceEngine([1 << 20, 4096, 3 << 20, 0], dev)
    red = GradReducer(eng, rccl_single_rank, reduce_dtype=wire, reduce_single_rank=True)
    sq = GradSquareSums(eng, red)
    g = torch.Generator(device=dev).manual_seed(5)
    for step in range(2):
        local = torch.randn(eng.flat_grads().numel(), device=dev, generator=g)
        eng.flat_grads().copy_(local)
        for n, s, e in reversed(eng.grad_ranges()):
            eng.on_layer_grads_ready(n, s, e)
        red.finish()
        want = local if wire is None else local.to(wire).float()           # one rank: the average is the (wire-rounded) value itself
        assert torch.equal(eng.flat_grads(), want)
        params = [torch.nn.Parameter(torch.zeros(e - s, device=dev)) for _, s, e in eng.grad_ranges() if e > s]
        for p, (_, s, e) in zip(params, [r for r in eng.grad_ranges() if r[2] > r[1]]):
            p.grad = eng.flat_grads()[s:e]
        norm, coef = clip_grad_norm(params, 8.0, flat=eng.flat_grads(), defer=True, sumsq=sq)
        ref = torch.linalg.vector_norm(eng.flat_grads().double()).float()
        assert torch.allclose(norm, ref, rtol=1e-5), (float(norm), float(ref))
        assert torch.allclose(coef, torch.clamp(8.0 / (ref + 1e-6), max=1.0), rtol=1e-5)


# ------------------------------------------------------------------ two ranks == one rank with the double batch (real engine)
_EQ = dict(dim=128, n_layers=2, n_heads=4, n_kv_heads=4, vocab_size=256, multiple_of=64, max_seq_len=256)


def _eq_model(dev):
    from a3vlm_amd.model.LLM import llama_ens5 as plugin
    from a3vlm_amd.util import promote_trainable_params_to_fp32
    from oracle import ref_cpu
    m = plugin.Transformer(plugin.ModelArgs(**_EQ), with_visual=False)
    m.load_state_dict(ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(**_EQ), seed=11, std=0.05))
    m.to(torch.float32).to(dev)
    promote_trainable_params_to_fp32(m)
    return m


def _eq_data():
    g = torch.Generator().manual_seed(77)
    ex = torch.randint(3, 256, (2, 4, 40), generator=g)      # [micro-step][sample of the global batch][token]; every label valid
    ex[:, :, 0] = 1
    return ex


def _eq_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from a3vlm_amd.train import TrainEngine
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        eng = TrainEngine(_eq_model(dev), torch.float32)
        red = GradReducer(eng, dist, reduce_dtype=None)
        ex = _eq_data()
        for micro in range(2):                                 # accum_iter = 2: the first micro-step is a no_sync step
            red.enabled = micro == 1
            mine = ex[micro, rank::world].to(dev)              # the sampler's strided shard of the global batch
            eng.forward_loss(mine, mine, None)
            eng.backward(0.5)
        red.finish()
        torch.cuda.synchronize()
        torch.save(eng.flat_grads().cpu(), os.path.join(outdir, f"grads{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_one_rank_with_the_double_batch(tmp_path):
    """The DP step on the real engine: two ranks (both on cuda:0, gloo), accum_iter 2, per-layer buckets all-reduced on the side
    stream during the boundary micro-step's backward -- the averaged gradients equal those of ONE rank that sees both ranks'
    samples in each micro-step (equal token counts per sample, fp32 kernels: the only difference is the summation order)."""
    import torch.multiprocessing as mp
    from a3vlm_amd.train import TrainEngine
    world, port = 2, 29900 + os.getpid() % 90
    mp.get_context("spawn")
    mp.spawn(_eq_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g0, g1 = (torch.load(os.path.join(str(tmp_path), f"grads{r}.pt")) for r in range(world))
    assert torch.equal(g0, g1), "replicas must hold identical gradients after the all-reduce"
    dev = torch.device("cuda", 0)
    eng = TrainEngine(_eq_model(dev), torch.float32)
    ex = _eq_data()
    for micro in range(2):
        both = ex[micro].to(dev)
        eng.forward_loss(both, both, None)
        eng.backward(0.5)
    want = eng.flat_grads().cpu()
    assert float(want.abs().max()) > 1e-4
    assert torch.allclose(g0, want, rtol=2e-4, atol=2e-6 * float(want.abs().max()) + 1e-7), float((g0 - want).abs().max())


def test_c_abi_bucket_allreduce_on_a_raw_rccl_communicator():
    """a3v_grad_bucket_allreduce with an ncclComm_t the HOST created (no torch.distributed): a one-rank communicator from the RCCL
    that ships with PyTorch-ROCm, fp32 wire (bucket unchanged by an average over one rank) and bf16 wire (bucket = its bf16
    rounding), sum and average, on a side stream."""
    import ctypes
    from a3vlm_amd import lib as _l
    lib = _l.load()
    assert lib.a3v_rccl_available() == 1
    rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        g = torch.Generator(device="cuda").manual_seed(4)
        n = (3 << 20) + 64
        grad = torch.randn(n, device="cuda", generator=g)
        want = grad.clone()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        for avg in (1, 0):
            _l.check(lib.a3v_grad_bucket_allreduce(comm, grad.data_ptr(), n, None, avg, st.cuda_stream), "a3v_grad_bucket_allreduce")
            st.synchronize()
            assert torch.equal(grad, want)
        wire = torch.empty(n, dtype=torch.bfloat16, device="cuda")
        _l.check(lib.a3v_grad_bucket_allreduce(comm, grad.data_ptr(), n, wire.data_ptr(), 1, st.cuda_stream), "a3v_grad_bucket_allreduce")
        st.synchronize()
        assert torch.equal(grad, want.to(torch.bfloat16).float()) and torch.equal(wire, want.to(torch.bfloat16))
        assert lib.a3v_grad_bucket_allreduce(None, grad.data_ptr(), n, None, 1, st.cuda_stream) != 0
        # ZeRO-1's two collectives through the same boundary (one rank: the slice is the whole span)
        assert lib.a3v_rccl_comm_count(comm) == 1 and lib.a3v_rccl_comm_count(None) == -1
        src = torch.randn(n, device="cuda", generator=g)
        shard = torch.zeros(n, device="cuda")
        st.wait_stream(torch.cuda.current_stream())
        _l.check(lib.a3v_grad_bucket_reduce_scatter(comm, src.data_ptr(), n, shard.data_ptr(), None, None, 1, st.cuda_stream), "a3v_grad_bucket_reduce_scatter")
        st.synchronize()
        assert torch.equal(shard, src)
        wshard = torch.empty(n, dtype=torch.bfloat16, device="cuda")
        _l.check(lib.a3v_grad_bucket_reduce_scatter(comm, src.data_ptr(), n, shard.data_ptr(), wire.data_ptr(), wshard.data_ptr(), 1, st.cuda_stream),
                 "a3v_grad_bucket_reduce_scatter")
        st.synchronize()
        assert torch.equal(shard, src.to(torch.bfloat16).float())
        assert lib.a3v_grad_bucket_reduce_scatter(comm, src.data_ptr(), n, shard.data_ptr(), wire.data_ptr(), None, 1, st.cuda_stream) != 0
        flat = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
        _l.check(lib.a3v_param_shard_all_gather(comm, wshard.data_ptr(), flat.data_ptr(), n, st.cuda_stream), "a3v_param_shard_all_gather")
        st.synchronize()
        assert torch.equal(flat, wshard)
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)
