"""CPU (-m "not gpu"): the data-parallel machinery with world_size 2 over gloo, and the host-side training
contracts (sampler / lr schedule / weight-decay groups) against fixtures captured from the reference."""
import json
import os
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from a3vlm_amd.dp import FinetuneDistSampler, GradReducer, GradSquareSums, clip_grad_norm
from a3vlm_amd.util import add_weight_decay, adjust_learning_rate_epoch


class FakeEngine:
    """Stands in for TrainEngine: a flat fp32 grad buffer with per-layer ranges (no kernels on CPU)."""

    def __init__(self, sizes):
        self._flat = torch.zeros(sum(sizes))
        self._ranges, o = [], 0
        for i, s in enumerate(sizes):
            self._ranges.append((f"layer{i}", o, o + s))
            o += s
        self.on_layer_grads_ready = None

    def flat_grads(self):
        return self._flat

    def grad_ranges(self):
        return list(self._ranges)


def _worker(rank, world, port, reduce_dtype, outdir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = FakeEngine([1000, 64, 5000])
        red = GradReducer(eng, dist, reduce_dtype=reduce_dtype)
        g = torch.Generator().manual_seed(100 + rank)
        local = torch.randn(eng.flat_grads().numel(), generator=g)
        # micro-step 1 of an accumulation pair: no_sync
        red.enabled = False
        eng.flat_grads().add_(local)
        for n, s, e in reversed(eng.grad_ranges()):
            eng.on_layer_grads_ready(n, s, e)
        assert torch.equal(eng.flat_grads(), local), "no_sync micro-step must not touch the gradients"
        # boundary micro-step: buckets are reduced as they become ready (reverse layer order, like backward)
        red.enabled = True
        eng.flat_grads().add_(local)
        for n, s, e in reversed(eng.grad_ranges()):
            eng.on_layer_grads_ready(n, s, e)
        red.finish()
        torch.save(eng.flat_grads().clone(), os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("reduce_dtype", [None, torch.bfloat16])
def test_grad_reducer_world2_gloo(reduce_dtype, tmp_path):
    world, port = 2, 29650 + (0 if reduce_dtype is None else 1) + (os.getpid() % 200) * 2
    mp.spawn(_worker, args=(world, port, reduce_dtype, str(tmp_path)), nprocs=world, join=True)
    res = {r: torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(world)}
    locals_ = [torch.randn(6064, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    want = sum(2 * l for l in locals_) / world          # FSDP semantics: average of the accumulated gradients
    tol = 1e-6 if reduce_dtype is None else 2e-2
    for r in range(world):
        assert torch.allclose(res[r], want, atol=tol, rtol=tol)
    assert torch.equal(res[0], res[1]), "replicas must hold identical gradients after the all-reduce"


def test_clip_grad_norm_matches_reference_coefficient():
    ps = [torch.nn.Parameter(torch.ones(10)), torch.nn.Parameter(torch.ones(6))]
    for p in ps:
        p.grad = torch.full_like(p, 3.0)
    norm = clip_grad_norm(ps, max_norm=8.0)
    assert abs(float(norm) - 12.0) < 1e-5                   # sqrt(16 * 9)
    coef = 8.0 / (12.0 + 1e-6)
    assert torch.allclose(ps[0].grad, torch.full((10,), 3.0 * coef))
    for p in ps:
        p.grad = torch.full_like(p, 0.1)
    clip_grad_norm(ps, max_norm=8.0)
    assert torch.allclose(ps[0].grad, torch.full((10,), 0.1)), "coefficient is clamped to 1"


def test_grad_square_sums_single_rank_follow_the_bucket_hooks():
    """Per-bucket sums of squares collected through the engine's bucket hook = the norm of the flat buffer; buckets that were not
    announced in a step are summed at norm() time; micro-steps that do not end an accumulation window are ignored."""
    eng = FakeEngine([1000, 64, 5000, 0])
    sq = GradSquareSums(eng)
    g = torch.Generator().manual_seed(3)
    params = [torch.nn.Parameter(torch.zeros(e - s)) for _, s, e in eng.grad_ranges() if e > s]
    for step in range(2):
        eng.flat_grads().copy_(torch.randn(eng.flat_grads().numel(), generator=g))
        for p, (_, s, e) in zip(params, [r for r in eng.grad_ranges() if r[2] > r[1]]):
            p.grad = eng.flat_grads()[s:e]
        sq.enabled = False                                   # a non-boundary micro-step: its (partial) gradients must not count
        eng.on_layer_grads_ready("layer2", 1064, 6064)
        eng.flat_grads()[1064:6064].mul_(2.0)
        sq.enabled = True
        for n, s, e in list(reversed(eng.grad_ranges()))[:-1 if step == 0 else None]:     # step 0: layer0 never announced
            eng.on_layer_grads_ready(n, s, e)
        norm, coef = clip_grad_norm(params, 8.0, flat=eng.flat_grads(), defer=True, sumsq=sq)
        want = torch.linalg.vector_norm(eng.flat_grads())
        assert torch.allclose(norm, want, rtol=1e-5), (step, float(norm), float(want))
        assert torch.allclose(coef, torch.clamp(8.0 / (want + 1e-6), max=1.0), rtol=1e-5)


def _sumsq_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = FakeEngine([1000, 64, 5000])
        red = GradReducer(eng, dist, reduce_dtype=torch.bfloat16)
        sq = GradSquareSums(eng, red)
        eng.flat_grads().copy_(torch.randn(6064, generator=torch.Generator().manual_seed(200 + rank)))
        for n, s, e in reversed(eng.grad_ranges()):
            eng.on_layer_grads_ready(n, s, e)
        red.finish()
        torch.save((sq.norm(), torch.linalg.vector_norm(eng.flat_grads())), os.path.join(outdir, f"n{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_grad_square_sums_world2_are_over_the_reduced_gradients(tmp_path):
    world, port = 2, 29750 + (os.getpid() % 200)
    mp.spawn(_sumsq_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), f"n{r}.pt")) for r in range(world)]
    for got, want in res:
        assert torch.allclose(got, want, rtol=1e-5)
    assert torch.equal(res[0][0], res[1][0]), "every rank must clip with the same norm"


@pytest.fixture(scope="module")
def host(golden_dir):
    with open(os.path.join(golden_dir, "host_tiny.json")) as f:
        return json.load(f)


def test_sampler_matches_reference(host):
    class DS:
        def __init__(self, sizes):
            self.g, o = [], 0
            for s in sizes:
                self.g.append(list(range(o, o + s)))
                o += s

        def groups(self):
            return self.g
    assert len(host["sampler"]) >= 20
    for c in host["sampler"]:
        s = FinetuneDistSampler(DS(c["sizes"]), num_replicas=c["ws"], rank=c["rank"], shuffle=c["shuffle"], seed=c["seed"],
                                batch_size=c["bs"], acc_grad=c["acc"])
        s.set_epoch(c["epoch"], c["start_iter"])
        assert list(iter(s)) == c["indices"] and len(s) == c["length"]
    with pytest.raises(ValueError):
        FinetuneDistSampler(DS([8]), num_replicas=2, rank=2, batch_size=1)


def test_sampler_shards_are_disjoint_and_cover():
    class DS:
        def groups(self):
            return [list(range(0, 96)), list(range(96, 160))]
    seen = []
    for r in range(4):
        s = FinetuneDistSampler(DS(), num_replicas=4, rank=r, shuffle=True, seed=5, batch_size=2, acc_grad=2)
        seen += list(iter(s))
    assert len(seen) == len(set(seen)) == 160


def test_sampler_shards_at_dp8_with_accumulation_and_resume():
    """DP 8 x accum 2 -- the eight ranks' index lists are disjoint, cover every sample that fits whole global
    windows (data/alpaca.py:246-328: groups are cut to multiples of batch x accum x world), are the same length, and a resume at
    `start_iter` yields exactly the tail of the un-resumed list on every rank."""
    class DS:
        def groups(self):
            return [list(range(0, 1000)), list(range(1000, 1700)), list(range(1700, 1733))]
    bs, acc, world = 2, 2, 8
    full = []
    for r in range(world):
        s = FinetuneDistSampler(DS(), num_replicas=world, rank=r, shuffle=True, seed=11, batch_size=bs, acc_grad=acc)
        s.set_epoch(1, 0)
        full.append(list(iter(s)))
        assert len(full[-1]) == len(s)
    assert len({len(x) for x in full}) == 1 and len(full[0]) % (bs * acc) == 0
    flat = [i for x in full for i in x]
    assert len(flat) == len(set(flat)), "two ranks drew the same sample"
    window = bs * acc * world
    assert len(flat) == (1000 // window) * window + (700 // window) * window + (33 // window) * window
    # every global window (one optimizer step of all ranks) comes from ONE group: sequence lengths are grouped (dataset.py groups())
    per = bs * acc
    gid = lambda i: 0 if i < 1000 else (1 if i < 1700 else 2)  # noqa: E731
    for w0 in range(0, len(full[0]), per):
        assert len({gid(i) for x in full for i in x[w0:w0 + per]}) == 1
    for start_iter in (1, 7, len(full[0]) // bs - 1):
        for r in range(world):
            s = FinetuneDistSampler(DS(), num_replicas=world, rank=r, shuffle=True, seed=11, batch_size=bs, acc_grad=acc)
            s.set_epoch(1, start_iter)
            assert list(iter(s)) == full[r][start_iter * bs:], (r, start_iter)


def test_lr_schedule_and_weight_decay_groups(host):
    opt = types.SimpleNamespace(param_groups=[{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.5}])
    for e, lr, g0, g1 in host["lr_table"]:
        got = adjust_learning_rate_epoch(opt, e, lr=2e-5, min_lr=0.0, warmup_epochs=0.03, epochs=3)
        assert abs(got - lr) < 1e-15 and abs(opt.param_groups[0]["lr"] - g0) < 1e-15 and abs(opt.param_groups[1]["lr"] - g1) < 1e-15
    m = torch.nn.Module()
    m.a = torch.nn.Linear(4, 4)
    m.attention_norm = torch.nn.LayerNorm(4)
    m.frozen = torch.nn.Linear(4, 4)
    for p in m.frozen.parameters():
        p.requires_grad = False
    names = {id(p): n for n, p in m.named_parameters()}
    got = [dict(weight_decay=g["weight_decay"], names=sorted(names[id(p)] for p in g["params"])) for g in add_weight_decay(m, 0.1)]
    assert got == host["wd_groups"]


def test_dp_replica_fits_288_gib():
    """The HBM budget of one pure-DP replica (masters + moments + flat gradients + bf16 images + the reducer's persistent wire
    buckets + activations) for BASELINE configs[2]/[3] sizing: the estimator follows the engine's own buffer layout, reproduces
    the peaks the bench measured on one rank (profiles/r02m_bench_7b.json: 158.6 GiB at 7B / bs 8 with stored activations, 224.1
    GiB at 13B / micro-batch 4 with recompute), and with the DP = 8 wire buckets on top both stay under 288 GiB."""
    from a3vlm_amd.train import hbm_budget
    G = 2 ** 30
    b7 = hbm_budget(4096, 32, 32, 11008, 32000, batch=8, seq=1091, text=512, stream_bytes=4)       # r02m ran the fp32 residual stream
    b13 = hbm_budget(5120, 40, 40, 13824, 32000, batch=4, seq=1091, text=512, recompute=True, stream_bytes=4)
    assert abs(b7["total"] / G - 158.6) < 0.02 * 158.6
    assert abs(b13["total"] / G - 224.1) < 0.02 * 224.1
    d7 = hbm_budget(4096, 32, 32, 11008, 32000, batch=8, seq=1091, text=512, world=8)
    d13 = hbm_budget(5120, 40, 40, 13824, 32000, batch=4, seq=1091, text=512, recompute=True, world=8)
    assert d7["wire_buckets"] == 2 * (b7["masters_fp32"] // 4) and d13["wire_buckets"] > 24 * G
    assert d7["total"] < 288 * G * 0.95 and d13["total"] < 288 * G * 0.95          # 5 % left for the allocator / RCCL buffers
    # round 5: + the transposed images of the NT input-gradient path (the engine falls back to the NN kernel below 8 GiB of headroom)
    t7 = hbm_budget(4096, 32, 32, 11008, 32000, batch=8, seq=1091, text=512, world=8, transposed_images=True)
    t13 = hbm_budget(5120, 40, 40, 13824, 32000, batch=4, seq=1091, text=512, recompute=True, world=8, transposed_images=True)
    assert t7["images_t_bf16"] == t7["images_bf16"] and t7["total"] == d7["total"] + t7["images_bf16"]
    assert t7["total"] < 288 * G * 0.95 and t13["total"] < 288 * G * 0.95
    # an fp32 wire needs no extra buckets (reduced in place); 13B with stored activations at micro-batch 8 does NOT fit -> recompute
    assert hbm_budget(5120, 40, 40, 13824, 32000, 4, 1091, 512, recompute=True, world=8, wire_bytes=4)["wire_buckets"] == 0
    assert hbm_budget(5120, 40, 40, 13824, 32000, 8, 1091, 512, recompute=False, world=8)["total"] > 288 * G
    # ZeRO-1 (--zero1) at DP 8: the sharded masters / moments free ~130 GB per GPU -- 13B runs at micro-batch 8 WITHOUT recompute
    z13 = hbm_budget(5120, 40, 40, 13824, 32000, 8, 1091, 512, recompute=False, world=8, zero1=True)
    assert z13["total"] < 288 * G * 0.90, z13["total"] / G
    assert d13["masters_fp32"] + d13["adamw_moments_fp32"] - z13["masters_fp32"] - z13["adamw_moments_fp32"] > 125 * G
    assert z13["wire_buckets"] < 2 * G


# ------------------------------------------------------------------ the non-finite flag is global under DP
class _ToyDP(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor([1.0, -2.0, 0.5]))

    def forward(self, examples, labels, images=None, depth_imgs=None):
        loss = ((examples.float() @ self.w - labels.float()) ** 2).mean()
        if bool((examples == 77).any()):
            loss = loss * float("nan")
        return loss, {}


class _ParamReducer(GradReducer):
    """The real reducer's flag collective; the toy model's one gradient is averaged in finish()."""

    def __init__(self, model, dist_):
        super().__init__(FakeEngine([4]), dist_)
        self.model = model

    def finish(self):
        super().finish()
        g = self.model.w.grad
        g.mul_(1.0 / self.world)
        self.dist.all_reduce(g)


def _nan_worker(rank, world, port, outdir):
    from a3vlm_amd.engine_finetune import train_one_epoch
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    code, saved = 0, []
    try:
        g = torch.Generator().manual_seed(3 + rank)
        data = [(torch.randint(0, 5, (4, 3), generator=g), torch.randint(0, 5, (4,), generator=g), torch.ones(4, 3)) for _ in range(4)]
        if rank == 0:
            data[2][0][0, 0] = 77                  # ONLY rank 0 sees a NaN loss, in the second optimizer step
        m = _ToyDP()
        args = types.SimpleNamespace(accum_iter=1, lr=1e-2, min_lr=0.0, warmup_epochs=0.0, epochs=1, clip_grad=-1, print_freq=100,
                                     save_iteration_interval=1)
        try:
            train_one_epoch(m, data, torch.optim.SGD(m.parameters(), lr=0.1), epoch=0, start_iter=0, args=args, reducer=_ParamReducer(m, dist),
                            log=lambda s: None, on_save=lambda step: saved.append((step, m.w.detach().clone())))
        except SystemExit as e:
            code = e.code
        torch.save({"code": code, "w": m.w.detach().clone(), "saved": saved}, os.path.join(outdir, f"nan{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_nan_loss_on_one_rank_stops_every_rank_before_the_update(tmp_path):
    """clip_grad <= 0 (the reference default): rank 0's loss is NaN in step 2, rank 1's own loss is finite but the averaged gradient
    it receives is NaN.  Both ranks must stop with exit code 1 BEFORE applying that update; only the two clean steps' checkpoints
    exist and the weights are the finite ones of step 1 on both ranks."""
    world, port = 2, 29650 + os.getpid() % 200
    mp.spawn(_nan_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f"nan{r}.pt")) for r in range(world))
    for r in (r0, r1):
        assert r["code"] == 1
        assert torch.isfinite(r["w"]).all()
        assert [s for s, _ in r["saved"]] == [0, 1]
    assert torch.equal(r0["w"], r1["w"]) and torch.equal(r0["w"], r0["saved"][-1][1])
