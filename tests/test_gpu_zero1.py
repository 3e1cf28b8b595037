"""-m gpu: ZeRO-1 (a3vlm_amd/zero1.py) on the real training engine: the big matrices live as bf16 views of one flat buffer, the GEMM images
alias them, the update runs on each rank's slice of the fp32 masters (a3v_adamw_scaled) and the all-gather refreshes every rank's
parameters.  One process (its slice = everything), two ranks on one device over gloo (the only two-process layout a one-GPU box allows:
the two-rank arithmetic of the collectives is pinned on CPU, tests/test_zero1_cpu.py), and the trainer entry point with --zero1."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV, BF = "cuda", torch.bfloat16
GEO = dict(dim=256, n_layers=2, n_heads=2, n_kv_heads=2, vocab_size=512, multiple_of=256, max_seq_len=256)


class _One:
    class ReduceOp:
        SUM, AVG, MAX = "sum", "avg", "max"
    get_world_size = staticmethod(lambda group=None: 1)
    get_rank = staticmethod(lambda group=None: 0)
    get_backend = staticmethod(lambda group=None: "none")


def _model(sharded: bool):
    from a3vlm_amd.model.LLM import llama_ens5 as plugin
    from a3vlm_amd.util import promote_trainable_params_to_fp32
    from oracle import ref_cpu
    m = plugin.Transformer(plugin.ModelArgs(**GEO), with_visual=False)
    m.load_state_dict(ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(**GEO), seed=3, std=0.05))
    for p in m.parameters():
        p.requires_grad = True
    m.to(BF).to(DEV)
    promote_trainable_params_to_fp32(m, keep_matrices_sharded=sharded)
    return m


def _data(n=4, T=48, seed=5):
    g = torch.Generator().manual_seed(seed)
    ex = torch.randint(3, GEO["vocab_size"], (n, T), generator=g)
    ex[:, 0] = 1
    lab = ex.clone()
    lab[:, :8] = 0
    return ex.to(DEV), lab.to(DEV)


def _adamw(p32, g, m, v, lr, b1, b2, eps, wd, step, coef):
    g = g * coef
    p32 = p32 * (1.0 - lr * wd)
    m = b1 * m + (1.0 - b1) * g
    v = b2 * v + (1.0 - b2) * g * g
    return p32 - lr * (m / (1.0 - b1 ** step)) / ((v / (1.0 - b2 ** step)).sqrt() + eps), m, v


def test_zero1_single_process_step_is_adamw_on_the_bf16_started_masters():
    from a3vlm_amd.optim import FusedAdamW
    from a3vlm_amd.train import TrainEngine
    from a3vlm_amd.util import add_weight_decay
    from a3vlm_amd.zero1 import Zero1Optimizer
    m = _model(True)
    eng = TrainEngine(m, BF, zero1_world=1)
    fp = eng.flat_params()
    big = {n: p for n, p in m.named_parameters() if p.dtype == BF}
    assert len(big) == 2 * 7 + 2 and all(p.untyped_storage().data_ptr() == fp.untyped_storage().data_ptr() for p in big.values())
    small_groups = [{**g, "params": [q for q in g["params"] if q.dtype == torch.float32]} for g in add_weight_decay(m, 0.02)]
    small = FusedAdamW([g for g in small_groups if g["params"]], lr=1e-2, betas=(0.9, 0.95), engine=eng)
    opt = Zero1Optimizer(eng, _One, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.02, reduce_dtype=BF, small=small)
    ex, lab = _data()
    state = {n: (p.detach().float().clone(), torch.zeros_like(p, dtype=torch.float32), torch.zeros_like(p, dtype=torch.float32)) for n, p in big.items()}
    losses = []
    for step in (1, 2, 3):
        loss = eng.forward_loss(ex, lab, None)
        losses.append(float(loss))
        eng.backward(1.0)
        opt.finish()
        norm, coef = opt.clip_coef(0.5)
        want_norm = torch.linalg.vector_norm(torch.cat([eng._views[n].flatten().to(BF).float() if n in big else eng._views[n].flatten()
                                                        for n in eng._views]))
        assert torch.allclose(norm, want_norm, rtol=2e-3), (float(norm), float(want_norm))
        grads = {n: eng._views[n].detach().to(BF).float().clone() for n in big}          # bf16 wire: what the slice update reads
        opt.step(grad_scale=coef.reshape(1))
        for n, p in big.items():
            p32, mm, vv = state[n]
            p32, mm, vv = _adamw(p32, grads[n], mm, vv, 1e-2, 0.9, 0.95, 1e-8, 0.02, step, float(coef))
            state[n] = (p32, mm, vv)
            want = p32.to(BF)
            diff = (p.detach().float() - want.float()).abs()
            assert float(diff.max()) <= 2.0 ** -7 * float(want.float().abs().max()), (n, step, float(diff.max()))
            assert float((p.detach() != want).float().mean()) < 0.02, (n, step)         # a differently-rounded last bit here and there, no more
        # the GEMM images ARE the parameter storage
        a = m.layers[0].attention
        assert eng._images()["qkv.0"].data_ptr() == a.wq.weight.data_ptr()
        m.zero_grad(set_to_none=True)
    assert losses[2] < losses[0], losses


def _two_rank_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from a3vlm_amd.optim import FusedAdamW
        from a3vlm_amd.train import TrainEngine
        from a3vlm_amd.util import add_weight_decay
        from a3vlm_amd.zero1 import Zero1Optimizer
        torch.cuda.set_device(0)
        m = _model(True)
        eng = TrainEngine(m, BF, zero1_world=world)
        sg = [{**g, "params": [q for q in g["params"] if q.dtype == torch.float32]} for g in add_weight_decay(m, 0.0)]
        small = FusedAdamW([g for g in sg if g["params"]], lr=5e-3, betas=(0.9, 0.95), engine=eng)
        opt = Zero1Optimizer(eng, dist, lr=5e-3, betas=(0.9, 0.95), reduce_dtype=BF, small=small)
        ex, lab = _data()
        losses = []
        for step in range(3):
            mine = slice(rank, None, world)
            losses.append(float(eng.forward_loss(ex[mine], lab[mine], None)))
            eng.backward(1.0)
            opt.finish()
            norm, coef = opt.clip_coef(1.0)
            # (round 5) the trainer's default: every bucket's all-gather on the side stream behind its AdamW launches; the next forward
            # waits bucket by bucket (engine.await_weights), the last step's gathers are joined by sync_params below
            opt.step(grad_scale=coef.reshape(1), overlap=True)
            m.zero_grad(set_to_none=True)
        opt.sync_params()
        torch.cuda.synchronize()
        held = sum(b["master"].numel() for b in opt.buckets if b["n"])
        torch.save({"params": {n: p.detach().float().cpu() for n, p in m.named_parameters()}, "losses": losses, "held": held,
                    "total": sum(b["shard"][1] - b["shard"][0] for b in opt.buckets)}, os.path.join(outdir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_zero1_two_ranks_track_the_replicated_step(tmp_path):
    """Two ranks (both on cuda:0, gloo): identical parameters on both after every step, each holds half of the masters, and the trajectory
    follows ONE replica that sees the whole batch with fp32 masters of everything (FusedAdamW): same loss curve, parameters within bf16."""
    import torch.multiprocessing as mp
    from a3vlm_amd.dp import clip_grad_norm
    from a3vlm_amd.optim import FusedAdamW
    from a3vlm_amd.train import TrainEngine
    world, port = 2, 29500 + os.getpid() % 90
    mp.spawn(_two_rank_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f"r{r}.pt"), weights_only=False) for r in range(world))
    for n in r0["params"]:
        assert torch.equal(r0["params"][n], r1["params"][n]), n
    assert r0["held"] * 2 == r0["total"] and r1["held"] == r0["held"]
    m = _model(False)
    eng = TrainEngine(m, BF)
    params = [p for p in m.parameters() if p.requires_grad]
    opt = FusedAdamW([{"params": params, "weight_decay": 0.0}], lr=5e-3, betas=(0.9, 0.95), engine=eng)
    ex, lab = _data()
    losses = []
    for step in range(3):
        losses.append(float(eng.forward_loss(ex, lab, None)))
        eng.backward(1.0)
        norm, coef = clip_grad_norm(params, 1.0, flat=eng.flat_grads(), defer=True)
        opt.step(grad_scale=coef.reshape(1))
        m.zero_grad(set_to_none=True)
    # (equal token counts per sample: the mean of the two ranks' losses is the whole-batch loss)
    for a, b, c in zip(r0["losses"], r1["losses"], losses):
        assert abs(0.5 * (a + b) - c) < 2e-2 * abs(c), (a, b, c)
    assert losses[2] < losses[0]
    for n, p in m.named_parameters():
        want = p.detach().float().cpu()
        got = r0["params"][n]
        scale = float(want.abs().max())
        # (AdamW moves an element by ~lr per step whatever the gradient's size: where the gradient is bf16-wire noise around zero the two
        #  runs may step in opposite directions -- at most 3 steps of 5e-3 apart)
        assert float((got - want).abs().max()) < 3e-2 * scale + 0.6 * 3 * 5e-3, n
        assert float((got - want).abs().mean()) < 2e-3 * scale + 1e-4, n


def test_main_finetune_zero1_runs_and_resumes(tmp_path):
    """The trainer entry point with --zero1 (one process): runs, logs a finite falling loss, writes the CONSOLIDATED optimizer state (the
    reference's consolidated.00-of-01.optimizer.pth: world-size independent) next to the model file and resumes from it; a resume from a
    checkpoint WITHOUT optimizer state says so and re-seeds the fp32 masters from the loaded weights instead of reverting them."""
    gd = os.path.join(ROOT, "tests", "golden")
    extra = tmp_path / "vit.json"
    extra.write_text(json.dumps(dict(vit_width=64, vit_layers=2, vit_heads=4, vit_crop=112, n_views=1)))
    out = tmp_path / "out"

    def run(extra_args):
        env = dict(os.environ, PYTHONPATH=ROOT)
        r = subprocess.run([sys.executable, "-m", "a3vlm_amd.main_finetune", "--llama_type", "llama_ens5", "--llama_config",
                            os.path.join(gd, "tiny_params.json"), str(extra), "--tokenizer_path", os.path.join(gd, "tokenizer.model"),
                            "--batch_size", "2", "--accum_iter", "2", "--warmup_epochs", "0.5", "--lr", "2e-3", "--min_lr", "0",
                            "--clip_grad", "8", "--weight_decay", "0.02", "--max_words", "120", "--precision", "bf16", "--output_dir", str(out),
                            "--synthetic", "16", "--num_workers", "0", "--dialog", "--model_parallel_size", "1", "--zero1"] + extra_args,
                           capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        return r.stdout + r.stderr
    log = run(["--epochs", "2"])
    assert "closs" in log
    lines = [json.loads(x) for x in open(out / "log.txt")]
    assert [ln["epoch"] for ln in lines] == [0, 1] and 0 < lines[1]["train_closs"] < lines[0]["train_closs"] + 0.5
    assert os.path.isfile(out / "epoch1" / "consolidated.00-of-01.optimizer.pth") and os.path.isfile(out / "epoch1" / "consolidated.00-of-01.model.pth")
    import torch
    sd = torch.load(out / "epoch1" / "consolidated.00-of-01.optimizer.pth", weights_only=False)["optimizer"]
    assert "zero1_full" in sd and sd["zero1_full"]["step"] > 0 and sd["small"] is not None
    log2 = run(["--epochs", "3", "--resume", str(out)])
    assert "resume:" in log2 and os.path.isdir(out / "epoch2") and "no optimizer state" not in log2
    lines = [json.loads(x) for x in open(out / "log.txt")]
    assert lines[-1]["epoch"] == 2 and 0 < lines[-1]["train_closs"] < lines[1]["train_closs"] + 0.5
    # weights-only checkpoint: the loss continues from the loaded weights (stale masters would throw it back to the epoch-0 level)
    os.remove(out / "epoch2" / "consolidated.00-of-01.optimizer.pth")
    log3 = run(["--epochs", "4", "--resume", str(out)])
    assert "no optimizer state" in log3
    lines = [json.loads(x) for x in open(out / "log.txt")]
    assert lines[-1]["epoch"] == 3 and 0 < lines[-1]["train_closs"] < lines[2]["train_closs"] + 0.5, lines
