"""-m gpu: the fine-tuning entry point end to end on one GPU (tiny model, synthetic data): epochs, gradient
accumulation, checkpoint layout on disk, resume."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env=None):
    e = dict(os.environ)
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "a3vlm_amd.main_finetune"] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("precision", ["tf32", "bf16"])
def test_main_finetune_synthetic(tmp_path, precision):
    gd = os.path.join(ROOT, "tests", "golden")
    extra = tmp_path / "vit.json"
    extra.write_text(json.dumps(dict(vit_width=64, vit_layers=2, vit_heads=4, vit_crop=112, n_views=1)))
    out = tmp_path / "out"
    base = ["--llama_type", "llama_ens5", "--llama_config", os.path.join(gd, "tiny_params.json"), str(extra),
            "--tokenizer_path", os.path.join(gd, "tokenizer.model"), "--batch_size", "2", "--accum_iter", "2",
            "--warmup_epochs", "0.5", "--lr", "1e-3", "--min_lr", "0", "--clip_grad", "8", "--weight_decay", "0",
            "--max_words", "120", "--precision", precision, "--output_dir", str(out), "--synthetic", "24", "--num_workers", "0",
            "--dialog", "--data_parallel", "sdp", "--model_parallel_size", "1", "--checkpointing"]
    log = run(base + ["--epochs", "2"])
    assert "closs" in log
    for ep in ("epoch0", "epoch1"):
        files = set(os.listdir(out / ep))
        assert {"consolidated.00-of-01.model.pth", "consolidated.00-of-01.optimizer.pth", "consolidated.00-of-01.other.pth",
                "config.json", "meta.json", "tokenizer.model", "rank-specific-00000-of-00001.pth"} <= files
    lines = [json.loads(x) for x in open(out / "log.txt")]
    assert [l["epoch"] for l in lines] == [0, 1] and all(0 < l["train_closs"] < 20 for l in lines)
    log2 = run(base + ["--epochs", "3", "--resume", str(out)])
    assert "resume:" in log2 and os.path.isdir(out / "epoch2")
    lines = [json.loads(x) for x in open(out / "log.txt")]
    assert [l["epoch"] for l in lines] == [0, 1, 2]


def test_main_finetune_dialog_dataset(tmp_path):
    """--data_config: the dialog dataset (image_text + text groups), PIL transform and FinetuneDistSampler feed the trainer."""
    from oracle.gen_golden import dialog_yaml
    gd = os.path.join(ROOT, "tests", "golden")
    extra = tmp_path / "vit.json"
    extra.write_text(json.dumps(dict(vit_width=64, vit_layers=2, vit_heads=4, vit_crop=112, n_views=1)))
    out = tmp_path / "out"
    log = run(["--llama_type", "llama_ens5", "--llama_config", os.path.join(gd, "tiny_params.json"), str(extra),
               "--tokenizer_path", os.path.join(gd, "tokenizer.model"), "--batch_size", "2", "--accum_iter", "1", "--epochs", "1",
               "--warmup_epochs", "0.5", "--lr", "1e-3", "--min_lr", "0", "--clip_grad", "8", "--weight_decay", "0.02",
               "--max_words", "220", "--precision", "bf16", "--output_dir", str(out), "--data_config", dialog_yaml(str(tmp_path)),
               "--image_transform", "padded_resize", "--num_workers", "0", "--dialog", "--model_parallel_size", "1"])
    assert "closs" in log and "total length: 13" in log
    line = json.loads(open(out / "log.txt").read().strip().splitlines()[-1])
    assert 0 < line["train_closs"] < 20
    # that run preprocessed on the device (--preprocess gpu, the default: workers decode to uint8 HWC); the PIL transform in the
    # workers feeds bit-identical image tensors, so the logged loss of the same seeded run is the same
    out2 = tmp_path / "out_cpu"
    run(["--llama_type", "llama_ens5", "--llama_config", os.path.join(gd, "tiny_params.json"), str(extra),
         "--tokenizer_path", os.path.join(gd, "tokenizer.model"), "--batch_size", "2", "--accum_iter", "1", "--epochs", "1",
         "--warmup_epochs", "0.5", "--lr", "1e-3", "--min_lr", "0", "--clip_grad", "8", "--weight_decay", "0.02",
         "--max_words", "220", "--precision", "bf16", "--output_dir", str(out2), "--data_config", dialog_yaml(str(tmp_path)),
         "--image_transform", "padded_resize", "--num_workers", "0", "--dialog", "--model_parallel_size", "1", "--preprocess", "cpu"])
    line2 = json.loads(open(out2 / "log.txt").read().strip().splitlines()[-1])
    assert abs(line2["train_closs"] - line["train_closs"]) < 2e-3 * line["train_closs"], (line, line2)      # (bf16 backward atomics: not bit-reproducible)


def test_main_finetune_lora_only_save_trainable(tmp_path):
    """--llama_type llama_ens5_peft: adapters + norms train, the base is frozen; --only_save_trainable writes just those keys
    (util/misc.py:340-350) and the run resumes / reloads on top of a base checkpoint by name."""
    import torch
    gd = os.path.join(ROOT, "tests", "golden")
    extra = tmp_path / "peft.json"
    extra.write_text(json.dumps(dict(vit_width=64, vit_layers=2, vit_heads=4, vit_crop=112, n_views=1, lora_rank=8)))
    out = tmp_path / "out"
    log = run(["--llama_type", "llama_ens5_peft", "--llama_config", os.path.join(gd, "tiny_params.json"), str(extra),
               "--tokenizer_path", os.path.join(gd, "tokenizer.model"), "--batch_size", "2", "--accum_iter", "1", "--epochs", "1",
               "--warmup_epochs", "0.5", "--lr", "1e-3", "--min_lr", "0", "--clip_grad", "8", "--weight_decay", "0.0",
               "--max_words", "120", "--precision", "bf16", "--output_dir", str(out), "--synthetic", "16", "--num_workers", "0",
               "--dialog", "--model_parallel_size", "1", "--only_save_trainable"])
    assert "closs" in log
    sd = torch.load(out / "epoch0" / "consolidated.00-of-01.model.pth", weights_only=False)["model"]
    keys = set(sd)
    assert any("lora_a" in k for k in keys) and any("attention_norm" in k for k in keys) and "llma.visual_proj.0.weight" in keys
    assert not any(k.endswith("attention.wq.weight") for k in keys) and "llma.tok_embeddings.weight" not in keys
    assert json.load(open(out / "epoch0" / "meta.json"))["llama_type"] == "llama_ens5_peft"


def _bench_two_rank_checks(r):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 4
    assert abs(d["value"] - 4 * 1000.0 / d["ms_per_step"]) / d["value"] < 0.02
    assert d["config"]["workload"].startswith("configs[2]") and abs(d["train_lora"]["samples_s"] - d["value"]) / d["value"] < 1e-3
    assert d["train"] and d["train"].get("samples_s"), d["train"]
    assert d["exposed_allreduce_ms"] is not None and d["allreduce"]["bucket_bytes_on_wire"] > 0
    assert d["train"]["allreduce"]["bucket_bytes_on_wire"] > d["allreduce"]["bucket_bytes_on_wire"]     # full fine-tune moves more than adapters
    assert d["decode_tok_s"] > 0


_BENCH_ARGS = ["--gpus", "2", "--model", "tiny", "--steps", "2", "--warmup", "1", "--batch", "2", "--prompt", "32", "--decode-steps", "4",
               "--no-cpu-baseline"]


def test_bench_gpus_2_without_a_launcher(tmp_path):
    """`python bench.py --gpus 2` exactly as the driver may call it -- no torchrun around it: bench.py starts the two ranks itself.
    Both ranks share cuda:0 and talk over gloo (this box has one GPU; real N > 1 runs use RCCL).  Checks n_gpus, the barrier /
    max-over-ranks timing, the weak-scaling value, the DP reducer of the training legs with real gradients, the exposed
    all-reduce fields, and that exactly one JSON line comes out."""
    e = dict(os.environ)
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    e.update(A3V_BENCH_ONE_DEVICE="1", A3V_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK", "TORCHELASTIC_RUN_ID"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + _BENCH_ARGS, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    _bench_two_rank_checks(r)


def test_bench_two_ranks_under_torchrun(tmp_path):
    """the driver's other form: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 (bench.py must not launch again)"""
    e = dict(os.environ)
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    e.update(A3V_BENCH_ONE_DEVICE="1", A3V_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    port = 29500 + os.getpid() % 2000
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py")] + _BENCH_ARGS,
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    _bench_two_rank_checks(r)


def _torchrun(module_args, tmp_env=None, nproc=2):
    e = dict(os.environ)
    e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
    e.update(A3V_ONE_DEVICE="1", A3V_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    port = 31500 + os.getpid() % 2000
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "-m"] + module_args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_main_finetune_two_ranks_dp(tmp_path):
    """Data-parallel trainer entry point with world_size 2 (both ranks on cuda:0, gloo): weight broadcast, sampler sharding,
    per-layer gradient buckets reduced on the side stream during backward, rank-0 checkpoint + per-rank files."""
    gd = os.path.join(ROOT, "tests", "golden")
    extra = tmp_path / "vit.json"
    extra.write_text(json.dumps(dict(vit_width=64, vit_layers=2, vit_heads=4, vit_crop=112, n_views=1)))
    out = tmp_path / "out"
    log = _torchrun(["a3vlm_amd.main_finetune", "--llama_type", "llama_ens5", "--llama_config", os.path.join(gd, "tiny_params.json"), str(extra),
                     "--tokenizer_path", os.path.join(gd, "tokenizer.model"), "--batch_size", "2", "--accum_iter", "2", "--epochs", "1",
                     "--warmup_epochs", "0.5", "--lr", "1e-3", "--min_lr", "0", "--clip_grad", "8", "--weight_decay", "0", "--max_words", "120",
                     "--precision", "bf16", "--output_dir", str(out), "--synthetic", "32", "--num_workers", "0", "--dialog",
                     "--model_parallel_size", "1"])
    assert "closs" in log
    files = set(os.listdir(out / "epoch0"))
    assert {"consolidated.00-of-01.model.pth", "rank-specific-00000-of-00002.pth", "rank-specific-00001-of-00002.pth"} <= files


def test_eval_entry_two_ranks(tmp_path):
    """Batch-inference entry point with world_size 2: the dataset is sharded by rank and the answers are gathered on rank 0."""
    import types
    from a3vlm_amd import checkpoint as ck
    from a3vlm_amd.model.meta import MetaModel
    from oracle import ref_cpu
    from oracle.gen_golden import TINY
    gd = os.path.join(ROOT, "tests", "golden")
    vit = dict(vit_width=64, vit_layers=2, vit_heads=4, vit_crop=112, n_views=5)
    cfgp = tmp_path / "cfg.json"
    cfgp.write_text(json.dumps({**{k: v for k, v in TINY.items() if k != "max_seq_len"}, **vit}))
    mm = MetaModel("llama_ens5", str(cfgp), os.path.join(gd, "tokenizer.model"), with_visual=True, max_seq_len=512)
    sd = ref_cpu.make_decoder_weights(ref_cpu.OracleArgs(vocab_size=mm.tokenizer.n_words, **TINY), seed=0, std=0.08)
    vsd = ref_cpu.make_vision_weights(64, width=64, layers=2, patch=14, grid=8, seed=1, std=0.05)
    mm.llma.load_state_dict({**sd, **vsd})
    ckdir = ck.save_checkpoint(str(tmp_path / "ck"), types.SimpleNamespace(precision="tf32", only_save_trainable=False), mm, None, None, None, epoch=0)
    _torchrun(["a3vlm_amd.eval_affordance_v2", "--llama_type", "llama_ens5", "--llama_config", str(cfgp), "--tokenizer_path",
               os.path.join(gd, "tokenizer.model"), "--pretrained_path", ckdir, "--batch_size", "1", "--num_workers", "0", "--dataset",
               os.path.join(gd, "demo", "demo.json"), "--input_size", "224", "--addition_flag", "t2", "--max_gen_len", "6", "--max_seq_len", "512",
               "--temperature", "0", "--image_root", os.path.join(gd, "demo"), "--output_root", str(tmp_path / "logs"), "--precision", "tf32"])
    recs = json.load(open(tmp_path / "logs" / "t2" / "demo.json"))
    assert len(recs) == 3 and recs[0]["answer"] == recs[1]["answer"] == recs[2]["answer"]


def test_fused_adamw_matches_torch_adamw():
    """a3vlm_amd.optim.FusedAdamW == torch.optim.AdamW over several steps (weight decay groups, odd sizes incl. a 4-element
    tail, bf16 image side output), and the two optimizers load each other's state_dict."""
    import torch
    from a3vlm_amd.optim import FusedAdamW
    DEV = "cuda"
    g = torch.Generator().manual_seed(1)
    shapes = [(300, 257), (4096,), (7,), (64, 64)]
    ref = [torch.randn(*s, generator=g).to(DEV).requires_grad_(True) for s in shapes]
    got = [r.detach().clone().requires_grad_(True) for r in ref]
    imgs = {id(p): torch.zeros(p.shape, dtype=torch.bfloat16, device=DEV) for p in got[:2]}
    groups = lambda ps: [dict(params=ps[:2], weight_decay=0.1), dict(params=ps[2:], weight_decay=0.0)]
    o_ref = torch.optim.AdamW(groups(ref), lr=3e-3, betas=(0.9, 0.95), eps=1e-8)
    o_got = FusedAdamW(groups(got), lr=3e-3, betas=(0.9, 0.95), eps=1e-8, image_of=lambda p: imgs.get(id(p)))
    for it in range(5):
        for a, b in zip(ref, got):
            gr = torch.randn(a.shape, generator=g).to(DEV) * (0.1 + it)
            a.grad, b.grad = gr.clone(), gr.clone()
        v0 = [b._version for b in got]
        o_ref.step()
        o_got.step()
        assert all(b._version > v for b, v in zip(got, v0))      # version-keyed caches (bf16 weight images) see the update
        for a, b in zip(ref, got):
            assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()) + 1e-7, it
    for p in got[:2]:
        assert torch.equal(imgs[id(p)], p.detach().to(torch.bfloat16))
    sd = o_got.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 5
    o_ref2 = torch.optim.AdamW(groups([r.detach().clone().requires_grad_(True) for r in ref]), lr=3e-3, betas=(0.9, 0.95))
    import copy
    o_ref2.load_state_dict(copy.deepcopy(sd))                       # torch's optimizer accepts the state the HIP optimizer wrote ...
    o_got.load_state_dict(copy.deepcopy(o_ref.state_dict()))        # ... and vice versa (deep copies: load_state_dict aliases tensors)
    for a, b in zip(ref, got):
        gr = torch.randn(a.shape, generator=g).to(DEV)
        a.grad, b.grad = gr.clone(), gr.clone()
    o_ref.step()
    o_got.step()
    for a, b in zip(ref, got):
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()) + 1e-7
    with pytest.raises(RuntimeError):
        bad = torch.zeros(4, device=DEV, dtype=torch.bfloat16, requires_grad=True)
        bad.grad = torch.zeros_like(bad)
        FusedAdamW([bad]).step()


def test_fused_adamw_multi_tensor_launch_equals_per_tensor_launches():
    """a3v_adamw_multi (all small tensors of a param group in one launch, bf16 values to same-shape images or to two strided sinks)
    against one a3v_adamw_scaled launch per tensor: parameters, moments and images bit-equal over 4 steps, with a clip
    coefficient, a skipped (negative-coefficient) step and odd sizes."""
    import torch
    from a3vlm_amd.optim import FusedAdamW
    DEV = "cuda"
    g = torch.Generator().manual_seed(2)
    shapes = [(16, 4096), (4096, 16), (4096,), (11008, 16), (7,), (16, 11008), (130, 6)]

    class Eng:          # stands in for TrainEngine: strided sinks for the 2-D "adapters", nothing for the rest
        def __init__(self, ps):
            self.A = {id(p): torch.zeros(p.shape[0] + 3, p.shape[1] + 5, dtype=torch.bfloat16, device=DEV) for p in ps if p.dim() == 2}
            self.At = {id(p): torch.zeros(p.shape[1] + 2, p.shape[0] + 9, dtype=torch.bfloat16, device=DEV) for p in ps if p.dim() == 2}

        def image_sink(self, p):
            return None

        def adapter_sink(self, p):
            if id(p) not in self.A:
                return None
            a, at = self.A[id(p)], self.At[id(p)]
            return (a.data_ptr() + 2 * (1 * a.stride(0) + 2), a.stride(0), 1, at.data_ptr() + 2 * (1 * at.stride(0) + 4), 1, at.stride(0))

        def images_adopted(self, w):
            pass

        def adapters_adopted(self, w):
            self.adopted = set(w)

        def sync_optimizer(self):
            pass
    base = [torch.randn(*s, generator=g).to(DEV) for s in shapes]
    runs = []
    for multi in (True, False):
        ps = [b.clone().requires_grad_(True) for b in base]
        eng = Eng(ps)
        opt = FusedAdamW([dict(params=ps[:4], weight_decay=0.05), dict(params=ps[4:], weight_decay=0.0)], lr=2e-3, betas=(0.9, 0.95), engine=eng)
        if not multi:
            opt.multi_threshold = 0
        gg = torch.Generator().manual_seed(3)
        for it in range(4):
            for p in ps:
                p.grad = (torch.randn(p.shape, generator=gg) * (0.1 + it)).to(DEV)
            coef = torch.tensor([-1.0 if it == 2 else 0.37], device=DEV)
            opt.step(grad_scale=coef)
        if multi:
            assert eng.adopted == {id(p) for p in ps if p.dim() == 2}
        runs.append((ps, [opt.state[p] for p in ps], eng))
    (pa, sa, ea), (pb, sb, eb) = runs
    for i in range(len(shapes)):
        assert torch.equal(pa[i], pb[i]) and torch.equal(sa[i]["exp_avg"], sb[i]["exp_avg"]) and torch.equal(sa[i]["exp_avg_sq"], sb[i]["exp_avg_sq"]), i
        assert float(sa[i]["step"]) == 4
        if pa[i].dim() == 2:
            a, at = ea.A[id(pa[i])], ea.At[id(pa[i])]
            r, c = pa[i].shape
            want = pa[i].detach().to(torch.bfloat16)
            assert torch.equal(a[1:1 + r, 2:2 + c], want) and torch.equal(at[1:1 + c, 4:4 + r], want.t())
            a2, at2 = a.clone(), at.clone()
            a2[1:1 + r, 2:2 + c] = 0
            at2[1:1 + c, 4:4 + r] = 0
            assert not bool(a2.any()) and not bool(at2.any())        # nothing outside the blocks was touched


@pytest.mark.parametrize("opt_kind", ["torch_fused", "torch_plain", "hip", "hip_engine"])
def test_engine_sees_updates_of_any_optimizer(opt_kind):
    """The engine's bf16 weight images are caches; torch.optim.AdamW(fused=True) updates parameters WITHOUT bumping
    Tensor._version, so the cache key also counts optimizer steps (global post-step hook).  All three optimizers must give the
    same loss trajectory on one repeated batch (they apply the same update)."""
    import torch
    import bench
    from a3vlm_amd.optim import FusedAdamW
    from a3vlm_amd.train import TrainEngine
    from a3vlm_amd.util import promote_trainable_params_to_fp32
    dev = torch.device("cuda", 0)

    def run(kind):
        torch.manual_seed(0)
        m, args = bench.build_model("tiny", dev, 256)
        for n, p in m.named_parameters():
            p.requires_grad = not n.startswith("clip.")
        promote_trainable_params_to_fp32(m)
        eng = TrainEngine(m, torch.bfloat16)
        params = [p for p in m.parameters() if p.requires_grad]
        if kind == "torch_fused":
            opt = torch.optim.AdamW(params, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.0, fused=True)
        elif kind == "torch_plain":
            opt = torch.optim.AdamW(params, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.0, fused=False, foreach=False)
        else:      # "hip_engine": the optimizer also writes the bf16 images the engine's next step reads
            opt = FusedAdamW(params, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.0, engine=eng if kind == "hip_engine" else None)
        g = torch.Generator(device=dev).manual_seed(1)
        tokens = torch.randint(3, args.vocab_size, (4, 64), device=dev, generator=g)
        tokens[:, 0] = 1
        labels = tokens.clone()
        labels[:, :32] = 0
        out = []
        for _ in range(4):
            loss = eng.forward_loss(tokens, labels, None)
            eng.backward(1.0)
            opt.step()
            opt.zero_grad(set_to_none=True)
            out.append(float(loss))
        if kind == "hip_engine":      # the images the optimizer wrote ARE the bf16 roundings of the updated parameters, and current
            im = eng._images()
            l0 = m.layers[0]
            want_qkv = torch.cat([l0.attention.wq.weight, l0.attention.wk.weight, l0.attention.wv.weight]).to(torch.bfloat16)
            ver0 = dict(im.ver)
            assert torch.equal(im["qkv.0"], want_qkv) and im.ver == ver0          # served from the adopted image, not rebuilt
        return out
    want = run("torch_plain")
    got = want if opt_kind == "torch_plain" else run(opt_kind)
    assert want[3] < want[0] - 1.0                      # the repeated batch is being fitted
    for a, b in zip(got, want):
        assert abs(a - b) < 2e-2 * abs(b), (opt_kind, got, want)
