#!/usr/bin/env python3
"""bench.py -- the A3VLM hot path on MI355X through liba3vlm_hip.so.

    python bench.py --gpus N --steps K --warmup W
    N > 1 without a launcher: this file re-executes itself under ``python -m torch.distributed.run --nproc-per-node N`` (one rank per
    GPU, RCCL); under a launcher (WORLD_SIZE set, the driver's own torchrun line) it is that launcher's rank.  Rank 0 prints ONE line.

Headline (`metric` / `value`, BASELINE.json: "image-text samples/sec (train) + articulation-decode tok/s"): ONE step = one LoRA
FINE-TUNING step of BASELINE configs[2] (ViT-L/14 @ 336x336 single crop, 577 + 2 image words, + Llama-2-7B, rank-16 adapters on the
seven decoder linears of every block, norms + projector trainable, base matrices frozen in bf16; global batch 64 = 8 per GPU x DP 8)
on this rank's micro-batch of 8 image + 512-token samples: multimodal forward (frozen ViT, projector, 32 decoder blocks over
8 x 1091 positions, LM head, CE), backward through the HIP kernels, DP all-reduce of the gradient buckets (N > 1; RCCL, overlapped
with the backward), global-norm clip, fused AdamW.  W warm-up steps, then EXACTLY K timed steps between barrier + synchronize
pairs, max over ranks; `value` = samples of ALL ranks per second.  Synthetic data, N(0, 0.02) weights.  At N > 1 the line also
carries `exposed_allreduce_ms` (the step timed again with the collective stubbed) and `rccl_ranks`.

Named legs on the same JSON line (same inputs, each with its own warm-up and barrier-bracketed timing):
  forward        configs[1]: the inference forward (prefill) step, samples/s and fraction of the MFMA peak
  decode         greedy decode model step (tok/s, HBM roofline); generate = MetaModel.generate() END TO END (tokenise ... stop match)
  decode_fp8     configs[4] semantics on the base plugin: weight-only fp8 decode, W8A8 prefill
  train_lora     configs[2]: LoRA r = 16 step (the headline)
  train          full fine-tune step of the same backbone (fp32 masters, AdamW over 6.7 G parameters)
  geometry_R     the reference-faithful geometry (448x448 -> 5 x 224 crops, 1455 image words, S = 1967) next to the headline's S
  m13b           configs[3] shapes: 13B forward + decode, and ONE DP replica of the 13B full fine-tune (288 GB sizing)
  config5        configs[4]: RGB + depth, 1024-token prompt, fp8 weights (and its bf16 twin)
  roofline       dominant kernel family of the headline step (MFMA GEMMs: NT forward, NN input-gradient, TN weight-gradient), HIP events
  recipe         the reference's OWN recipes at its geometry (round 5): LoRA step at max_words 2048 on geometry R (scripts/a3vlm_train.sh:45-55:
                 448^2 input, 1455 image words + 593 text tokens, micro-batch 4 x accum 2) with attention's share; sampled generate on the eval
                 recipe (eval_affordance_v2.py:46-49, a3vlm_infer.sh:37-44: bs 8, max_seq_len 4096, T 0.1, top-p 0.75) -- tok/s at context 1.5 k /
                 2.5 k / 3.5 k and one long run; m13b.train_zero1_recipe = the 13B ZeRO-1 shard at S = 2048, micro-batch 4 x accum 2
  cpu_baseline   the oracle on the host cores: one forward sample + 16 decode steps, and configs[0] (C1) greedy ids GPU == CPU;
                 parity_full_depth_rel_err = the oracle's full-depth, full-width logits against the HIP forward on the same aliased weights
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_BF16 = 2.5e15      # dense, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_PEAK_FP8 = 5.0e15
HBM_PEAK = 8.0e12            # spec; 6.3e12 achievable

ALL_LEGS = ("forward", "decode", "generate", "fp8", "geometry_r", "config5", "recipe", "lora", "loader", "train", "m13b", "cpu")
CORE_LEGS = ("forward", "decode", "lora", "train")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--prompt", type=int, default=512)
    ap.add_argument("--decode-steps", type=int, default=32)
    ap.add_argument("--gen-len", type=int, default=64, help="new tokens of the end-to-end generate() leg")
    ap.add_argument("--model", default="7b", choices=["7b", "13b", "tiny"])
    ap.add_argument("--legs", default=None, help="comma list of " + ",".join(ALL_LEGS) + " (default: all at 1 GPU, core legs at N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the fine-tuning legs (profiling runs); the headline is then the forward step")
    ap.add_argument("--train-steps", type=int, default=None, help="(deprecated) the training legs use --steps / --warmup")
    ap.add_argument("--cpu-seconds", type=float, default=30.0)
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-shape GEMM timing loop (kernel-table runs of the step alone)")
    return ap.parse_args()


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(a):
    """``python bench.py --gpus N`` with N > 1 and no rank environment: re-exec this file under ``torch.distributed.run`` with one
    rank per GPU (RCCL over xGMI) and pass its output and exit code through.  Under a launcher (WORLD_SIZE set) this is a no-op, so
    both the driver's ``python -m torch.distributed.run ... bench.py --gpus N`` and a bare ``python bench.py --gpus N`` end in the
    same N-rank run (reference: one process per GPU, main_finetune.py:241-263 / util/misc.py:138-147)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


GEOM = {
    "7b": dict(dim=4096, n_layers=32, n_heads=32, multiple_of=256),
    "13b": dict(dim=5120, n_layers=40, n_heads=40, multiple_of=256),
    "tiny": dict(dim=256, n_layers=2, n_heads=2, multiple_of=64),
}


def _vit(name):
    return dict(vit_width=1024, vit_layers=24, vit_heads=16) if name != "tiny" else dict(vit_width=128, vit_layers=2, vit_heads=2)


def _init_params(m, dev, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)      # identical weights on every rank
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("norm.weight") or (".ln_" in n and n.endswith("weight")) or n.endswith(".1.weight"):
                p.fill_(1.0)
            elif n.endswith("bias"):
                p.zero_()
            else:
                p.normal_(0.0, 0.02, generator=g)


def build_model(name, dev, max_seq_len):
    from a3vlm_amd.model.LLM import llama_ens5 as plugin
    args = plugin.ModelArgs(vocab_size=32000 if name != "tiny" else 512, max_seq_len=max_seq_len,
                            vit_patch=14, vit_crop=336, n_views=1, **GEOM[name], **_vit(name))
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(dev):
            m = plugin.Transformer(args, with_visual=True)
    finally:
        torch.set_default_dtype(old)
    _init_params(m, dev)
    return m, args


def share_into(cls, new_args, base_model, dev, seed=7):
    """A second plugin instance (another geometry / the two-image / adapter plugin) built around the SAME parameter tensors as
    ``base_model`` wherever name and shape agree (no second copy of the 7B weights); the rest is N(0, 0.02) bf16."""
    with torch.device("meta"):
        m = cls(new_args, with_visual=True)
    base = dict(base_model.named_parameters())
    g = torch.Generator(device=dev).manual_seed(seed)
    for name, p in list(m.named_parameters()):
        mod = m
        parts = name.split(".")
        for q in parts[:-1]:
            mod = getattr(mod, q)
        if name in base and tuple(base[name].shape) == tuple(p.shape):
            setattr(mod, parts[-1], base[name])
        else:
            t = torch.empty(p.shape, dtype=torch.bfloat16, device=dev)
            if name.endswith(".1.weight"):
                t.fill_(1.0)
            elif name.endswith("bias"):
                t.zero_()
            else:
                t.normal_(0.0, 0.02, generator=g)
            setattr(mod, parts[-1], torch.nn.Parameter(t))
    from a3vlm_amd.model.LLM.llama_ens5 import precompute_cos_sin
    if new_args.max_seq_len == base_model.args.max_seq_len:
        m._cos_sin_cpu = base_model._cos_sin_cpu
    else:
        m._cos_sin_cpu = precompute_cos_sin(m.head_dim, new_args.max_seq_len * 2, new_args.rope_theta, new_args.rope_scaling)
    return m


def _ffn(args):
    from a3vlm_amd.model.LLM.llama_ens5 import _ffn_hidden
    return _ffn_hidden(args.dim, args.multiple_of, args.ffn_dim_multiplier)


def flops_forward(args, B, T, W, n_images=1):
    """Algorithmic FLOPs of one forward step (SURVEY.md 8(d) conventions: 2 FLOP/MAC, causal attention at 1/2, LM head on the
    positions actually computed -- the last one for the inference step).  Vision: ``n_views`` crops of ``vit_crop`` per image
    through the frozen ViT, projector Linear(vit_width + extra_feat_dim -> dim) on every ViT token, Q-Former projector if present."""
    S = T + W
    d, L = args.dim, args.n_layers
    hd = d // args.n_heads
    nkv = args.n_kv_heads or args.n_heads
    ffn = _ffn(args)
    p_layer = d * (args.n_heads + 2 * nkv) * hd + d * d + 3 * d * ffn
    f_lin = 2 * p_layer * L * S * B
    f_att = L * 4 * S * S * d * 0.5 * B
    f_head = 2 * d * args.vocab_size * B
    w, Lv = args.vit_width, args.vit_layers
    g = args.vit_crop // args.vit_patch
    Ltok = g * g + 1
    crops = B * args.n_views * n_images
    f_vit_gemm = crops * (2 * 3 * args.vit_patch ** 2 * w * g * g + Lv * 2 * 12 * w * w * Ltok)
    f_vit_att = crops * Lv * 4 * Ltok * Ltok * w
    extra = getattr(args, "extra_feat_dim", 0)
    f_proj = 2 * (w + extra) * d * Ltok * crops + 2 * 768 * d * getattr(args, "qformer_tokens", 0) * crops
    gemm = f_lin + f_head + f_proj + f_vit_gemm
    return dict(total=gemm + f_att + f_vit_att, gemm=gemm, att=f_att + f_vit_att, vit=f_vit_gemm + f_vit_att)


def p_decoder(args):
    d, L = args.dim, args.n_layers
    hd = d // args.n_heads
    nkv = args.n_kv_heads or args.n_heads
    return L * (d * (args.n_heads + 2 * nkv) * hd + d * d + 3 * d * _ffn(args))


def bytes_decode_step(args, B, ctx):
    d, L = args.dim, args.n_layers
    hd = d // args.n_heads
    nkv = args.n_kv_heads or args.n_heads
    return 2 * (p_decoder(args) + d * args.vocab_size) + B * 2 * L * ctx * nkv * hd * 2


def bytes_decoder_matrices(args):
    """bytes SAVED per decode step by 1-byte decoder matrices (wqkv, wo, w1|w3, w2 of every layer) relative to bf16"""
    return p_decoder(args)


class Timer:
    """W warm-up calls, then K calls between (barrier + synchronize) pairs; seconds per call, max over ranks."""

    def __init__(self, dist, dev):
        self.dist, self.dev = dist, dev

    def sync(self):
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def __call__(self, fn, steps, warmup):
        for _ in range(warmup):
            fn()
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        self.sync()
        el = time.perf_counter() - t0
        if self.dist is not None:
            t = torch.tensor([el], device=self.dev, dtype=torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            el = float(t.item())
        return el / steps


def _events(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def time_gemm_shapes(args, B, T, W, dev, train=True):
    """HIP-event time of every distinct MFMA GEMM shape of one step on the stream the step uses (torch's current stream):
    NT = forward linears (a3v_gemm_nt: ring ping-pong kernel / 128x128 kernel), and for the training step NN = input gradients
    (a3v_gemm_nn) and TN = weight gradients (a3v_gemm_tn) of the trainable decoder linears.  Returns per-family
    (FLOP per step, seconds per step) and the per-shape rows."""
    from a3vlm_amd import ops
    S = T + W
    rows = B * S
    d, Lyr, ffn = args.dim, args.n_layers, _ffn(args)
    w, Lv = args.vit_width, args.vit_layers
    g = args.vit_crop // args.vit_patch
    vr = B * (g * g + 1)
    nt_shapes = [  # (M, N, K, count per step, epilogue)
        (rows, 3 * d, d, Lyr, 0), (rows, d, d, Lyr, ops.EPI_RESIDUAL), (rows, 2 * ffn, d, Lyr, ops.EPI_SWIGLU),
        (rows, d, ffn, Lyr, ops.EPI_RESIDUAL),
        (vr, 3 * w, w, Lv, 0), (vr, w, w, Lv, ops.EPI_RESIDUAL), (vr, 4 * w, w, Lv, ops.EPI_GELU), (vr, w, 4 * w, Lv, ops.EPI_RESIDUAL),
        (vr, d, w, 1, 0), (B * g * g, w, 640, 1, 0),
    ]
    fam = {"nt": [0.0, 0.0], "nn": [0.0, 0.0], "tn": [0.0, 0.0], "nt_dgrad": [0.0, 0.0], "nt_dgrad_ft": [0.0, 0.0]}
    table = []
    # the legs before this one end with frees / allocator work on the host: bring the part back to its loaded clock first
    # (the first shape timed after an idle gap read 15 % low)
    wa = torch.randn(rows, d, device=dev, dtype=torch.bfloat16)
    ww = torch.randn(3 * d, d, device=dev, dtype=torch.bfloat16) * 0.02
    wo = torch.empty(rows, 3 * d, device=dev, dtype=torch.bfloat16)
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        for _ in range(8):
            ops.gemm_nt(wa, ww, wo)
        torch.cuda.synchronize()
    del wa, ww, wo
    for (M, N, K, cnt, epi) in nt_shapes:
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        wt = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
        out = torch.zeros(M, N // 2 if epi & ops.EPI_SWIGLU else N, device=dev, dtype=torch.bfloat16)
        res = out if epi & ops.EPI_RESIDUAL else None
        dt = _events(lambda: ops.gemm_nt(a, wt, out, residual=res, epilogue=epi & ~ops.EPI_RESIDUAL))
        fl = 2.0 * M * N * K
        fam["nt"][0] += fl * cnt
        fam["nt"][1] += dt * cnt
        table.append(dict(kind="nt", M=M, N=N, K=K, count=cnt, us=round(dt * 1e6, 1), tflops=round(fl / dt / 1e12, 1)))
        del a, wt, out
    if train:
        # decoder linears only (the ViT is frozen: no gradient GEMMs); (rows, out features, in features)
        for (M, N, K) in [(rows, 3 * d, d), (rows, d, d), (rows, 2 * ffn, d), (rows, d, ffn)]:
            dy = torch.randn(M, N + 64, device=dev, dtype=torch.bfloat16)
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            wimg = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
            wt_img = torch.randn(K, N + 64, device=dev, dtype=torch.bfloat16) * 0.02     # [W^T | A^T] of a LoRA run (frozen base: built once)
            dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
            gw = torch.zeros(N, K, device=dev, dtype=torch.float32)
            fl = 2.0 * M * N * K
            dt = _events(lambda: ops.gemm_nn(dy[:, :N], wimg, dx))
            fam["nn"][0] += fl * Lyr
            fam["nn"][1] += dt * Lyr
            table.append(dict(kind="nn", M=M, N=K, K=N, count=Lyr, us=round(dt * 1e6, 1), tflops=round(fl / dt / 1e12, 1)))
            # the LoRA step's form of the same input gradient: dx = [dy | dt] . [W^T | A^T]^T on the NT ring kernel (K extended by the
            # 64 adapter columns; the algorithmic FLOPs counted are those of the base product)
            dt = _events(lambda: ops.gemm_nt(dy, wt_img, dx))
            fam["nt_dgrad"][0] += fl * Lyr
            fam["nt_dgrad"][1] += dt * Lyr
            table.append(dict(kind="nt_dgrad", M=M, N=K, K=N + 64, count=Lyr, us=round(dt * 1e6, 1), tflops=round(fl / dt / 1e12, 1)))
            # the full fine-tune's form since round 5: dx = dy . (W^T)^T on the NT ring kernel over the transposed image the AdamW pass
            # keeps current (a3v_adamw_scaled_t); the NN family above is what the ZeRO-1 / recompute engines still run
            dt = _events(lambda: ops.gemm_nt(dy[:, :N], wt_img[:, :N], dx))
            fam["nt_dgrad_ft"][0] += fl * Lyr
            fam["nt_dgrad_ft"][1] += dt * Lyr
            table.append(dict(kind="nt_dgrad_ft", M=M, N=K, K=N, count=Lyr, us=round(dt * 1e6, 1), tflops=round(fl / dt / 1e12, 1)))
            dt = _events(lambda: ops.gemm_tn(dy[:, :N], x, gw, epilogue=ops.EPI_OUT_F32))
            fam["tn"][0] += fl * Lyr
            fam["tn"][1] += dt * Lyr
            table.append(dict(kind="tn", M=N, N=K, K=M, count=Lyr, us=round(dt * 1e6, 1), tflops=round(fl / dt / 1e12, 1)))
            del dy, x, wimg, wt_img, dx, gw
    return fam, table


# ---------------------------------------------------------------------------------------------------------------- training legs
def _train_step_fn(eng, opt, red, params, tokens, labels, image):
    from a3vlm_amd.dp import clip_grad_norm, GradSquareSums
    sq = GradSquareSums(eng, red) if os.environ.get("A3V_CLIP_SUMSQ", "1") != "0" else None   # =0: one norm pass after the backward (A/B)
    # =1: the update runs on the optimizer's stream under the next step's forward, bucket by bucket (optim.py).  Off by default:
    # +2 ms at best (the GEMMs leave it no registers to co-reside with), and -40 ms when the runtime maps the two streams onto one
    # hardware queue (seen when another engine's optimizer stream existed earlier in the process)
    overlap = os.environ.get("A3V_ADAMW_OVERLAP", "0") == "1"

    def one():
        loss = eng.forward_loss(tokens, labels, image)
        eng.backward(1.0)
        if red is not None:
            red.finish()
        # global-norm clip of the reference recipe (--clip_grad 8, a3vlm_train.sh:47-55; util/misc.py:302-315): per-bucket sums of
        # squares taken on a side stream while the backward runs, coefficient applied inside the optimizer kernel
        _, coef = clip_grad_norm(params, 8.0, flat=eng.flat_grads(), defer=True, sumsq=sq)
        opt.step(grad_scale=coef, overlap=overlap)
        opt.zero_grad(set_to_none=True)
        one.loss = loss
    return one


def _exposed_allreduce(one, red, sec, steps, timer):  # noqa: D401
    """SURVEY 8(d) "exposed (non-overlapped) all-reduce time": the same step timed again with the collective itself stubbed out
    (the bucket hand-over, wire casts and side-stream events still run), the difference is what the all-reduce adds to the step
    beyond what the backward hides.  None on one rank."""
    if red is None:
        return None
    red.stub_collective = True
    try:
        stub = timer(one, max(2, steps // 2), 1)
    finally:
        red.stub_collective = False
    return {"exposed_allreduce_ms": round((sec - stub) * 1e3, 3), "ms_per_step_without_collective": round(stub * 1e3, 3),
            "bucket_bytes_on_wire": int(red.wire_bytes_last_step)}


def train_leg(m, B, T, image, tokens, steps, warmup, timer, recompute=None):
    """Full fine-tune step on this rank's micro-batch: fp32 masters for the trainables (decoder + projector), frozen bf16 ViT,
    bf16 GEMMs, AdamW(0.9, 0.95), clip 8; with N > 1 the per-layer gradient buckets are averaged over RCCL (bf16 on the wire =
    the reference's FSDP reduce_dtype, main_finetune.py:251-255) on a side stream while earlier layers still back-propagate.
    DESTRUCTIVE for ``m`` (its trainable parameters become fp32)."""
    from a3vlm_amd.train import TrainEngine
    from a3vlm_amd.util import promote_trainable_params_to_fp32
    from a3vlm_amd.dp import GradReducer
    from a3vlm_amd.optim import FusedAdamW
    for n, p in m.named_parameters():
        p.requires_grad = not n.startswith("clip.")
    m._ws.clear(); m._packed.clear(); m._packed_version = None; m._destroy_kv_cache()
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    promote_trainable_params_to_fp32(m)
    eng = TrainEngine(m, torch.bfloat16) if recompute is None else TrainEngine(m, torch.bfloat16, recompute=recompute)
    params = [p for p in m.parameters() if p.requires_grad]
    opt = FusedAdamW(params, lr=2e-5, betas=(0.9, 0.95), weight_decay=0.0, engine=eng)
    red = GradReducer(eng, timer.dist, reduce_dtype=torch.bfloat16) if timer.dist is not None else None
    labels = tokens.clone()
    labels[:, :T // 2] = 0
    one = _train_step_fn(eng, opt, red, params, tokens, labels, image)
    sec = timer(one, steps, max(1, warmup))
    train_leg.exposed = _exposed_allreduce(one, red, sec, steps, timer)
    return sec, float(one.loss), torch.cuda.max_memory_allocated() / 2 ** 30, sum(p.numel() for p in params), bool(eng.recompute)


train_leg.exposed = None


class _ShardOf:
    """torch.distributed's surface of rank 0 of an N-rank job for ``Zero1Optimizer(stub_collective=True)``: on ONE GPU this allocates and
    updates exactly what rank 0 of a DP-N ZeRO-1 job holds (1/N of the masters and moments), with the collectives replaced by the local
    slice copies -- the memory and the compute of one DP-N shard, not its wire time."""
    class ReduceOp:
        SUM, AVG, MAX = "sum", "avg", "max"

    def __init__(self, world):
        self.world = world

    def get_world_size(self, group=None):
        return self.world

    def get_rank(self, group=None):
        return 0

    def get_backend(self, group=None):
        return "stub"


def zero1_leg(m, B, T, image, tokens, steps, warmup, timer, shard_of=8, recompute=False, accum=1):
    """configs[3] under ZeRO-1 (a3vlm_amd/zero1.py = the reference's FSDP(SHARD_GRAD_OP) sizing): full fine-tune step of this rank's
    micro-batch with the big matrices kept only in bf16 (flat buffer = parameters = GEMM images), fp32 masters + AdamW moments of 1/N of
    them.  With a real process group (N = world > 1) the collectives run; on one GPU (``shard_of`` = 8) rank 0's shard of a DP-8 job is
    emulated: same allocations, same kernels, collectives stubbed.  DESTRUCTIVE for ``m`` (small parameters become fp32, the big ones move
    into the engine's flat buffer)."""
    from a3vlm_amd.train import TrainEngine
    from a3vlm_amd.util import promote_trainable_params_to_fp32, add_weight_decay
    from a3vlm_amd.optim import FusedAdamW
    from a3vlm_amd.zero1 import Zero1Optimizer
    for n, p in m.named_parameters():
        p.requires_grad = not n.startswith("clip.")
    m._ws.clear(); m._packed.clear(); m._packed_version = None; m._destroy_kv_cache()
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    promote_trainable_params_to_fp32(m, keep_matrices_sharded=True)
    real = timer.dist is not None
    zd = timer.dist if real else _ShardOf(shard_of)
    world = zd.get_world_size()
    eng = TrainEngine(m, torch.bfloat16, recompute=recompute, zero1_world=world)
    groups = [{**g, "params": [q for q in g["params"] if q.dtype == torch.float32]} for g in add_weight_decay(m, 0.0)]
    small = FusedAdamW([g for g in groups if g["params"]], lr=2e-5, betas=(0.9, 0.95), engine=eng)
    opt = Zero1Optimizer(eng, zd, lr=2e-5, betas=(0.9, 0.95), reduce_dtype=torch.bfloat16, small=small)
    opt.stub_collective = not real
    labels = tokens.clone()
    labels[:, :T // 2] = 0

    def one():
        for micro in range(accum):              # an accumulation window: the exchange runs on its last micro-step only (no_sync before)
            opt.enabled = micro == accum - 1
            loss = eng.forward_loss(tokens, labels, image)
            eng.backward(1.0 / accum)
        opt.finish()
        _, coef = opt.clip_coef(8.0)
        opt.step(grad_scale=coef.reshape(1), overlap=real)
        m.zero_grad(set_to_none=True)
        one.loss = loss
    try:
        sec = timer(one, steps, max(1, warmup))
    finally:
        for p in m.parameters():               # whatever happens, the legs that follow start from parameters without gradient views
            p.grad = None
    mem = torch.cuda.max_memory_allocated() / 2 ** 30
    n_big = sum(b["shard"][1] - b["shard"][0] for b in opt.buckets)
    out = {"sec": sec, "loss": float(one.loss), "hbm_gib": mem, "sharded_params": n_big, "shard_state_gib": opt.shard_bytes() / 2 ** 30,
           "world": world, "emulated": not real, "recompute": bool(eng.recompute), "wire_bytes_per_step": int(opt.wire_bytes_last_step)}
    del eng, opt, small, one
    gc.collect()
    torch.cuda.empty_cache()
    return out


def lora_leg(m, args, B, T, image, tokens, steps, warmup, timer, dev, rank=16):
    """configs[2]: LoRA fine-tune step (rank-16 adapters on the seven decoder linears of every block + norms + projector
    trainable, base matrices frozen in bf16) on this rank's micro-batch; adapter plugin built around the SAME base parameters."""
    import dataclasses
    from a3vlm_amd.model.LLM import llama_ens5_peft as peft
    from a3vlm_amd.train import TrainEngine
    from a3vlm_amd.util import promote_trainable_params_to_fp32
    from a3vlm_amd.dp import GradReducer
    from a3vlm_amd.optim import FusedAdamW
    torch.cuda.reset_peak_memory_stats()
    pm = share_into(peft.Transformer, peft.ModelArgs(**dataclasses.asdict(args), lora_rank=rank), m, dev)   # B != 0: every adapter GEMM works
    train = pm.get_trainable_params()
    for n, p in pm.named_parameters():
        p.requires_grad = n in train
    promote_trainable_params_to_fp32(pm)
    n_train = sum(p.numel() for p in pm.parameters() if p.requires_grad)
    eng = TrainEngine(pm, torch.bfloat16)
    params = [p for p in pm.parameters() if p.requires_grad]
    opt = FusedAdamW(params, lr=2e-5, betas=(0.9, 0.95), weight_decay=0.0, engine=eng)
    red = GradReducer(eng, timer.dist, reduce_dtype=torch.bfloat16) if timer.dist is not None else None
    labels = tokens.clone()
    labels[:, :T // 2] = 0
    one = _train_step_fn(eng, opt, red, params, tokens, labels, image)
    sec = timer(one, steps, max(1, warmup))
    lora_leg.exposed = _exposed_allreduce(one, red, sec, steps, timer)
    mem = torch.cuda.max_memory_allocated() / 2 ** 30
    loss = float(one.loss)
    for n, p in m.named_parameters():      # restore the shared parameters' state for the legs that follow
        p.grad = None
    del eng, opt, red, pm, one
    gc.collect()                           # the engine holds lazy weight-image objects that point back at it: a cycle, not a leak
    torch.cuda.empty_cache()
    return sec, loss, mem, n_train


lora_leg.exposed = None


def loader_leg(m, args, B, T, steps, warmup, timer, dev, rank=16, workers=4):
    """The headline step with the trainer's REAL input pipeline in the loop (reference: engine_finetune.py:13-105 over
    data/conversation/dataset.py:210-273 + data/transform.py:59-68): PNG files on disk -> FinetuneDialogDataset (render the conversation,
    tokenise, label the answers) and PIL decode in DataLoader worker processes -> FinetuneDistSampler -> pinned uint8 batch ->
    a3v_preprocess_batch (PadToSquare / bicubic 336 / normalise on the device) -> engine_finetune.train_one_epoch (LoRA r = 16 step with clip
    and FusedAdamW, the host loop included).  Same model, same step as `train_lora`; what is added is everything between the disk and
    MetaModel.forward.  Synthetic 640 x 480 PNGs and conversations, padded to the same T text tokens."""
    import argparse
    import dataclasses
    import json as _json
    import tempfile
    import numpy as np
    from PIL import Image
    from a3vlm_amd.data.conversation.dataset import FinetuneDialogDataset
    from a3vlm_amd.data.transform import DevicePreprocessLoader, collate_raw_images, get_transform
    from a3vlm_amd.dp import FinetuneDistSampler, GradReducer
    from a3vlm_amd.engine_finetune import train_one_epoch
    from a3vlm_amd.model.LLM import llama_ens5_peft as peft
    from a3vlm_amd.model.meta import MetaModel
    from a3vlm_amd.model.tokenizer import Tokenizer
    from a3vlm_amd.optim import FusedAdamW
    from a3vlm_amd.util import promote_trainable_params_to_fp32
    tok_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "tokenizer.model")
    world, rk = timer_world(timer), (timer.dist.get_rank() if timer.dist is not None else 0)
    n_items = B * world * (steps + warmup)
    tmp = tempfile.mkdtemp(prefix="a3v_loader_")
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:480, 0:640]
    ann = []
    for i in range(n_items):                      # smooth renders with a little noise: PNG sizes like a PartNet render, cheap to write
        if rk == 0 or timer.dist is None:
            base = np.stack([(xx * (1 + i % 3) + yy) % 256, (yy * 2 + 7 * i) % 256, (xx + yy * (1 + i % 5)) % 256], -1).astype(np.float32)
            img = np.clip(base * 0.5 + 64 + rng.normal(0, 2, base.shape), 0, 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(tmp, f"r{i}.png"), compress_level=1)
        box = ",".join(f"[{rng.random():.2f},{rng.random():.2f},{rng.random():.2f}]" for _ in range(8))
        ann.append({"image": f"r{i}.png", "conversations": [
            {"from": "human", "value": "<image>\nDetect all manipulable object parts and provide their 3D bounding boxes."},
            # (six boxes: with the fixture tokenizer the conversation fills all T text tokens, labels to the end -- the same trimmed length
            #  as the device-resident synthetic step)
            {"from": "gpt", "value": "There are six manipulable object parts with their 3d bounding boxes: " + " ".join(f"<box>door</box>[{box}]" for _ in range(6))}]})
    if timer.dist is not None:
        timer.dist.barrier()
    with open(os.path.join(tmp, "mm.json"), "w") as f:
        _json.dump(ann, f)
    with open(os.path.join(tmp, "data.yaml"), "w") as f:
        f.write(f"META:\n  - path: '{tmp}/mm.json'\n    type: 'image_text'\n    root: '{tmp}'\n")
    pm = share_into(peft.Transformer, peft.ModelArgs(**dataclasses.asdict(args), lora_rank=rank), m, dev)
    train = pm.get_trainable_params()
    for n, p in pm.named_parameters():
        p.requires_grad = n in train
    promote_trainable_params_to_fp32(pm)
    mm = MetaModel.__new__(MetaModel)
    torch.nn.Module.__init__(mm)
    mm.llma, mm.tokenizer, mm.llama_type, mm.is_peft = pm, Tokenizer(tok_path), "llama_ens5_peft", True
    eng = mm.train_engine(torch.bfloat16)
    params = [p for p in mm.parameters() if p.requires_grad]
    opt = FusedAdamW(params, lr=2e-5, betas=(0.9, 0.95), weight_decay=0.0, engine=eng)
    red = GradReducer(eng, timer.dist, reduce_dtype=torch.bfloat16) if timer.dist is not None else None
    W = pm.image_words
    ds = FinetuneDialogDataset(os.path.join(tmp, "data.yaml"), get_transform("padded_resize", 336, on_device=True), max_words=T + W, image_words=W,
                               tokenizer=mm.tokenizer, cache_on_disk=False, rank=rk)
    sampler = FinetuneDistSampler(ds, num_replicas=world, rank=rk, shuffle=True, batch_size=B, acc_grad=1, seed=0)
    # what the host side delivers WITHOUT a model behind it, per worker count: PNG decode + conversation render +
    # tokenise + collate + pin, batches of B; the step needs B / step-time samples/s per rank, a DP-8 node eight times that from one host
    loader_only = {}
    if rk == 0:
        for nw in (2, 4, 8):
            lo = torch.utils.data.DataLoader(ds, batch_size=B, sampler=sampler, num_workers=nw, pin_memory=True, drop_last=True,
                                             collate_fn=collate_raw_images, persistent_workers=True, prefetch_factor=4)
            sampler.set_epoch(0, 0)
            it, n_b = iter(lo), 0
            for _ in range(min(3, len(lo))):               # workers started, first batches through
                next(it)
            t_lo = time.perf_counter()
            for _b in it:
                n_b += 1
            if n_b:
                loader_only[str(nw)] = round(n_b * B / (time.perf_counter() - t_lo), 1)
            del it, lo
    if timer.dist is not None:
        timer.dist.barrier()
    inner = torch.utils.data.DataLoader(ds, batch_size=B, sampler=sampler, num_workers=workers, pin_memory=True, drop_last=True,
                                        collate_fn=collate_raw_images, persistent_workers=True, prefetch_factor=4)
    loader = DevicePreprocessLoader(inner, 336, dev, torch.float32)
    marks = {}

    class Timed:
        """the epoch's loader with a timestamp in front of batch `warmup` (worker start-up and the first steps are not the steady state)"""
        def __len__(self):
            return len(loader)

        def __iter__(self):
            for i, b in enumerate(loader):
                if i == warmup:
                    torch.cuda.synchronize()
                    if timer.dist is not None:
                        timer.dist.barrier()
                    marks["t0"] = time.perf_counter()
                yield b
    targs = argparse.Namespace(accum_iter=1, clip_grad=8, lr=2e-5, min_lr=0.0, warmup_epochs=0.0, epochs=1, print_freq=10 ** 6, save_iteration_interval=0)
    torch.cuda.reset_peak_memory_stats()
    try:
        sampler.set_epoch(0, 0)
        stats = train_one_epoch(mm, Timed(), opt, 0, 0, targs, reducer=red, log=lambda *_: None)
        torch.cuda.synchronize()
        if timer.dist is not None:
            timer.dist.barrier()
        t1 = time.perf_counter()
    finally:
        for p in m.parameters():
            p.grad = None
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    n_timed = len(loader) - warmup
    sec = (t1 - marks["t0"]) / max(n_timed, 1)
    del eng, opt, red, pm, mm, loader, inner
    gc.collect()
    torch.cuda.empty_cache()
    need = B / sec                                          # samples/s one rank consumes
    per_worker = max(loader_only.values()) / int(max(loader_only, key=loader_only.get)) if loader_only else None
    return {"samples_s": round(B * world / sec, 3), "ms_per_step": round(sec * 1e3, 2), "steps_timed": n_timed, "workers": workers,
            "prefetch_factor": 4, "persistent_workers": True, "host_cpus": os.cpu_count(),
            "loader_only_samples_s": loader_only or None,
            "samples_s_needed_per_rank": round(need, 1),
            "workers_per_rank_for_dp8": (max(2, int(-(-1.5 * need // per_worker))) if per_worker else None),
            "closs": round(float(stats["closs"]), 4),
            "pipeline": "640x480 PNG files -> FinetuneDialogDataset + PIL decode in DataLoader workers -> FinetuneDistSampler -> pinned uint8 batch -> "
                        "a3v_preprocess_batch on the device -> engine_finetune.train_one_epoch (LoRA r=16, clip 8, FusedAdamW)"}


# ---------------------------------------------------------------------------------------------------------------- inference legs
def decode_leg(m, fwd_prefill, B, T, n_steps, timer, dev, image_words=None):
    """Greedy decode MODEL steps after a prefill (KV cache holds T + W positions): argmax + one-token forward_inference."""
    from a3vlm_amd import ops
    nt = torch.empty(B, dtype=torch.long, device=dev)
    cur = torch.empty(B, 1, dtype=torch.long, device=dev)
    state = {"logits": fwd_prefill(), "pos": T}

    def one():
        ops.argmax(state["logits"], nt)
        cur[:, 0] = nt
        state["logits"] = m.forward_inference(cur, state["pos"], None)
        state["pos"] += 1
    return timer(one, n_steps, 2)


class _SynthTokenizer:
    """Stands in for a 32000-word SentencePiece model (none ships with the repo, no network): deterministic ids per string.
    Only the generate() leg uses it; what is timed is MetaModel.generate's own loop, whatever the ids are."""
    bos_id, eos_id, n_words = 1, 2, 32000

    def __init__(self, prompt_len):
        self.prompt_len = prompt_len

    def encode(self, s, bos, eos):
        import zlib
        g = torch.Generator().manual_seed(zlib.crc32(s.encode()) & 0x7fffffff)
        ids = torch.randint(3, self.n_words, (self.prompt_len - 1,), generator=g).tolist()
        return ([self.bos_id] if bos else []) + ids + ([self.eos_id] if eos else [])

    def encode_segment(self, s):
        return self.encode(s, False, False)[:2]

    def encode_wo_prefix_space(self, s):
        return self.encode(s, False, False)[1:3]

    def decode(self, t):
        return " ".join(map(str, t))


def generate_leg(m, B, T, image, gen_len, timer, dev):
    """MetaModel.generate(temperature=0) end to end: tokenise, left-truncate, prefill (image + T-token prompts), gen_len greedy
    steps with teacher forcing / stop matching / stop_pos bookkeeping (one a3v_generate_step launch per token), detokenise."""
    from a3vlm_amd.model.meta import MetaModel
    mm = MetaModel.__new__(MetaModel)
    torch.nn.Module.__init__(mm)
    mm.llma, mm.tokenizer, mm.llama_type = m, _SynthTokenizer(T), "llama_ens5"
    prompts = [f"synthetic prompt {i}" for i in range(B)]
    out = {}

    def one():
        _, ids = mm.generate(prompts, image, max_gen_len=gen_len, temperature=0.0, additional_stop_symbols=["###"], return_ids=True)
        out["n"] = sum(len(t) for t in ids)
    sec = timer(one, 2, 1)
    # the prefill's share, timed alone, so that the decode-phase rate can be stated next to the model-step number
    tok = torch.tensor([mm.tokenizer.encode(p, True, False) for p in prompts], device=dev)
    pre = timer(lambda: m.forward_inference(tok, 0, image), 2, 1)
    mm.llma = None
    return sec, pre, out["n"]


def geometry_r_leg(m, args, B, T, steps, warmup, timer, dev):
    """Reference-faithful geometry R: 448x448 input -> bicubic 224 global view + 4 quadrant crops through the ViT,
    32 Q-Former tokens + 257 ViT tokens + 2 tags per view = 1455 image words, visual_proj over the 5632-wide feature concat.  The
    three out-of-scope encoders (Q-Former / ConvNeXt-XXL / DINOv2) enter as synthetic feature tensors through the plugin hooks."""
    import dataclasses
    from a3vlm_amd.model.LLM import llama_ens5 as plugin
    ra = plugin.ModelArgs(**{**dataclasses.asdict(args), "vit_crop": 224, "n_views": 5, "extra_feat_dim": 3072 + 1536, "qformer_tokens": 32})
    mr = share_into(plugin.Transformer, ra, m, dev)
    W = mr.image_words
    g = torch.Generator(device=dev).manual_seed(3)
    img = torch.randn(B, 3, 448, 448, device=dev, generator=g).bfloat16()
    tok = torch.randint(3, args.vocab_size, (B, T), device=dev, generator=g)
    tok[:, 0] = 1
    qf = torch.randn(5 * B, 32, 768, device=dev, generator=g).bfloat16()
    extra = [torch.randn(5 * B, 257, 3072, device=dev, generator=g).bfloat16(), torch.randn(5 * B, 257, 1536, device=dev, generator=g).bfloat16()]
    fwd = lambda: mr.forward_inference(tok, 0, img, qformer_feats=qf, extra_feats=extra)  # noqa: E731
    sec = timer(fwd, steps, warmup)
    dsec = decode_leg(mr, fwd, B, T, 16, timer, dev)
    fl = flops_forward(ra, B, T, W)
    res = {"image_words": W, "seq_len": T + W, "forward_samples_s": round(B * timer_world(timer) / sec, 2), "forward_ms": round(sec * 1e3, 2),
           "forward_mfma_frac": round(fl["total"] / sec / MFMA_PEAK_BF16, 4), "decode_tok_s": round(B * timer_world(timer) / dsec, 1),
           "decode_ms_per_step": round(dsec * 1e3, 3),
           "decode_hbm_frac": round(bytes_decode_step(ra, B, T + W + 10) / dsec / HBM_PEAK, 4),
           "note": "geometry R: image_size 448, 5 x 224 crops, W = 1455 (CLIP ViT-L/14 on the HIP path; Q-Former / ConvNeXt-XXL / DINOv2 "
                   "features synthetic, injected through the plugin hooks: out-of-scope frozen encoders)"}
    mr._ws.clear(); mr._destroy_kv_cache()
    del mr
    gc.collect(); torch.cuda.empty_cache()
    return res


def _attention_per_layer(B, S, H, hd, dev):
    """Prefill (with lse) and backward of ONE layer's causal self-attention, timed alone with HIP events: us per layer."""
    from a3vlm_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    spad = (S + 63) // 64 * 64
    q = torch.randn(B, S, H, hd, device=dev, generator=g).bfloat16()
    kc = torch.randn(B, H, spad, hd, device=dev, generator=g).bfloat16()
    vc = torch.randn(B, H, hd, spad, device=dev, generator=g).bfloat16()
    vrows = torch.randn(B, S, H, hd, device=dev, generator=g).bfloat16()
    do = torch.randn(B, S, H, hd, device=dev, generator=g).bfloat16()
    o = torch.empty_like(q)
    lse = torch.empty(B, H, S, device=dev)
    strides = (S * H * hd, H * hd, hd, H * spad * hd, spad * hd, hd, H * hd * spad, hd * spad, spad, S * H * hd, H * hd, hd)
    dq, dk, dv = torch.empty_like(q), torch.empty(B, H, S, hd, dtype=torch.bfloat16, device=dev), torch.empty(B, H, S, hd, dtype=torch.bfloat16, device=dev)
    D = torch.empty(B, S, H, device=dev)
    ws = torch.empty(ops.attention_bwd_workspace_bytes(B, S, H, H, hd), dtype=torch.uint8, device=dev)
    fwd = _events(lambda: ops.attention_lse(q, kc, vc, o, lse, B, S, S, H, H, hd, strides, True), reps=6, warm=2)
    bwd = _events(lambda: ops.attention_bwd(q, kc, H * spad * hd, spad * hd, vrows, S * H * hd, H * hd, hd, o, do, lse, D, dq, dk, dv, B, S, H, H, hd,
                                            True, workspace=ws), reps=6, warm=2)
    fl = 2.0 * B * H * S * S * hd               # causal: half of 4 B H S^2 hd per product pair
    return {"prefill_us": round(fwd * 1e6, 1), "backward_us": round(bwd * 1e6, 1),
            "prefill_mfma_frac": round(fl / fwd / MFMA_PEAK_BF16, 4), "backward_mfma_frac": round(2.5 * fl / bwd / MFMA_PEAK_BF16, 4)}


def recipe_leg(m, args, B, timer, dev, gen_long=1024):
    """The reference's own recipes, at the reference's geometry (VERDICT r4 'missing' 1-2).
    train: scripts/a3vlm_train.sh:45-55 -- max_words 2048 on 448^2 input (W = 1455 image words + 593 text tokens = S 2048), accumulation;
      here the 7B LoRA step (configs[2] adapters) at micro-batch 4 x accum 2 = the headline's 8 samples per optimizer step.
    eval: eval_affordance_v2.py:46-49,258,271 + scripts/a3vlm_infer.sh:37-44 -- bs 8, max_seq_len 4096, max_gen_len 2048, temperature 0.1,
      top-p 0.75: MetaModel.generate end to end with the nucleus sampler in the loop, 256 new tokens from contexts 1.5 k / 2.5 k / 3.5 k
      (KV bytes overtake the weight bytes at ~3.1 k) and one run of `gen_long` new tokens from 1.5 k."""
    import dataclasses
    from a3vlm_amd.model.LLM import llama_ens5 as plugin
    from a3vlm_amd.model.LLM import llama_ens5_peft as peft
    from a3vlm_amd.model.meta import MetaModel
    from a3vlm_amd.train import TrainEngine
    from a3vlm_amd.util import promote_trainable_params_to_fp32
    from a3vlm_amd.dp import GradReducer, clip_grad_norm, GradSquareSums
    from a3vlm_amd.optim import FusedAdamW
    world = timer_world(timer)
    rk = {**dataclasses.asdict(args), "vit_crop": 224, "n_views": 5, "extra_feat_dim": 3072 + 1536, "qformer_tokens": 32, "max_seq_len": 4096}
    g = torch.Generator(device=dev).manual_seed(9)
    out = {}
    part = os.environ.get("A3V_RECIPE_PART", "all")          # "train" / "eval": one half only (step-alone kernel tables: tools/prof_leg.sh)
    # ---------------- eval recipe
    mr = share_into(plugin.Transformer, plugin.ModelArgs(**rk), m, dev)
    W = mr.image_words
    img = torch.randn(B, 3, 448, 448, device=dev, generator=g).bfloat16()
    qf = torch.randn(5 * B, 32, 768, device=dev, generator=g).bfloat16()
    extra = [torch.randn(5 * B, 257, 3072, device=dev, generator=g).bfloat16(), torch.randn(5 * B, 257, 1536, device=dev, generator=g).bfloat16()]
    mr.qformer_fn = lambda views: qf[:views.shape[0]]            # the out-of-scope frozen encoders enter through the plugin's hooks
    mr.extra_feat_fns = [lambda views, e=e: e[:views.shape[0]] for e in extra]
    mm = MetaModel.__new__(MetaModel)
    torch.nn.Module.__init__(mm)
    mm.llma, mm.llama_type = mr, "llama_ens5"
    ev = {"image_words": W, "temperature": 0.1, "top_p": 0.75, "batch": B, "max_seq_len": 4096}
    for ctx in ((1500, 2500, 3500) if part != "train" else ()):
        Tp = ctx - W
        mm.tokenizer = _SynthTokenizer(Tp)
        prompts = [f"eval prompt {i} {ctx}" for i in range(B)]
        n_new = 256
        cnt = {}

        def one():
            _, ids = mm.generate(prompts, img, max_gen_len=n_new, temperature=0.1, top_p=0.75, additional_stop_symbols=["###"], return_ids=True)
            cnt["n"] = sum(len(t) for t in ids)
            cnt["steps"] = max(len(t) for t in ids)
        sec = timer(one, 1, 1)
        tok = torch.tensor([mm.tokenizer.encode(p, True, False) for p in prompts], device=dev)
        pre = timer(lambda: mr.forward_inference(tok, 0, img), 1, 1)
        dphase = max(sec - pre, 1e-9)
        # the batch steps until its LONGEST sequence ends: model steps = that length.  A sampled end-of-sequence shortens some rows (random
        # weights at T = 0.1: run-dependent; r06d drew 1835 of 2048 tokens and the old tokens / B step count made every step look 11 %
        # slower), so the rate is quoted per batch SLOT (B x steps) and the tokens actually kept are reported beside it
        steps = max(cnt["steps"], 1)
        mid = ctx + steps // 2
        ev[f"ctx_{ctx}"] = {"prompt_tokens": Tp, "new_tokens": cnt["n"], "model_steps": steps,
                            "tok_s_after_prefill": round(B * steps * world / dphase, 1),
                            "kept_tok_s_after_prefill": round(cnt["n"] * world / dphase, 1),
                            "ms_per_step": round(dphase / steps * 1e3, 3), "prefill_ms": round(pre * 1e3, 1),
                            "hbm_frac": round(bytes_decode_step(args, B, mid) / (dphase / steps) / HBM_PEAK, 4)}
    mm.tokenizer = _SynthTokenizer(1500 - W)
    prompts = [f"eval prompt {i} long" for i in range(B)]
    cnt = {}

    def long_one():
        _, ids = mm.generate(prompts, img, max_gen_len=gen_long, temperature=0.1, top_p=0.75, additional_stop_symbols=["###"], return_ids=True)
        cnt["n"] = sum(len(t) for t in ids)
    if part != "train":
        sec = timer(long_one, 1, 0)
        ev["long_run"] = {"from_ctx": 1500, "new_tokens": cnt["n"], "seconds": round(sec, 3), "tok_s_end_to_end": round(cnt["n"] * world / sec, 1)}
    ev["note"] = ("MetaModel.generate(temperature=0.1, top_p=0.75) end to end on geometry R (448^2 -> 5 x 224 crops, W = 1455; Q-Former / ConvNeXt / DINOv2 "
                  "features synthetic through the plugin hooks): a3v_sample_top_p + a3v_generate_step per token; hbm_frac = bf16 weights + KV of all "
                  "sequences at the run's middle context / 8 TB/s")
    out["eval"] = ev
    mm.llma = None
    mr._ws.clear(); mr._destroy_kv_cache()
    del mr, mm
    gc.collect(); torch.cuda.empty_cache()
    if part == "eval":
        return out
    # ---------------- training recipe
    mb, accum, S = 4, 2, 2048
    Tt = S - W
    torch.cuda.reset_peak_memory_stats()
    pm = share_into(peft.Transformer, peft.ModelArgs(**rk, lora_rank=16), m, dev)
    train = pm.get_trainable_params()
    for n, p in pm.named_parameters():
        p.requires_grad = n in train
    promote_trainable_params_to_fp32(pm)
    eng = TrainEngine(pm, torch.bfloat16)
    params = [p for p in pm.parameters() if p.requires_grad]
    opt = FusedAdamW(params, lr=2e-5, betas=(0.9, 0.95), weight_decay=0.0, engine=eng)
    red = GradReducer(eng, timer.dist, reduce_dtype=torch.bfloat16) if timer.dist is not None else None
    toks = [torch.randint(3, args.vocab_size, (mb, Tt), device=dev, generator=g) for _ in range(accum)]
    for t in toks:
        t[:, 0] = 1
    labs = [t.clone() for t in toks]
    for l in labs:
        l[:, :Tt // 2] = 0
    imgs = [img[i * mb:(i + 1) * mb].contiguous() for i in range(accum)]
    qfs = [qf[i * 5 * mb:(i + 1) * 5 * mb].contiguous() for i in range(accum)]
    exs = [[e[i * 5 * mb:(i + 1) * 5 * mb].contiguous() for e in extra] for i in range(accum)]
    sq = GradSquareSums(eng, red)

    def step():
        for i in range(accum):
            if red is not None:
                red.enabled = i == accum - 1
            sq.enabled = i == accum - 1
            loss = eng.forward_loss(toks[i], labs[i], imgs[i], qformer_feats=qfs[i], extra_feats=exs[i])
            eng.backward(1.0 / accum)
        if red is not None:
            red.finish()
        _, coef = clip_grad_norm(params, 8.0, flat=eng.flat_grads(), defer=True, sumsq=sq)
        opt.step(grad_scale=coef)
        opt.zero_grad(set_to_none=True)
        step.loss = loss
    try:
        sec = timer(step, 3, 2)
    finally:
        sq.detach()
    fl = flops_forward(plugin.ModelArgs(**rk), mb, Tt, W)
    att = _attention_per_layer(mb, S, args.n_heads, args.dim // args.n_heads, dev)
    att_ms = accum * args.n_layers * (att["prefill_us"] + att["backward_us"]) * 1e-3
    out["train_lora"] = {"seq_len": S, "image_words": W, "text_tokens": Tt, "micro_batch": mb, "accum": accum,
                         "ms_per_optimizer_step": round(sec * 1e3, 2), "ms_per_micro_step": round(sec / accum * 1e3, 2),
                         "samples_s": round(mb * accum * world / sec, 3), "tokens_s": round(mb * accum * S * world / sec, 0),
                         "loss": round(float(step.loss), 4), "hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                         "mfma_frac": round(2 * accum * fl["total"] / sec / MFMA_PEAK_BF16, 4),
                         "attention_per_layer": att, "attention_ms_per_step": round(att_ms, 2), "attention_share": round(att_ms / (sec * 1e3), 4),
                         "note": "scripts/a3vlm_train.sh:45-55 geometry: max_words 2048 (W = 1455 image words of a 448^2 input + 593 text tokens), "
                                 "LoRA r = 16 on the 7B backbone, micro-batch 4 x accum 2 (the reference: 13B, bs 2 x accum 8 under FSDP + checkpointing); "
                                 "attention timed alone per layer at this geometry x layers x micro-steps"}
    for n, p in m.named_parameters():
        p.grad = None
    del eng, opt, red, pm, step, sq
    gc.collect(); torch.cuda.empty_cache()
    return out


def wire_prediction(n_big_7b_lora, p13, world=8):
    """What the first real DP-8 run should show (stated BEFORE it is measured).  xGMI: 7 links x ~153 GB/s per GPU,
    point to point.  A ring collective moves 2 (N-1)/N x bytes per GPU through ONE link per direction; RCCL's direct all-to-all forms can
    use all 7.  exposed = wire time minus the compute it hides under (reduce-scatter / all-reduce under the backward ~ 2/3 of the step,
    all-gather under the next forward ~ 1/3)."""
    link, nl = 153e9, 7
    per = (world - 1) / world

    def row(phases, step_ms):          # phases: [(bytes, fraction of the step the phase can hide under)]
        t1 = [b * per / link * 1e3 for b, _ in phases]
        t7 = [b * per / (link * nl) * 1e3 for b, _ in phases]
        return {"wire_ms_one_link_ring": round(sum(t1), 1), "wire_ms_all_7_links": round(sum(t7), 1),
                "exposed_ms_overlapped_one_link_ring": round(sum(max(0.0, t - f * step_ms) for t, (_, f) in zip(t1, phases)), 1),
                "exposed_ms_overlapped_all_7_links": round(sum(max(0.0, t - f * step_ms) for t, (_, f) in zip(t7, phases)), 1),
                "exposed_ms_not_overlapped_one_link_ring": round(sum(t1), 1), "exposed_ms_not_overlapped_all_7_links": round(sum(t7), 1),
                "one_gpu_step_ms": round(step_ms, 1)}
    return {"assumptions": "DP 8, bf16 wire, 7 xGMI links x 153 GB/s per GPU; ring = (N-1)/N x bytes per phase through one link, all-links = the same "
                           "bytes over 7; the gradient phase(s) hide under the backward (2/3 of the measured one-GPU step), ZeRO-1's parameter "
                           "all-gather under the AdamW tail + the next forward (1/3)",
            "configs2_lora": None if n_big_7b_lora is None else row([(n_big_7b_lora[0] * 2, 1 / 3), (n_big_7b_lora[0] * 2, 1 / 3)], n_big_7b_lora[1]),
            "configs3_zero1": None if p13 is None else row([(p13[0] * 2, 2 / 3), (p13[0] * 2, 1 / 3)], p13[1])}


def timer_world(timer):
    return timer.dist.get_world_size() if timer.dist is not None else 1


def config5_leg(m, args, B, steps, warmup, timer, dev):
    """BASELINE configs[4]: RGB + depth images (two-image plugin), 1024-token prompt, fp8 weight path (W8A8 prefill, weight-only
    fp8 decode); the bf16 run of the same workload beside it."""
    import dataclasses
    from a3vlm_amd.model.LLM import llama_ens5_2images as p2
    a2 = p2.ModelArgs(**{**dataclasses.asdict(args), "max_seq_len": 4096})
    m2 = share_into(p2.Transformer, a2, m, dev)
    T = 1024
    g = torch.Generator(device=dev).manual_seed(5)
    img = torch.randn(B, 3, 336, 336, device=dev, generator=g).bfloat16()
    dep = torch.randn(B, 3, 336, 336, device=dev, generator=g).bfloat16()
    tok = torch.randint(3, args.vocab_size, (B, T), device=dev, generator=g)
    tok[:, 0] = 1
    W = m2.image_words
    fl = flops_forward(a2, B, T, W, n_images=2)
    world = timer_world(timer)
    res = {"image_words": W, "seq_len": T + W}
    fwd = lambda: m2.forward_inference(tok, 0, img, dep)  # noqa: E731
    for tag in ("bf16", "fp8"):
        if tag == "fp8":
            m2.quantize_decode_weights("fp8", prefill=True)
        sec = timer(fwd, steps, warmup)
        dsec = decode_leg(m2, fwd, B, T, 16, timer, dev)
        db = bytes_decode_step(a2, B, T + W + 10) - (bytes_decoder_matrices(a2) if tag == "fp8" else 0)
        res[tag] = {"forward_samples_s": round(B * world / sec, 2), "forward_ms": round(sec * 1e3, 2),
                    "forward_tflops": round(fl["total"] * world / sec / 1e12, 1), "decode_tok_s": round(B * world / dsec, 1),
                    "decode_ms_per_step": round(dsec * 1e3, 3), "decode_hbm_frac": round(db / dsec / HBM_PEAK, 4)}
    res["fp8"]["forward_frac_of_fp8_peak"] = round(fl["total"] / (res["fp8"]["forward_ms"] * 1e-3) / MFMA_PEAK_FP8, 4)
    res["note"] = ("configs[4]: llama_ens5_2images, one 336x336 crop per image (2 x 579 words), 1024-token prompt; fp8 = "
                   "quantize_decode_weights('fp8', prefill=True): W8A8 decoder GEMMs on the MX-scaled MFMA + weight-only fp8 decode. "
                   "No reference oracle exists for fp8 (SURVEY 8(a) row Q); parity: tests/test_gpu_two_image.py, tests/test_gpu_fp8.py")
    m2.quantize_decode_weights(None)
    m2._ws.clear(); m2._destroy_kv_cache()
    del m2
    gc.collect(); torch.cuda.empty_cache()
    return res


def m13b_leg(B, T, steps, warmup, timer, dev):
    """configs[3] shapes: Llama-2-13B geometry.  Inference forward + decode at bs 8, then ONE DP replica of the full fine-tune
    (every rank of a DP = 8 job holds exactly this: fp32 masters + flat fp32 grads + AdamW state + bf16 images ~ 234 GB, block
    activations recomputed) at micro-batch 4."""
    world = timer_world(timer)
    m, args = build_model("13b", dev, 2048)
    W = m.image_words
    g = torch.Generator(device=dev).manual_seed(1)
    img = torch.randn(B, 3, 336, 336, device=dev, generator=g).bfloat16()
    tok = torch.randint(3, args.vocab_size, (B, T), device=dev, generator=g)
    tok[:, 0] = 1
    fwd = lambda: m.forward_inference(tok, 0, img)  # noqa: E731
    sec = timer(fwd, max(2, steps // 2), 1)
    dsec = decode_leg(m, fwd, B, T, 16, timer, dev)
    fl = flops_forward(args, B, T, W)
    res = {"forward_samples_s": round(B * world / sec, 2), "forward_ms": round(sec * 1e3, 1), "forward_mfma_frac": round(fl["total"] / sec / MFMA_PEAK_BF16, 4),
           "decode_tok_s": round(B * world / dsec, 1), "decode_ms_per_step": round(dsec * 1e3, 3),
           "decode_hbm_frac": round(bytes_decode_step(args, B, T + W + 10) / dsec / HBM_PEAK, 4)}
    # configs[3] under ZeRO-1: rank 0's shard of a DP-8 job (or the real N ranks), micro-batch 8, block activations kept (no recompute)
    try:
        z = zero1_leg(m, B, T, img, tok, 2, 1, timer, shard_of=8, recompute=False)
        flz = flops_forward(args, B, T, W)
        res["train_zero1"] = {"micro_batch": B, "samples_s": round(B * world / z["sec"], 2), "ms_per_step": round(z["sec"] * 1e3, 1), "loss": round(z["loss"], 4),
                              "hbm_gib": round(z["hbm_gib"], 1), "shard_of": z["world"], "collectives": "stubbed (one-GPU emulation of rank 0 of DP-8)" if z["emulated"] else "RCCL",
                              "optimizer_state_gib_per_rank": round(z["shard_state_gib"], 1), "sharded_params": z["sharded_params"], "recompute": z["recompute"],
                              "wire_bytes_per_step": z["wire_bytes_per_step"], "mfma_frac_3x": round(3 * flz["total"] / z["sec"] / MFMA_PEAK_BF16, 4),
                              "note": "ZeRO-1 (main_finetune --zero1; the reference's FSDP(SHARD_GRAD_OP), main_finetune.py:241-263): bf16 parameters = GEMM "
                                      "images in one flat buffer, reduce-scatter -> AdamW on 1/N of the fp32 masters + moments -> all-gather"}
    except Exception as e:
        res["train_zero1"] = {"samples_s": None, "error": repr(e)[:300]}
    # the same shard at the reference recipe's length (scripts/a3vlm_train.sh:45-55: max_words 2048, accumulation): S = 2048 = 579 image words +
    # 1469 text tokens, micro-batch 4 x accum 2
    try:
        mb2, acc2, T2 = 4, 2, 2048 - W
        tok2 = torch.randint(3, args.vocab_size, (mb2, T2), device=dev, generator=g)
        tok2[:, 0] = 1
        z = zero1_leg(m, mb2, T2, img[:mb2].contiguous(), tok2, 2, 1, timer, shard_of=8, recompute=False, accum=acc2)
        flz = flops_forward(args, mb2, T2, W)
        res["train_zero1_recipe"] = {"seq_len": 2048, "micro_batch": mb2, "accum": acc2, "samples_s": round(mb2 * acc2 * world / z["sec"], 2),
                                     "ms_per_optimizer_step": round(z["sec"] * 1e3, 1), "ms_per_micro_step": round(z["sec"] / acc2 * 1e3, 1),
                                     "hbm_gib": round(z["hbm_gib"], 1), "loss": round(z["loss"], 4),
                                     "mfma_frac_3x": round(3 * acc2 * flz["total"] / z["sec"] / MFMA_PEAK_BF16, 4),
                                     "collectives": "stubbed (one-GPU emulation of rank 0 of DP-8)" if z["emulated"] else "RCCL",
                                     "note": "13B ZeRO-1 shard at the reference recipe's max_words 2048 with an accumulation window of 2 (the exchange "
                                             "and the update once per window)"}
    except Exception as e:
        res["train_zero1_recipe"] = {"samples_s": None, "error": repr(e)[:300]}
    # ... and at the recipe's own memory regime (scripts/a3vlm_train.sh:45-55 + main_finetune.py --checkpointing: batch 2 x accum 8 with
    # activation checkpointing): per-block recompute, micro-batch 2, eight micro-steps per exchange / update (VERDICT r5 'missing' 4)
    try:
        mb3, acc3, T3 = 2, 8, 2048 - W
        tok3 = torch.randint(3, args.vocab_size, (mb3, T3), device=dev, generator=g)
        tok3[:, 0] = 1
        z = zero1_leg(m, mb3, T3, img[:mb3].contiguous(), tok3, 1, 1, timer, shard_of=8, recompute=True, accum=acc3)
        flz = flops_forward(args, mb3, T3, W)
        res["train_zero1_recipe_ckpt"] = {"seq_len": 2048, "micro_batch": mb3, "accum": acc3, "recompute": z["recompute"],
                                          "samples_s": round(mb3 * acc3 * world / z["sec"], 2),
                                          "ms_per_optimizer_step": round(z["sec"] * 1e3, 1), "ms_per_micro_step": round(z["sec"] / acc3 * 1e3, 1),
                                          "hbm_gib": round(z["hbm_gib"], 1), "loss": round(z["loss"], 4),
                                          "mfma_frac_3x": round(3 * acc3 * flz["total"] / z["sec"] / MFMA_PEAK_BF16, 4),
                                          "collectives": "stubbed (one-GPU emulation of rank 0 of DP-8)" if z["emulated"] else "RCCL",
                                          "note": "13B ZeRO-1 shard at the reference recipe's own regime: max_words 2048, batch 2 x accum 8, activation "
                                                  "checkpointing (per-block recompute); 3 x forward FLOP convention although a recomputing step executes 4 x"}
    except Exception as e:
        res["train_zero1_recipe_ckpt"] = {"samples_s": None, "error": repr(e)[:300]}
    mb = 4
    try:
        tsec, loss, mem, ntr, rec = train_leg(m, mb, T, img[:mb].contiguous(), tok[:mb].contiguous(), 2, 1, timer, recompute=True)
        flt = flops_forward(args, mb, T, W)
        res["train_replica"] = {"micro_batch": mb, "samples_s": round(mb * world / tsec, 2), "ms_per_step": round(tsec * 1e3, 1), "loss": round(loss, 4),
                                "hbm_gib": round(mem, 1), "trainable_params": ntr, "recompute": rec,
                                "mfma_frac_3x": round(3 * flt["total"] / tsec / MFMA_PEAK_BF16, 4),
                                "note": "one pure-DP replica of configs[3] (13B full fine-tune) in 288 GB HBM; gradients would be "
                                        "all-reduced in bf16 per layer bucket exactly as in the 7B headline leg"}
    except Exception as e:
        res["train_replica"] = {"samples_s": None, "error": repr(e)[:300]}
    del m
    gc.collect(); torch.cuda.empty_cache()
    return res


# ---------------------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(args, T, W, seconds, dev):
    """The CPU oracle (oracle/ref_cpu.py, kind "port": the reference has no compiled code and cannot travel) on the host cores,
    bf16: (i) ONE sample of the headline workload's forward (full ViT-L/14@336 + projector + all decoder layers over 1091
    positions + LM head) and 16 greedy decode steps on the cache it leaves; every decoder layer aliases ONE set of weights so the
    host RAM stays bounded (identical FLOPs / bytes per layer); if the time budget runs out the decoder is cut after k layers and
    the rate scaled by the layer ratio (stated in `sample`).  (ii) configs[0] (C1): tiny weights, one 336x336 image + 64-token
    prompt, greedy decode on the CPU oracle and through the HIP path in fp32 -- token ids must be identical."""
    from oracle import ref_cpu
    ncpu = os.cpu_count() or 1
    xs = torch.randn(T + W, args.dim)
    ws = torch.randn(4096, args.dim) * 0.02
    best = None
    for th in sorted({min(ncpu, t) for t in (16, 32, 64, 128, ncpu)}):
        torch.set_num_threads(th)
        for dtc in (torch.float32, torch.bfloat16):
            xa, wa = xs.to(dtc), ws.to(dtc)
            torch.nn.functional.linear(xa, wa)
            t0 = time.perf_counter()
            for _ in range(2):
                torch.nn.functional.linear(xa, wa)
            el = (time.perf_counter() - t0) / 2
            if best is None or el < best[0]:
                best = (el, th, dtc)
    _, cores, dt = best
    torch.set_num_threads(cores)
    oargs = ref_cpu.OracleArgs(dim=args.dim, n_layers=1, n_heads=args.n_heads, n_kv_heads=args.n_kv_heads,
                               vocab_size=args.vocab_size, multiple_of=args.multiple_of, max_seq_len=2048)
    sd1 = ref_cpu.make_decoder_weights(oargs, seed=0, std=0.02, dtype=dt)
    g = args.vit_crop // args.vit_patch
    vsd = ref_cpu.make_vision_weights(args.dim, width=args.vit_width, layers=args.vit_layers, patch=args.vit_patch,
                                      grid=g, seed=1, std=0.02, dtype=dt)
    dec = ref_cpu.OracleDecoder(oargs, sd1)
    img = torch.randn(1, 3, args.vit_crop, args.vit_crop).to(dt)
    tok = torch.randint(3, args.vocab_size, (1, T))
    tok[:, 0] = 1
    t0 = time.perf_counter()
    with torch.no_grad():
        views = ref_cpu.encode_image(img, vsd, vit_layers=args.vit_layers, vit_heads=args.vit_heads, n_views=1)
        itok = ref_cpu.assemble_image_tokens(views, vsd["start_img"], vsd["end_img"])
        h = dec.embed(tok)
        h = torch.cat((h[:, :1], itok.to(h.dtype), h[:, 1:]), dim=1)
        S = h.shape[1]
        fc = dec.freqs_cis[:S]
        t_vit = time.perf_counter() - t0
        done = 0
        for i in range(args.n_layers):
            h = dec.block(0, h, 0, fc, "causal")
            done += 1
            if time.perf_counter() - t0 > seconds and done < args.n_layers:
                break
        hn = ref_cpu.rmsnorm(h, sd1["norm.weight"], oargs.norm_eps)
        oracle_logits = torch.nn.functional.linear(hn[:, -1, :], sd1["output.weight"]).float()
        el = time.perf_counter() - t0
        n_iw = itok.shape[1]
        # (outside the timed sample) every text position's logits of this sample and of a second one, for the full-depth parity check
        par_samples = [(tok, img, torch.nn.functional.linear(hn[0, n_iw:, :], sd1["output.weight"]).float())]
        if os.environ.get("A3V_BENCH_PARITY_SAMPLES", "2") != "1":
            g2 = torch.Generator().manual_seed(77)
            img2 = torch.randn(1, 3, args.vit_crop, args.vit_crop, generator=g2).to(dt)
            tok2 = torch.randint(3, args.vocab_size, (1, T), generator=g2)
            tok2[:, 0] = 1
            it2 = ref_cpu.assemble_image_tokens(ref_cpu.encode_image(img2, vsd, vit_layers=args.vit_layers, vit_heads=args.vit_heads, n_views=1),
                                                vsd["start_img"], vsd["end_img"])
            h2 = dec.embed(tok2)
            h2 = torch.cat((h2[:, :1], it2.to(h2.dtype), h2[:, 1:]), dim=1)
            for i in range(done):
                h2 = dec.block(0, h2, 0, fc, "causal")
            hn2 = ref_cpu.rmsnorm(h2, sd1["norm.weight"], oargs.norm_eps)
            par_samples.append((tok2, img2, torch.nn.functional.linear(hn2[0, n_iw:, :], sd1["output.weight"]).float()))
            del h2, hn2
        # 16 decode steps: one new token against S cached positions; the same aliased layer n_layers times per step
        dec.allocate_kv_cache(1)
        dec.k_cache[0] = torch.randn(dec.k_cache[0].shape).to(dt)      # a context of S positions (contents irrelevant for timing)
        dec.v_cache[0] = torch.randn(dec.v_cache[0].shape).to(dt)
        t1 = time.perf_counter()
        nd = 0
        for stp in range(16):
            x = dec.embed(tok[:, :1])
            for i in range(args.n_layers):
                x = dec.block(0, x, S + stp, dec.freqs_cis[S + stp:S + stp + 1], None)
            _ = torch.nn.functional.linear(ref_cpu.rmsnorm(x, sd1["norm.weight"], oargs.norm_eps)[:, -1, :], sd1["output.weight"]).float()
            nd += 1
            if time.perf_counter() - t1 > seconds / 2:
                break
        d_el = (time.perf_counter() - t1) / nd
    t_dec = el - t_vit
    full = t_vit + t_dec * (args.n_layers / done)
    out = dict(value=round(1.0 / full, 5), unit="samples/s", cores=cores, kind="port",
               sample=(f"oracle/ref_cpu.py {str(dt).split('.')[-1]} on {cores} of {ncpu} host threads (fastest of a threads x dtype probe), 1 sample "
                       f"(336x336 image, {T}-token prompt, S={S}): ViT-L/14 24 blocks {t_vit:.1f}s + {done}/{args.n_layers} decoder layers {t_dec:.1f}s "
                       f"(weights aliased across layers)" + ("" if done == args.n_layers else f"; decoder time scaled x{args.n_layers / done:.2f} by layer count")
                       + f"; then {nd} greedy decode steps at context {S} (batch 1)"),
               seconds=round(el, 1), decode_tok_s=round(1.0 / d_el, 3), decode_ms_per_step=round(d_el * 1e3, 1))
    # full depth at full width (VERDICT r4 'missing' 3): the logits the oracle just computed through `done` aliased 7B-width layers against the
    # HIP forward of a `done`-layer plugin whose layers alias the SAME weights (LLM/llama_ens5.py:461-487)
    try:
        out["parity_full_depth"] = full_depth_parity(args, done, sd1, vsd, par_samples, dev)
        out["parity_full_depth_rel_err"] = out["parity_full_depth"]["rel_err"]
        out["parity_full_depth_decided_frac"] = out["parity_full_depth"]["decided_frac"]
        out["parity_full_depth_agreement"] = out["parity_full_depth"]["agreement_on_decided"]
    except Exception as e:
        out["parity_full_depth"] = {"error": repr(e)[:300]}
        out["parity_full_depth_rel_err"] = out["parity_full_depth_decided_frac"] = out["parity_full_depth_agreement"] = None
    out["c1"] = c1_check(dev)
    return out


def _parity_stats(got, want):
    """got / want: fp32 logits [P, V] of P positions.  `noise` is the largest |difference| anywhere; a position is DECIDED when the
    oracle's top-2 margin exceeds 2 x noise (the greedy token cannot depend on the arithmetic then)."""
    diff = (got - want).abs()
    scale = float(want.abs().max())
    noise = float(diff.max())
    top2 = torch.topk(want, 2, dim=-1).values
    margin = (top2[:, 0] - top2[:, 1])
    decided = margin > 2.0 * noise
    same = got.argmax(-1) == want.argmax(-1)
    sigma = float(diff.pow(2).mean().sqrt())
    dec4 = margin > 2.0 * 4.0 * sigma                  # non-tautological variant: 4 sigma of the RMS difference instead of the maximum
    return {"positions": int(want.shape[0]), "rel_err": round(noise / scale, 5), "rms_rel_err": round(sigma / scale, 6), "max_abs_logit": round(scale, 4),
            "decided_frac": round(float(decided.float().mean()), 4),
            "agreement_on_decided": round(float(same[decided].float().mean()), 4) if bool(decided.any()) else None,
            "decided_frac_4sigma": round(float(dec4.float().mean()), 4),
            "agreement_on_decided_4sigma": round(float(same[dec4].float().mean()), 4) if bool(dec4.any()) else None,
            "argmax_agreement_all_positions": round(float(same.float().mean()), 4),
            "median_margin_over_noise": round(float(margin.median()) / max(noise, 1e-12), 3)}


def full_depth_parity(args, n_layers, sd1, vsd, samples, dev):
    """HIP forward (bf16) of ViT-L/14 + projector + n_layers decoder layers ALIASING one layer's weights + LM head over EVERY text
    position (llama_ens5.py:485-487 slices h[:, image_words:]) of the oracle's samples -- `samples` = [(tokens [1, T], image, oracle
    fp32 logits [T, V])] -- and the same comparison of the fp8 (W8A8) image of the model against its bf16 image."""
    import dataclasses
    import re
    from a3vlm_amd import ops
    from a3vlm_amd.model.LLM import llama_ens5 as plugin
    hargs = dataclasses.replace(args, n_layers=n_layers)
    with torch.device("meta"):
        hm = plugin.Transformer(hargs, with_visual=True)
    cache = {}
    src_all = {**sd1, **vsd}
    for name, p in list(hm.named_parameters()):
        key = re.sub(r"^layers\.\d+\.", "layers.0.", name)
        if key not in cache:
            if key not in src_all:
                raise KeyError(f"no oracle weight for {name}")
            cache[key] = torch.nn.Parameter(src_all[key].to(torch.bfloat16).to(dev), requires_grad=False)
        mod = hm
        parts = name.split(".")
        for q in parts[:-1]:
            mod = getattr(mod, q)
        setattr(mod, parts[-1], cache[key])
    hm._cos_sin_cpu = plugin.precompute_cos_sin(hm.head_dim, hargs.max_seq_len * 2, hargs.rope_theta, hargs.rope_scaling)
    tok = torch.cat([s_[0] for s_ in samples]).to(dev)
    img = torch.cat([s_[1] for s_ in samples]).to(torch.bfloat16).to(dev)
    want = torch.cat([s_[2].float() for s_ in samples])                     # [n T, V]
    B, T = tok.shape
    W = hm.image_words

    def hip_logits():
        # the plugin's forward rounds its logits to the model dtype; the comparison wants the fp32 accumulators of the same LM-head GEMM
        with torch.no_grad():
            hm.forward(tok, img)
            xn = hm._buf("xn_final", (B * (T + W), hargs.dim)).view(B, T + W, hargs.dim)
            out = torch.empty(B, T, hargs.vocab_size, dtype=torch.float32, device=dev)
            for b_ in range(B):
                ops.gemm_nt(xn[b_, W:], hm.output.weight, out[b_], epilogue=ops.EPI_OUT_F32)
        return out.view(B * T, -1).cpu()

    got = hip_logits()
    res = {"layers": n_layers, "samples": len(samples), **_parity_stats(got, want)}
    res["argmax_equal"] = res["agreement_on_decided"] in (None, 1.0)
    try:
        hm.quantize_decode_weights("fp8", prefill=True)
        got8 = hip_logits()
        res["fp8_vs_bf16"] = _parity_stats(got8, got)
    except Exception as e:
        res["fp8_vs_bf16"] = {"error": repr(e)[:200]}
    finally:
        try:
            hm.quantize_decode_weights(None)
        except Exception:
            pass
    res["note"] = ("LM head over every text position of the samples (336x336 image + prompt, S = 1091), oracle in its timed dtype on the host vs the "
                   "HIP bf16 path; every decoder layer aliases ONE set of 7B-width weights on both sides; noise = largest |logit difference| "
                   "anywhere, decided = oracle top-2 margin > 2 x noise; fp8_vs_bf16 = the W8A8 image of the same model against its bf16 image")
    hm._ws.clear(); hm._destroy_kv_cache()
    del hm, cache
    gc.collect(); torch.cuda.empty_cache()
    return res


def c1_check(dev):
    """configs[0]: single 336x336 render + 64-token prompt through the CPU eager path (the oracle), greedy decode; the HIP path in
    fp32 must produce the same ids (north_star: token ids bit-exact under greedy decode)."""
    from a3vlm_amd.model.LLM import llama_ens5 as plugin
    from oracle import ref_cpu
    kw = dict(dim=128, n_layers=2, n_heads=2, vocab_size=256, multiple_of=64, max_seq_len=1024)
    oargs = ref_cpu.OracleArgs(**kw)
    sd = ref_cpu.make_decoder_weights(oargs, seed=3, std=0.05)
    vsd = ref_cpu.make_vision_weights(128, width=128, layers=2, patch=14, grid=24, seed=4, std=0.05)
    g = torch.Generator().manual_seed(0)
    img = torch.randn(1, 3, 336, 336, generator=g).half().float()
    tok = torch.randint(3, 256, (1, 64), generator=g)
    tok[:, 0] = 1
    n_new = 16
    t0 = time.perf_counter()
    dec = ref_cpu.OracleDecoder(oargs, sd)
    itok = ref_cpu.assemble_image_tokens(ref_cpu.encode_image(img, vsd, vit_layers=2, vit_heads=2, n_views=1), vsd["start_img"], vsd["end_img"])
    lg = dec.forward_inference(tok, 0, itok)
    want = []
    for i in range(n_new):
        nt = int(lg.argmax(-1))
        want.append(nt)
        lg = dec.forward_inference(torch.tensor([[nt]]), 64 + i)
    cpu_s = time.perf_counter() - t0
    m = plugin.Transformer(plugin.ModelArgs(**kw, vit_width=128, vit_layers=2, vit_heads=2, vit_crop=336, n_views=1), with_visual=True)
    m.load_state_dict({**sd, **vsd})
    m.to(dev)
    lg = m.forward_inference(tok.to(dev), 0, img.to(dev))
    got = []
    for i in range(n_new):
        nt = int(lg.argmax(-1))
        got.append(nt)
        lg = m.forward_inference(torch.tensor([[nt]], device=dev), 64 + i)
    return {"config": "configs[0]: 336x336 image + 64-token prompt, tiny weights, greedy, fp32", "ids_equal": got == want, "n_tokens": n_new,
            "cpu_seconds": round(cpu_s, 2)}


def pmc_traffic():
    """HBM-side bytes per launch of the dominant GEMM from the newest committed rocprofv3 PMC summary (counters cannot be read
    from inside the process: separate --pmc passes, tools/pmc_gemm.sh); None when no summary is there."""
    try:
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
        with open(files[-1]) as f:
            d = json.load(f)
        return {"bytes_per_launch": d["traffic_bytes_per_launch"], "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
                "shape": d["shape"], "kernel": d.get("kernel"), "l2_hit_rate": d.get("l2_hit_rate"),
                "mfma_busy_frac_of_gui_active": d.get("mfma_busy_frac_of_gui_active"), "source": d["source"],
                "file": os.path.basename(files[-1]),
                "other_kernels": [{k: o.get(k) for k in ("what", "kernel", "shape", "traffic_bytes_per_launch", "algorithmic_bytes_per_launch",
                                                         "l2_hit_rate", "mfma_busy_frac_of_gui_active")} for o in d.get("other_kernels", [])]}
    except Exception:
        return None


def in_step_ring_ms(headline):
    """ms per step of the ring GEMM kernels INSIDE the headline step, from the newest step-alone rocprofv3 table under profiles/
    (tools/round_evidence.sh; `roofline.gemm_ms_per_step` is each shape timed alone): the two should agree to a few per cent."""
    import glob
    name = {"lora": "lora_step", "train": "train_step", "forward": "forward_step"}[headline]
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"r*_kernel_stats_{name}.txt")))
    if not files:
        return None
    tot, calls = 0.0, 0.0
    for line in open(files[-1]):
        if "gemm_nt_bf16_ring_kernel" in line or (headline == "train" and "gemm_tn_bf16_pp_kernel" in line):
            parts = line.rsplit(None, 6)
            try:
                calls += float(parts[-6]); tot += float(parts[-5])
            except (ValueError, IndexError):
                pass
    return {"ms_per_step": round(tot, 2), "calls_per_step": int(calls), "file": os.path.basename(files[-1]),
            "kernels": "gemm_nt_bf16_ring_kernel<...>" + (" + gemm_tn_bf16_pp_kernel<...>" if headline == "train" else "")}


_NOTE_KEYS = ("note", "flop_convention", "pipeline", "assumptions", "source")


def _strip_notes(o, path, notes):
    """Move the long prose fields of every leg out of the JSON line (they go to profiles/bench_notes.md): the driver keeps the last
    8 KB of stdout, and round 5's line had grown past it (the forward / decode legs fell out of the driver-held record)."""
    if isinstance(o, dict):
        for k in list(o):
            v = o[k]
            if k in _NOTE_KEYS and isinstance(v, str) and len(v) > 60:
                notes[path + k] = o.pop(k)
            elif k == "config" and isinstance(v, str) and len(v) > 60 and path:
                notes[path + k] = o.pop(k)
            else:
                _strip_notes(v, path + k + ".", notes)
    elif isinstance(o, list):
        for i, v in enumerate(o):
            _strip_notes(v, path + f"{i}.", notes)


def _g(d, *path):
    for k in path:
        if not isinstance(d, dict) or d.get(k) is None:
            return None
        d = d[k]
    return d


def finalize_line(out):
    """(line, notes): the bench line as it is printed.  Scalars of the north-star legs are copied INTO `roofline` (the driver's parsed
    record keeps that object's scalars), the prose leaves the line, and a compact `summary` of every leg's ms / fraction is the LAST
    key so that any tail of the line a few KB long contains it."""
    notes = {}
    _strip_notes(out, "", notes)
    roof = out.get("roofline")
    if isinstance(roof, dict) and "error" not in roof:
        tr = roof.get("traffic") or {}
        ins = roof.get("in_step") or {}
        scal = {
            "forward_ms": _g(out, "forward", "ms_per_step"), "forward_mfma_frac": _g(out, "forward", "mfma_frac"),
            "decode_tok_s": _g(out, "decode", "tok_s"), "decode_hbm_frac": _g(out, "decode", "roofline", "frac"),
            "in_step_ms": ins.get("ms_per_step"), "in_step_file": ins.get("file"),
            "in_step_frac": (round(roof["achieved"] * roof.get("gemm_ms_per_step", 0) / ins["ms_per_step"] / roof["peak"], 4)
                             if ins.get("ms_per_step") and roof.get("gemm_ms_per_step") else None),
            "traffic_ratio": (round(tr["bytes_per_launch"] / tr["algorithmic_bytes_per_launch"], 3)
                              if tr.get("bytes_per_launch") and tr.get("algorithmic_bytes_per_launch") else None),
            "l2_hit": tr.get("l2_hit_rate"), "mfma_busy": tr.get("mfma_busy_frac_of_gui_active"),
        }
        # scalars first, the big tables (traffic / families / shapes) behind them
        big = {k: roof.pop(k) for k in list(roof) if isinstance(roof[k], (dict, list))}
        roof.update(scal)
        roof.update(big)
    rnd = lambda v, n=4: (round(v, n) if isinstance(v, float) else v)  # noqa: E731
    S = {
        "headline_ms": out.get("ms_per_step"), "headline_samples_s": out.get("value"),
        "forward": [_g(out, "forward", "ms_per_step"), _g(out, "forward", "mfma_frac")],
        "decode": [_g(out, "decode", "ms_per_step"), _g(out, "decode", "tok_s"), _g(out, "decode", "roofline", "frac")],
        "generate": [_g(out, "generate", "tok_s_end_to_end"), _g(out, "generate", "tok_s_after_prefill")],
        "decode_fp8": [_g(out, "decode_fp8", "ms_per_step"), _g(out, "decode_fp8", "tok_s"), _g(out, "decode_fp8", "hbm_frac")],
        "forward_w8a8": [_g(out, "decode_fp8", "forward_w8a8", "ms_per_step"), _g(out, "decode_fp8", "forward_w8a8", "tflops")],
        "geometry_R": [_g(out, "geometry_R", "forward_ms"), _g(out, "geometry_R", "forward_mfma_frac"), _g(out, "geometry_R", "decode_tok_s"),
                       _g(out, "geometry_R", "decode_hbm_frac")],
        "config5_bf16": [_g(out, "config5", "bf16", "forward_ms"), _g(out, "config5", "bf16", "decode_tok_s")],
        "config5_fp8": [_g(out, "config5", "fp8", "forward_ms"), _g(out, "config5", "fp8", "forward_frac_of_fp8_peak"), _g(out, "config5", "fp8", "decode_tok_s")],
        "train_lora": [_g(out, "train_lora", "ms_per_step"), _g(out, "train_lora", "mfma_frac")],
        "train_lora_with_loader": [_g(out, "train_lora_with_loader", "ms_per_step"), _g(out, "train_lora_with_loader", "vs_device_resident_synthetic_step")],
        "train": [_g(out, "train", "ms_per_step"), _g(out, "train", "mfma_frac")],
        "recipe_train_lora": [_g(out, "recipe", "train_lora", "ms_per_micro_step"), _g(out, "recipe", "train_lora", "mfma_frac"),
                              _g(out, "recipe", "train_lora", "attention_share")],
        "recipe_eval_ctx1500_tok_s": _g(out, "recipe", "eval", "ctx_1500", "tok_s_after_prefill"),
        "m13b": [_g(out, "m13b", "forward_ms"), _g(out, "m13b", "forward_mfma_frac"), _g(out, "m13b", "decode_tok_s"), _g(out, "m13b", "decode_hbm_frac")],
        "m13b_train_zero1": [_g(out, "m13b", "train_zero1", "ms_per_step"), _g(out, "m13b", "train_zero1", "mfma_frac_3x")],
        "m13b_train_replica": [_g(out, "m13b", "train_replica", "ms_per_step"), _g(out, "m13b", "train_replica", "mfma_frac_3x")],
        "m13b_zero1_recipe_ckpt": [_g(out, "m13b", "train_zero1_recipe_ckpt", "ms_per_optimizer_step"), _g(out, "m13b", "train_zero1_recipe_ckpt", "mfma_frac_3x"),
                                   _g(out, "m13b", "train_zero1_recipe_ckpt", "hbm_gib")],
        "roofline": [_g(out, "roofline", "achieved"), _g(out, "roofline", "frac"), _g(out, "roofline", "in_step_frac")],
        "families": {k: v.get("frac") for k, v in (_g(out, "roofline", "families") or {}).items()},
        "cpu": [_g(out, "cpu_baseline", "value"), _g(out, "cpu_baseline", "decode_tok_s"), _g(out, "cpu_baseline", "parity_full_depth_rel_err"),
                _g(out, "cpu_baseline", "parity_full_depth_decided_frac"), _g(out, "cpu_baseline", "parity_full_depth_agreement")],
        "errors": sorted(k for k, v in out.items() if isinstance(v, dict) and "error" in v),
        "legend": "[ms, frac] per leg; decode [ms, tok/s, hbm_frac]; notes: profiles/bench_notes.md",
    }
    out.pop("summary", None)
    out["summary"] = {k: ([rnd(x) for x in v] if isinstance(v, list) else v) for k, v in S.items()}
    return out, notes


def write_notes(notes, path):
    with open(path, "w") as f:
        f.write("# bench.py: the prose of every leg (what is measured, conventions, caveats)\n\n"
                "Moved out of the JSON line in round 6 so that the driver-held tail of the line always contains the numbers.\n"
                "Keys are the paths the fields had in the line.\n\n")
        for k in sorted(notes):
            f.write(f"- `{k}`: {notes[k]}\n")


def check_rank_devices(rank_devices, allow_shared=False):
    """Every rank of a SCALE run must sit on its own GPU (one process per GPU, main_finetune.py:241-263): two ranks on one device
    ordinal (or one uuid) would time N model replicas sharing a chip and call it N-GPU throughput.  Raises unless ``allow_shared``
    (the one-GPU emulation the tests use)."""
    if not rank_devices:
        return
    seen = {}
    for d in rank_devices:
        key = d.get("uuid") or ("ordinal", d.get("device"))
        for k in (("ordinal", d.get("device")), key):
            if k in seen and seen[k] != d.get("rank") and not allow_shared:
                raise RuntimeError(f"bench.py: ranks {seen[k]} and {d.get('rank')} share GPU {k}: refusing to report an N-GPU number "
                                   f"(LOCAL_RANK / device map: {[(x.get('rank'), x.get('local_rank'), x.get('device')) for x in rank_devices]})")
            seen[k] = d.get("rank")
    locs = [d.get("local_rank") for d in rank_devices]
    if len(set(locs)) != len(locs) and not allow_shared:
        raise RuntimeError(f"bench.py: duplicate LOCAL_RANK among the ranks: {locs}")


def launch_only(a, rank, world):
    """A3V_BENCH_LAUNCH_ONLY=1 (CPU test of the launch contract, no GPU work): every rank joins a gloo group, one all-reduce
    counts them, rank 0 prints the line's launch fields."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    mine = {"rank": rank, "local_rank": local, "device": local, "pid": os.getpid()}      # the ordinal main() would set_device() to
    devs = [mine]
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1)
        dist.all_reduce(t)
        seen = int(t.item())
        devs = [None] * world
        dist.all_gather_object(devs, mine)
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen = 1
    check_rank_devices(devs)
    if rank == 0:
        print(json.dumps({"launch_only": True, "n_gpus": world, "rccl_ranks": seen, "gpus_arg": a.gpus, "rank_devices": devs}), flush=True)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(a.gpus, 1) and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks; reporting n_gpus={world}", file=sys.stderr)
    if os.environ.get("A3V_BENCH_LAUNCH_ONLY") == "1":
        return launch_only(a, rank, world)
    # stdout carries the ONE JSON line and nothing else: whatever the legs print on the way (the dataset code echoes its config as the
    # reference's does, data/dataset.py) goes to stderr
    json_out, sys.stdout = sys.stdout, sys.stderr
    try:
        return _main_legs(a, rank, world, local, json_out)
    finally:                                  # an exception outside a guarded() leg must not leave stdout pointing at stderr
        sys.stdout = json_out


def _main_legs(a, rank, world, local, json_out):
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # A3V_BENCH_ONE_DEVICE=1 + A3V_DIST_BACKEND=gloo: exercise the multi-rank code path on a single-GPU box (tests only)
        if os.environ.get("A3V_BENCH_ONE_DEVICE") == "1":
            local = 0
        torch.cuda.set_device(local)
        dist.init_process_group(os.environ.get("A3V_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if world > 1 else 0)
    # the SCALE run must prove itself: the number of ranks comes OUT OF THE COMMUNICATOR (an all-reduce of ones on the device the rank
    # computes on), and every rank reports the device it sits on
    rccl_ranks, rank_devices = 1, None
    if dist is not None:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local, "device": int(dev.index), "name": props.name, "uuid": str(getattr(props, "uuid", "")),
                "backend": dist.get_backend()}
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, mine)
        check_rank_devices(rank_devices, allow_shared=os.environ.get("A3V_BENCH_ONE_DEVICE") == "1")
    timer = Timer(dist, dev)
    legs = set((a.legs.split(",") if a.legs else (ALL_LEGS if world == 1 else CORE_LEGS)))
    if a.model != "7b":
        legs -= {"m13b", "geometry_r", "config5"} if a.model == "13b" else {"m13b"}
    if a.no_train:
        legs -= {"train", "lora", "loader", "m13b"}
    if a.no_cpu_baseline:
        legs.discard("cpu")

    B, T = a.batch, a.prompt
    m, args = build_model(a.model, dev, 2048)
    W = m.image_words
    S = T + W
    gen = torch.Generator(device=dev).manual_seed(100 + rank)
    img_u8 = torch.randint(0, 256, (B, 3, 336, 336), device=dev, generator=gen).float()
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073], device=dev).view(1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711], device=dev).view(1, 3, 1, 1)
    image = ((img_u8 / 255.0 - mean) / std).contiguous()          # transform.py:59-68 (resize is a no-op at 336)
    tokens = torch.randint(3, args.vocab_size, (B, T), device=dev, generator=gen)
    tokens[:, 0] = 1
    fl = flops_forward(args, B, T, W)
    res = {}

    def guarded(name, fn):
        try:
            return fn()
        except Exception as e:             # a failing side leg must never hide the headline
            return {"error": f"{name}: {e!r}"[:400]}

    fwd = lambda: m.forward_inference(tokens, 0, image)  # noqa: E731
    if "forward" in legs:
        sec = timer(fwd, a.steps, a.warmup)
        res["forward"] = {"samples_s": round(B * world / sec, 3), "ms_per_step": round(sec * 1e3, 3), "tflops": round(fl["total"] * world / sec / 1e12, 1),
                          "mfma_frac": round(fl["total"] / sec / MFMA_PEAK_BF16, 4),
                          "config": f"configs[1]: ViT-L/14@336 (577+2 image words) + Llama-2-{a.model.upper()} bf16 inference forward, bs={B}/GPU, {T}-token prompt, S={S}"}
    if "decode" in legs:
        dsec = decode_leg(m, fwd, B, T, a.decode_steps, timer, dev)
        ctx = S + 2 + a.decode_steps // 2
        db = bytes_decode_step(args, B, ctx)
        res["decode"] = {"tok_s": round(B * world / dsec, 1), "ms_per_step": round(dsec * 1e3, 3), "steps": a.decode_steps,
                         "roofline": {"bound": "hbm", "achieved": round(db / dsec / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                      "frac": round(db / dsec / HBM_PEAK, 4), "bytes_per_step": db,
                                      "note": "bf16 weights once per step + KV of all sequences (SURVEY 8(d)); whole model step incl. host launch gaps"}}
    if "generate" in legs:
        def _gen():
            gsec, pre, n_new = generate_leg(m, B, T, image, a.gen_len, timer, dev)
            dphase = max(gsec - pre, 1e-9)
            return {"tok_s_end_to_end": round(n_new * world / gsec, 1), "seconds": round(gsec, 4), "new_tokens": n_new, "prefill_seconds": round(pre, 4),
                    "tok_s_after_prefill": round(n_new * world / dphase, 1),
                    "note": f"MetaModel.generate(temperature=0): {B} prompts x {T} tokens + image, {a.gen_len} new tokens each; host tokenise / "
                            "detokenise, one C call per model step and ONE a3v_generate_step launch per token (teacher forcing, stop match, "
                            "stop_pos on the device), `live` polled every 4 steps"}
        res["generate"] = guarded("generate", _gen)
    if "fp8" in legs:
        def _fp8():
            out = {}
            try:
                m.quantize_decode_weights("fp8")
                dsec8 = decode_leg(m, fwd, B, T, a.decode_steps, timer, dev)
                f_bytes = bytes_decode_step(args, B, S + 2 + a.decode_steps // 2) - bytes_decoder_matrices(args)
                out = {"tok_s": round(B * world / dsec8, 1), "ms_per_step": round(dsec8 * 1e3, 3), "hbm_frac": round(f_bytes / dsec8 / HBM_PEAK, 4),
                       "bytes_per_step": f_bytes,
                       "note": "weight-only OCP e4m3fn decoder matrices with per-row fp32 scales (embeddings, norms, LM head, KV cache bf16); no "
                               "reference oracle exists for fp8 (SURVEY 8(a) row Q): parity vs the bf16 kernels on the dequantised weights"}
                m.quantize_decode_weights("fp8", prefill=True)
                psec = timer(fwd, a.steps, 2)
                out["forward_w8a8"] = {"samples_s": round(B * world / psec, 2), "ms_per_step": round(psec * 1e3, 2), "tflops": round(fl["total"] * world / psec / 1e12, 1)}
            finally:
                m.quantize_decode_weights(None)
            return out
        res["decode_fp8"] = guarded("fp8", _fp8)
    if "geometry_r" in legs:
        res["geometry_R"] = guarded("geometry_r", lambda: geometry_r_leg(m, args, B, T, max(2, a.steps // 2), 1, timer, dev))
    if "config5" in legs:
        res["config5"] = guarded("config5", lambda: config5_leg(m, args, B, max(2, a.steps // 2), 1, timer, dev))
    if "recipe" in legs and a.model == "7b":
        res["recipe"] = guarded("recipe", lambda: recipe_leg(m, args, B, timer, dev))
    if "lora" in legs:
        def _lora():
            sec, tl, mem, ntr = lora_leg(m, args, B, T, image, tokens, a.steps, a.warmup, timer, dev)
            return {"samples_s": round(B * world / sec, 3), "ms_per_step": round(sec * 1e3, 2), "seconds_per_step": sec, "loss": round(tl, 4),
                    "hbm_gib": round(mem, 1), "allreduce": lora_leg.exposed,
                    "trainable_params": ntr, "tflops": round(2 * fl["total"] * world / sec / 1e12, 1), "mfma_frac": round(2 * fl["total"] / sec / MFMA_PEAK_BF16, 4),
                    "config": f"configs[2]: LoRA r=16 on all 7 decoder linears + norms + projector trainable, base frozen bf16, bs={B}/GPU, dp{world}",
                    "flop_convention": "2 x forward FLOPs (forward + input-gradient GEMMs; no weight-gradient GEMMs for frozen matrices)"}
        res["train_lora"] = guarded("lora", _lora)
    if "loader" in legs:
        def _loader():
            out = loader_leg(m, args, B, T, max(a.steps, 20) if a.model != "tiny" else a.steps, max(4, a.warmup) if a.model != "tiny" else a.warmup, timer, dev)
            syn = (res.get("train_lora") or {}).get("samples_s")
            if syn:
                out["vs_device_resident_synthetic_step"] = round(out["samples_s"] / syn, 4)
            return out
        res["train_lora_with_loader"] = guarded("loader", _loader)
    train = None
    if "train" in legs:
        def _train():
            sec, tl, mem, ntr, rec = train_leg(m, B, T, image, tokens, a.steps, a.warmup, timer)
            return {"samples_s": round(B * world / sec, 3), "ms_per_step": round(sec * 1e3, 2), "seconds_per_step": sec, "loss": round(tl, 4),
                    "hbm_gib": round(mem, 1), "allreduce": train_leg.exposed, "trainable_params": ntr,
                    "tflops": round(3 * fl["total"] * world / sec / 1e12, 1),
                    "mfma_frac": round(3 * fl["total"] / sec / MFMA_PEAK_BF16, 4),
                    "config": f"full fine-tune of decoder+projector, bs={B}/GPU, {T}-token prompts + {W} image words, fp32 masters + bf16 GEMMs, "
                              + ("per-block recompute" if rec else "block activations kept in HBM (no recompute)")
                              + f", global-norm clip 8 + AdamW (a3v_adamw_scaled), dp{world}"
                              + (" with RCCL all-reduce (bf16 wire) of per-layer grad buckets overlapped with backward" if world > 1 else ""),
                    "flop_convention": "3 x forward FLOPs of the step (SURVEY 8(d)); LM head on all text positions and the frozen ViT counted once are ignored"}
        train = guarded("train", _train)
        res["train"] = train
    # ---- roofline of the headline step's dominant kernel family (rank 0)
    roof = None
    headline = "lora" if (res.get("train_lora") or {}).get("seconds_per_step") else ("train" if train and train.get("seconds_per_step") else "forward")
    if rank == 0 and not a.no_roofline:
        def _roof():
            fam, table = time_gemm_shapes(args, B, T, W, dev, train=bool({"train", "lora"} & legs))
            # families of the headline step: LoRA = forward (NT) + input gradients, which the LoRA engine runs on the NT ring kernel over
            # the transposed frozen images (nt_dgrad); the frozen matrices have no weight-gradient GEMMs.  Full fine-tune = NT + NT input gradients over
            # the AdamW-written transposed images (round 5; NN on the forward image before) + TN (weight gradients); inference forward = NT only
            use = {"lora": ("nt", "nt_dgrad"), "train": ("nt", "nt_dgrad_ft", "tn"), "forward": ("nt",)}[headline]
            tot_f = sum(fam[k][0] for k in use)
            tot_t = sum(fam[k][1] for k in use)
            ft = ("nt", "nt_dgrad_ft", "tn")
            all_f = sum(fam[k][0] for k in ft)
            all_t = sum(fam[k][1] for k in ft)
            try:      # what the matrix pipe delivers on this box under its power limit (bare MFMA stream; context, not the price)
                from tools.ubench.probe import probe_mfma_tflops
                pc, pr = probe_mfma_tflops(20000)
                pipe = {"constant_operands_tflops": round(pc, 1), "random_operands_tflops": round(pr, 1),
                        "frac_of_random_operand_rate": round(tot_f / tot_t / 1e12 / pr, 4) if pr > 0 else None,
                        "note": "bare v_mfma_f32_16x16x32_bf16 stream on every CU, HIP events (tools/ubench/liba3v_probe.so): the chip clocks to its power "
                                "budget, so random operands run slower than constants; `frac` above stays priced against the nominal 2.5 PF/s"}
            except Exception as e:
                pipe = {"error": repr(e)}
            return {"kernel": "MFMA GEMM family of the step: gemm_nt_bf16_ring_kernel (256x256x64 ping-pong over a 160-KiB LDS ring; forward linears), "
                              "gemm_tn_bf16_pp_kernel<A_ROWS> (NN input gradients / TN weight gradients), gemm_nt_bf16_kernel<128,128> on tail rows and small shapes",
                    "bound": "mfma", "achieved": round(tot_f / tot_t / 1e12, 1), "peak": MFMA_PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                    "frac": round(tot_f / tot_t / MFMA_PEAK_BF16, 4), "traffic": pmc_traffic(), "families_of_the_headline_step": list(use),
                    "frac_full_fine_tune_mix": round(all_f / all_t / MFMA_PEAK_BF16, 4),
                    "families": {k: {"tflops": round(v[0] / v[1] / 1e12, 1), "frac": round(v[0] / v[1] / MFMA_PEAK_BF16, 4), "ms_per_step": round(v[1] * 1e3, 2)}
                                 for k, v in fam.items() if v[1] > 0},
                    "in_step": in_step_ring_ms(headline), "mfma_pipe_measured": pipe,
                    "gemm_ms_per_step": round(tot_t * 1e3, 2), "gemm_calls_per_step": sum(r["count"] for r in table if r["kind"] in use),
                    "note": "achieved = algorithmic 2MNK of every GEMM call of one headline step / its HIP-event duration on the launch stream "
                            "(FLOP-weighted over the step's shapes = total GEMM FLOP / total GEMM time), each shape timed alone after a clock "
                            "warm-up; the in-step kernel table of the same command is the newest profiles/r*_kernel_stats_*_step.txt (`in_step`); traffic = rocprofv3 PMC bytes "
                            "per launch of the dominant w1|w3 forward shape (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 fetch "
                            "correction), newest summary under profiles/",
                    "shapes": table}
        del m
        gc.collect(); torch.cuda.empty_cache()
        roof = guarded("roofline", _roof)
    else:
        del m
        gc.collect(); torch.cuda.empty_cache()
    if "m13b" in legs:
        res["m13b"] = guarded("m13b", lambda: m13b_leg(B, T, a.steps, a.warmup, timer, dev))
    if rank == 0:
        lora = res.get("train_lora") or {}
        if headline == "lora":
            head_s = lora.pop("seconds_per_step")
            (train or {}).pop("seconds_per_step", None)
            metric = "image-text samples/sec (train) + articulation-decode tok/s"
            wl = (f"configs[2]: ViT-L/14@336 (577+2 image words) + Llama-2-{a.model.upper()} LoRA r=16 fine-tune step bf16 (adapters on all 7 decoder "
                  f"linears, norms + projector trainable, base frozen), bs={B} per GPU (global {B * world} at dp{world}), {T}-token prompt, S={S}")
        elif headline == "train":
            head_s = train.pop("seconds_per_step")
            metric = "image-text samples/sec (train) + articulation-decode tok/s"
            wl = (f"configs[1] backbone (ViT-L/14@336, 577+2 image words, + Llama-2-{a.model.upper()}), FULL FINE-TUNE step, bs={B} per GPU, "
                  f"{T}-token prompt, S={S}")
        else:
            head_s = res.get("forward", {}).get("ms_per_step", float("nan")) * 1e-3
            metric = "image-text samples/sec (inference forward; training legs skipped) + articulation-decode tok/s"
            wl = (f"configs[1]: ViT-L/14@336 (577+2 image words) + Llama-2-{a.model.upper()} bf16 inference forward step, bs={B} per GPU, "
                  f"{T}-token prompt, S={S}")
        exposed = (lora.get("allreduce") if headline == "lora" else train.get("allreduce") if headline == "train" else None) if world > 1 else None
        out = {
            "metric": metric,
            "value": round(B * world / head_s, 3), "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(head_s * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic (uint8-uniform 336x336 images, uniform token ids, N(0,0.02) weights)",
            "config": {"workload": wl,
                       "geometry": "S (single 336x336 crop); geometry R (W=1455) in `geometry_R`",
                       "global_batch": B * world, "seq_len": S,
                       "parallelism": f"dp{world}" + (" (bucketed RCCL gradient all-reduce overlapped with backward)" if world > 1 else "")},
            "rccl_ranks": rccl_ranks,                  # counted by an all-reduce of ones through the process group, not copied from WORLD_SIZE
            "rank_devices": rank_devices,
            "exposed_allreduce_ms": exposed["exposed_allreduce_ms"] if exposed else None,
            "allreduce": exposed,
            "decode_tok_s": res.get("decode", {}).get("tok_s"),
            "generate_tok_s": (res.get("generate") or {}).get("tok_s_end_to_end"),
            "roofline": roof,
        }
        out.update(res)
        try:
            z13 = (res.get("m13b") or {}).get("train_zero1") or {}
            lo = res.get("train_lora") or {}
            out["wire_prediction_dp8"] = wire_prediction((lo["trainable_params"], lo["ms_per_step"]) if lo.get("trainable_params") else None,
                                                         (z13["sharded_params"], z13["ms_per_step"]) if z13.get("sharded_params") else None)
        except Exception as e:
            out["wire_prediction_dp8"] = {"error": repr(e)[:200]}
        if "cpu" in legs:
            try:
                out["cpu_baseline"] = cpu_baseline(args, T, W, a.cpu_seconds, dev)
            except Exception as e:  # the baseline must never hide the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
        out, notes = finalize_line(out)
        try:
            npath = os.environ.get("A3V_BENCH_NOTES", os.path.join(ROOT, "gpurun_out", "bench_notes.md"))
            os.makedirs(os.path.dirname(npath), exist_ok=True)
            write_notes(notes, npath)
        except OSError:
            pass
        print(json.dumps(out), file=json_out, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
