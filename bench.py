#!/usr/bin/env python3
"""bench.py -- the A3VLM hot path on MI355X through liba3vlm_hip.so.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], geometry S of SURVEY.md 8(d)): ViT-L/14 @ 336x336 single crop
(577 + 2 image words) + Llama-2-7B decoder, bf16, batch 8 per GPU, 512-token prompts, synthetic
data, N(0, 0.02) random weights.  One "step" = ONE pass of the multimodal forward hot path over one
batch: patch-embed + 24 ViT blocks + projector + [BOS|image|text] assembly + 32 decoder blocks over
8 x 1091 positions (KV cache written) + final norm + LM head on the last position.
`value` = image-text samples/s (whole job).  The same JSON line carries the greedy-decode rate
(`decode_tok_s`, HBM-bound, its own roofline in `decode_roofline`), the MFMA roofline of the dominant
kernel (gemm_nt_bf16_kernel) and the CPU oracle timed on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_BF16 = 2.5e15      # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK = 8.0e12            # spec; 6.3e12 achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--prompt", type=int, default=512)
    ap.add_argument("--decode-steps", type=int, default=32)
    ap.add_argument("--model", default="7b", choices=["7b", "13b", "tiny"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the fine-tuning step measurement")
    ap.add_argument("--train-steps", type=int, default=3)
    ap.add_argument("--cpu-seconds", type=float, default=30.0)
    return ap.parse_args()


GEOM = {
    "7b": dict(dim=4096, n_layers=32, n_heads=32, multiple_of=256),
    "13b": dict(dim=5120, n_layers=40, n_heads=40, multiple_of=256),
    "tiny": dict(dim=256, n_layers=2, n_heads=2, multiple_of=64),
}


def build_model(name, dev, max_seq_len):
    from a3vlm_amd.model.LLM import llama_ens5 as plugin
    vit = dict(vit_width=1024, vit_layers=24, vit_heads=16) if name != "tiny" else dict(vit_width=128, vit_layers=2, vit_heads=2)
    args = plugin.ModelArgs(vocab_size=32000 if name != "tiny" else 512, max_seq_len=max_seq_len,
                            vit_patch=14, vit_crop=336, n_views=1, **GEOM[name], **vit)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(dev):
            m = plugin.Transformer(args, with_visual=True)
    finally:
        torch.set_default_dtype(old)
    g = torch.Generator(device=dev).manual_seed(0)      # identical weights on every rank
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("norm.weight") or (".ln_" in n and n.endswith("weight")) or n.endswith(".1.weight"):
                p.fill_(1.0)
            elif n.endswith("bias"):
                p.zero_()
            else:
                p.normal_(0.0, 0.02, generator=g)
    return m, args


def flops_forward(args, B, T, W):
    """Algorithmic FLOPs of one step (SURVEY.md 8(d) conventions: 2 FLOP/MAC, causal attention at
    1/2, LM head on the positions actually computed -- the last one for the inference step)."""
    S = T + W
    d, L, ffn = args.dim, args.n_layers, None
    hd = d // args.n_heads
    nkv = args.n_kv_heads or args.n_heads
    from a3vlm_amd.model.LLM.llama_ens5 import _ffn_hidden
    ffn = _ffn_hidden(d, args.multiple_of, args.ffn_dim_multiplier)
    p_layer = d * (args.n_heads + 2 * nkv) * hd + d * d + 3 * d * ffn
    f_lin = 2 * p_layer * L * S * B
    f_att = L * 4 * S * S * d * 0.5 * B
    f_head = 2 * d * args.vocab_size * B
    w, Lv = args.vit_width, args.vit_layers
    g = args.vit_crop // args.vit_patch
    Ltok = g * g + 1
    f_vit = B * (2 * 3 * args.vit_patch ** 2 * w * g * g + Lv * (2 * 12 * w * w * Ltok + 4 * Ltok * Ltok * w))
    f_proj = 2 * w * d * Ltok * B
    gemm = f_lin + f_head + f_proj + B * (2 * 3 * args.vit_patch ** 2 * w * g * g + Lv * 2 * 12 * w * w * Ltok)
    return dict(total=f_lin + f_att + f_head + f_vit + f_proj, gemm=gemm, att=f_att + B * Lv * 4 * Ltok * Ltok * w)


def bytes_decode_step(args, B, ctx):
    from a3vlm_amd.model.LLM.llama_ens5 import _ffn_hidden
    d, L = args.dim, args.n_layers
    hd = d // args.n_heads
    nkv = args.n_kv_heads or args.n_heads
    ffn = _ffn_hidden(d, args.multiple_of, args.ffn_dim_multiplier)
    p_dec = L * (d * (args.n_heads + 2 * nkv) * hd + d * d + 3 * d * ffn)
    return 2 * (p_dec + d * args.vocab_size) + B * 2 * L * ctx * nkv * hd * 2


def bytes_decoder_matrices(args):
    """bytes SAVED per decode step by 1-byte decoder matrices (wqkv, wo, w1|w3, w2 of every layer) relative to bf16"""
    from a3vlm_amd.model.LLM.llama_ens5 import _ffn_hidden
    d, L = args.dim, args.n_layers
    hd = d // args.n_heads
    nkv = args.n_kv_heads or args.n_heads
    ffn = _ffn_hidden(d, args.multiple_of, args.ffn_dim_multiplier)
    return L * (d * (args.n_heads + 2 * nkv) * hd + d * d + 3 * d * ffn)


def time_gemm_shapes(m, args, B, T, W, dev):
    """Event-time every distinct gemm_nt_bf16_kernel shape of one step (same stream the step uses);
    returns (flops per step in that kernel, seconds per step in that kernel, per-shape rows)."""
    from a3vlm_amd import ops
    from a3vlm_amd.model.LLM.llama_ens5 import _ffn_hidden
    S = T + W
    rows = B * S
    d, Lyr = args.dim, args.n_layers
    ffn = _ffn_hidden(d, args.multiple_of, args.ffn_dim_multiplier)
    w, Lv = args.vit_width, args.vit_layers
    g = args.vit_crop // args.vit_patch
    vr = B * (g * g + 1)
    shapes = [  # (M, N, K, count per step, epilogue)
        (rows, 3 * d, d, Lyr, 0), (rows, d, d, Lyr, ops.EPI_RESIDUAL), (rows, 2 * ffn, d, Lyr, ops.EPI_SWIGLU),
        (rows, d, ffn, Lyr, ops.EPI_RESIDUAL),
        (vr, 3 * w, w, Lv, 0), (vr, w, w, Lv, ops.EPI_RESIDUAL), (vr, 4 * w, w, Lv, ops.EPI_GELU), (vr, w, 4 * w, Lv, ops.EPI_RESIDUAL),
        (vr, d, w, 1, 0), (B * g * g, w, 640, 1, 0),
    ]
    tot_f, tot_t, table = 0.0, 0.0, []
    for (M, N, K, cnt, epi) in shapes:
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        wt = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
        ncol = N // 2 if epi & ops.EPI_SWIGLU else N
        out = torch.zeros(M, ncol, device=dev, dtype=torch.bfloat16)
        res = out if epi & ops.EPI_RESIDUAL else None
        e = epi & ~ops.EPI_RESIDUAL
        for _ in range(2):
            ops.gemm_nt(a, wt, out, residual=res, epilogue=e)
        reps = 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.gemm_nt(a, wt, out, residual=res, epilogue=e)
        e1.record()
        torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) * 1e-3 / reps
        fl = 2.0 * M * N * K
        tot_f += fl * cnt
        tot_t += dt * cnt
        table.append(dict(M=M, N=N, K=K, count=cnt, us=round(dt * 1e6, 1), tflops=round(fl / dt / 1e12, 1)))
        del a, wt, out
    return tot_f, tot_t, table


def train_leg(m, args, B, T, image, tokens, steps, dist, dev):
    """Full fine-tune step of configs[2]/[3] semantics on this rank's micro-batch: fp32 masters for the trainables
    (decoder + projector), frozen bf16 ViT, bf16 GEMMs, per-block recompute, AdamW(0.9, 0.95); with N > 1 the
    gradient buckets are all-reduced over RCCL on a side stream while earlier layers still back-propagate."""
    from a3vlm_amd.train import TrainEngine
    from a3vlm_amd.util import promote_trainable_params_to_fp32
    from a3vlm_amd.dp import GradReducer, clip_grad_norm
    for n, p in m.named_parameters():
        p.requires_grad = not n.startswith("clip.")
    m._ws.clear(); m._packed.clear(); m._packed_version = None; m._destroy_kv_cache()
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    promote_trainable_params_to_fp32(m)
    eng = TrainEngine(m, torch.bfloat16)
    params = [p for p in m.parameters() if p.requires_grad]
    params_list = params
    from a3vlm_amd.optim import FusedAdamW
    opt = FusedAdamW(params, lr=2e-5, betas=(0.9, 0.95), weight_decay=0.0, engine=eng)
    # bf16 on the wire = the reference's FSDP MixedPrecision(reduce_dtype=bf16) (main_finetune.py:251-255); fp32 accumulation buffers
    red = GradReducer(eng, dist, reduce_dtype=torch.bfloat16) if dist is not None else None
    labels = tokens.clone()
    labels[:, :T // 2] = 0

    def one():
        loss = eng.forward_loss(tokens, labels, image)
        eng.backward(1.0)
        if red is not None:
            red.finish()
        # global-norm clip of the reference recipe (--clip_grad 8, a3vlm_train.sh:47-55; util/misc.py:302-315): one reduction
        # over the flat gradient buffer, coefficient applied inside the optimizer kernel
        _, coef = clip_grad_norm(params_list, 8.0, flat=eng.flat_grads(), defer=True)
        opt.step(grad_scale=coef)
        opt.zero_grad(set_to_none=True)
        return loss

    one()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el / steps, float(loss), torch.cuda.max_memory_allocated() / 2 ** 30


def lora_leg(m, args, B, T, image, tokens, steps, dist, dev, rank=16):
    """configs[2]: LoRA fine-tune step (rank-16 adapters on the seven decoder linears of every block + norms + projector
    trainable, base matrices frozen in bf16) on this rank's micro-batch.  The adapter plugin is built around the SAME base
    parameter tensors as ``m`` (no second copy of the 7B weights)."""
    import dataclasses
    from a3vlm_amd.model.LLM import llama_ens5_peft as peft
    from a3vlm_amd.train import TrainEngine
    from a3vlm_amd.util import promote_trainable_params_to_fp32
    from a3vlm_amd.dp import GradReducer, clip_grad_norm
    torch.cuda.reset_peak_memory_stats()
    pargs = peft.ModelArgs(**dataclasses.asdict(args), lora_rank=rank)
    with torch.device("meta"):
        pm = peft.Transformer(pargs, with_visual=True)
    base = dict(m.named_parameters())
    g = torch.Generator(device=dev).manual_seed(7)
    for name, p in list(pm.named_parameters()):
        mod = pm
        parts = name.split(".")
        for q in parts[:-1]:
            mod = getattr(mod, q)
        if name in base:
            setattr(mod, parts[-1], base[name])                      # shared storage
        else:
            t = torch.empty(p.shape, dtype=torch.bfloat16, device=dev)
            t.normal_(0.0, 0.02, generator=g)                        # B != 0 so that every adapter GEMM does real work
            setattr(mod, parts[-1], torch.nn.Parameter(t))
    pm._cos_sin_cpu = m._cos_sin_cpu
    train = pm.get_trainable_params()
    for n, p in pm.named_parameters():
        p.requires_grad = n in train
    promote_trainable_params_to_fp32(pm)
    n_train = sum(p.numel() for p in pm.parameters() if p.requires_grad)
    eng = TrainEngine(pm, torch.bfloat16)
    from a3vlm_amd.optim import FusedAdamW
    params_list = [p for p in pm.parameters() if p.requires_grad]
    opt = FusedAdamW(params_list, lr=2e-5, betas=(0.9, 0.95), weight_decay=0.0, engine=eng)
    # bf16 on the wire = the reference's FSDP MixedPrecision(reduce_dtype=bf16) (main_finetune.py:251-255); fp32 accumulation buffers
    red = GradReducer(eng, dist, reduce_dtype=torch.bfloat16) if dist is not None else None
    labels = tokens.clone()
    labels[:, :T // 2] = 0

    def one():
        loss = eng.forward_loss(tokens, labels, image)
        eng.backward(1.0)
        if red is not None:
            red.finish()
        # global-norm clip of the reference recipe (--clip_grad 8, a3vlm_train.sh:47-55; util/misc.py:302-315): one reduction
        # over the flat gradient buffer, coefficient applied inside the optimizer kernel
        _, coef = clip_grad_norm(params_list, 8.0, flat=eng.flat_grads(), defer=True)
        opt.step(grad_scale=coef)
        opt.zero_grad(set_to_none=True)
        return loss

    one()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    mem = torch.cuda.max_memory_allocated() / 2 ** 30
    # restore the shared parameters' state for the legs that follow
    for n, p in m.named_parameters():
        p.grad = None
    del eng, opt, red, pm, one
    import gc
    gc.collect()                        # the engine holds lazy weight-image objects that point back at it: a cycle, not a leak
    torch.cuda.empty_cache()
    return el / steps, float(loss), mem, n_train


def cpu_baseline(args, T, W, seconds):
    """The CPU oracle (oracle/ref_cpu.py, kind "port") on the host cores, bf16, ONE sample of the same
    workload: full ViT-L/14@336 + projector + all decoder layers over 1091 positions + LM head.  To keep
    host RAM bounded every decoder layer aliases ONE set of N(0,0.02) weights (identical FLOPs/bytes per
    layer); if the time budget runs out the decoder is cut after k layers and the rate is reported for the
    layers actually run, scaled by the algorithmic FLOP ratio (stated in `sample`)."""
    from oracle import ref_cpu
    ncpu = os.cpu_count() or 1
    # pick the fastest (threads, dtype) for torch's CPU kernels on this host with one FFN-sized matmul
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
        _ = torch.nn.functional.linear(hn[:, -1, :], sd1["output.weight"]).float()
    el = time.perf_counter() - t0
    t_dec = el - t_vit
    full = t_vit + t_dec * (args.n_layers / done)
    return dict(value=round(1.0 / full, 5), unit="samples/s", cores=cores, kind="port",
                sample=(f"oracle/ref_cpu.py {str(dt).split('.')[-1]} on {cores} of {ncpu} host threads (fastest of a threads x dtype probe), 1 sample (336x336 image, {T}-token prompt, S={S}): ViT-L/14 24 blocks "
                        f"{t_vit:.1f}s + {done}/{args.n_layers} decoder layers {t_dec:.1f}s (weights aliased across layers)"
                        + ("" if done == args.n_layers else f"; decoder time scaled x{args.n_layers / done:.2f} by layer count")),
                seconds=round(el, 1))


def pmc_traffic():
    """HBM-side bytes per launch of the dominant GEMM from the committed rocprofv3 PMC passes (counters cannot be read
    from inside the process); None when the summary is absent."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01i_pmc_traffic.json")) as f:
            d = json.load(f)
        return {"bytes_per_launch": d["traffic_bytes_per_launch"], "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
                "shape": d["shape"], "source": d["source"]}
    except Exception:
        return None


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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

    B, T = a.batch, a.prompt
    max_seq = 2048
    m, args = build_model(a.model, dev, max_seq)
    W = m.image_words
    S = T + W
    gen = torch.Generator(device=dev).manual_seed(100 + rank)
    img_u8 = torch.randint(0, 256, (B, 3, 336, 336), device=dev, generator=gen).float()
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073], device=dev).view(1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711], device=dev).view(1, 3, 1, 1)
    image = ((img_u8 / 255.0 - mean) / std).contiguous()          # transform.py:59-68 (resize is a no-op at 336)
    tokens = torch.randint(3, args.vocab_size, (B, T), device=dev, generator=gen)
    tokens[:, 0] = 1

    def step():
        return m.forward_inference(tokens, 0, image)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync_all()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    ms_step = el / a.steps * 1e3
    value = B * world * a.steps / el

    # ---- decode: greedy steps after the prefill above (KV cache holds S positions) ----
    nt = torch.empty(B, dtype=torch.long, device=dev)
    from a3vlm_amd import ops
    logits = step()
    cur = torch.empty(B, 1, dtype=torch.long, device=dev)
    for _ in range(2):
        ops.argmax(logits, nt)
        cur[:, 0] = nt
        logits = m.forward_inference(cur, T, None)
    sync_all()
    t0 = time.perf_counter()
    for i in range(a.decode_steps):
        ops.argmax(logits, nt)
        cur[:, 0] = nt
        logits = m.forward_inference(cur, T + 2 + i, None)
    sync_all()
    d_el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([d_el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        d_el = float(t.item())
    dec_ms = d_el / a.decode_steps * 1e3
    dec_tok_s = B * world * a.decode_steps / d_el
    ctx = S + 2 + a.decode_steps // 2
    dec_bytes = bytes_decode_step(args, B, ctx)

    # ---- decode with weight-only fp8 images of the decoder matrices (BASELINE config 5 semantics; opt-in, separate from the
    # bf16 headline: the reference has no fp8 path, so this leg is reported beside it, never instead of it)
    fp8 = None
    try:
        m.quantize_decode_weights("fp8")
        logits = step()
        for _ in range(2):
            ops.argmax(logits, nt)
            cur[:, 0] = nt
            logits = m.forward_inference(cur, T, None)
        sync_all()
        t0 = time.perf_counter()
        for i in range(a.decode_steps):
            ops.argmax(logits, nt)
            cur[:, 0] = nt
            logits = m.forward_inference(cur, T + 2 + i, None)
        sync_all()
        f_el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([f_el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            f_el = float(t.item())
        f_ms = f_el / a.decode_steps * 1e3
        f_bytes = bytes_decode_step(args, B, ctx) - bytes_decoder_matrices(args)          # the four matrices per layer at 1 B/weight
        fp8 = {"tok_s": round(B * world * a.decode_steps / f_el, 1), "ms_per_step": round(f_ms, 3),
               "hbm_frac": round(f_bytes / (f_ms * 1e-3) / HBM_PEAK, 4), "bytes_per_step": f_bytes,
               "note": "weight-only OCP e4m3fn decoder matrices with per-row fp32 scales (embeddings, norms, LM head, KV cache bf16); "
                       "no reference oracle exists for fp8 (SURVEY 8(a) row Q): parity is stated against the bf16 kernels on the "
                       "dequantised weights (tests/test_gpu_fp8.py)"}
        # W8A8 prefill: the same forward step with every decoder GEMM on fp8 operands (MX-scaled MFMA); ViT and LM head bf16
        m.quantize_decode_weights("fp8", prefill=True)
        for _ in range(2):
            step()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        sync_all()
        p_el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([p_el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            p_el = float(t.item())
        fp8["forward_w8a8"] = {"samples_s": round(B * world * a.steps / p_el, 2), "ms_per_step": round(p_el / a.steps * 1e3, 2),
                               "tflops": round(flops_forward(args, B, T, W)["total"] * world * a.steps / p_el / 1e12, 1),
                               "note": "opt-in quantize_decode_weights('fp8', prefill=True): per-token dynamic e4m3 activations x "
                                       "per-row e4m3 weights, fp32 accumulate; never the headline value"}
    except Exception as e:
        fp8 = dict(fp8 or {}, error=repr(e)[:300])
    finally:
        m.quantize_decode_weights(None)

    lora = None
    if not a.no_train and a.model != "13b":
        try:
            sec, tl, mem, ntr = lora_leg(m, args, B, T, image, tokens, a.train_steps, dist, dev)
            fl_t = flops_forward(args, B, T, W)
            lora = {"samples_s": round(B * world / sec, 2), "ms_per_step": round(sec * 1e3, 1), "loss": round(tl, 4), "hbm_gib": round(mem, 1),
                    "trainable_params": ntr, "tflops": round(2 * fl_t["total"] * world / sec / 1e12, 1),
                    "mfma_frac": round(2 * fl_t["total"] / sec / MFMA_PEAK_BF16, 4),
                    "config": f"configs[2]: LoRA r=16 on all 7 decoder linears + norms + projector trainable, base frozen bf16, bs={B}/GPU, dp{world}",
                    "flop_convention": "2 x forward FLOPs (forward + input-gradient GEMMs; no weight-gradient GEMMs for frozen matrices)"}
        except Exception as e:
            lora = {"samples_s": None, "error": repr(e)[:300]}
    train = None
    if not a.no_train and a.model != "13b":
        try:
            sec, tl, mem = train_leg(m, args, B, T, image, tokens, a.train_steps, dist, dev)
            fl_t = flops_forward(args, B, T, W)
            # 3x (fwd + dgrad + wgrad) of the trainable part + 1x recompute + frozen ViT forward once
            vit = fl_t["total"] - (fl_t["gemm"] + fl_t["att"]) + 0.0
            train = {"samples_s": round(B * world / sec, 2), "ms_per_step": round(sec * 1e3, 1), "loss": round(tl, 4),
                     "hbm_gib": round(mem, 1), "tflops": round((3 * fl_t["total"]) * world / sec / 1e12, 1),
                     "mfma_frac": round(3 * fl_t["total"] / sec / MFMA_PEAK_BF16, 4),
                     "config": f"full fine-tune of decoder+projector (6.7 G trainable), bs={B}/GPU, {T}-token prompts + 579 image words, fp32 masters + "
                               f"bf16 GEMMs, block activations kept in HBM (no recompute), global-norm clip 8 + AdamW (a3v_adamw_scaled), dp{world}"
                               + (" with RCCL all-reduce of per-layer fp32 grad buckets overlapped with backward" if world > 1 else ""),
                     "flop_convention": "3 x forward FLOPs of the step (SURVEY 8(d)); LM head on all text positions and the frozen ViT counted once are ignored"}
        except Exception as e:
            train = {"samples_s": None, "error": repr(e)[:300]}
    out = None
    if rank == 0:
        fl = flops_forward(args, B, T, W)
        gf, gt, table = time_gemm_shapes(m, args, B, T, W, dev)
        out = {
            "metric": "image-text samples/sec (ViT-L/14 + Llama-2-7B multimodal forward, bf16, bs=8/GPU) + articulation-decode tok/s",
            "value": round(value, 3), "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic (uint8-uniform 336x336 images, uniform token ids, N(0,0.02) weights)",
            "config": {"workload": f"configs[1]: ViT-L/14@336 (577+2 image words) + Llama-2-{a.model.upper()} bf16 inference forward, "
                                   f"bs={B} per GPU, {T}-token prompt, S={S}", "geometry": "S (single 336x336 crop)",
                       "global_batch": B * world, "seq_len": S, "parallelism": f"dp{world} replicas (no data-path collective)"},
            "forward_tflops": round(fl["total"] * world / (ms_step * 1e-3) / 1e12, 1),
            "forward_mfma_frac": round(fl["total"] / (ms_step * 1e-3) / MFMA_PEAK_BF16, 4),
            "roofline": {"kernel": "gemm_nt_bf16_pp_kernel (256x256x64 ping-pong; gemm_nt_bf16_kernel<128,128> on tail rows / small shapes)",
                         "bound": "mfma", "achieved": round(gf / gt / 1e12, 1), "peak": MFMA_PEAK_BF16 / 1e12,
                         "unit": "TFLOP/s", "frac": round(gf / gt / MFMA_PEAK_BF16, 4), "traffic": pmc_traffic(),
                         "note": "algorithmic 2MNK per GEMM call / HIP-event duration on the launch stream, FLOP-weighted over the step's GEMM "
                                 "shapes (= total GEMM FLOP / total GEMM time of one step); traffic = rocprofv3 PMC bytes per launch of the "
                                 "dominant w1|w3 shape (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 fetch correction), from profiles/",
                         "gemm_ms_per_step": round(gt * 1e3, 2), "gemm_calls_per_step": sum(r["count"] for r in table),
                         "avg_call_us": round(gt * 1e6 / max(1, sum(r["count"] for r in table)), 1), "shapes": table},
            "decode_tok_s": round(dec_tok_s, 1), "decode_ms_per_step": round(dec_ms, 3), "decode_steps": a.decode_steps,
            "decode_roofline": {"bound": "hbm", "achieved": round(dec_bytes / (dec_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                "frac": round(dec_bytes / (dec_ms * 1e-3) / HBM_PEAK, 4), "bytes_per_step": dec_bytes,
                                "note": "bf16 weights once per step + KV of all sequences (SURVEY 8(d)); whole step incl. host launch gaps"},
        }
        out["decode_fp8"] = fp8
        out["train"] = train
        out["train_lora"] = lora
        if not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, T, W, a.cpu_seconds)
            except Exception as e:  # the baseline must never hide the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
