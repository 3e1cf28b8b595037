#!/usr/bin/env python3
"""One-off functional checks at full scale: 13B geometry forward/decode, and the two-image plugin at 7B with a
1024-token prompt (BASELINE configs[3]/[4] shapes, bf16)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from a3vlm_amd import ops

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
which = sys.argv[1]
if which == "13b":
    m, args = bench.build_model("13b", dev, 2048)
    B, T = 8, 512
    gen = torch.Generator(device=dev).manual_seed(1)
    img = torch.randn(B, 3, 336, 336, device=dev, generator=gen)
    tok = torch.randint(3, args.vocab_size, (B, T), device=dev, generator=gen); tok[:, 0] = 1
    for _ in range(2): lg = m.forward_inference(tok, 0, img)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): lg = m.forward_inference(tok, 0, img)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    fl = bench.flops_forward(args, B, T, m.image_words)["total"]
    nt = torch.empty(B, dtype=torch.long, device=dev); cur = torch.empty(B, 1, dtype=torch.long, device=dev)
    for i in range(2):
        ops.argmax(lg, nt); cur[:, 0] = nt; lg = m.forward_inference(cur, T + i, None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(16):
        ops.argmax(lg, nt); cur[:, 0] = nt; lg = m.forward_inference(cur, T + 2 + i, None)
    torch.cuda.synchronize(); dd = (time.perf_counter() - t0) / 16
    print(json.dumps({"model": "13b", "fwd_ms": round(dt * 1e3, 1), "samples_s": round(B / dt, 2), "mfma_frac": round(fl / dt / 2.5e15, 4),
                      "decode_ms": round(dd * 1e3, 3), "tok_s": round(B / dd, 1), "finite": bool(torch.isfinite(lg).all()),
                      "hbm_gib": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
elif which == "13b_train":
    # BASELINE configs[3]: 13B full fine-tune replica in 288 GB (fp32 masters + flat grads + AdamW state + bf16 images = 234 GB,
    # block activations recomputed): one DP replica's step at micro-batch sys.argv[2] (default 1).
    from a3vlm_amd.optim import FusedAdamW
    from a3vlm_amd.train import TrainEngine
    from a3vlm_amd.util import promote_trainable_params_to_fp32
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    T = 512
    m, args = bench.build_model("13b", dev, 2048)
    for n, p in m.named_parameters():
        p.requires_grad = not n.startswith("clip.")
    promote_trainable_params_to_fp32(m)
    eng = TrainEngine(m, torch.bfloat16, recompute=True)
    params = [p for p in m.parameters() if p.requires_grad]
    opt = FusedAdamW(params, lr=2e-5, betas=(0.9, 0.95), weight_decay=0.0, engine=eng)
    gen = torch.Generator(device=dev).manual_seed(1)
    img = torch.randn(B, 3, 336, 336, device=dev, generator=gen)
    tok = torch.randint(3, args.vocab_size, (B, T), device=dev, generator=gen); tok[:, 0] = 1
    lab = tok.clone(); lab[:, :T // 2] = 0
    times, losses = [], []
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = eng.forward_loss(tok, lab, img)
        eng.backward(1.0)
        opt.step(); opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0); losses.append(round(float(loss), 4))
        print(json.dumps({"step": it, "ms": round(times[-1] * 1e3, 1), "loss": losses[-1], "alloc_gib": round(torch.cuda.max_memory_allocated() / 2**30, 1),
                          "reserved_gib": round(torch.cuda.max_memory_reserved() / 2**30, 1)}), flush=True)
    ntr = sum(p.numel() for p in params)
    print(json.dumps({"model": "13b full fine-tune replica", "micro_batch": B, "trainable_params": ntr, "ms_per_step": round(min(times[1:]) * 1e3, 1),
                      "samples_s": round(B / min(times[1:]), 3), "losses": losses, "hbm_gib": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
else:
    import dataclasses
    from a3vlm_amd.model.LLM import llama_ens5_2images as p2
    m0, args = bench.build_model("7b", dev, 4096)
    with torch.device("meta"):
        m = p2.Transformer(p2.ModelArgs(**dataclasses.asdict(args)), with_visual=True)
    base = dict(m0.named_parameters())
    for name, p in list(m.named_parameters()):
        mod = m
        parts = name.split(".")
        for q in parts[:-1]: mod = getattr(mod, q)
        setattr(mod, parts[-1], base[name] if name in base else torch.nn.Parameter(torch.rand(p.shape, dtype=torch.bfloat16, device=dev)))
    m._cos_sin_cpu = m0._cos_sin_cpu
    B, T = 8, 1024
    gen = torch.Generator(device=dev).manual_seed(1)
    img = torch.randn(B, 3, 336, 336, device=dev, generator=gen); dep = torch.randn(B, 3, 336, 336, device=dev, generator=gen)
    tok = torch.randint(3, args.vocab_size, (B, T), device=dev, generator=gen); tok[:, 0] = 1
    for _ in range(2): lg = m.forward_inference(tok, 0, img, dep)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): lg = m.forward_inference(tok, 0, img, dep)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    S = T + m.image_words
    nt = torch.empty(B, dtype=torch.long, device=dev); cur = torch.empty(B, 1, dtype=torch.long, device=dev)
    for i in range(2):
        ops.argmax(lg, nt); cur[:, 0] = nt; lg = m.forward_inference(cur, T + i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(16):
        ops.argmax(lg, nt); cur[:, 0] = nt; lg = m.forward_inference(cur, T + 2 + i)
    torch.cuda.synchronize(); dd = (time.perf_counter() - t0) / 16
    print(json.dumps({"model": "7b two-image", "S": S, "image_words": m.image_words, "fwd_ms": round(dt * 1e3, 1), "samples_s": round(B / dt, 2),
                      "decode_ms": round(dd * 1e3, 3), "tok_s": round(B / dd, 1), "finite": bool(torch.isfinite(lg).all())}))
    # BASELINE configs[4]: the same two-image, 1024-token workload on the fp8 weight path (W8A8 prefill, weight-only fp8 decode)
    m.quantize_decode_weights("fp8", prefill=True)
    for _ in range(2): lg = m.forward_inference(tok, 0, img, dep)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): lg = m.forward_inference(tok, 0, img, dep)
    torch.cuda.synchronize(); dt8 = (time.perf_counter() - t0) / 3
    for i in range(2):
        ops.argmax(lg, nt); cur[:, 0] = nt; lg = m.forward_inference(cur, T + i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(16):
        ops.argmax(lg, nt); cur[:, 0] = nt; lg = m.forward_inference(cur, T + 2 + i)
    torch.cuda.synchronize(); dd8 = (time.perf_counter() - t0) / 16
    print(json.dumps({"model": "7b two-image fp8 (configs[4])", "S": S, "fwd_ms": round(dt8 * 1e3, 1), "samples_s": round(B / dt8, 2),
                      "decode_ms": round(dd8 * 1e3, 3), "tok_s": round(B / dd8, 1), "finite": bool(torch.isfinite(lg).all())}))
