#!/usr/bin/env python3
"""Fine-tuning entry point with the CLI surface of the reference's ``main_finetune.py`` (flags of
model/accessory/main_finetune.py:57-136 as used by scripts/a3vlm_train.sh:46-55), running on pure DP replicas:

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m a3vlm_amd.main_finetune \
        --llama_type llama_ens5 --llama_config params.json --tokenizer_path tokenizer.model \
        --pretrained_path /ckpt --batch_size 2 --accum_iter 8 --epochs 3 --warmup_epochs 0.03 --lr 2e-5 --min_lr 0 \
        --clip_grad 8 --weight_decay 0 --max_words 2048 --precision bf16 --output_dir out [--synthetic 256]

Call order follows main_finetune.py:141-362: dist init (one process per GPU, backend "nccl" = RCCL) -> MetaModel in
bf16 on the GPU -> trainables promoted to fp32 -> checkpoint load -> identical weights on all ranks (broadcast) ->
AdamW(add_weight_decay, betas 0.9/0.95) -> FinetuneDistSampler -> epochs of train_one_epoch -> save_checkpoint.
FSDP / tensor parallelism / ``--checkpointing`` / ``--quant`` are accepted and ignored or rejected as documented
in DESIGN.md (DP replicas, activations kept in HBM).  ``--synthetic N`` substitutes a seeded synthetic dataset of N
items of the dialog dataset's output shape (the real dataset classes are the next host-side row).
"""
from __future__ import annotations

import argparse
import datetime
import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .checkpoint import latest_checkpoint_dir, load_tensor_parallel_model_list, save_checkpoint
from .dp import FinetuneDistSampler, GradReducer
from .engine_finetune import train_one_epoch
from .model.meta import MetaModel
from .util import add_weight_decay, promote_trainable_params_to_fp32


def get_args_parser():
    p = argparse.ArgumentParser("LLaMA2-Accessory style fine-tuning on MI355X (DP replicas)", add_help=False)
    p.add_argument("--batch_size", default=16, type=int)
    p.add_argument("--accum_iter", default=4, type=int)
    p.add_argument("--llama_type", default="llama_ens5", type=str)
    p.add_argument("--llama_config", default=[], nargs="*")
    p.add_argument("--no_visual", action="store_true")
    p.add_argument("--tokenizer_path", type=str, default="../tokenizer.model")
    p.add_argument("--pretrained_path", default=[], type=str, nargs="*")
    p.add_argument("--pretrained_type", type=str, default=None, choices=["consolidated", "meta_ori"])
    p.add_argument("--weight_decay", type=float, default=0.02)
    p.add_argument("--lr", type=float, default=0.001)
    p.add_argument("--min_lr", type=float, default=0.0001)
    p.add_argument("--epochs", default=400, type=int)
    p.add_argument("--warmup_epochs", type=float, default=1.0)
    p.add_argument("--clip_grad", type=int, default=-1)
    p.add_argument("--max_words", default=1024, type=int)
    p.add_argument("--dialog", action="store_true", default=False)
    p.add_argument("--data_config", default=None, type=str)
    p.add_argument("--image_transform", default="padded_resize", type=str)
    p.add_argument("--preprocess", default="gpu", choices=["gpu", "cpu"],
                   help="gpu (padded_resize only): DataLoader workers decode to uint8 HWC, PadToSquare / bicubic resize / normalise run on the "
                        "device per batch (a3v_preprocess_batch), bit-identical to the PIL transform; cpu: the PIL transform in the workers")
    p.add_argument("--cache_ann_on_disk", action="store_true")
    p.add_argument("--output_dir", default="./output_dir")
    p.add_argument("--save_interval", default=1, type=int)
    p.add_argument("--save_iteration_interval", default=10000, type=int)
    p.add_argument("--only_save_trainable", default=False, action="store_true")
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--resume", default="")
    p.add_argument("--num_workers", default=2, type=int)
    p.add_argument("--pin_mem", action="store_true")
    p.add_argument("--no_pin_mem", action="store_false", dest="pin_mem")
    p.add_argument("--dist_on_itp", action="store_true")
    p.add_argument("--model_parallel_size", type=int, default=1)
    p.add_argument("--data_parallel", type=str, choices=["sdp", "fsdp"], default="sdp")
    p.add_argument("--precision", type=str, choices=["fp16", "bf16", "tf32"], default="bf16")
    p.add_argument("--checkpointing", action="store_true", default=False)
    p.add_argument("--quant", action="store_true", default=False)
    p.add_argument("--synthetic", type=int, default=0, help="use a seeded synthetic dataset of this many items")
    p.add_argument("--max_seq_len", type=int, default=None)
    p.add_argument("--zero1", action="store_true", default=False,
                   help="shard the fp32 masters and the AdamW state of the big matrices over the DP ranks (reduce-scatter -> AdamW on 1/N -> "
                        "all-gather of bf16 parameters): the reference's FSDP(SHARD_GRAD_OP) sizing for the 13B full fine-tune (configs[3])")
    return p


class _OneRank:
    """torch.distributed's surface for a single process (``--zero1`` without a launcher: one slice = everything)."""
    class ReduceOp:
        SUM, AVG, MAX = "sum", "avg", "max"

    @staticmethod
    def get_world_size(group=None):
        return 1

    @staticmethod
    def get_rank(group=None):
        return 0

    @staticmethod
    def get_backend(group=None):
        return "none"


class SyntheticDialogDataset(torch.utils.data.Dataset):
    """Items shaped like FinetuneDialogDataset's (data/conversation/dataset.py:210-273): (tokens[T], labels[T],
    mask[T], image[3,H,W]) with T = max_words - image_words, BOS first, labels on the second half, pad id 0."""

    def __init__(self, n, T, vocab, image_size, with_image=True, seed=0):
        self.n, self.T, self.vocab, self.size, self.with_image, self.seed = n, T, vocab, image_size, with_image, seed

    def __len__(self):
        return self.n

    def groups(self):
        return [list(range(self.n))]

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        tok = torch.randint(3, self.vocab, (self.T,), generator=g)
        tok[0] = 1
        lab = tok.clone()
        lab[: self.T // 2] = 0
        mask = torch.ones(self.T)
        if not self.with_image:
            return tok, lab, mask
        return tok, lab, mask, torch.randn(3, self.size, self.size, generator=g)


def main(args):
    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if args.model_parallel_size != 1:
        raise SystemExit("tensor parallelism is not part of this build (DP replicas only, SURVEY 8(e)): use --model_parallel_size 1")
    if args.quant:
        raise SystemExit("--quant (bitsandbytes NF4) is CUDA-only and out of scope (SURVEY 8(a) row Q)")
    if os.environ.get("A3V_ONE_DEVICE") == "1":      # tests: several ranks on a single-GPU box (with A3V_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("A3V_DIST_BACKEND", "nccl"), rank=rank, world_size=world)     # "nccl" on ROCm is RCCL (util/misc.py:141-145)
    seed = args.seed + rank                                              # main_finetune.py:152-154
    torch.manual_seed(seed)
    np.random.seed(seed)
    dev = torch.device("cuda", local)

    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)                              # default_tensor_type(bf16, "cuda"), :212-215
    with torch.device(dev):
        model = MetaModel(args.llama_type, args.llama_config, args.tokenizer_path, with_visual=not args.no_visual,
                          max_seq_len=args.max_seq_len or args.max_words)
    torch.set_default_dtype(old)
    promote_trainable_params_to_fp32(model, keep_matrices_sharded=args.zero1)        # :217 (ZeRO-1: the big matrices' masters live in 1/N slices)
    if args.pretrained_path and rank == 0:
        print("load result:", load_tensor_parallel_model_list(model, args.pretrained_path))
    if distributed:                                                      # :237-239 (every parameter, trainable or frozen)
        for p in model.parameters():
            dist.broadcast(p.data, src=0)
    model.llma.invalidate_packed_weights()
    if args.precision == "tf32":
        model.train_compute_dtype = torch.float32

    # AdamW(0.9, 0.95) over the reference's two weight-decay groups; the HIP kernel streams each parameter's state once
    # (5.7 TB/s against 4.3 for torch's multi-tensor kernel); A3V_TORCH_ADAMW=1 selects torch.optim.AdamW (same state layout)
    reducer = None
    if args.zero1:
        from .optim import FusedAdamW
        from .zero1 import Zero1Optimizer
        if args.precision != "bf16":
            raise SystemExit("--zero1 needs --precision bf16 (the sharded matrices are the bf16 compute parameters)")
        eng = model.train_engine(zero1_world=world)
        groups = [{**g, "params": [q for q in g["params"] if q.dtype == torch.float32]} for g in add_weight_decay(model, args.weight_decay)]
        small = FusedAdamW([g for g in groups if g["params"]], lr=args.lr, betas=(0.9, 0.95), engine=eng)
        zd = dist if distributed else _OneRank()
        optimizer = reducer = Zero1Optimizer(eng, zd, lr=args.lr, betas=(0.9, 0.95), weight_decay=args.weight_decay,
                                             reduce_dtype=torch.bfloat16, small=small)
    elif os.environ.get("A3V_TORCH_ADAMW", "0") == "1":
        optimizer = torch.optim.AdamW(add_weight_decay(model, args.weight_decay), lr=args.lr, betas=(0.9, 0.95), fused=True)
    else:
        from .optim import FusedAdamW
        optimizer = FusedAdamW(add_weight_decay(model, args.weight_decay), lr=args.lr, betas=(0.9, 0.95), engine=model.train_engine())
    # gradient wire dtype = FSDP's MixedPrecision(reduce_dtype) of the reference (:251-255): bf16 under --precision bf16
    if reducer is None:
        reducer = GradReducer(model.train_engine(), dist, reduce_dtype=torch.bfloat16 if args.precision == "bf16" else None) if distributed else None

    image_words = model.get_image_words()
    if args.synthetic:
        dataset = SyntheticDialogDataset(args.synthetic, args.max_words - image_words, model.tokenizer.n_words,
                                         getattr(model.llma, "image_size", 224), with_image=not args.no_visual, seed=args.seed)
    elif args.data_config:                                               # main_finetune.py:289-293
        from .data.conversation.dataset import FinetuneDialogDataset
        from .data.transform import get_transform
        dataset = FinetuneDialogDataset(args.data_config, get_transform(args.image_transform, getattr(model.llma, "image_size", 224),
                                                                        on_device=args.preprocess == "gpu"),
                                        max_words=args.max_words, image_words=image_words, tokenizer=model.tokenizer,
                                        cache_on_disk=args.cache_ann_on_disk, rank=rank)
    else:
        raise SystemExit("give --data_config <yaml> (dialog dataset) or --synthetic N")
    sampler = FinetuneDistSampler(dataset, num_replicas=world, rank=rank, shuffle=True, batch_size=args.batch_size,
                                  acc_grad=args.accum_iter, seed=args.seed)
    deferred = getattr(getattr(dataset, "transform", None), "deferred", None) == "padded_resize"
    if deferred:
        from .data.transform import DevicePreprocessLoader, collate_raw_images
        loader = torch.utils.data.DataLoader(dataset, batch_size=args.batch_size, sampler=sampler, num_workers=args.num_workers,
                                             pin_memory=args.pin_mem, drop_last=True, collate_fn=collate_raw_images)
        # workers decode to uint8 HWC; the device pads / resizes / normalises the batch (same tensor as T_padded_resize, bit for bit)
        loader = DevicePreprocessLoader(loader, getattr(model.llma, "image_size", 224), dev, torch.float32)
    else:
        loader = torch.utils.data.DataLoader(dataset, batch_size=args.batch_size, sampler=sampler, num_workers=args.num_workers,
                                             pin_memory=args.pin_mem, drop_last=True)
    start_epoch, start_iter = 0, 0
    if args.resume:
        d = latest_checkpoint_dir(args.resume) or args.resume
        print("resume:", load_tensor_parallel_model_list(model, [d]))
        other = torch.load(os.path.join(d, "consolidated.00-of-01.other.pth"), weights_only=False)
        # optimizer state: one consolidated file for every mode (ZeRO-1 re-slices it for this run's DP size); the per-rank slice files of
        # round 4 are still read when every rank has its own.  Whether a file is used is decided COLLECTIVELY (a rank that alone skipped the
        # load would leave the others in a collective, or train on with different state).
        opt_path = os.path.join(d, "consolidated.00-of-01.optimizer.pth")
        legacy = os.path.join(d, f"zero1-optimizer.{rank:05d}-of-{world:05d}.pth")
        have = torch.tensor([int(os.path.isfile(opt_path)), int(args.zero1 and os.path.isfile(legacy))], dtype=torch.int32)
        if distributed:
            have = have.to(dev)
            dist.all_reduce(have, op=dist.ReduceOp.MIN)
        have_full, have_legacy = (bool(x) for x in have.tolist())
        opt_sd = None
        if have_full:
            # FusedAdamW / torch.optim and ZeRO-1 write the same file name in two formats: a checkpoint of the other mode is
            # reported and skipped (every rank reads the same file, so the decision is the same everywhere), not handed to the wrong loader
            opt_sd = torch.load(opt_path, weights_only=False, mmap=True)["optimizer"]
            is_zero1 = isinstance(opt_sd, dict) and ("zero1_full" in opt_sd or "zero1" in opt_sd)
            if is_zero1 != bool(args.zero1):
                if rank == 0:
                    print(f"resume: {opt_path} holds {'ZeRO-1' if is_zero1 else 'replicated AdamW'} optimizer state but this run is "
                          f"{'--zero1' if args.zero1 else 'not --zero1'}: the optimizer state is NOT loaded")
                opt_sd, have_full, have_legacy = None, False, False
        if have_full:
            optimizer.load_state_dict(opt_sd)
        elif have_legacy:
            optimizer.load_state_dict(torch.load(legacy, weights_only=False)["optimizer"])
        else:
            if rank == 0:
                print(f"resume: no optimizer state under {d} on every rank -- AdamW moments start from zero")
            if args.zero1:
                optimizer.resync_from_params()       # the masters were snapshotted BEFORE the checkpoint's weights were loaded
        if other.get("iter") is not None:
            start_epoch, start_iter = other["epoch"], other["iter"] + 1
        else:
            start_epoch = other["epoch"] + 1

    t0 = time.time()
    for epoch in range(start_epoch, args.epochs):
        sampler.set_epoch(epoch, start_iter)
        def _save(step, epoch=epoch):
            save_checkpoint(args.output_dir, args, model, optimizer, None, None, epoch=epoch, iteration=step, rank=rank, world_size=world)
        stats = train_one_epoch(model, loader, optimizer, epoch, start_iter, args, reducer=reducer,
                                log=(print if rank == 0 else (lambda *_: None)), on_save=_save)
        if args.output_dir and (epoch % args.save_interval == 0 or epoch + 1 == args.epochs):
            save_checkpoint(args.output_dir, args, model, optimizer, None, None, epoch=epoch, rank=rank, world_size=world)
        if rank == 0 and args.output_dir:
            with open(os.path.join(args.output_dir, "log.txt"), "a") as f:
                f.write(json.dumps({**{f"train_{k}": v for k, v in stats.items()}, "epoch": epoch}) + "\n")
        start_iter = 0
    if rank == 0:
        print("Training time", str(datetime.timedelta(seconds=int(time.time() - t0))))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    a = argparse.ArgumentParser(parents=[get_args_parser()]).parse_args()
    if a.output_dir:
        os.makedirs(a.output_dir, exist_ok=True)
    main(a)
