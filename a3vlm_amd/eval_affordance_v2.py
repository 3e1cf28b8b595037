#!/usr/bin/env python3
"""Batch inference entry point with the CLI surface of the reference's ``eval_affordance_v2.py`` (:236-259 as used by
scripts/a3vlm_infer.sh:37-44), on DP replicas: every rank takes a contiguous shard of the request list (the reference's
dormant InferenceSampler, :182-205) and runs ``MetaModel.generate`` on it; results are gathered on rank 0 and written
to ``vqa_logs/{addition_flag}/{dataset}.json`` in the reference's record format (:363-370).

The reference broadcasts every batch from rank 0 to its tensor-parallel peers (:333-334, 378-380); there are no
TP peers here, so there is no data-path collective.  ``--temperature/--top_p`` default to the reference's 0.1 / 0.75
(:46-49); ``--temperature 0`` gives the greedy (bit-exact) path.
"""
from __future__ import annotations

import argparse
import itertools
import json
import os
import random
import re

import torch
import torch.distributed as dist
from PIL import Image

from .checkpoint import load_tensor_parallel_model_list
from .data.conversation import default_conversation
from .data.transform import GpuPaddedResize, T_decode_rgb, T_padded_resize
from .model.meta import MetaModel


def normalize_number(x: float) -> float:
    """eval_affordance_v2.py:207-215"""
    if x > 100:
        return x / 1000
    if x > 10:
        return x / 100
    if x >= 1:
        return x / 10
    return x


def format_bounding_box(answer: str):
    """eval_affordance_v2.py:217-232: digits and commas only, a dot before the last three digits of every 4+ digit run,
    then per-number magnitude normalisation."""
    cleaned = re.sub(r"[^\d,]", "", answer.replace(" ", ""))
    formatted = re.sub(r"\d{4,}", lambda m: m.group(0)[:-3] + "." + m.group(0)[-3:], cleaned)
    return [normalize_number(float(n)) for n in formatted.split(",") if n]


def postprocess_answer(answer: str) -> str:
    """eval_affordance_v2.py:343-357"""
    answer = answer.split("###")[0]
    answer = answer.replace(".", "").strip()
    if "answer is" in answer:
        try:
            ext = re.findall(r"answer is[ ]*[a-zA-Z0-9.]+", answer)[0]
            answer = re.sub("answer is", "", ext).strip()
        except Exception:
            answer = answer.strip()
    return answer


class VQADataset(torch.utils.data.Dataset):
    """eval_affordance_v2.py:109-180 (without the resample-on-corrupt-image fallback's undefined-field access)."""

    def __init__(self, test: str, img_size: int = 224, sampled_num: int = 5000, result=None, image_root: str = "", decode_only: bool = False):
        with open(test, "r") as f:
            self.test = json.load(f)
        if len(self.test) > sampled_num:
            random.shuffle(self.test)
            self.test = self.test[:sampled_num]
        if result is not None:
            done = {r["image"] for r in result}
            self.test = [t for t in self.test if t["image"] not in done]
        # decode_only: the worker hands over the decoded pixels (uint8 HWC); pad / resize / normalise run on the device per batch
        self.transform_val = T_decode_rgb(img_size) if decode_only else T_padded_resize(img_size)
        self.image_root = image_root

    def __len__(self):
        return len(self.test)

    def __getitem__(self, idx):
        data = self.test[idx]
        path = data["image"]
        local = path if os.path.isfile(path) or not self.image_root else os.path.join(self.image_root, os.path.basename(path))
        image = self.transform_val(Image.open(local).convert("RGB"))
        conv = default_conversation()
        conv.load_qas([[data["conversations"][0]["value"], None]])
        return {"question": conv.get_prompt(), "question_id": idx, "annotation": data["conversations"][1]["value"],
                "image": image, "image_path": path}


def collate_fn(batches):
    images = [b["image"] for b in batches]
    if images[0].dtype != torch.uint8:
        images = torch.stack(images)          # CPU transform: [B, 3, size, size]; decode-only: a list of uint8 HWC images
    return (images, [b["question_id"] for b in batches], [b["question"] for b in batches],
            [b["annotation"] for b in batches], [b["image_path"] for b in batches])


def shard_range(total: int, world: int, rank: int) -> range:
    """InferenceSampler._get_local_indices, eval_affordance_v2.py:191-199."""
    size, left = total // world, total % world
    sizes = [size + int(r < left) for r in range(world)]
    return range(sum(sizes[:rank]), min(sum(sizes[:rank + 1]), total))


def get_args_parser():
    p = argparse.ArgumentParser("A3VLM batch inference on MI355X (DP replicas)", add_help=False)
    p.add_argument("--llama_type", default="llama_ens5", type=str)
    p.add_argument("--llama_config", type=str, default=None, nargs="*")
    p.add_argument("--tokenizer_path", type=str, default="../tokenizer.model")
    p.add_argument("--pretrained_path", default=[], type=str, nargs="*")
    p.add_argument("--device", default="cuda")
    p.add_argument("--model_parallel_size", default=1, type=int)
    p.add_argument("--batch_size", default=4, type=int)
    p.add_argument("--num_workers", default=4, type=int)
    p.add_argument("--seed", default=1, type=int)
    p.add_argument("--dataset", default="path_to_eval_json", type=str)
    p.add_argument("--input_size", type=int, default=224)
    p.add_argument("--addition_flag", default=None, type=str)
    p.add_argument("--remove_space", action="store_true", default=False)
    p.add_argument("--sampled_num", type=int, default=200)
    p.add_argument("--max_gen_len", type=int, default=2048)
    p.add_argument("--max_seq_len", type=int, default=4096)
    p.add_argument("--temperature", type=float, default=0.1)
    p.add_argument("--top_p", type=float, default=0.75)
    p.add_argument("--image_root", type=str, default="", help="directory to look images up by basename (demo.json paths are absolute)")
    p.add_argument("--output_root", type=str, default="vqa_logs")
    p.add_argument("--precision", type=str, choices=["bf16", "tf32"], default="bf16", help="tf32 = fp32 parity path")
    p.add_argument("--preprocess", type=str, choices=["gpu", "cpu"], default="gpu",
                   help="gpu: workers decode only, PadToSquare / bicubic resize / normalise run on the device per batch (a3v_preprocess_batch, "
                        "bit-identical to the PIL transform); cpu: the PIL transform in the workers (data/transform.py:59-68)")
    return p


def main(args):
    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if args.model_parallel_size != 1:
        raise SystemExit("tensor parallelism is not part of this build (DP replicas only): use --model_parallel_size 1")
    if os.environ.get("A3V_ONE_DEVICE") == "1":      # tests: several ranks on a single-GPU box (with A3V_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("A3V_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    dev = torch.device("cuda", local)
    cfg = args.llama_config or []
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16 if args.precision == "bf16" else torch.float32)   # model.bfloat16().cuda(), :282
    with torch.device(dev):
        model = MetaModel(args.llama_type, cfg, args.tokenizer_path, with_visual=True, max_seq_len=args.max_seq_len)
    torch.set_default_dtype(old)
    if args.pretrained_path:
        print(f"load pretrained from {args.pretrained_path}:", load_tensor_parallel_model_list(model, args.pretrained_path))
    model.eval()

    name = os.path.basename(args.dataset).split(".")[0]
    save_dir = os.path.join(args.output_root, str(args.addition_flag))
    os.makedirs(save_dir, exist_ok=True)
    results_file = os.path.join(save_dir, f"{name}.json")
    result = json.load(open(results_file)) if os.path.exists(results_file) else None
    random.seed(args.seed)
    on_gpu = getattr(args, "preprocess", "gpu") == "gpu"
    dataset = VQADataset(args.dataset, img_size=args.input_size, sampled_num=args.sampled_num, result=result, image_root=args.image_root,
                         decode_only=on_gpu)
    pre = GpuPaddedResize(args.input_size, dev, torch.float32) if on_gpu else None
    idx = list(shard_range(len(dataset), world, rank))
    loader = torch.utils.data.DataLoader(torch.utils.data.Subset(dataset, idx), batch_size=args.batch_size, shuffle=False,
                                         num_workers=args.num_workers, pin_memory=True, drop_last=False, collate_fn=collate_fn)
    outputs = []
    # the sampled recipe (default: temperature 0.1 / top-p 0.75) draws its uniform numbers from torch's device generator inside
    # generate(): seeded here, per rank, so that a run with the same --seed reproduces its answers (the reference seeds only the
    # dataset shuffle, :304; its sampling stream is whatever the process left in the generator)
    torch.manual_seed(args.seed + rank)
    with torch.no_grad():
        for image, qids, prompts, annotations, paths in loader:
            image = pre.batch(image) if isinstance(image, list) else image.to(dev)
            answers = model.generate(prompts, image, max_gen_len=args.max_gen_len, temperature=args.temperature, top_p=args.top_p)
            for answer, annotation, question, path in zip(answers, annotations, prompts, paths):
                answer = postprocess_answer(answer)
                box = format_bounding_box(answer)
                fail = len(box) != 4 or box[0] > box[2] or box[1] > box[3]
                outputs.append({"answer": answer, "format_answer": box, "annotation": annotation, "question": question,
                                "image": path, "fail": fail})
    if distributed:
        gathered = [None] * world
        dist.all_gather_object(gathered, outputs)
        outputs = list(itertools.chain.from_iterable(gathered))
    if rank == 0:
        if result:
            outputs.extend(result)
        with open(results_file, "w") as f:
            json.dump(outputs, f, ensure_ascii=False)
        print(f"{len(outputs)} records -> {results_file}")
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return outputs


if __name__ == "__main__":
    main(argparse.ArgumentParser(parents=[get_args_parser()]).parse_args())
