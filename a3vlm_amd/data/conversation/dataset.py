"""Dialog fine-tuning dataset with the item contract of the reference's ``data/conversation/dataset.py``:
yaml ``META`` list of .json/.jsonl annotation files (optional ``ratio`` sub-sampling with ``random.seed(0)``, optional
image ``root``), items grouped by ``type`` and sorted by total conversation length inside a group (:150-160), and
``__getitem__`` -> (tokens[T], labels[T], mask[T] [, image]) with T = max_words - image_words for image items
(:210-273): labels carry the token ids of assistant spans only (sub-list search from the previous match), pads are 0.
A failing item falls back to its predecessor inside the group (:275-291).  Host-side only; feeds ``FinetuneDistSampler``
through ``groups()``.  The optional on-disk annotation cache is a jsonl file (the reference uses h5py, absent here)."""
from __future__ import annotations

import copy
import json
import os
import random
import traceback
import warnings
from pathlib import Path
from time import sleep
from typing import Callable, List

import torch
import yaml
from PIL import Image
from torch.utils.data import Dataset

from ...model.tokenizer import Tokenizer
from . import lib as conversation_lib

IGNORE_INDEX = -100


class LabelAllZeroError(Exception):
    def __init__(self, message=None):
        self.message = message

    def __str__(self):
        return f"LabelAllZeroError: {self.message}"


def read_img_general(img_path: str):
    """data/data_reader.py:7-18 without the ceph/point-cloud branches (not on the path)."""
    if ".npy" in img_path or "s3://" in img_path:
        raise NotImplementedError(f"unsupported image source: {img_path}")
    return Image.open(img_path).convert("RGB")


class ConversationGenerator:
    def __init__(self, tokenizer, conv_template_func: Callable = conversation_lib.default_conversation):
        self.tokenizer = tokenizer
        self.conv_func = conv_template_func

    def add_speaker_and_signal(self, source: List):
        """[{"from": "human"|"gpt"|"assistant", "value": str}, ...] -> (conversation text, assistant spans) (:37-64)."""
        conv = self.conv_func()
        for sentence in source:
            who = sentence["from"].lower()
            if who in ["human"]:
                role = conv.roles[0]
            elif who in ["gpt", "assistant"]:
                role = conv.roles[1]
            else:
                raise ValueError(f"unknown dialog role: {who}")
            conv.append_message(role, sentence["value"])
        processed = conv.process()
        return processed["conv"], processed["to_predict"]


def find_sublist(a: list, b: list) -> int:
    for i in range(len(a) - len(b) + 1):
        if a[i:i + len(b)] == b:
            return i
    return -1


def _read_meta(meta_path: str) -> list:
    ext = os.path.splitext(meta_path)[-1]
    if ext == ".json":
        with open(meta_path) as f:
            return json.load(f)
    if ext == ".jsonl":
        out = []
        with open(meta_path) as f:
            for i, line in enumerate(f):
                try:
                    out.append(json.loads(line))
                except json.decoder.JSONDecodeError:
                    print(f"Error decoding the following jsonl line ({i}):\n{line.rstrip()}")
                    raise
        return out
    raise NotImplementedError(f'Unknown meta file extension: "{ext}". Currently, .json, .jsonl are supported. '
                              "If you are using a supported format, please set the file extension so that the proper "
                              "parsing routine can be called.")


class FinetuneDialogDataset(Dataset):
    def __init__(self, config_path, transform, max_words=30, image_words=257, tokenizer=None, cache_on_disk=False, rank=0):
        print(f"read dataset config from {config_path}")
        with open(config_path, "r") as f:
            self.config = yaml.load(f, Loader=yaml.FullLoader)
        print("DATASET CONFIG:")
        print(self.config)
        self.cache_on_disk = cache_on_disk
        self.cache_dir = None
        if cache_on_disk:
            ident = config_path
            for ch in ["/", "\\", ".", "?", "!"]:
                ident = ident.replace(ch, "-")
            self.cache_dir = Path(f"./accessory_data_cache/{ident}")
            if rank == 0:
                self.cache_dir.mkdir(parents=True, exist_ok=True)
        need_collect = (not cache_on_disk) or (rank == 0 and not ((self.cache_dir / "data.jsonl").exists()
                                                                  and (self.cache_dir / "ready").exists()))
        if need_collect:
            group_ann = {}
            for meta in self.config["META"]:
                meta_path, meta_type = meta["path"], meta["type"]
                meta_l = _read_meta(meta_path)
                print(f"{meta_path}, type{meta_type}: len {len(meta_l)}")
                if "ratio" in meta:
                    random.seed(0)
                    meta_l = random.sample(meta_l, int(len(meta_l) * meta["ratio"]))
                    print(f"sample (ratio = {meta['ratio']}) {len(meta_l)} items")
                if "root" in meta:
                    for item in meta_l:
                        if "image" in item:
                            item["image"] = str(Path(meta["root"]) / item["image"])
                for item in meta_l:
                    for turn in item["conversations"]:
                        if not isinstance(turn["value"], str):
                            turn["value"] = str(turn["value"])
                group_ann.setdefault(meta_type, [])
                group_ann[meta_type] += meta_l
            for meta_l in group_ann.values():          # similar lengths inside a global batch
                meta_l.sort(key=lambda it: sum(len(t["value"]) for t in it["conversations"]))
            ann = sum(list(group_ann.values()), start=[])
            ranges, pos = {}, 0
            for meta_type, meta_l in group_ann.items():
                ranges[meta_type] = [pos, pos + len(meta_l)]
                pos += len(meta_l)
            if cache_on_disk:
                with open(self.cache_dir / "data.jsonl", "w") as f:
                    for a in ann:
                        f.write(json.dumps(a) + "\n")
                with open(self.cache_dir / "ranges.json", "w") as f:
                    json.dump(ranges, f)
                with open(self.cache_dir / "ready", "w") as f:
                    f.write("ready")
        if cache_on_disk:
            while not (self.cache_dir / "ready").exists():
                assert rank != 0
                sleep(1)
            with open(self.cache_dir / "data.jsonl") as f:
                ann = [json.loads(line) for line in f]
            with open(self.cache_dir / "ranges.json") as f:
                ranges = json.load(f)
        self.ann = ann
        self.group_indices = {k: list(range(v[0], v[1])) for k, v in ranges.items()}
        print(f"total length: {len(self)}")
        self.transform = transform
        print(f"transform:\n{self.transform}")
        self.max_words = max_words
        self.image_words = image_words
        self.tokenizer = Tokenizer(model_path=tokenizer) if isinstance(tokenizer, str) else copy.deepcopy(tokenizer)
        self.conversation_generator = ConversationGenerator(self.tokenizer)

    def __len__(self):
        return len(self.ann)

    def get_item_func(self, index):
        data_item = self.ann[index]
        if "image" in data_item.keys():
            image = self.transform(read_img_general(data_item["image"]))
        else:
            image = None
        source = data_item["conversations"]
        for turn in source:
            turn["value"] = turn["value"].replace("<image>", "").strip()
        conversation, to_predict_values = self.conversation_generator.add_speaker_and_signal(source)
        if len(to_predict_values) == 0:
            warnings.warn(f"see dialog data with nothing to predict, data: {data_item}")
            return self[index - 1]
        tokenized = self.tokenizer.encode(conversation, bos=True, eos=True)
        labels = [IGNORE_INDEX for _ in tokenized]
        check_pos = 0
        for value in to_predict_values:
            tv = self.tokenizer.encode_segment(value)
            # reference quirk kept (:238-241): find() + check_pos is compared with -1, so a miss only returns early
            # when check_pos == 0; later misses write at check_pos-1 and trip the assert -> predecessor fallback.
            value_pos = find_sublist(tokenized[check_pos:], tv) + check_pos
            if value_pos == -1:
                print("a sentence mismatches the corresponding piece in the conversation")
                return self[index - 1]
            labels[value_pos:value_pos + len(tv)] = tv
            assert labels[value_pos:value_pos + len(tv)] == tokenized[value_pos:value_pos + len(tv)]
            check_pos = value_pos + len(tv)
        input2 = torch.tensor(tokenized, dtype=torch.int64)
        labels = torch.tensor(labels, dtype=torch.int64)
        max_words = self.max_words - self.image_words if image is not None else self.max_words
        padding = max_words - input2.shape[0]
        if padding > 0:
            input2 = torch.cat((input2, torch.zeros(padding, dtype=torch.int64) - 1))
            labels = torch.cat((labels, torch.zeros(padding, dtype=torch.int64) - 1))
        elif padding < 0:
            input2 = input2[:max_words]
            labels = labels[:max_words]
        input2_mask = input2.ge(0)
        label_mask = labels.ge(0)
        input2[~input2_mask] = 0
        labels[~label_mask] = 0
        input2_mask = input2_mask.float()
        if torch.count_nonzero(labels) == 0:
            raise LabelAllZeroError()
        if image is None:
            return input2, labels, input2_mask
        return input2, labels, input2_mask, image

    def __getitem__(self, index):
        try:
            return self.get_item_func(index)
        except Exception as e:
            if not isinstance(e, LabelAllZeroError):
                print(f"Item {index} errored, annotation:\n{self.ann[index]}\nError:\n{traceback.format_exc()}")
            for indices in self.group_indices.values():
                if indices[0] <= index <= indices[-1]:
                    new_index = indices[-1] if index == indices[0] else index - 1
                    return self[new_index]

    def groups(self):
        return list(self.group_indices.values())
