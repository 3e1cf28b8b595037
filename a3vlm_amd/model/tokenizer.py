"""Tokenizer wrapper with the interface of the reference's ``accessory.model.tokenizer``
(reference: model/accessory/model/tokenizer.py:15-156).  Host-side plumbing only: delegates
to sentencepiece (``*.model``) or a HuggingFace tokenizer directory."""
from __future__ import annotations

import os
from pathlib import Path
from typing import List, Optional

__all__ = ["Tokenizer", "probe_tokenizer_path_from_pretrained"]


class Tokenizer:
    def __init__(self, model_path: str):
        if model_path.endswith(".model"):
            from sentencepiece import SentencePieceProcessor
            assert os.path.isfile(model_path), model_path
            self.tokenizer_type = "spm"
            self.tokenizer = SentencePieceProcessor(model_file=model_path)
            self.bos_id: int = self.tokenizer.bos_id()
            self.eos_id: int = self.tokenizer.eos_id()
            assert self.tokenizer.vocab_size() == self.tokenizer.get_piece_size()
        else:
            from transformers import AutoTokenizer
            self.tokenizer_type = "transformers"
            self.tokenizer = AutoTokenizer.from_pretrained(model_path, trust_remote_code=True)
            self.bos_id = self.tokenizer.bos_token_id
            if self.bos_id is None:
                self.bos_id = self.tokenizer.eos_token_id
            self.eos_id = self.tokenizer.eos_token_id
            assert self.eos_id is not None
        self._probe_tokenizer_style()

    def encode(self, s: str, bos: bool, eos: bool) -> List[int]:
        assert type(s) is str
        if self.tokenizer_type == "transformers":
            t = self.tokenizer.encode(s, truncation=False, add_special_tokens=False)
        else:
            t = self.tokenizer.encode(s)
        if bos:
            t = [self.bos_id] + t
        if eos:
            t = t + [self.eos_id]
        return t

    def encode_segment(self, s: str) -> List[int]:
        s = s.lstrip(" ")
        if self.need_space_before_segment:
            return self.encode(" " + s, bos=False, eos=False)
        return self.encode(s, bos=False, eos=False)

    def encode_wo_prefix_space(self, s: str) -> List[int]:
        if self.need_space_before_segment:
            return self.encode(s, bos=False, eos=False)
        # tokenizer.py:76-88: find a prefix that stays a separate token, then strip it
        for prefix in ["@", "\n", "\\", "=", ">", "`"]:
            pt = self.encode(prefix, bos=False, eos=False)
            ct = self.encode(prefix + s, bos=False, eos=False)
            if ct[:len(pt)] == pt:
                return ct[len(pt):]
        raise NotImplementedError(f"All prefixes are merged into {s} during tokenization")

    def _probe_tokenizer_style(self) -> None:
        """tokenizer.py:90-112: does a cut-out segment need an explicit leading space?"""
        s1 = self.encode("Hi my darling", bos=False, eos=False)
        s2 = self.encode("my darling", bos=False, eos=False)
        if s1[-len(s2):] == s2:
            self.need_space_before_segment = False
        else:
            s3 = self.encode(" my darling", bos=False, eos=False)
            assert s1[-len(s3):] == s3
            self.need_space_before_segment = True

    def decode(self, t: List[int]) -> str:
        return self.tokenizer.decode(t)

    def save(self, save_dir: str) -> None:
        if self.tokenizer_type == "transformers":
            self.tokenizer.save_pretrained(save_dir)
        else:
            with open(Path(save_dir) / "tokenizer.model", "wb") as f:
                f.write(self.tokenizer.serialized_model_proto())

    @property
    def n_words(self) -> int:
        return self.tokenizer.vocab_size() if self.tokenizer_type == "spm" else len(self.tokenizer)


def probe_tokenizer_path_from_pretrained(pretrained_path: str) -> Optional[str]:
    """tokenizer.py:134-156."""
    p = Path(pretrained_path)
    if (p / "tokenizer.model").exists():
        return str(p / "tokenizer.model")
    if (p / "tokenizer.json").exists() and (p / "tokenizer_config.json").exists():
        return str(pretrained_path)
    return None
