"""Text <-> token ids for the path's host side (prompt encoding in ``MetaModel.generate``, label building in the dialog
dataset, stop-string matching).  Public surface = what ``MetaModel`` / the dataset / the checkpoint writer call on the
reference's ``accessory.model.tokenizer.Tokenizer``: ``encode(s, bos, eos)``, ``encode_segment``, ``encode_wo_prefix_space``,
``decode``, ``save``, ``n_words``, ``bos_id``, ``eos_id``, ``need_space_before_segment``, ``tokenizer_type``, ``tokenizer``.
Behaviour is pinned by ``tests/golden/meta_tiny.json`` (ids, segment ids, vocabulary facts captured from the reference).

Two vocabularies are supported, each behind a small adapter so that the wrapper itself has no branches:
a SentencePiece ``*.model`` file (Llama) and a HuggingFace tokenizer directory."""
from __future__ import annotations

import os
from pathlib import Path
from typing import List, Optional, Sequence

__all__ = ["Tokenizer", "probe_tokenizer_path_from_pretrained"]


class _SentencePieceVocab:
    kind = "spm"

    def __init__(self, model_file: str):
        from sentencepiece import SentencePieceProcessor
        if not os.path.isfile(model_file):
            raise AssertionError(model_file)
        self.impl = SentencePieceProcessor(model_file=model_file)
        self.bos, self.eos = self.impl.bos_id(), self.impl.eos_id()
        self.size = self.impl.vocab_size()
        assert self.size == self.impl.get_piece_size()

    def ids(self, text: str) -> List[int]:
        return self.impl.encode(text)

    def write(self, directory: str) -> None:
        (Path(directory) / "tokenizer.model").write_bytes(self.impl.serialized_model_proto())


class _HuggingFaceVocab:
    kind = "transformers"

    def __init__(self, directory: str):
        from transformers import AutoTokenizer
        self.impl = AutoTokenizer.from_pretrained(directory, trust_remote_code=True)
        self.eos = self.impl.eos_token_id
        assert self.eos is not None
        self.bos = self.impl.bos_token_id if self.impl.bos_token_id is not None else self.eos
        self.size = len(self.impl)

    def ids(self, text: str) -> List[int]:
        return self.impl.encode(text, truncation=False, add_special_tokens=False)

    def write(self, directory: str) -> None:
        self.impl.save_pretrained(directory)


def _ends_with(seq: Sequence[int], tail: Sequence[int]) -> bool:
    return len(tail) <= len(seq) and list(seq[len(seq) - len(tail):]) == list(tail)


# single characters that a vocabulary normally keeps as their own token in front of arbitrary text
_SENTINELS = ("@", "\n", "\\", "=", ">", "`")


class Tokenizer:
    def __init__(self, model_path: str):
        self._vocab = _SentencePieceVocab(model_path) if model_path.endswith(".model") else _HuggingFaceVocab(model_path)
        self.tokenizer = self._vocab.impl                 # the wrapped object (the reference exposes it under this name)
        self.tokenizer_type = self._vocab.kind
        self.bos_id: int = self._vocab.bos
        self.eos_id: int = self._vocab.eos
        self.need_space_before_segment = self._segments_need_leading_space()

    # ---- whole strings -------------------------------------------------------------------------------------------
    def encode(self, s: str, bos: bool, eos: bool) -> List[int]:
        assert type(s) is str
        ids = list(self._vocab.ids(s))
        return ([self.bos_id] if bos else []) + ids + ([self.eos_id] if eos else [])

    def decode(self, t: List[int]) -> str:
        return self.tokenizer.decode(t)

    # ---- pieces of a longer string (their ids must equal the ids they get INSIDE that string) -------------------------
    def _segments_need_leading_space(self) -> bool:
        """SentencePiece puts the word-boundary marker on the token AFTER a space, so "my darling" cut out of
        "Hi my darling" encodes identically with or without its own leading space; byte-level BPE vocabularies attach the
        space to the word and need it spelled out.  Probe which kind this is."""
        whole = self._vocab.ids("Hi my darling")
        if _ends_with(whole, self._vocab.ids("my darling")):
            return False
        assert _ends_with(whole, self._vocab.ids(" my darling"))
        return True

    def encode_segment(self, s: str) -> List[int]:
        """ids of a word-initial piece (an answer, a stop string that follows a space)."""
        body = s.lstrip(" ")
        return self.encode(" " + body if self.need_space_before_segment else body, bos=False, eos=False)

    def encode_wo_prefix_space(self, s: str) -> List[int]:
        """ids of a piece that continues a word (no boundary marker on its first token)."""
        if self.need_space_before_segment:
            return self.encode(s, bos=False, eos=False)
        for mark in _SENTINELS:
            head = self._vocab.ids(mark)
            joined = self._vocab.ids(mark + s)
            if list(joined[:len(head)]) == list(head):       # the sentinel survived as its own token(s): drop it
                return list(joined[len(head):])
        raise NotImplementedError(f"All prefixes are merged into {s} during tokenization")

    # ---- bookkeeping ---------------------------------------------------------------------------------------------
    def save(self, save_dir: str) -> None:
        self._vocab.write(save_dir)

    @property
    def n_words(self) -> int:
        return self._vocab.size


def probe_tokenizer_path_from_pretrained(pretrained_path: str) -> Optional[str]:
    """Where a checkpoint directory keeps its vocabulary: ``tokenizer.model`` wins, else an HF pair of json files."""
    root = Path(pretrained_path)
    if (root / "tokenizer.model").is_file():
        return str(root / "tokenizer.model")
    if all((root / name).is_file() for name in ("tokenizer.json", "tokenizer_config.json")):
        return str(pretrained_path)
    return None
