"""Tokenizer front end.  The reference uses transformers' CLIPTokenizer loaded from the SD
checkpoint (training/coach.py:600-611); neither the vocabulary files nor network access exist on
the build/GPU boxes, so:
  * `load_tokenizer(path)` returns the real CLIPTokenizer when `path` holds vocab.json/merges.txt;
  * otherwise a deterministic stand-in with the same call surface the train path uses
    (`add_tokens`, `convert_tokens_to_ids`, `encode`, `__call__(padding="max_length")`, `__len__`,
    `model_max_length`, `unk_token_id`): lower-cased whitespace/punctuation words hashed into the
    49 406 regular ids, BOS 49406, EOS/pad 49407, added tokens appended from 49408.
Token ids feed an embedding gather only, so throughput and kernel parity do not depend on BPE.
"""
from __future__ import annotations

import os
import re
import zlib
from typing import Dict, List, Union

import torch


class HashTokenizer:
    model_max_length = 77

    def __init__(self, vocab_size: int = 49408):
        self.base_vocab = vocab_size
        self.bos_token_id = vocab_size - 2
        self.eos_token_id = vocab_size - 1
        self.pad_token_id = vocab_size - 1
        self.unk_token_id = vocab_size - 1
        self.added: Dict[str, int] = {}

    def __len__(self):
        return self.base_vocab + len(self.added)

    def add_tokens(self, tokens: Union[str, List[str]]) -> int:
        tokens = [tokens] if isinstance(tokens, str) else list(tokens)
        n = 0
        for t in tokens:
            if t not in self.added:
                self.added[t] = self.base_vocab + len(self.added)
                n += 1
        return n

    def _word_id(self, w: str) -> int:
        return zlib.crc32(w.lower().encode()) % (self.base_vocab - 2)

    def _split(self, text: str) -> List[str]:
        if self.added:
            pat = "(" + "|".join(re.escape(t) for t in sorted(self.added, key=len, reverse=True)) + ")"
            parts = re.split(pat, text)
        else:
            parts = [text]
        out: List[str] = []
        for p in parts:
            if p in self.added:
                out.append(p)
            else:
                out += re.findall(r"[A-Za-z0-9]+|[^\sA-Za-z0-9]", p)
        return out

    def convert_tokens_to_ids(self, tokens):
        if isinstance(tokens, str):
            return self.added.get(tokens, self._word_id(tokens))
        return [self.convert_tokens_to_ids(t) for t in tokens]

    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        ids = [self.convert_tokens_to_ids(t) for t in self._split(text)]
        return [self.bos_token_id] + ids + [self.eos_token_id] if add_special_tokens else ids

    def __call__(self, text, padding="max_length", truncation=True, max_length=None, return_tensors="pt"):
        texts = [text] if isinstance(text, str) else list(text)
        L = max_length or self.model_max_length
        rows = []
        for t in texts:
            ids = self.encode(t)[:L]
            ids[-1] = self.eos_token_id if len(ids) == L else ids[-1]
            rows.append(ids + [self.pad_token_id] * (L - len(ids)))
        out = torch.tensor(rows, dtype=torch.int64)

        class _Enc:
            input_ids = out
        return _Enc()


def load_tokenizer(path: str = None, vocab_size: int = 49408):
    for sub in ("tokenizer", ""):
        d = os.path.join(path, sub) if path else None
        if d and os.path.isdir(d) and os.path.exists(os.path.join(d, "vocab.json")):
            from transformers import CLIPTokenizer
            return CLIPTokenizer.from_pretrained(d)
    return HashTokenizer(vocab_size)
