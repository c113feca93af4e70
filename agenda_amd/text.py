"""Prompt side stand-ins.  The CLIP tokenizer/text-encoder are SURVEY.md §8(f) rank-2 "next"
(no vocab or weights exist offline); these keep the reference's call surface
(`pipeline.tokenizer.add_tokens / convert_tokens_to_ids / tokenize / __len__`,
`text_encoder.resize_token_embeddings / get_input_embeddings().weight`, reference
data_generation/data_generation.py:47-52) with deterministic synthetic embeddings, so token
selection, learned-embedding injection and `compute_token_merge_indices` run end to end.
"""
from __future__ import annotations

import hashlib
import re
from typing import Dict, List

import torch


class SimpleTokenizer:
    """Word-level tokenizer with CLIP-style '</w>' end-of-word markers and added-token support."""
    bos_token, eos_token = "<|startoftext|>", "<|endoftext|>"

    def __init__(self, model_max_length: int = 77):
        self.model_max_length = model_max_length
        self.vocab: Dict[str, int] = {self.bos_token: 0, self.eos_token: 1}
        self.added: List[str] = []

    def __len__(self):
        return len(self.vocab)

    def add_tokens(self, tokens) -> int:
        if isinstance(tokens, str):
            tokens = [tokens]
        n = 0
        for t in tokens:
            if t not in self.vocab:
                self.vocab[t] = len(self.vocab)
                self.added.append(t)
                n += 1
        return n

    def tokenize(self, text: str) -> List[str]:
        out: List[str] = []
        text = text.lower()
        if self.added:
            pat = "(" + "|".join(re.escape(a.lower()) for a in sorted(self.added, key=len, reverse=True)) + ")"
            parts = re.split(pat, text)
        else:
            parts = [text]
        added_l = {a.lower(): a for a in self.added}
        for part in parts:
            if part in added_l:
                out.append(added_l[part])          # added tokens are kept whole, no '</w>'
                continue
            for w in re.findall(r"[a-z0-9]+|[^\sa-z0-9]", part):
                out.append(w + "</w>")
        return out

    def convert_tokens_to_ids(self, tokens):
        single = isinstance(tokens, str)
        toks = [tokens] if single else list(tokens)
        ids = []
        for t in toks:
            if t not in self.vocab:
                self.vocab[t] = len(self.vocab)
            ids.append(self.vocab[t])
        return ids[0] if single else ids

    def encode(self, text: str) -> List[int]:
        ids = [0] + self.convert_tokens_to_ids(self.tokenize(text))[: self.model_max_length - 2] + [1]
        return ids + [1] * (self.model_max_length - len(ids))


class _Emb:
    def __init__(self, weight):
        self.weight = weight


class SyntheticTextEncoder:
    """Deterministic embedding table + positional code -> [B, T, D] fp32 'encoder_hidden_states'."""

    def __init__(self, tokenizer: SimpleTokenizer, dim: int, seed: int = 7):
        self.tokenizer, self.dim, self.seed = tokenizer, dim, seed
        self._table = torch.zeros(0, dim)
        self.resize_token_embeddings(max(len(tokenizer), 2))
        g = torch.Generator().manual_seed(seed)
        self._pos = torch.randn(tokenizer.model_max_length, dim, generator=g) * 0.3

    def _row(self, idx: int) -> torch.Tensor:
        h = int.from_bytes(hashlib.sha256(f"{self.seed}:{idx}".encode()).digest()[:8], "little") % (2 ** 31)
        return torch.randn(self.dim, generator=torch.Generator().manual_seed(h))

    def resize_token_embeddings(self, n: int):
        old = self._table.shape[0]
        if n > old:
            self._table = torch.cat([self._table, torch.stack([self._row(i) for i in range(old, n)])])
        self._emb = _Emb(self._table)
        return self._emb

    def get_input_embeddings(self):
        self._emb.weight.data = self._table      # same storage: writes to .weight.data[ids] land in the table
        return self._emb

    def __call__(self, prompts: List[str]) -> torch.Tensor:
        self.resize_token_embeddings(len(self.tokenizer))
        rows = []
        for p in prompts:
            ids = self.tokenizer.encode(p)
            self.resize_token_embeddings(len(self.tokenizer))
            rows.append(self._table[torch.tensor(ids)] + self._pos)
        x = torch.stack(rows)
        return x.to(torch.bfloat16).to(torch.float32)


class HipCLIPTextEncoder:
    """`pipeline.text_encoder` on the GPU (agd_text_encode: CLIP text transformer as HIP kernels).  Keeps the
    surface the reference touches (data_generation.py:49-52): `resize_token_embeddings(n)`,
    `get_input_embeddings().weight.data[ids] = rows`; rows of added tokens are pushed to the device before each
    encode.  Works with a real CLIPTokenizer (has `__call__` returning input_ids) or the SimpleTokenizer stand-in."""

    def __init__(self, engine, tokenizer, text_sd):
        self.engine, self.tokenizer = engine, tokenizer
        key = next(k for k in text_sd if k.endswith("embeddings.token_embedding.weight"))
        self.base_vocab = text_sd[key].shape[0]
        self.max_len = engine.cfg.text.max_position_embeddings
        self._emb = _Emb(text_sd[key].detach().float().clone())     # host master copy (fp32), like nn.Embedding.weight

    def resize_token_embeddings(self, n: int):
        cur = self._emb.weight.shape[0]
        if n > cur:
            if n > self.base_vocab + 256:
                raise ValueError("at most 256 added tokens")
            self._emb.weight = torch.cat([self._emb.weight, torch.zeros(n - cur, self._emb.weight.shape[1])])
            self._emb.weight.data = self._emb.weight
        return self._emb

    def get_input_embeddings(self):
        return self._emb

    def _ids(self, prompts):
        tok = self.tokenizer
        if hasattr(tok, "encode") and isinstance(tok, SimpleTokenizer):
            vocab = self._emb.weight.shape[0]
            rows = []
            for p in prompts:
                ids = tok.encode(p)
                rows.append([min(i, vocab - 1) for i in ids])
            return torch.tensor(rows, dtype=torch.int32)
        enc = tok(prompts, padding="max_length", max_length=self.max_len, truncation=True, return_tensors="pt")
        return enc.input_ids.to(torch.int32)

    def __call__(self, prompts):
        for tid in range(self.base_vocab, self._emb.weight.shape[0]):        # learned / added tokens
            self.engine.text_set_embedding_row(tid, self._emb.weight[tid])
        return self.engine.text_encode(self._ids(prompts)).cpu()
