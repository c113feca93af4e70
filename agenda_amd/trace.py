"""`daam.trace` surface (reference data_generation/data_generation.py:57,64,74-77) over the fused
HIP recorder.  Semantics restated from the public `daam` package [upstream-knowledge, SURVEY.md
§8a rows D1-D4]: per-(layer, head) time-summed conditional-half maps, mid block excluded,
bicubic -> clamp -> mean, rows truncated to len(tokenize(prompt)) + 2."""
from __future__ import annotations

from typing import List, Optional

import torch


def compute_token_merge_indices(tokenizer, prompt: str, word: str, word_idx: Optional[int] = None, offset_idx: int = 0):
    """`daam.utils.compute_token_merge_indices` (also used by reference dataset.py:93)."""
    merge_idxs: List[int] = []
    tokens = [x.replace("</w>", "") for x in tokenizer.tokenize(prompt.lower())]
    if word_idx is None:
        word = word.lower()
        search = [x.replace("</w>", "") for x in tokenizer.tokenize(word)]
        starts = [x + offset_idx for x in range(len(tokens)) if tokens[x:x + len(search)] == search]
        for s in starts:
            merge_idxs += [i + s for i in range(len(search))]
        if not merge_idxs:
            raise ValueError(f"Search word {word} not found in prompt!")
    else:
        merge_idxs.append(word_idx)
    return [x + 1 for x in merge_idxs], word_idx


class WordHeatMap:
    def __init__(self, heatmap: torch.Tensor, word: str):
        self.heatmap = heatmap
        self.word = word

    @property
    def value(self):
        return self.heatmap


class GlobalHeatMap:
    def __init__(self, tokenizer, prompt: str, heat_maps: torch.Tensor):
        self.tokenizer, self.prompt, self.heat_maps = tokenizer, prompt, heat_maps

    def compute_word_heat_map(self, word: str, word_idx: Optional[int] = None, offset_idx: int = 0) -> WordHeatMap:
        merge_idxs, _ = compute_token_merge_indices(self.tokenizer, self.prompt, word, word_idx, offset_idx)
        if max(merge_idxs) >= self.heat_maps.shape[0]:
            raise ValueError(f"token index {max(merge_idxs)} beyond the {self.heat_maps.shape[0]} recorded rows")
        # (integer indexing: views, no index tensor uploaded through a blocking copy)
        return WordHeatMap(torch.stack([self.heat_maps[int(i)] for i in merge_idxs]).mean(0), word)


class trace:
    """`with trace(pipe) as trc: pipe(...); trc.compute_global_heat_map()`."""

    def __init__(self, pipe, rec_tokens: Optional[int] = None):
        self.pipe = pipe
        # recording fewer rows than 77 is safe: daam only ever reads the first len(tokens)+2 rows
        self.rec_tokens = rec_tokens or pipe.cfg.max_tokens
        self.batch = 0
        self.latent_side = 0
        self.last_prompt = None
        self._ran = False

    def __enter__(self):
        if self.pipe._trace is not None:
            raise RuntimeError("a trace is already active on this pipeline")
        self.pipe._trace = self
        self.pipe._apply_record_mode()
        return self

    def __exit__(self, *exc):
        self.pipe._trace = None
        self.pipe._apply_record_mode()
        return False

    def _on_generate(self, batch: int, latent_side: int, prompt: Optional[str]):
        self.batch, self.latent_side, self.last_prompt, self._ran = batch, latent_side, prompt, True

    def compute_global_heat_map(self, prompt: Optional[str] = None, image_index: int = 0, normalize: bool = False) -> GlobalHeatMap:
        if not self._ran:
            raise RuntimeError("No heat maps found. Did you forget to call `with trace(...)`?")
        prompt = prompt if prompt is not None else self.last_prompt
        rows = self.rec_tokens
        if prompt is not None:
            rows = min(rows, len(self.pipe.tokenizer.tokenize(prompt)) + 2)   # 1 for SOS and 1 for padding
        maps = self.pipe.engine.daam_global(image_index, rows, self.latent_side)
        if normalize:
            maps = maps / (maps[1:-1].sum(0, keepdim=True) + 1e-6)
        return GlobalHeatMap(self.pipe.tokenizer, prompt or "", maps)
