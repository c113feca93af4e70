"""`UNetCrossAttentionHooker` -- drop-in for reference data_generation/hook.py:14-122.

Same constructor (`is_train`, `latent_hw` = SIDE length, hook.py:15-22), `clear()` (hook.py:25-26),
`compute_global_heat_map()` (hook.py:59-81) and the diffusers attention-processor call signature
(hook.py:83-89).  Installed with `unet.set_attn_processor(hooker)` (reference
finetune_sd_token.py:755-757) it switches the fused HIP cross-attention kernel into hook-mode
recording: per call, head-mean probability maps (hook.py:55) of the kept batch rows
(hook.py:48-49) are bicubic-upsampled, clamped and summed on the GPU as they are produced, so the
[B*H, N, 77] tensor of hook.py:108 and the python list of hook.py:112 never exist.
Called directly as a processor it routes one `Attention` module (attn1 or attn2, with or without an
attention mask) through the C-ABI seam `agd_attn_processor`.
"""
from __future__ import annotations

import math

import torch


class UNetCrossAttentionHooker:
    def __init__(self, is_train: bool = True, latent_hw: int = 64):
        self.is_train = is_train
        self.latent_hw = latent_hw
        self.cross_attn_maps = []          # hook.py:19 -- filled by direct (seam) calls; the fused UNet walk streams instead
        self._pipe = None
        self._bp = 0
        self._tokens = 0
        self._side = 0

    # -- wiring ---------------------------------------------------------------------------
    def _bind(self, pipe):
        self._pipe = pipe

    def _on_generate(self, batch: int, latent_side: int, tokens: int):
        self._bp = 2 * batch if self.is_train else batch
        self._tokens, self._side = tokens, latent_side

    def _need_pipe(self):
        if self._pipe is None:
            raise RuntimeError("hooker is not installed: call unet.set_attn_processor(hooker) first")

    # -- reference surface ----------------------------------------------------------------
    def clear(self):
        """hook.py:25-26"""
        self.cross_attn_maps.clear()
        if self._pipe is not None and self._bp:
            self._pipe._apply_record_mode()
            self._pipe.engine.record_reset(self._bp // 2 if self.is_train else self._bp, self._side)

    @property
    def num_recorded(self) -> int:
        return self._pipe.engine.hook_count() if self._pipe is not None else 0

    def compute_global_heat_map(self) -> torch.Tensor:
        """hook.py:59-81 -> [B', T, latent_hw, latent_hw] fp32; RuntimeError('No heat maps found.')
        when nothing was recorded (hook.py:74-77)."""
        if self._pipe is None or self._bp == 0 or self.num_recorded == 0:
            raise RuntimeError("No heat maps found.")
        if self._side != self.latent_hw:
            raise ValueError(f"latent_hw={self.latent_hw} but the UNet ran at latent side {self._side}")
        return self._pipe.engine.hook_global(self._bp, self._tokens, self._side)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None):
        """hook.py:83-122 for one `Attention` module through the C-ABI seam (`agd_attn_processor`):
        cross-attention (`encoder_hidden_states` given; the head-mean map is appended, hook.py:110-112) or
        self-attention (`encoder_hidden_states is None`, hook.py:95-99; records nothing).  `attention_mask` is the
        additive float mask of diffusers' `prepare_attention_mask` (hook.py:92): [B, keys] or [B, 1, keys]."""
        self._need_pipe()
        b2, n, _ = hidden_states.shape
        mask = None
        if attention_mask is not None:
            mask = attention_mask.to(torch.float32)
            if mask.ndim == 3 and mask.shape[1] == 1:
                mask = mask[:, 0]
            n_keys = n if encoder_hidden_states is None else encoder_hidden_states.shape[1]
            if mask.ndim != 2 or tuple(mask.shape) != (b2, n_keys):
                raise ValueError(f"attention_mask must be [B, keys] or [B, 1, keys] = ({b2}, {n_keys}), got {tuple(attention_mask.shape)}")
        if encoder_hidden_states is None:                          # is_cross_attn False: nothing is recorded
            if getattr(attn, "is_cross", False):
                raise ValueError(f"{attn.name} is a cross-attention module: encoder_hidden_states is required "
                                 "(its to_k/to_v take the text width)")
            return self._pipe.engine.attn_processor(attn.name, hidden_states, None, mask, record=False)
        side = int(math.sqrt(n))
        if side * side != n:                                       # hook.py:43-46: map_.view(.., h, w) fails the same way
            raise RuntimeError(f"shape '[-1, {side}, {side}]' is invalid for input of size {n} per map")
        bp = b2 if self.is_train else b2 // 2
        if self._bp != bp or self._side != self.latent_hw or self._tokens != encoder_hidden_states.shape[1]:
            self._bp, self._side, self._tokens = bp, self.latent_hw, encoder_hidden_states.shape[1]
            self._pipe._apply_record_mode()
            self._pipe.engine.set_context(encoder_hidden_states)
            self._pipe.engine.record_reset(bp // 2 if self.is_train else bp, self.latent_hw)
        out = self._pipe.engine.attn_processor(attn.name, hidden_states, encoder_hidden_states, mask, record=True)
        self.cross_attn_maps.append(self._pipe.engine.hook_last_map(bp, self._tokens, n))     # hook.py:110-112
        return out
