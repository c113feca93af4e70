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


class _SeamCall(torch.autograd.Function):
    """One cross-attention call of the seam with autograd to its two inputs: forward = `agd_attn_processor` (+ the recorded
    map), backward = `agd_attn_processor_backward` (HIP: recompute P, dS, dQ/dK/dV, input-gradient GEMMs).  This is the path
    the reference's attention regulariser takes from `cross_attn_maps` back to the learned token embeddings
    (finetune_sd_token.py:1043-1069,1089)."""

    @staticmethod
    def forward(ctx, hidden, enc, hooker, name, mask):
        eng = hooker._pipe.engine
        out = eng.attn_processor(name, hidden, enc, mask, record=True)
        k = eng.hook_num_maps() - 1
        amap = eng.hook_map(k) if hooker.is_train else eng.hook_last_map(hooker._bp, hooker._tokens, hidden.shape[1])
        ctx.hooker, ctx.name, ctx.devs = hooker, name, (hidden.device, enc.device)
        ctx.save_for_backward(hidden.detach(), enc.detach())
        return out.to(hidden.device), amap

    @staticmethod
    def backward(ctx, d_out, d_map):
        hidden, enc = ctx.saved_tensors
        hk = ctx.hooker
        zero = lambda g: g is None or not bool(torch.count_nonzero(g))
        if zero(d_out) and zero(d_map):
            return torch.zeros_like(hidden), torch.zeros_like(enc), None, None, None
        dh, dc = hk._pipe.engine.attn_processor_backward(ctx.name, hidden, enc, None if zero(d_out) else d_out, None if zero(d_map) else d_map,
                                                         hk.is_train, want_hidden=ctx.needs_input_grad[0], want_ctx=ctx.needs_input_grad[1])
        return (dh.to(ctx.devs[0]) if dh is not None else None), (dc.to(ctx.devs[1]) if dc is not None else None), None, None, None


class UNetCrossAttentionHooker:
    def __init__(self, is_train: bool = True, latent_hw: int = 64):
        self.is_train = is_train
        self.latent_hw = latent_hw
        self._seam_maps = []               # inference mode: maps of direct (seam) calls (the fused UNet walk streams instead)
        self._grad_maps = {}               # train mode: index in the device store -> autograd-connected map of a seam call
        self._cache = (-1, [])
        self._pipe = None
        self._bp = 0
        self._tokens = 0
        self._side = 0

    @property
    def cross_attn_maps(self):
        """hook.py:19,110-112: the per-call maps `[B', T, h, w]` since the last `clear()`.  Train mode (`is_train=True`, what
        finetune_sd_token.py:755-757 installs) keeps every recorded call of the fused UNet walk and of direct seam calls on the
        device, in call order; inference mode streams the fused walk's maps into the global heat map and lists seam calls only."""
        if not self.is_train or self._pipe is None:
            return self._seam_maps
        n = self._pipe.engine.hook_num_maps()
        if self._cache[0] != n:
            self._cache = (n, [self._grad_maps[k] if k in self._grad_maps else self._pipe.engine.hook_map(k) for k in range(n)])
        return self._cache[1]

    # -- wiring ---------------------------------------------------------------------------
    def _bind(self, pipe):
        self._pipe = pipe

    def _on_generate(self, batch: int, latent_side: int, tokens: int):
        """pipe(...) reset the device recorder (agd_record_reset): the Python-side views of the old store go with it, so that
        `cross_attn_maps[k]` can never return the autograd map of an earlier seam call for a freshly recorded index k."""
        self._bp = 2 * batch if self.is_train else batch
        self._tokens, self._side = tokens, latent_side
        self._seam_maps.clear(); self._grad_maps.clear(); self._cache = (-1, [])

    def _ensure(self, rows: int, side: int, tokens: int):
        """(Re)size the device recorder when the kept batch rows / latent side / token count change; otherwise maps keep
        accumulating until `clear()` (hook.py:25-26)."""
        if self._bp != rows or self._side != side or self._tokens != tokens:
            self._bp, self._side, self._tokens = rows, side, tokens
            self._pipe._apply_record_mode()
            self._pipe.engine.hook_reset(rows, side)
            self._seam_maps.clear(); self._grad_maps.clear(); self._cache = (-1, [])

    def _need_pipe(self):
        if self._pipe is None:
            raise RuntimeError("hooker is not installed: call unet.set_attn_processor(hooker) first")

    # -- reference surface ----------------------------------------------------------------
    def clear(self):
        """hook.py:25-26"""
        self._seam_maps.clear(); self._grad_maps.clear(); self._cache = (-1, [])
        if self._pipe is not None and self._bp:
            self._pipe._apply_record_mode()
            self._pipe.engine.hook_reset(self._bp, self._side)

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
            self._pipe.engine.set_context(encoder_hidden_states)
            self._ensure(bp, self.latent_hw, encoder_hidden_states.shape[1])
        if torch.is_grad_enabled() and (hidden_states.requires_grad or encoder_hidden_states.requires_grad):
            if mask is not None:
                raise NotImplementedError("autograd through the seam with an attention_mask")
            out, amap = _SeamCall.apply(hidden_states, encoder_hidden_states, self, attn.name, None)
        else:
            out = self._pipe.engine.attn_processor(attn.name, hidden_states, encoder_hidden_states, mask, record=True)
            amap = None if self.is_train else self._pipe.engine.hook_last_map(bp, self._tokens, n)
        if self.is_train:                                          # hook.py:110-112: kept on the device, listed by the property
            if amap is not None:
                self._grad_maps[self._pipe.engine.hook_num_maps() - 1] = amap
            self._cache = (-1, [])
        else:
            self._seam_maps.append(amap)
        return out

    def attention_regulariser(self, new_tokens_start_indices, n_object_embedding: int, reg_weight: float, want_grads: bool = False):
        """The cross-attention loss of finetune_sd_token.py:1040-1069 over the kept maps, on the device (`agd_op_attn_reg_loss`):
        returns (attn_loss, bg_attn_loss, fg_attn_loss) as 0-d cuda tensors [+ the list of d attn_loss / d map when `want_grads`].
        `new_tokens_start_indices`: int [B, n_new_tokens], -1 = token absent (dataset.py:88-97)."""
        from . import ops
        maps = [m.detach() for m in self.cross_attn_maps]
        if not maps:
            raise RuntimeError("No heat maps found.")
        idx = torch.as_tensor(new_tokens_start_indices).to(torch.int64).cpu()
        if idx.ndim != 2 or idx.shape[0] != maps[0].shape[0]:
            raise ValueError(f"new_tokens_start_indices must be [B={maps[0].shape[0]}, n_tokens], got {tuple(idx.shape)}")
        has = idx[:, 0] > 0                                                        # :1048 "whether there is a car in the image"
        T = maps[0].shape[1]
        # the reference indexes sample_attn_map[obj_index] (:1049-1060) and raises IndexError on a token row the map does not have
        # (a mis-set n_object_embedding, a truncated prompt); only "no object in this sample" is a silent skip
        worst = int(torch.where(has, idx[:, 0] + n_object_embedding, torch.zeros_like(idx[:, 0])).max()) if bool(has.any()) else 0
        worst = max(worst, int(idx[has].max()) if bool(has.any()) else 0)
        if worst >= T:
            raise IndexError(f"index {worst} is out of bounds for dimension 0 with size {T}")
        obj = torch.where(has, idx[:, 0] + n_object_embedding, torch.full_like(idx[:, 0], -1))          # :1049
        fg = torch.where(has, idx[:, 0], torch.full_like(idx[:, 0], -1))                                 # :1055
        last = torch.tensor([int(r[r > -1][-1]) if bool((r > -1).any()) else -1 for r in idx])          # :1059
        bg = torch.where(has, last, torch.full_like(last, -1))
        cnt = int(has.sum())
        dev = maps[0].device
        bg_l, fg_l, grads = torch.zeros((), device=dev), torch.zeros((), device=dev), []
        for m in maps:
            if cnt == 0:
                grads.append(torch.zeros_like(m)); continue
            loss, dmap = ops.attn_reg_loss(m, obj, fg, bg, reg_weight / cnt, want_grad=want_grads)           # :1064-1065
            bg_l, fg_l = bg_l + loss[:, 0].sum(), fg_l + loss[:, 1].sum()
            if want_grads:
                grads.append(dmap / len(maps))                                     # :1068 attn_loss / len(cross_attn_maps)
        attn = (bg_l + fg_l) / len(maps)
        return (attn, bg_l, fg_l, grads) if want_grads else (attn, bg_l, fg_l)
