"""`StableDiffusionPipeline`-shaped surface over libagenda_hip.so.

Mirrors what the reference touches on `diffusers.StableDiffusionPipeline`
(reference data_generation/data_generation.py:30-31,47-52,59): `from_pretrained`, `.to("cuda")`,
`.tokenizer`, `.text_encoder`, `.unet`, `.vae`, `.scheduler`, `set_progress_bar_config`,
`__call__(prompt, num_inference_steps=, generator=).images[0]`.

Differences that are deliberate:
 * batched prompts/seeds per call (the reference runs batch 1);
 * DDIM (eta 0) instead of the checkpoint's default scheduler (BASELINE.json's metric);
 * initial latents come from a CPU `torch.Generator` (or are passed explicitly) so the CPU oracle
   and the GPU run share them; a CUDA generator cannot be reproduced on the host;
 * no safety checker (no weights offline): nothing is ever blacked out.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from .config import SDConfig, CONFIGS, UNetConfig, VAEConfig, SchedulerConfig, cross_attn_layer_names
from .scheduler import DDIMScheduler, PNDMScheduler, SCHEDULERS
from .text import SimpleTokenizer, SyntheticTextEncoder


@dataclass
class UNetOutput:
    """diffusers' UNet2DConditionOutput: `unet(...).sample`."""
    sample: torch.Tensor


@dataclass
class PipelineOutput:
    images: list
    latents: Optional[torch.Tensor] = None
    nsfw_content_detected: Optional[list] = None


def _make_cfg(cfg: SDConfig, workspace_bytes: int) -> _lib.AgdConfig:
    a = _lib.AgdConfig()
    a.struct_size = C.sizeof(_lib.AgdConfig)
    u, v = cfg.unet, cfg.vae
    a.in_channels, a.out_channels, a.n_levels = u.in_channels, u.out_channels, len(u.block_out_channels)
    for i, c in enumerate(u.block_out_channels):
        a.block_out_channels[i] = c
        a.down_cross[i] = int(u.down_cross[i])
        a.num_heads[i] = u.num_heads[i]
    a.layers_per_block, a.cross_attention_dim = u.layers_per_block, u.cross_attention_dim
    a.use_linear_projection, a.norm_num_groups = int(u.use_linear_projection), u.norm_num_groups
    a.vae_latent_channels, a.vae_out_channels, a.vae_n_levels = v.latent_channels, v.out_channels, len(v.block_out_channels)
    for i, c in enumerate(v.block_out_channels):
        a.vae_block_out_channels[i] = c
    a.vae_layers_per_block, a.vae_norm_num_groups = v.layers_per_block, v.norm_num_groups
    a.vae_scaling_factor = v.scaling_factor
    a.max_tokens = cfg.max_tokens
    a.prediction_type = 1 if cfg.sched.prediction_type == "v_prediction" else 0
    a.workspace_bytes = workspace_bytes
    t = getattr(cfg, "text", None)
    if t is not None:
        a.text_hidden, a.text_layers, a.text_heads = t.hidden_size, t.num_hidden_layers, t.num_attention_heads
        a.text_intermediate, a.text_vocab, a.text_max_pos = t.intermediate_size, t.vocab_size, t.max_position_embeddings
        a.text_act = 0 if t.hidden_act == "quick_gelu" else 1
        a.text_eps = t.layer_norm_eps
    return a


class Engine:
    """Owns one `agd_ctx` (one per GPU per process)."""

    def __init__(self, cfg: SDConfig, device: int = 0, workspace_bytes: int = 0):
        if not torch.cuda.is_available():
            raise _lib.AgendaHipError("agenda_amd needs an MI355X (no CPU fallback)")
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = device
        self._acfg = _make_cfg(cfg, workspace_bytes)
        self.ctx = self.lib.agd_create(device, C.byref(self._acfg))
        if not self.ctx:
            raise _lib.AgendaHipError("agd_create failed: " + self.lib.agd_last_error(None).decode())
        self.finalized = False

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.agd_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        _lib.check(rc, self.ctx, what)

    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str):
        for k, t in sd.items():
            t = t.detach().to(torch.float32).contiguous()
            shape = (C.c_longlong * t.ndim)(*t.shape)
            self._ck(self.lib.agd_load_tensor(self.ctx, (prefix + k).encode(), C.c_void_p(t.data_ptr()), 0, t.ndim, shape),
                     f"agd_load_tensor({prefix + k})")

    def finalize(self):
        self._ck(self.lib.agd_finalize(self.ctx), "agd_finalize")
        self.finalized = True

    @staticmethod
    def _stream():
        return _lib.current_stream_ptr()

    def _h2d(self, t: torch.Tensor) -> torch.Tensor:
        """fp32 copy of `t` on the engine's device.  A host tensor goes through one of two persistent pinned staging buffers per shape with a
        NON-blocking transfer: a pageable `.to(device)` blocks the launching thread until the copy ran -- behind everything already queued on
        the stream -- so every batch ended with the GPU idle while the host prepared the next one (1 - 1.5 ms per batch in the kernel trace);
        a fresh `pin_memory()` per call is worse (the pinned allocation stalls for tens of ms).  A staging buffer is rewritten only after the
        event behind its previous copy has completed."""
        t = t.detach().to(torch.float32)
        if t.device.type != "cpu":
            return t.to(device=f"cuda:{self.device}").contiguous()
        pool = self.__dict__.setdefault("_stage", {})
        slot = pool.get(tuple(t.shape))
        if slot is None:                                   # both buffers at the first use of a shape (a warm-up batch): pinning stalls for tens of ms
            slot = pool[tuple(t.shape)] = {"bufs": [torch.empty(t.shape, dtype=torch.float32).pin_memory() for _ in range(2)],
                                           "evs": [torch.cuda.Event() for _ in range(2)], "next": 0}
            for e in slot["evs"]:
                e.record()
        k = slot["next"]; slot["next"] = (k + 1) % 2
        slot["evs"][k].synchronize()
        slot["bufs"][k].copy_(t)
        out = slot["bufs"][k].to(device=f"cuda:{self.device}", non_blocking=True)
        slot["evs"][k].record()
        return out

    def set_context(self, ctx_emb: torch.Tensor):
        ctx_emb = self._h2d(ctx_emb)
        b2, t, _ = ctx_emb.shape
        self._ck(self.lib.agd_set_context(self.ctx, _lib.ptr(ctx_emb), b2, t, self._stream()), "agd_set_context")
        self._ctx_keepalive = ctx_emb

    def text_encode(self, input_ids: torch.Tensor) -> torch.Tensor:
        """`text_encoder(input_ids)[0]`: int [B,T] -> fp32 [B,T,H] (cuda)."""
        ids = input_ids.to(torch.int32).contiguous()
        b, t = ids.shape
        out = torch.empty(b, t, self.cfg.text.hidden_size, device=f"cuda:{self.device}", dtype=torch.float32)
        self._ck(self.lib.agd_text_encode(self.ctx, C.c_void_p(ids.data_ptr()), b, t, _lib.ptr(out), self._stream()), "agd_text_encode")
        torch.cuda.synchronize()          # `ids` may be a host tensor: keep it alive until the copy ran
        return out

    def text_set_embedding_row(self, token_id: int, row: torch.Tensor):
        row = row.detach().to(torch.float32).contiguous()
        self._ck(self.lib.agd_text_set_embedding_row(self.ctx, int(token_id), C.c_void_p(row.data_ptr())), "agd_text_set_embedding_row")

    def unet_forward(self, sample: torch.Tensor, timestep) -> torch.Tensor:
        """`timestep`: a number (one timestep for the batch) or a [B] tensor / sequence (one per image, the training call)."""
        sample = sample.to(device=f"cuda:{self.device}", dtype=torch.float32).contiguous()
        out = torch.empty_like(sample)
        b2, _, L, _ = sample.shape
        ts = timestep.detach().flatten().tolist() if torch.is_tensor(timestep) else (list(timestep) if isinstance(timestep, (list, tuple)) else [timestep])
        if len(ts) == 1:
            self._ck(self.lib.agd_unet_forward(self.ctx, _lib.ptr(sample), b2, L, float(ts[0]), _lib.ptr(out), self._stream()),
                     "agd_unet_forward")
        else:
            if len(ts) != b2:
                raise ValueError(f"timestep has {len(ts)} entries for a batch of {b2}")
            arr = (C.c_float * b2)(*[float(t) for t in ts])
            self._ck(self.lib.agd_unet_forward_ts(self.ctx, _lib.ptr(sample), b2, L, arr, _lib.ptr(out), self._stream()), "agd_unet_forward_ts")
        return out

    def denoise(self, latents: torch.Tensor, timesteps, a_t, a_p, guidance: float):
        assert latents.is_cuda and latents.dtype == torch.float32 and latents.is_contiguous()
        n = len(timesteps)
        ts = (C.c_float * n)(*[float(t) for t in timesteps])
        at = (C.c_float * n)(*[float(x) for x in a_t])
        ap = (C.c_float * n)(*[float(x) for x in a_p])
        b, _, L, _ = latents.shape
        self._ck(self.lib.agd_denoise(self.ctx, _lib.ptr(latents), b, L, n, ts, at, ap, float(guidance), self._stream()),
                 "agd_denoise")
        return latents

    def cfg_ddim_step(self, eps: torch.Tensor, latents: torch.Tensor, guidance: float, alpha_t: float, alpha_prev: float):
        """`scheduler.step` of the call-by-call loop: CFG combine of eps [2B,4,L,L] (rows [0,B) unconditional) + one DDIM
        (eta 0) update of `latents` [B,4,L,L] in place (`agd_cfg_ddim_step`; the fused loop is `denoise`)."""
        assert latents.is_cuda and latents.dtype == torch.float32 and latents.is_contiguous()
        eps = eps.to(device=latents.device, dtype=torch.float32).contiguous()
        b, _, L, _ = latents.shape
        if eps.shape[0] != 2 * b:
            raise ValueError(f"eps batch {eps.shape[0]} != 2 x latents batch {b}")
        self._ck(self.lib.agd_cfg_ddim_step(self.ctx, _lib.ptr(eps), _lib.ptr(latents), b, L, float(guidance), float(alpha_t),
                                            float(alpha_prev), self._stream()), "agd_cfg_ddim_step")
        return latents

    def denoise_plms(self, latents: torch.Tensor, timesteps, sample_coeff, eps_coeff, guidance: float):
        """The fused loop under PNDM/PLMS (`agd_denoise_plms`): len(timesteps) = num_inference_steps + 1 model evaluations."""
        assert latents.is_cuda and latents.dtype == torch.float32 and latents.is_contiguous()
        n = len(timesteps)
        ts = (C.c_float * n)(*[float(t) for t in timesteps])
        ca = (C.c_float * n)(*[float(x) for x in sample_coeff])
        cb = (C.c_float * n)(*[float(x) for x in eps_coeff])
        b, _, L, _ = latents.shape
        self._ck(self.lib.agd_denoise_plms(self.ctx, _lib.ptr(latents), b, L, n, ts, ca, cb, float(guidance), self._stream()),
                 "agd_denoise_plms")
        return latents

    def vae_decode(self, latents: torch.Tensor, want_f32: bool = False):
        latents = latents.to(device=f"cuda:{self.device}", dtype=torch.float32).contiguous()
        b, _, L, _ = latents.shape
        S = L * (2 ** (len(self.cfg.vae.block_out_channels) - 1))
        u8 = torch.empty(b, S, S, 3, device=latents.device, dtype=torch.uint8)
        f32 = torch.empty(b, S, S, 3, device=latents.device, dtype=torch.float32) if want_f32 else None
        self._ck(self.lib.agd_vae_decode(self.ctx, _lib.ptr(latents), b, L, _lib.ptr(u8), _lib.ptr(f32), self._stream()),
                 "agd_vae_decode")
        return (u8, f32) if want_f32 else u8

    def vae_encode(self, image: torch.Tensor):
        """`vae.encode(image).latent_dist` moments: image [B,3,S,S] in [-1,1] -> (mean, logvar) fp32 [B,4,S/8,S/8]."""
        image = image.to(device=f"cuda:{self.device}", dtype=torch.float32).contiguous()
        b, _, S, _ = image.shape
        L = S // (2 ** (len(self.cfg.vae.block_out_channels) - 1))
        mean = torch.empty(b, self.cfg.vae.latent_channels, L, L, device=image.device, dtype=torch.float32)
        logvar = torch.empty_like(mean)
        self._ck(self.lib.agd_vae_encode(self.ctx, _lib.ptr(image), b, S, _lib.ptr(mean), _lib.ptr(logvar), self._stream()), "agd_vae_encode")
        return mean, logvar.clamp_(-30.0, 20.0)

    def set_option(self, name: str, value: int):
        self._ck(self.lib.agd_set_option(self.ctx, name.encode(), int(value)), f"agd_set_option({name})")

    # recorder
    def record_config(self, mode: int, is_train: bool = False, rec_tokens: int = 0):
        self._ck(self.lib.agd_record_config(self.ctx, mode, int(is_train), rec_tokens), "agd_record_config")

    def record_reset(self, batch: int, L: int):
        self._ck(self.lib.agd_record_reset(self.ctx, batch, L, self._stream()), "agd_record_reset")

    def daam_global(self, img: int, rows: int, S: int) -> torch.Tensor:
        out = torch.empty(rows, S, S, device=f"cuda:{self.device}", dtype=torch.float32)
        rc = self.lib.agd_daam_global(self.ctx, img, rows, _lib.ptr(out), self._stream())
        if rc == -2:
            raise RuntimeError(self.lib.agd_last_error(self.ctx).decode())
        self._ck(rc, "agd_daam_global")
        return out

    def hook_global(self, bp: int, T: int, S: int) -> torch.Tensor:
        out = torch.empty(bp, T, S, S, device=f"cuda:{self.device}", dtype=torch.float32)
        rc = self.lib.agd_hook_global(self.ctx, _lib.ptr(out), self._stream())
        if rc == -2:
            raise RuntimeError("No heat maps found.")          # hook.py:74-77
        self._ck(rc, "agd_hook_global")
        return out

    def hook_last_map(self, bp: int, T: int, n_query: int) -> torch.Tensor:
        side = int(round(n_query ** 0.5))
        out = torch.empty(bp, T, side, side, device=f"cuda:{self.device}", dtype=torch.float32)
        self._ck(self.lib.agd_hook_last_map(self.ctx, _lib.ptr(out), n_query, self._stream()), "agd_hook_last_map")
        return out

    def hook_count(self) -> int:
        return int(self.lib.agd_hook_count(self.ctx))

    def cross_attn(self, layer: str, hidden: torch.Tensor, ctx_emb: Optional[torch.Tensor], record: bool) -> torch.Tensor:
        hidden = hidden.to(device=f"cuda:{self.device}", dtype=torch.float32).contiguous()
        b2, n, _ = hidden.shape
        t = self.cfg.max_tokens
        if ctx_emb is not None:
            ctx_emb = ctx_emb.to(device=hidden.device, dtype=torch.float32).contiguous()
            t = ctx_emb.shape[1]
        out = torch.empty_like(hidden)
        self._ck(self.lib.agd_cross_attn(self.ctx, layer.encode(), _lib.ptr(hidden), _lib.ptr(ctx_emb), b2, n, t,
                                         _lib.ptr(out), int(record), self._stream()), "agd_cross_attn")
        return out

    def attn_processor(self, layer: str, hidden: torch.Tensor, ctx_emb: Optional[torch.Tensor], mask: Optional[torch.Tensor],
                       record: bool) -> torch.Tensor:
        """One `Attention` module call (hook.py:83-122): attn2 with `ctx_emb`, attn1 with `ctx_emb=None`; additive mask [B2, keys]."""
        dev = f"cuda:{self.device}"
        hidden = hidden.to(device=dev, dtype=torch.float32).contiguous()
        b2, n, _ = hidden.shape
        t = 0
        if ctx_emb is not None:
            ctx_emb = ctx_emb.to(device=dev, dtype=torch.float32).contiguous()
            t = ctx_emb.shape[1]
        if mask is not None:
            mask = mask.to(device=dev, dtype=torch.float32).contiguous()
        out = torch.empty_like(hidden)
        self._ck(self.lib.agd_attn_processor(self.ctx, layer.encode(), _lib.ptr(hidden), _lib.ptr(ctx_emb), _lib.ptr(mask), b2, n, t,
                                             _lib.ptr(out), int(record), self._stream()), "agd_attn_processor")
        self._keep = (hidden, ctx_emb, mask)
        return out

    # training-mode seam
    def hook_reset(self, rows: int, L: int):
        self._ck(self.lib.agd_hook_reset(self.ctx, rows, L, self._stream()), "agd_hook_reset")

    def hook_num_maps(self) -> int:
        return int(self.lib.agd_hook_num_maps(self.ctx))

    def hook_map(self, k: int) -> torch.Tensor:
        """The k-th per-call map kept since the last reset (train mode): [B', T, h, w] fp32 (hook.py:110-112)."""
        d = (C.c_int * 3)()
        self._ck(self.lib.agd_hook_map_dims(self.ctx, k, d), "agd_hook_map_dims")
        side = int(round(d[2] ** 0.5))
        out = torch.empty(d[0], d[1], side, side, device=f"cuda:{self.device}", dtype=torch.float32)
        self._ck(self.lib.agd_hook_map(self.ctx, k, _lib.ptr(out), self._stream()), "agd_hook_map")
        return out

    def attn_processor_backward(self, layer: str, hidden: torch.Tensor, ctx_emb: Optional[torch.Tensor], d_out: Optional[torch.Tensor],
                                d_map: Optional[torch.Tensor], is_train: bool, want_hidden: bool = True, want_ctx: bool = True):
        """Backward of one cross-attention seam call: (d_hidden [B2,N,C], d_ctx [B2,T,ctx_dim]) from d_out and/or d_map."""
        dev = f"cuda:{self.device}"
        f = lambda t: None if t is None else t.detach().to(device=dev, dtype=torch.float32).contiguous()
        hidden, ctx_emb, d_out, d_map = f(hidden), f(ctx_emb), f(d_out), f(d_map)
        b2, n, _ = hidden.shape
        t = ctx_emb.shape[1] if ctx_emb is not None else self.cfg.max_tokens
        if d_map is not None:
            d_map = d_map.reshape(d_map.shape[0], d_map.shape[1], -1).contiguous()
            if d_map.shape[0] != (b2 if is_train else b2 // 2) or d_map.shape[2] != n:
                raise ValueError(f"d_map shape {tuple(d_map.shape)} does not match the recorded map of this call")
        dh = torch.empty_like(hidden) if want_hidden else None
        dc = torch.empty(b2, t, self.cfg.unet.cross_attention_dim, device=dev, dtype=torch.float32) if want_ctx else None
        self._ck(self.lib.agd_attn_processor_backward(self.ctx, layer.encode(), _lib.ptr(hidden), _lib.ptr(ctx_emb), _lib.ptr(d_out), _lib.ptr(d_map),
                                                      int(is_train), b2, n, t, _lib.ptr(dh), _lib.ptr(dc), self._stream()), "agd_attn_processor_backward")
        self._keep = (hidden, ctx_emb, d_out, d_map)
        return dh, dc

    def profile_begin(self):
        self._ck(self.lib.agd_profile_begin(self.ctx), "agd_profile_begin")

    def profile_end(self, mfma_peak_flops: float = 2.5e15, hbm_peak_bytes: float = 8.0e12):
        """Per kernel class: HIP-event ms, algorithmic flop / HBM bytes, launches, and `roof_ms` = the time the binding roof
        (max of flop / MFMA peak and bytes / HBM peak, per launch) allows."""
        n = _lib.AGD_N_CLASSES
        ms, fl, by, rf, rh = ((C.c_double * n)() for _ in range(5))
        ln = (C.c_longlong * n)()
        self._ck(self.lib.agd_profile_end_ex(self.ctx, mfma_peak_flops, hbm_peak_bytes, ms, fl, by, rf, rh, ln), "agd_profile_end_ex")
        return {self.lib.agd_profile_class_name(i).decode(): {"ms": ms[i], "flops": fl[i], "bytes": by[i], "roof_ms": rf[i],
                                                              "roof_ms_hbm_bound": rh[i], "launches": ln[i]} for i in range(n)}


class AttnHandle:
    """Stand-in for a diffusers `Attention` module of the UNet: identifies the layer for the seam."""

    def __init__(self, unet, name: str, heads: int, is_cross: bool):
        self.unet, self.name, self.heads, self.is_cross = unet, name, heads, is_cross
        self.norm_cross = None


class UNetHandle:
    """`pipeline.unet`: config + the attention-processor registry (`set_attn_processor`,
    `attn_processors`; reference finetune_sd_token.py:755-757 installs hook.py's hooker this way)."""

    def __init__(self, pipe):
        self._pipe = pipe
        self.config = pipe.cfg.unet
        self._default = "AttnProcessor(fused-hip)"
        names = cross_attn_layer_names(pipe.cfg.unet, include_mid=True)
        self._attn2 = {n: AttnHandle(self, n, 0, True) for n in names}
        self._attn1 = {n.replace("attn2", "attn1"): AttnHandle(self, n.replace("attn2", "attn1"), 0, False) for n in names}
        self._procs = {}
        for n in names:
            self._procs[n + ".processor"] = self._default
            self._procs[n.replace("attn2", "attn1") + ".processor"] = self._default

    @property
    def attn_processors(self):
        return dict(self._procs)

    def set_attn_processor(self, proc):
        from .hook import UNetCrossAttentionHooker
        if isinstance(proc, dict):
            procs = proc
        else:
            procs = {k: proc for k in self._procs}
        hookers = {id(p): p for p in procs.values() if isinstance(p, UNetCrossAttentionHooker)}
        if len(hookers) > 1:
            raise ValueError("only one UNetCrossAttentionHooker instance may be installed (it is shared by all layers)")
        self._procs.update(procs)
        hooker = next(iter(hookers.values()), None)
        self._pipe._install_hooker(hooker)

    def attn2(self, name: str) -> AttnHandle:
        return self._attn2[name]

    def attn1(self, name: str) -> AttnHandle:
        return self._attn1[name]

    def attn(self, name: str) -> AttnHandle:
        """The `Attention` module handle (attn1 = self-, attn2 = cross-attention) the processor is called with."""
        return self._attn2[name] if name in self._attn2 else self._attn1[name]

    def __call__(self, sample, timestep, encoder_hidden_states=None, class_labels=None, return_dict: bool = True, **unused):
        """`unet(sample, t, encoder_hidden_states)`: one fused forward; an installed hooker records its attn2 calls (train mode keeps
        the 16 per-call maps, hook.py:110-112).  finetune_sd_token.py:1027 calls it as
        `unet(noisy_latents, timesteps, encoder_hidden_states, class_labels=None, return_dict=False)[0]` with a per-sample
        [bsz] timestep tensor and without CFG: `timestep` may be a number or a [B] tensor, the result is `UNetOutput(sample=...)`
        (diffusers' UNet2DConditionOutput) or, with return_dict=False, the tuple `(sample,)`.  The maps recorded by this fused
        call are detached (no UNet backward here -- INTEGRATION.md); gradients flow only through direct seam calls."""
        if class_labels is not None:
            raise NotImplementedError("class-conditional UNets (class_labels) are not part of this path")
        if unused:
            raise TypeError(f"unsupported unet() arguments: {sorted(unused)}")
        out = self._forward(sample, timestep, encoder_hidden_states)
        return UNetOutput(sample=out) if return_dict else (out,)

    def _forward(self, sample, timestep, encoder_hidden_states=None):
        if encoder_hidden_states is not None:
            self._pipe.engine.set_context(encoder_hidden_states)
        hk = self._pipe._hooker
        if hk is not None and self._pipe._trace is None:
            hk._ensure(sample.shape[0] if hk.is_train else sample.shape[0] // 2, sample.shape[-1],
                       encoder_hidden_states.shape[1] if encoder_hidden_states is not None else self._pipe.cfg.max_tokens)
        out = self._pipe.engine.unet_forward(sample, timestep)
        if hk is not None:
            hk._cache = (-1, [])
        return out


class VAEHandle:
    def __init__(self, pipe):
        self._pipe = pipe
        self.config = pipe.cfg.vae

    def decode(self, z):
        """`vae.decode(z)`: z already divided by scaling_factor by the caller (diffusers convention)."""
        u8, f32 = self._pipe.engine.vae_decode(z * self.config.scaling_factor, want_f32=True)
        return f32.permute(0, 3, 1, 2)


class StableDiffusionPipeline:
    def __init__(self, cfg: SDConfig, unet_sd: Dict[str, torch.Tensor], vae_sd: Dict[str, torch.Tensor],
                 tokenizer=None, text_encoder=None, device: Union[int, str] = 0, workspace_bytes: int = 0,
                 text_sd: Optional[Dict[str, torch.Tensor]] = None, scheduler: str = "DDIMScheduler"):
        self.cfg = cfg
        dev = int(str(device).split(":")[-1]) if not isinstance(device, int) and ":" in str(device) else (device if isinstance(device, int) else 0)
        self.engine = Engine(cfg, dev, workspace_bytes)
        self.engine.load_state_dict(unet_sd, "unet.")
        self.engine.load_state_dict(vae_sd, "vae.")
        if text_sd is not None:
            if cfg.text is None:
                raise ValueError("text_sd given but cfg.text is None")
            self.engine.load_state_dict({k[len("text_model."):] if k.startswith("text_model.") else k: v
                                         for k, v in text_sd.items() if "position_ids" not in k}, "text.")
        self.engine.finalize()
        self.device = torch.device(f"cuda:{dev}")
        self.tokenizer = tokenizer or SimpleTokenizer(cfg.max_tokens)
        if text_encoder is None and text_sd is not None:
            from .text import HipCLIPTextEncoder
            text_encoder = HipCLIPTextEncoder(self.engine, self.tokenizer, text_sd)
        self.text_encoder = text_encoder or SyntheticTextEncoder(self.tokenizer, cfg.unet.cross_attention_dim)
        if scheduler not in SCHEDULERS:
            raise ValueError(f"scheduler '{scheduler}' is not implemented (have: {sorted(SCHEDULERS)})")
        self.scheduler = SCHEDULERS[scheduler].from_config(cfg.sched)
        self.unet = UNetHandle(self)
        self.vae = VAEHandle(self)
        self.vae_scale_factor = cfg.vae_scale_factor
        self._trace = None
        self._hooker = None
        self._last_prompt = None
        self._progress = {}
        self._source_path = None          # from_pretrained: the checkpoint directory (save_pretrained re-exports from it)
        # diffusers' `pipeline.safety_checker` slot: None (no checker weights ship with this repo) or a callable
        # images uint8 [B,H,W,3] (cuda tensor) -> sequence of B bools; flagged images are returned black, which the generation
        # driver then skips exactly as data_generation.py:61-62 does
        self.safety_checker = None

    # ---- construction -------------------------------------------------------------------
    @classmethod
    def from_synthetic(cls, cfg: Union[str, SDConfig] = "sd15", seed: int = 1234, device=0, workspace_bytes: int = 0,
                       weights_device: str = "cpu", keep_weights: bool = False, scheduler: str = "DDIMScheduler", **kw):
        from . import synthetic
        cfg = CONFIGS[cfg]() if isinstance(cfg, str) else cfg
        usd = synthetic.make_unet_weights(cfg, seed, device=weights_device, **kw)
        vsd = synthetic.make_vae_weights(cfg, seed + 1, device=weights_device, **kw)
        pipe = cls(cfg, usd, vsd, device=device, workspace_bytes=workspace_bytes, scheduler=scheduler)
        if keep_weights:
            pipe.synthetic_weights = (usd, vsd)
        return pipe

    @classmethod
    def from_pretrained(cls, path: str, device=0, workspace_bytes: int = 0, scheduler: Optional[str] = None):
        """Reads the diffusers on-disk layout (`unet/config.json`, `unet/diffusion_pytorch_model.safetensors`,
        `vae/...`) that `save_pretrained` writes (reference finetune_sd_token.py:164-187)."""
        from safetensors.torch import load_file

        def jload(p):
            with open(p) as f:
                return json.load(f)

        uc = jload(os.path.join(path, "unet", "config.json"))
        vc = jload(os.path.join(path, "vae", "config.json"))
        boc = tuple(uc["block_out_channels"])
        ahd = uc.get("attention_head_dim", 8)
        heads = tuple(ahd) if isinstance(ahd, (list, tuple)) else (ahd,) * len(boc)
        ucfg = UNetConfig(in_channels=uc.get("in_channels", 4), out_channels=uc.get("out_channels", 4), block_out_channels=boc,
                          down_cross=tuple("CrossAttn" in t for t in uc["down_block_types"]),
                          layers_per_block=uc.get("layers_per_block", 2), num_heads=heads,
                          cross_attention_dim=uc.get("cross_attention_dim", 768),
                          use_linear_projection=uc.get("use_linear_projection", False),
                          norm_num_groups=uc.get("norm_num_groups", 32))
        vcfg = VAEConfig(latent_channels=vc.get("latent_channels", 4), out_channels=vc.get("out_channels", 3),
                         block_out_channels=tuple(vc["block_out_channels"]), layers_per_block=vc.get("layers_per_block", 2),
                         norm_num_groups=vc.get("norm_num_groups", 32), scaling_factor=vc.get("scaling_factor", 0.18215))
        sc = SchedulerConfig()
        sched_name = scheduler or "DDIMScheduler"
        sp = os.path.join(path, "scheduler", "scheduler_config.json")
        if os.path.exists(sp):
            sj = jload(sp)
            if scheduler is None:           # the checkpoint's own scheduler, as `from_pretrained` of the reference gives it
                sched_name = sj.get("_class_name", "DDIMScheduler")
                if sched_name not in SCHEDULERS:
                    raise _lib.AgendaHipError(f"{sp}: scheduler '{sched_name}' is not implemented (have: {sorted(SCHEDULERS)}); "
                                              "pass from_pretrained(..., scheduler='DDIMScheduler') to override")
            sc = SchedulerConfig(sj.get("num_train_timesteps", 1000), sj.get("beta_start", 0.00085), sj.get("beta_end", 0.012),
                                 sj.get("steps_offset", 1), sj.get("set_alpha_to_one", False), sj.get("prediction_type", "epsilon"),
                                 # diffusers' PNDMScheduler defaults skip_prk_steps to False; SD checkpoints store True
                                 # (that default applies only when the JSON itself is a PNDMScheduler config: an explicit scheduler="PNDMScheduler"
                                 # override on a DDIM checkpoint has no such key and means the SD form, skip_prk_steps=True)
                                 bool(sj.get("skip_prk_steps", False)) if sj.get("_class_name") == "PNDMScheduler" else True)
        cfg = SDConfig(name=os.path.basename(path.rstrip("/")), unet=ucfg, vae=vcfg, sched=sc,
                       default_sample_size=uc.get("sample_size", 64))
        src_path = path

        def wload(sub):
            for fn in ("diffusion_pytorch_model.safetensors", "model.safetensors"):
                p = os.path.join(path, sub, fn)
                if os.path.exists(p):
                    return load_file(p)
            raise FileNotFoundError(f"no safetensors weights under {path}/{sub}")

        usd = wload("unet")
        # decoder side for txt2img, encoder side (`vae.encode`, img2img front end) as well
        vsd = {k: v for k, v in wload("vae").items()
               if k.startswith(("decoder.", "post_quant_conv.", "encoder.", "quant_conv."))}
        # pre-0.18 VAE attention naming
        ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
        vsd = {(".".join(ren.get(p, p) for p in k.split(".")) if ".attentions." in k else k): v for k, v in vsd.items()}
        # prompt side: text_encoder/ runs on the device (agd_text_encode); tokenizer/ via transformers when present
        tok = None
        tsd = None
        tcj = os.path.join(path, "text_encoder", "config.json")
        if os.path.exists(tcj):
            from .config import TextConfig
            tj = jload(tcj)
            cfg.text = TextConfig(hidden_size=tj.get("hidden_size", 768), num_hidden_layers=tj.get("num_hidden_layers", 12),
                                  num_attention_heads=tj.get("num_attention_heads", 12), intermediate_size=tj.get("intermediate_size", 3072),
                                  vocab_size=tj.get("vocab_size", 49408), max_position_embeddings=tj.get("max_position_embeddings", 77),
                                  hidden_act=tj.get("hidden_act", "quick_gelu"), layer_norm_eps=tj.get("layer_norm_eps", 1e-5))
            tsd = wload("text_encoder")
        if os.path.isdir(os.path.join(path, "tokenizer")):
            try:
                from transformers import CLIPTokenizer
                tok = CLIPTokenizer.from_pretrained(os.path.join(path, "tokenizer"))
            except Exception as e:
                if tsd is not None:        # real CLIP weights + a word-level stand-in tokenizer = garbage embeddings: refuse
                    raise _lib.AgendaHipError(f"text_encoder/ weights found but tokenizer/ could not be loaded ({e}); "
                                              "no silent fallback to the synthetic tokenizer") from e
                tok = None
        elif tsd is not None:
            raise _lib.AgendaHipError(f"{path}: text_encoder/ weights found but no tokenizer/ directory; "
                                      "no silent fallback to the synthetic tokenizer")
        pipe = cls(cfg, usd, vsd, tokenizer=tok, device=device, workspace_bytes=workspace_bytes, text_sd=tsd, scheduler=sched_name)
        pipe._source_path = src_path
        return pipe

    def save_pretrained(self, save_directory: str):
        """`pipeline.save_pretrained(dir)` (finetune_sd_token.py:164-187 writes its result this way): the diffusers layout
        `from_pretrained` reads.  This pipeline is an inference engine -- UNet / VAE / scheduler are exactly what was loaded, so
        their directories are re-exported from the source checkpoint; what CAN have changed is the prompt side
        (`tokenizer.add_tokens` + rows written into `text_encoder.get_input_embeddings().weight`, data_generation.py:45-52): the
        tokenizer is saved with its added tokens and the text encoder with its current (resized) embedding table."""
        import shutil
        from safetensors.torch import load_file, save_file
        if self._source_path is None:
            raise ValueError("save_pretrained: this pipeline was built from in-memory weights (no checkpoint directory to re-export)")
        src, dst = self._source_path, save_directory
        os.makedirs(dst, exist_ok=True)
        same = os.path.realpath(src) == os.path.realpath(dst)          # saving over the source: nothing to copy, only the rewritten files
        for sub in ("unet", "vae", "scheduler"):
            if not same and os.path.isdir(os.path.join(src, sub)):
                shutil.copytree(os.path.join(src, sub), os.path.join(dst, sub), dirs_exist_ok=True)
        sched_cls = type(self.scheduler).__name__
        sp = os.path.join(src, "scheduler", "scheduler_config.json")
        sj = {}
        if os.path.exists(sp):
            with open(sp) as f:
                sj = json.load(f)
        if sj.get("_class_name") != sched_cls:
            # the pipeline runs a different scheduler than the source directory names (from_pretrained(..., scheduler=...)): a reload
            # must give the scheduler this pipeline ran, so its config is written from the live objects
            sc = self.cfg.sched
            sj = {"_class_name": sched_cls, "num_train_timesteps": sc.num_train_timesteps, "beta_start": sc.beta_start, "beta_end": sc.beta_end,
                  "beta_schedule": "scaled_linear", "steps_offset": sc.steps_offset, "set_alpha_to_one": sc.set_alpha_to_one,
                  "prediction_type": sc.prediction_type}
            if sched_cls == "PNDMScheduler":
                sj["skip_prk_steps"] = bool(sc.skip_prk_steps)
            os.makedirs(os.path.join(dst, "scheduler"), exist_ok=True)
            with open(os.path.join(dst, "scheduler", "scheduler_config.json"), "w") as f:
                json.dump(sj, f, indent=2)
        mi = os.path.join(src, "model_index.json")
        mj = None
        if os.path.exists(mi):
            with open(mi) as f:
                mj = json.load(f)
        if mj is None:
            mj = {"_class_name": "StableDiffusionPipeline", "unet": ["diffusers", "UNet2DConditionModel"], "vae": ["diffusers", "AutoencoderKL"],
                  "text_encoder": ["transformers", "CLIPTextModel"], "tokenizer": ["transformers", "CLIPTokenizer"], "safety_checker": [None, None]}
        mj["scheduler"] = ["diffusers", sched_cls]
        with open(os.path.join(dst, "model_index.json"), "w") as f:
            json.dump(mj, f, indent=2)
        te = os.path.join(src, "text_encoder")
        if os.path.isdir(te):
            os.makedirs(os.path.join(dst, "text_encoder"), exist_ok=True)
            cands = ("model.safetensors", "diffusion_pytorch_model.safetensors")
            fn = next((f for f in cands if os.path.exists(os.path.join(te, f))), None)
            if fn is None:
                raise _lib.AgendaHipError(f"save_pretrained: no single-file safetensors text encoder under {te} (looked for {', '.join(cands)}; "
                                          "pytorch_model.bin and sharded checkpoints are not re-exported)")
            tsd = load_file(os.path.join(te, fn))
            key = next(k for k in tsd if k.endswith("embeddings.token_embedding.weight"))
            w = self.text_encoder.get_input_embeddings().weight
            tsd[key] = w.detach().to(tsd[key].dtype).contiguous().clone()
            save_file(tsd, os.path.join(dst, "text_encoder", fn))
            with open(os.path.join(te, "config.json")) as f:
                tj = json.load(f)
            tj["vocab_size"] = int(w.shape[0])
            with open(os.path.join(dst, "text_encoder", "config.json"), "w") as f:
                json.dump(tj, f, indent=2)
        if hasattr(self.tokenizer, "save_pretrained"):
            self.tokenizer.save_pretrained(os.path.join(dst, "tokenizer"))
        elif os.path.isdir(os.path.join(src, "tokenizer")):
            shutil.copytree(os.path.join(src, "tokenizer"), os.path.join(dst, "tokenizer"), dirs_exist_ok=True)


    def to(self, device):
        if "cuda" not in str(device):
            raise _lib.AgendaHipError("agenda_amd runs on the GPU only")
        return self

    def set_progress_bar_config(self, **kw):
        self._progress = kw

    # ---- recorder plumbing --------------------------------------------------------------
    def _install_hooker(self, hooker):
        self._hooker = hooker
        if hooker is not None:
            hooker._bind(self)
        self._apply_record_mode()

    def _apply_record_mode(self):
        if self._trace is not None:
            self.engine.record_config(1, False, self._trace.rec_tokens)
        elif self._hooker is not None:
            self.engine.record_config(2, self._hooker.is_train, 0)
        else:
            self.engine.record_config(0)

    # ---- prompt -------------------------------------------------------------------------
    def encode_prompt(self, prompts: List[str], negative: Optional[List[str]] = None) -> torch.Tensor:
        """[uncond x B, cond x B] context, the CFG batch order (hook.py:48-49)."""
        neg = negative if negative is not None else [""] * len(prompts)
        self._last_prompt = prompts[0]
        return torch.cat([self.text_encoder(neg), self.text_encoder(prompts)], 0)

    # ---- txt2img ------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str], None] = None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5, negative_prompt=None,
                 generator: Union[torch.Generator, Sequence[torch.Generator], None] = None, latents: Optional[torch.Tensor] = None,
                 prompt_embeds: Optional[torch.Tensor] = None, output_type: str = "pil", num_images_per_prompt: int = 1):
        side = self.cfg.default_sample_size * self.vae_scale_factor
        height, width = height or side, width or side
        if height != width or height % 64:
            raise ValueError("height == width, multiple of 64 required")
        L = height // self.vae_scale_factor
        if prompt_embeds is None:
            prompts = [prompt] if isinstance(prompt, str) else list(prompt)
            prompts = [p for p in prompts for _ in range(num_images_per_prompt)]
            negs = None if negative_prompt is None else ([negative_prompt] * len(prompts) if isinstance(negative_prompt, str) else list(negative_prompt))
            prompt_embeds = self.encode_prompt(prompts, negs)
        B = prompt_embeds.shape[0] // 2
        if latents is None:
            # data_generation.py:58 seeds `torch.Generator(device="cuda")`: accepted (torch's device Philox stream; whether it is
            # bit-identical to an NVIDIA run of the reference is not verifiable here).  CPU generators give host-reproducible latents.
            Cl = self.cfg.unet.in_channels
            if isinstance(generator, (list, tuple)):           # diffusers randn_tensor: one (1, C, L, L) draw per generator
                if len(generator) != B:
                    raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch size of {B}.")
                parts = [torch.randn(1, Cl, L, L, generator=g, device=g.device if g is not None else "cpu") for g in generator]
                if len({p_.device for p_ in parts}) > 1:
                    parts = [p_.cpu() for p_ in parts]
                latents = torch.cat(parts, 0)
            else:                                              # ONE (B, C, L, L) draw, kept on the generator's device (no host round trip)
                latents = torch.randn(B, Cl, L, L, generator=generator, device=generator.device if generator is not None else "cpu")
        expect = (B, self.cfg.unet.in_channels, L, L)
        if tuple(latents.shape) != expect:                 # diffusers prepare_latents raises the same way
            raise ValueError(f"Unexpected latents shape, got {tuple(latents.shape)}, expected {expect}")
        lat = self.engine._h2d(latents.to(torch.float32) * self.scheduler.init_noise_sigma).clone()
        self.engine.set_context(prompt_embeds)
        self._apply_record_mode()
        if self._trace is not None or self._hooker is not None:
            self.engine.record_reset(B, L)
            if self._trace is not None:
                self._trace._on_generate(B, L, self._last_prompt)
            if self._hooker is not None:
                self._hooker._on_generate(B, L, prompt_embeds.shape[1])
        self._denoise(lat, num_inference_steps, guidance_scale)
        if output_type == "latent":
            return PipelineOutput(images=[], latents=lat)
        return self._finish(lat, B, output_type)

    def _denoise(self, lat, num_inference_steps, guidance_scale):
        """`for t in scheduler.timesteps: unet -> CFG -> scheduler.step`, fused on the device, under the pipeline's scheduler."""
        ts = self.scheduler.set_timesteps(num_inference_steps)
        if isinstance(self.scheduler, PNDMScheduler):
            tsf, ca, cb = self.scheduler.plms_program()
            self.engine.denoise_plms(lat, tsf, ca, cb, guidance_scale)
        else:
            a_t, a_p = self.scheduler.step_coeffs()
            self.engine.denoise(lat, ts, a_t, a_p, guidance_scale)

    def _finish(self, lat, B, output_type):
        u8 = self.engine.vae_decode(lat)
        flags = [False] * B
        if self.safety_checker is not None:        # diffusers run_safety_checker: flagged images come back black
            flags = [bool(f) for f in self.safety_checker(u8)]
            if len(flags) != B:
                raise ValueError(f"safety_checker returned {len(flags)} flags for {B} images")
            if any(flags):
                u8 = u8.clone()
                u8[torch.tensor(flags, device=u8.device)] = 0
        if output_type == "pt":
            return PipelineOutput(images=u8, latents=lat, nsfw_content_detected=flags)
        arr = u8.cpu().numpy()
        if output_type == "np":
            return PipelineOutput(images=arr, latents=lat, nsfw_content_detected=flags)
        from PIL import Image
        return PipelineOutput(images=[Image.fromarray(a) for a in arr], latents=lat, nsfw_content_detected=flags)

    # ---- img2img (diffusers StableDiffusionImg2ImgPipeline semantics, parity-unpinned) ------
    @torch.no_grad()
    def img2img(self, prompt=None, image: torch.Tensor = None, strength: float = 0.8, num_inference_steps: int = 50,
                guidance_scale: float = 7.5, generator: Optional[torch.Generator] = None, prompt_embeds: Optional[torch.Tensor] = None,
                noise_enc: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None, output_type: str = "pil"):
        """image: float [B,3,S,S] in [-1,1] (or uint8 [B,S,S,3]).  Noise draws come from a CPU generator (or are passed
        explicitly) for the same host-reproducibility reason as the txt2img latents."""
        # img2img runs the strength-truncated DDIM schedule.  A checkpoint whose own scheduler is PNDM (SD-1.x) gets a DDIM scheduler
        # built from the same scheduler config for this call (the reference has no img2img call site; strength-truncated PLMS is not
        # implemented)
        sched = self.scheduler if isinstance(self.scheduler, DDIMScheduler) else DDIMScheduler.from_config(self.cfg.sched)
        if image.dtype == torch.uint8:
            image = image.permute(0, 3, 1, 2).float() / 127.5 - 1.0
        B, _, S, _ = image.shape
        if prompt_embeds is None:
            prompts = [prompt] * B if isinstance(prompt, str) else list(prompt)
            prompt_embeds = self.encode_prompt(prompts)
        L = S // self.vae_scale_factor
        shape = (B, self.cfg.unet.in_channels, L, L)
        if generator is not None and generator.device.type != "cpu":
            raise ValueError("use a CPU torch.Generator")
        noise_enc = noise_enc if noise_enc is not None else torch.randn(shape, generator=generator)
        noise = noise if noise is not None else torch.randn(shape, generator=generator)
        ts = sched.set_timesteps(num_inference_steps)
        a_t, a_p = sched.step_coeffs()
        init = min(int(num_inference_steps * strength), num_inference_steps)
        t0 = max(num_inference_steps - init, 0)
        ts, a_t, a_p = ts[t0:], a_t[t0:], a_p[t0:]
        mean, logvar = self.engine.vae_encode(image)
        x0 = (mean + torch.exp(0.5 * logvar) * noise_enc.to(mean.device)) * self.cfg.vae.scaling_factor
        a = float(sched.alphas_cumprod[int(ts[0])])
        lat = (a ** 0.5 * x0 + (1 - a) ** 0.5 * noise.to(mean.device)).contiguous()
        self.engine.set_context(prompt_embeds)
        self._apply_record_mode()
        if self._trace is not None or self._hooker is not None:
            self.engine.record_reset(B, L)
            if self._trace is not None:
                self._trace._on_generate(B, L, self._last_prompt)
            if self._hooker is not None:
                self._hooker._on_generate(B, L, prompt_embeds.shape[1])
        self.engine.denoise(lat, ts, a_t, a_p, guidance_scale)
        if output_type == "latent":
            return PipelineOutput(images=[], latents=lat)
        return self._finish(lat, B, output_type)
