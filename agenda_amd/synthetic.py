"""Seeded synthetic weights / inputs in real SD shapes (no checkpoints exist offline).

Recipe follows SURVEY.md §8(d) "Synthetic inputs": conv/linear ~ N(0, 1/fan_in), norm gamma=1
beta=0, biases 0, `torch.Generator('cpu')`.  Values are rounded to bf16-representable floats so
the fp32 CPU oracle and the bf16 HIP path see *identical* weight values (the comparison then
measures kernel arithmetic, not weight quantisation).
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from .config import SDConfig, unet_param_shapes, vae_decoder_param_shapes, vae_encoder_param_shapes, text_param_shapes


def _bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def _fill(shapes: Dict[str, tuple], gen: torch.Generator, gain: float, bias_std: float,
          perturb_norm: float) -> Dict[str, torch.Tensor]:
    sd = {}
    dev = gen.device
    for k, shp in shapes.items():
        is_norm = ("norm" in k.split(".")[-2]) or k.split(".")[-2] in ("norm", "group_norm")
        if k.endswith(".weight") and len(shp) >= 2:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            w = torch.randn(shp, generator=gen, device=dev) * (gain / math.sqrt(fan_in))
        elif k.endswith(".weight"):
            w = torch.ones(shp, device=dev)
            if is_norm and perturb_norm:
                w = w + perturb_norm * torch.randn(shp, generator=gen, device=dev)
        else:
            if is_norm:
                w = perturb_norm * torch.randn(shp, generator=gen, device=dev) if perturb_norm else torch.zeros(shp, device=dev)
            else:
                w = bias_std * torch.randn(shp, generator=gen, device=dev) if bias_std else torch.zeros(shp, device=dev)
        sd[k] = _bf16_round(w)
    return sd


def make_unet_weights(cfg: SDConfig, seed: int = 1234, gain: float = 1.0, bias_std: float = 0.0,
                      perturb_norm: float = 0.0, device: str = "cpu") -> Dict[str, torch.Tensor]:
    """`device="cuda"` draws with a CUDA generator (fast; different values than the CPU stream)."""
    g = torch.Generator(device).manual_seed(seed)
    return _fill(unet_param_shapes(cfg.unet), g, gain, bias_std, perturb_norm)


def make_vae_weights(cfg: SDConfig, seed: int = 4321, gain: float = 1.0, bias_std: float = 0.0,
                     perturb_norm: float = 0.0, device: str = "cpu", with_encoder: bool = False) -> Dict[str, torch.Tensor]:
    g = torch.Generator(device).manual_seed(seed)
    sd = _fill(vae_decoder_param_shapes(cfg.vae), g, gain, bias_std, perturb_norm)
    if with_encoder:            # drawn after the decoder so decoder weights do not depend on the flag
        sd.update(_fill(vae_encoder_param_shapes(cfg.vae), g, gain, bias_std, perturb_norm))
    return sd


def make_context(cfg: SDConfig, batch: int, seed: int = 7) -> torch.Tensor:
    """Text context stand-in `[2*batch, T, ctx_dim]` ordered [uncond..., cond...] (the CFG order
    the reference relies on, hook.py:48-49)."""
    g = torch.Generator("cpu").manual_seed(seed)
    c = torch.randn(2 * batch, cfg.max_tokens, cfg.unet.cross_attention_dim, generator=g)
    return _bf16_round(c)


def make_latents(cfg: SDConfig, seeds, latent_side: int) -> torch.Tensor:
    """`randn([1,4,L,L])` from a CPU generator per image seed (SURVEY.md §7 'RNG parity': the
    reference seeds a CUDA generator, data_generation.py:58, which a CPU oracle cannot
    reproduce, so initial latents are an explicit shared input)."""
    out = []
    for s in seeds:
        g = torch.Generator("cpu").manual_seed(int(s))
        out.append(torch.randn(1, cfg.unet.in_channels, latent_side, latent_side, generator=g))
    return torch.cat(out, 0)


def make_text_weights(cfg: SDConfig, seed: int = 99, device: str = "cpu") -> Dict[str, torch.Tensor]:
    """Random CLIP text-encoder weights (transformers key names): linears ~ N(0, 1/fan_in), embeddings ~ N(0, 0.02),
    LayerNorm gamma ~ 1, small biases; bf16-representable."""
    g = torch.Generator(device).manual_seed(seed)
    sd = {}
    for k, shp in text_param_shapes(cfg.text).items():
        if "embedding" in k:
            w = torch.randn(shp, generator=g, device=device) * 0.02
        elif k.endswith(".weight") and len(shp) == 2:
            w = torch.randn(shp, generator=g, device=device) / math.sqrt(shp[1])
        elif k.endswith(".weight"):
            w = 1.0 + 0.1 * torch.randn(shp, generator=g, device=device)
        else:
            w = 0.05 * torch.randn(shp, generator=g, device=device)
        sd[k] = _bf16_round(w)
    return sd
