"""Architecture configs for the Stable-Diffusion UNet / VAE decoder on the AGenDA generation path.

The reference never spells these out: it loads them from the checkpoint's ``unet/config.json`` /
``vae/config.json`` via ``StableDiffusionPipeline.from_pretrained`` (reference
data_generation/data_generation.py:30).  The values below restate the public SD-1.x / SD-2.1
configs [upstream-knowledge, SURVEY.md §8a/§8d] and give every parameter the diffusers
state-dict key so a real checkpoint maps 1:1 (`agd_load_tensor`).
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import Dict, List, Tuple


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    # True = CrossAttnDownBlock2D / CrossAttnUpBlock2D at that level
    down_cross: Tuple[bool, ...] = (True, True, True, False)
    layers_per_block: int = 2
    # diffusers' `attention_head_dim` is in fact the NUMBER of heads for SD-1.x/2.x
    num_heads: Tuple[int, ...] = (8, 8, 8, 8)
    cross_attention_dim: int = 768
    use_linear_projection: bool = False
    norm_num_groups: int = 32
    time_embed_dim_mult: int = 4

    @property
    def up_cross(self) -> Tuple[bool, ...]:
        return tuple(reversed(self.down_cross))


@dataclass
class VAEConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215


@dataclass
class SchedulerConfig:
    """DDIM as configured for SD [upstream-knowledge, SURVEY.md §8a row S1]."""
    num_train_timesteps: int = 1000
    beta_start: float = 0.00085
    beta_end: float = 0.012
    steps_offset: int = 1
    set_alpha_to_one: bool = False
    prediction_type: str = "epsilon"  # "v_prediction" for SD-2.1 768
    skip_prk_steps: bool = True       # PNDM only: SD checkpoints set it (pure PLMS); False (Runge-Kutta warm-up) is refused, not ignored


@dataclass
class TextConfig:
    """CLIP text encoder (`pipeline.text_encoder`; transformers CLIPTextConfig fields). Defaults = SD-1.x ViT-L/14."""
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    vocab_size: int = 49408
    max_position_embeddings: int = 77
    hidden_act: str = "quick_gelu"       # "gelu" for the OpenCLIP encoder of SD-2.x
    layer_norm_eps: float = 1e-5


@dataclass
class SDConfig:
    name: str = "sd15"
    unet: UNetConfig = field(default_factory=UNetConfig)
    vae: VAEConfig = field(default_factory=VAEConfig)
    sched: SchedulerConfig = field(default_factory=SchedulerConfig)
    max_tokens: int = 77
    vae_scale_factor: int = 8
    default_sample_size: int = 64
    text: "TextConfig | None" = None     # None: no device text encoder (synthetic / external embeddings)

    def to_dict(self):
        return asdict(self)


def sd15() -> SDConfig:
    return SDConfig(name="sd15")


def sd21() -> SDConfig:
    return SDConfig(
        name="sd21",
        unet=UNetConfig(num_heads=(5, 10, 20, 20), cross_attention_dim=1024, use_linear_projection=True),
        sched=SchedulerConfig(prediction_type="v_prediction"),
        default_sample_size=96,
    )


def tiny(cross_dim: int = 64) -> SDConfig:
    """Small config (same topology, 64-multiples of channels) for fast CPU/GPU plumbing tests."""
    return SDConfig(
        name="tiny",
        unet=UNetConfig(block_out_channels=(64, 128, 128, 128), num_heads=(2, 2, 2, 2),
                        cross_attention_dim=cross_dim),
        vae=VAEConfig(block_out_channels=(64, 64, 128, 128)),
        default_sample_size=16,
    )


def tiny40() -> SDConfig:
    """Small config with SD-1.x head dims 40/80 (C=320/640 at 8 heads) but only two levels."""
    return SDConfig(
        name="tiny40",
        unet=UNetConfig(block_out_channels=(320, 640), down_cross=(True, False), num_heads=(8, 8),
                        cross_attention_dim=128, layers_per_block=1),
        vae=VAEConfig(block_out_channels=(64, 128), layers_per_block=1),
        default_sample_size=16,
    )


def tiny21() -> SDConfig:
    """SD-2.1-style small config: linear proj_in/out, head dim 64, v-prediction (run it at latent 24 to get
    the ragged 768-px-like token counts 576 / 144 / 36 / 9)."""
    return SDConfig(
        name="tiny21",
        unet=UNetConfig(block_out_channels=(64, 128, 128, 128), num_heads=(1, 2, 2, 2), cross_attention_dim=128,
                        use_linear_projection=True),
        vae=VAEConfig(block_out_channels=(64, 64, 128, 128)),
        sched=SchedulerConfig(prediction_type="v_prediction"),
        default_sample_size=24,
    )


CONFIGS = {"sd15": sd15, "sd21": sd21, "tiny": tiny, "tiny40": tiny40, "tiny21": tiny21}


def text_param_shapes(t: TextConfig) -> Dict[str, tuple]:
    """transformers CLIPTextModel state-dict keys (without the `text_model.` prefix of transformers 4.x)."""
    H, I = t.hidden_size, t.intermediate_size
    p: Dict[str, tuple] = {"embeddings.token_embedding.weight": (t.vocab_size, H),
                           "embeddings.position_embedding.weight": (t.max_position_embeddings, H)}
    for l in range(t.num_hidden_layers):
        L = f"encoder.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            p[L + f"self_attn.{n}.weight"] = (H, H)
            p[L + f"self_attn.{n}.bias"] = (H,)
        for n in ("layer_norm1", "layer_norm2"):
            p[L + n + ".weight"] = (H,)
            p[L + n + ".bias"] = (H,)
        p[L + "mlp.fc1.weight"] = (I, H); p[L + "mlp.fc1.bias"] = (I,)
        p[L + "mlp.fc2.weight"] = (H, I); p[L + "mlp.fc2.bias"] = (H,)
    p["final_layer_norm.weight"] = (H,)
    p["final_layer_norm.bias"] = (H,)
    return p


# ------------------------------------------------------------------------------------------
# Parameter inventory: diffusers state-dict key -> shape
# ------------------------------------------------------------------------------------------
def _resnet(p: Dict[str, tuple], pre: str, cin: int, cout: int, temb: int | None):
    p[pre + "norm1.weight"] = (cin,)
    p[pre + "norm1.bias"] = (cin,)
    p[pre + "conv1.weight"] = (cout, cin, 3, 3)
    p[pre + "conv1.bias"] = (cout,)
    if temb:
        p[pre + "time_emb_proj.weight"] = (cout, temb)
        p[pre + "time_emb_proj.bias"] = (cout,)
    p[pre + "norm2.weight"] = (cout,)
    p[pre + "norm2.bias"] = (cout,)
    p[pre + "conv2.weight"] = (cout, cout, 3, 3)
    p[pre + "conv2.bias"] = (cout,)
    if cin != cout:
        p[pre + "conv_shortcut.weight"] = (cout, cin, 1, 1)
        p[pre + "conv_shortcut.bias"] = (cout,)


def _transformer(p: Dict[str, tuple], pre: str, c: int, ctx: int, linear_proj: bool):
    p[pre + "norm.weight"] = (c,)
    p[pre + "norm.bias"] = (c,)
    pw = (c, c) if linear_proj else (c, c, 1, 1)
    p[pre + "proj_in.weight"] = pw
    p[pre + "proj_in.bias"] = (c,)
    t = pre + "transformer_blocks.0."
    for n in ("norm1", "norm2", "norm3"):
        p[t + n + ".weight"] = (c,)
        p[t + n + ".bias"] = (c,)
    for a, kd in (("attn1", c), ("attn2", ctx)):
        p[t + a + ".to_q.weight"] = (c, c)
        p[t + a + ".to_k.weight"] = (c, kd)
        p[t + a + ".to_v.weight"] = (c, kd)
        p[t + a + ".to_out.0.weight"] = (c, c)
        p[t + a + ".to_out.0.bias"] = (c,)
    p[t + "ff.net.0.proj.weight"] = (8 * c, c)
    p[t + "ff.net.0.proj.bias"] = (8 * c,)
    p[t + "ff.net.2.weight"] = (c, 4 * c)
    p[t + "ff.net.2.bias"] = (c,)
    p[pre + "proj_out.weight"] = pw
    p[pre + "proj_out.bias"] = (c,)


def unet_param_shapes(cfg: UNetConfig) -> Dict[str, tuple]:
    p: Dict[str, tuple] = {}
    boc = cfg.block_out_channels
    temb = boc[0] * cfg.time_embed_dim_mult
    p["conv_in.weight"] = (boc[0], cfg.in_channels, 3, 3)
    p["conv_in.bias"] = (boc[0],)
    p["time_embedding.linear_1.weight"] = (temb, boc[0])
    p["time_embedding.linear_1.bias"] = (temb,)
    p["time_embedding.linear_2.weight"] = (temb, temb)
    p["time_embedding.linear_2.bias"] = (temb,)
    ch = boc[0]
    for i, co in enumerate(boc):
        for j in range(cfg.layers_per_block):
            _resnet(p, f"down_blocks.{i}.resnets.{j}.", ch, co, temb)
            ch = co
            if cfg.down_cross[i]:
                _transformer(p, f"down_blocks.{i}.attentions.{j}.", co, cfg.cross_attention_dim,
                             cfg.use_linear_projection)
        if i != len(boc) - 1:
            p[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (co, co, 3, 3)
            p[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (co,)
    mid = boc[-1]
    _resnet(p, "mid_block.resnets.0.", mid, mid, temb)
    _transformer(p, "mid_block.attentions.0.", mid, cfg.cross_attention_dim, cfg.use_linear_projection)
    _resnet(p, "mid_block.resnets.1.", mid, mid, temb)
    rev = list(reversed(boc))
    ch = rev[0]
    for i, co in enumerate(rev):
        prev_out = ch
        in_ch = rev[min(i + 1, len(rev) - 1)]
        for j in range(cfg.layers_per_block + 1):
            skip = in_ch if j == cfg.layers_per_block else co
            rin = prev_out if j == 0 else co
            _resnet(p, f"up_blocks.{i}.resnets.{j}.", rin + skip, co, temb)
            if cfg.up_cross[i]:
                _transformer(p, f"up_blocks.{i}.attentions.{j}.", co, cfg.cross_attention_dim,
                             cfg.use_linear_projection)
        ch = co
        if i != len(rev) - 1:
            p[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (co, co, 3, 3)
            p[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (co,)
    p["conv_norm_out.weight"] = (boc[0],)
    p["conv_norm_out.bias"] = (boc[0],)
    p["conv_out.weight"] = (cfg.out_channels, boc[0], 3, 3)
    p["conv_out.bias"] = (cfg.out_channels,)
    return p


def vae_decoder_param_shapes(cfg: VAEConfig) -> Dict[str, tuple]:
    """Keys of `AutoencoderKL` that `decode()` touches (post_quant_conv + decoder.*), with the
    diffusers>=0.18 attention naming (`to_q/to_k/to_v/to_out.0/group_norm`)."""
    p: Dict[str, tuple] = {}
    lc = cfg.latent_channels
    p["post_quant_conv.weight"] = (lc, lc, 1, 1)
    p["post_quant_conv.bias"] = (lc,)
    rev = list(reversed(cfg.block_out_channels))
    top = rev[0]
    p["decoder.conv_in.weight"] = (top, lc, 3, 3)
    p["decoder.conv_in.bias"] = (top,)
    _resnet(p, "decoder.mid_block.resnets.0.", top, top, None)
    a = "decoder.mid_block.attentions.0."
    p[a + "group_norm.weight"] = (top,)
    p[a + "group_norm.bias"] = (top,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        p[a + n + ".weight"] = (top, top)
        p[a + n + ".bias"] = (top,)
    _resnet(p, "decoder.mid_block.resnets.1.", top, top, None)
    ch = top
    for i, co in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            _resnet(p, f"decoder.up_blocks.{i}.resnets.{j}.", ch, co, None)
            ch = co
        if i != len(rev) - 1:
            p[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (co, co, 3, 3)
            p[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (co,)
    p["decoder.conv_norm_out.weight"] = (ch,)
    p["decoder.conv_norm_out.bias"] = (ch,)
    p["decoder.conv_out.weight"] = (cfg.out_channels, ch, 3, 3)
    p["decoder.conv_out.bias"] = (cfg.out_channels,)
    return p


def vae_encoder_param_shapes(cfg: VAEConfig) -> Dict[str, tuple]:
    """Keys of `AutoencoderKL` that `encode()` touches (encoder.* + quant_conv) -- the img2img front end."""
    p: Dict[str, tuple] = {}
    lc, boc = cfg.latent_channels, cfg.block_out_channels
    p["encoder.conv_in.weight"] = (boc[0], cfg.out_channels, 3, 3)
    p["encoder.conv_in.bias"] = (boc[0],)
    ch = boc[0]
    for i, co in enumerate(boc):
        for j in range(cfg.layers_per_block):
            _resnet(p, f"encoder.down_blocks.{i}.resnets.{j}.", ch, co, None)
            ch = co
        if i != len(boc) - 1:
            p[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (co, co, 3, 3)
            p[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (co,)
    _resnet(p, "encoder.mid_block.resnets.0.", ch, ch, None)
    a = "encoder.mid_block.attentions.0."
    p[a + "group_norm.weight"] = (ch,)
    p[a + "group_norm.bias"] = (ch,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        p[a + n + ".weight"] = (ch, ch)
        p[a + n + ".bias"] = (ch,)
    _resnet(p, "encoder.mid_block.resnets.1.", ch, ch, None)
    p["encoder.conv_norm_out.weight"] = (ch,)
    p["encoder.conv_norm_out.bias"] = (ch,)
    p["encoder.conv_out.weight"] = (2 * lc, ch, 3, 3)
    p["encoder.conv_out.bias"] = (2 * lc,)
    p["quant_conv.weight"] = (2 * lc, 2 * lc, 1, 1)
    p["quant_conv.bias"] = (2 * lc,)
    return p


def cross_attn_layer_names(cfg: UNetConfig, include_mid: bool = True) -> List[str]:
    """Module paths of every `attn2`, in daam's locator order (up blocks, then down blocks, then
    mid) [upstream-knowledge, SURVEY.md §8a row D1]."""
    names = []
    n = len(cfg.block_out_channels)
    for i in range(n):
        if cfg.up_cross[i]:
            for j in range(cfg.layers_per_block + 1):
                names.append(f"up_blocks.{i}.attentions.{j}.transformer_blocks.0.attn2")
    for i in range(n):
        if cfg.down_cross[i]:
            for j in range(cfg.layers_per_block):
                names.append(f"down_blocks.{i}.attentions.{j}.transformer_blocks.0.attn2")
    if include_mid:
        names.append("mid_block.attentions.0.transformer_blocks.0.attn2")
    return names
