"""CPU fp32 ORACLE for the AGenDA generation hot path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module -- as the checker, never as the product.  The product path (`agenda_amd/`) never imports
it and fails loudly when the HIP library is missing.

Pinning status
--------------
* PINNED against the reference's own code: `hooker_unravel_attn`, `hooker_global_heat_map`,
  `explicit_attention_processor` -- checked in tests/test_oracle_golden.py against golden vectors
  produced by running /root/reference/data_generation/hook.py itself
  (tests/golden/make_golden_hook.py).
* PARITY UNPINNED: UNet / VAE / DDIM / daam aggregation.  Their arithmetic lives in un-vendored
  third-party packages (`diffusers==0.21.2`, `daam` unpinned; reference requirements.txt:4-5)
  that are absent here and for which the reference holds no tests or fixtures.
  They are restated from the published algorithms [upstream-knowledge] and anchored on the
  reference's call sites (data_generation/data_generation.py:57-86).

All functions are plain PyTorch fp32 on CPU; weights are passed as a diffusers-keyed state dict.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ==========================================================================================
# 1. Attention-processor seam  (reference data_generation/hook.py:83-122)
# ==========================================================================================
def head_to_batch_dim(t: Tensor, heads: int) -> Tensor:
    """diffusers `Attention.head_to_batch_dim` as used at hook.py:104-106: [B,N,C]->[B*H,N,C/H],
    head-minor on the batch axis (index = b*H + h)."""
    b, n, c = t.shape
    return t.reshape(b, n, heads, c // heads).permute(0, 2, 1, 3).reshape(b * heads, n, c // heads)


def batch_to_head_dim(t: Tensor, heads: int) -> Tensor:
    """hook.py:115."""
    bh, n, d = t.shape
    return t.reshape(bh // heads, heads, n, d).permute(0, 2, 1, 3).reshape(bh // heads, n, d * heads)


def attention_scores(q: Tensor, k: Tensor, scale: float, mask: Optional[Tensor] = None) -> Tensor:
    """`attn.get_attention_scores` (hook.py:108): softmax(scale * q k^T [+ mask]) over keys, fp32.
    `mask`: the additive mask `prepare_attention_mask` returns (hook.py:92), [B*H, 1, keys] [upstream-knowledge]."""
    s = scale * torch.bmm(q, k.transpose(1, 2))
    if mask is not None:
        s = s + mask
    return torch.softmax(s, dim=-1)


def explicit_attention_processor(x: Tensor, ctx: Optional[Tensor], wq: Tensor, wk: Tensor, wv: Tensor,
                                 wo: Tensor, bo: Optional[Tensor], heads: int,
                                 recorder: Optional[Callable[[Tensor, int], None]] = None,
                                 bq=None, bk=None, bv=None, attention_mask: Optional[Tensor] = None) -> Tensor:
    """Body of `UNetCrossAttentionHooker.__call__` (hook.py:91-120): explicit-softmax attention,
    probabilities handed to `recorder` only for cross-attention (hook.py:110-112)."""
    is_cross = ctx is not None                                   # hook.py:95
    kv_src = ctx if is_cross else x                              # hook.py:96-99 (norm_cross is None for SD)
    q = F.linear(x, wq, bq)                                      # hook.py:93
    k = F.linear(kv_src, wk, bk)                                 # hook.py:101
    v = F.linear(kv_src, wv, bv)                                 # hook.py:102
    d = q.shape[-1] // heads
    q, k, v = (head_to_batch_dim(t, heads) for t in (q, k, v))   # hook.py:104-106
    if attention_mask is not None:                               # hook.py:92: [B, 1, keys] repeated per head (head-minor)
        attention_mask = attention_mask.reshape(attention_mask.shape[0], 1, -1).repeat_interleave(heads, dim=0)
    p = attention_scores(q, k, d ** -0.5, attention_mask)        # hook.py:108
    if is_cross and recorder is not None:
        recorder(p, heads)                                       # hook.py:110-112
    o = batch_to_head_dim(torch.bmm(p, v), heads)                # hook.py:114-115
    return F.linear(o, wo, bo)                                   # hook.py:118-120 (dropout p=0)


def hooker_unravel_attn(p: Tensor, n_heads: int, is_train: bool) -> Tensor:
    """`_unravel_attn` (hook.py:28-56): [B*H, h*w, T] -> [B' , T, h, w], mean over heads, where
    inference mode keeps only the second (conditional) half of the B*H axis (hook.py:48-49)."""
    bh, n, t = p.shape
    h = w = int(math.sqrt(n))                                    # hook.py:43
    m = p.permute(2, 0, 1).reshape(t, bh, h, w)                  # hook.py:44-47
    if not is_train:
        m = m[:, bh // 2:]                                       # hook.py:48-49
    m = m.permute(1, 0, 2, 3)                                    # hook.py:53
    m = m.reshape(m.shape[0] // n_heads, n_heads, *m.shape[1:])  # hook.py:54
    return m.mean(dim=1)                                         # hook.py:55


def hooker_global_heat_map(maps: Sequence[Tensor], latent_side: int) -> Tensor:
    """`compute_global_heat_map` (hook.py:59-81): bicubic to (side, side), clamp>=0, mean over
    every recorded call.  `latent_hw` there is a SIDE length (hook.py:18,72)."""
    if len(maps) == 0:
        raise RuntimeError("No heat maps found.")                # hook.py:74-77
    ups = [F.interpolate(m.float(), size=(latent_side, latent_side), mode="bicubic").clamp_(min=0)
           for m in maps]                                        # hook.py:71-72
    return torch.stack(ups, 0).mean(0)                           # hook.py:75,79


class HookRecorder:
    """State of `UNetCrossAttentionHooker` (hook.py:15-26)."""

    def __init__(self, is_train: bool = True, latent_hw: int = 64):
        self.cross_attn_maps: List[Tensor] = []
        self.is_train = is_train
        self.latent_hw = latent_hw

    def clear(self):
        self.cross_attn_maps.clear()

    def __call__(self, p: Tensor, heads: int, layer: str = ""):
        self.cross_attn_maps.append(hooker_unravel_attn(p, heads, self.is_train))

    def compute_global_heat_map(self) -> Tensor:
        return hooker_global_heat_map(self.cross_attn_maps, self.latent_hw)


def attention_regulariser(cross_attn_maps: Sequence[Tensor], new_tokens_start_indices: Tensor, n_object_embedding: int,
                          reg_weight: float):
    """The cross-attention loss of the token fine-tuning step, restated from reference
    data_generation/finetune_sd_token.py:1040-1069 (inline in its training loop, so it cannot be imported; torch ops and their
    order kept, so autograd through the maps gives the reference's gradients).  `cross_attn_maps`: the hooker's list of
    [B, T, h, w] maps (is_train=True); `new_tokens_start_indices`: int [B, n_new_tokens], -1 = absent (dataset.py:88-97).
    Returns (attn_loss, bg_attn_loss, fg_attn_loss)."""
    dev = cross_attn_maps[0].device
    attn_loss = torch.tensor(0.0, device=dev)                                      # :1041-1043
    bg_attn_loss = torch.tensor(0.0, device=dev)
    fg_attn_loss = torch.tensor(0.0, device=dev)
    idx = new_tokens_start_indices
    for attn_maps in cross_attn_maps:                                              # :1046
        for sample_attn_map, sample_start_indices in zip(attn_maps, idx):          # :1047
            if sample_start_indices[0] > 0:                                        # :1048
                obj_index = sample_start_indices[0] + n_object_embedding           # :1049
                obj = sample_attn_map[obj_index]
                norm_obj = (obj - obj.min()) / (obj.max() - obj.min() + 1e-8)      # :1051
                norm_bg_ref = 1 - norm_obj                                         # :1052
                norm_bg_ref = norm_bg_ref / torch.sum(norm_bg_ref)                 # :1053
                norm_obj = norm_obj / torch.sum(norm_obj)                          # :1054
                fg = sample_attn_map[sample_start_indices[0]]                      # :1056
                norm_fg = (fg - fg.min()) / (fg.max() - fg.min() + 1e-8)           # :1057
                norm_fg = norm_fg / torch.sum(norm_fg)                             # :1058
                bg_index = sample_start_indices[sample_start_indices > -1][-1]     # :1060
                bg = sample_attn_map[bg_index]
                norm_bg = (bg - bg.min()) / (bg.max() - bg.min() + 1e-8)           # :1062
                norm_bg = norm_bg / torch.sum(norm_bg)                             # :1063
                n_obj = torch.sum(idx[:, 0] > 0)
                bg_attn_loss = bg_attn_loss + reg_weight * torch.mean(torch.abs(norm_bg_ref - norm_bg)) / n_obj       # :1065
                fg_attn_loss = fg_attn_loss + reg_weight * torch.mean(torch.abs(norm_obj - norm_fg)) / n_obj          # :1066
                attn_loss = bg_attn_loss + fg_attn_loss                            # :1067
    attn_loss = attn_loss / len(cross_attn_maps)                                   # :1069
    return attn_loss, bg_attn_loss, fg_attn_loss


# ==========================================================================================
# 2. DAAM accumulation (third-party `daam`, call sites data_generation.py:57,64,74)  [upstream-knowledge]
# ==========================================================================================
class DaamRecorder:
    """Per-(layer, head) time-summed heat maps, one set per image.

    daam semantics: its locator hooks the attn2 of up/down blocks only (mid block never hooked,
    `locate_middle_block=False`); `factor = sqrt(latent_area / N)`; a hooked call is recorded iff
    keys == 77 context tokens and `factor != 8`; the conditional half of the B*H axis is
    reshaped to [H, T, h, w] and added into `acc[(factor, layer, head)]`.  Upstream assumes one
    image per call; a batch here is treated as independent images (image-major on the B axis),
    i.e. exactly what running daam once per image would give.
    """

    def __init__(self, latent_area: int, context_size: int = 77):
        self.latent_area = latent_area
        self.context_size = context_size
        self.acc: Dict[Tuple[int, str, int], Tensor] = {}   # (factor, layer, head) -> [B', T, h, w]

    def __call__(self, p: Tensor, heads: int, layer: str = ""):
        bh, n, t = p.shape
        if "mid_block" in layer:                               # locator: mid block is not hooked
            return
        factor = int(math.sqrt(self.latent_area // n))
        if t != self.context_size or factor == 8:
            return
        h = w = int(math.sqrt(n))
        cond = p[bh // 2:]                                     # conditional half
        b = cond.shape[0] // heads
        m = cond.reshape(b, heads, n, t).permute(1, 0, 3, 2).reshape(heads, b, t, h, w)
        for hd in range(heads):
            key = (factor, layer, hd)
            self.acc[key] = self.acc.get(key, 0) + m[hd]

    def compute_global_heat_map(self, n_rows: Optional[int] = None) -> Tensor:
        """`trace.compute_global_heat_map()` (data_generation.py:64): every accumulator ->
        bicubic to sqrt(latent_area) -> clamp>=0 -> mean over all (layer, head) -> first
        `len(tokenize(prompt)) + 2` rows.  Returns [B', T', S, S]."""
        if not self.acc:
            raise RuntimeError("No heat maps found. Did you forget to call `with trace(...)`?")
        s = int(math.sqrt(self.latent_area))
        ups = [F.interpolate(m, size=(s, s), mode="bicubic").clamp_(min=0) for m in self.acc.values()]
        g = torch.stack(ups, 0).mean(0)
        return g if n_rows is None else g[:, :n_rows]


def compute_token_merge_indices(tokenize: Callable[[str], List[str]], prompt: str, word: str,
                                word_idx: Optional[int] = None, offset_idx: int = 0):
    """`daam.utils.compute_token_merge_indices` (call sites data_generation.py:74 via
    `compute_word_heat_map`, dataset.py:93) [upstream-knowledge]."""
    merge_idxs: List[int] = []
    tokens = [x.replace("</w>", "") for x in tokenize(prompt.lower())]
    if word_idx is None:
        word = word.lower()
        search = [x.replace("</w>", "") for x in tokenize(word)]
        starts = [x + offset_idx for x in range(len(tokens)) if tokens[x:x + len(search)] == search]
        for s in starts:
            merge_idxs += [i + s for i in range(len(search))]
        if not merge_idxs:
            raise ValueError(f"Search word {word} not found in prompt!")
    else:
        merge_idxs.append(word_idx)
    return [x + 1 for x in merge_idxs], word_idx                # +1: SOS token


def word_heat_map(global_map: Tensor, merge_idxs: Sequence[int]) -> Tensor:
    """`GlobalHeatMap.compute_word_heat_map(word).heatmap`: mean of the word's token rows."""
    return global_map[..., list(merge_idxs), :, :].mean(-3)


# ==========================================================================================
# 3. Driver-side export semantics (reference data_generation.py:76-86, postprocess_heatmap.py:36-50)
# ==========================================================================================
def export_heatmap_u8(hm: np.ndarray) -> np.ndarray:
    """data_generation.py:82-84: min-max with +1e-8, x255, TRUNCATING uint8 cast."""
    hm = np.asarray(hm, dtype=np.float32)
    hm = (hm - hm.min()) / (hm.max() - hm.min() + 1e-8) * 255
    return hm.astype(np.uint8)


def stack_heatmaps(obj: np.ndarray, fg: np.ndarray, bg: np.ndarray):
    """postprocess_heatmap.py:44-48: RGB = [obj, fg, 255 - bg]; also returns the inverted bg."""
    inv = 255 - bg
    return np.stack([obj, fg, inv], axis=-1), inv


def select_learned_tokens(prompt_template: str, initialize_token: Sequence[str], learned: Sequence[str],
                          word_token_heatmaps: Optional[List[str]], store_learnable: bool):
    """data_generation.py:36-43,54: a learned token is used iff its init word is a substring of
    the *unformatted* template; the heat-map word list aliases the CLI list and grows in place."""
    words = word_token_heatmaps if word_token_heatmaps is not None else []
    new_tokens = []
    for t, n in zip(initialize_token, learned):
        if t in prompt_template:
            if store_learnable:
                words.append(n)
            new_tokens.append(n)
    return new_tokens, words, prompt_template.format(*new_tokens)


def postprocess_image(x: Tensor) -> np.ndarray:
    """diffusers image post-process [upstream-knowledge, SURVEY §8a V2]: (x/2+0.5).clamp(0,1),
    NHWC, round(255 x) -> uint8."""
    x = (x / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).float().numpy()
    return (x * 255).round().astype("uint8")


# ==========================================================================================
# 4. DDIM scheduler  [upstream-knowledge]
# ==========================================================================================
class DDIM:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1,
                 set_alpha_to_one=False, prediction_type="epsilon"):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.T = num_train_timesteps
        self.steps_offset = steps_offset
        self.prediction_type = prediction_type
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n: int):
        ratio = self.T // n                                     # "leading" spacing
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.num_inference_steps = n
        self.timesteps = ts
        return ts

    def coeffs(self, t: int):
        prev_t = t - self.T // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return float(a_t), float(a_p)

    def step(self, eps: Tensor, t: int, x: Tensor) -> Tensor:
        """eta = 0, clip_sample False."""
        a_t, a_p = self.coeffs(t)
        if self.prediction_type == "epsilon":
            x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
            e = eps
        elif self.prediction_type == "v_prediction":
            x0 = a_t ** 0.5 * x - (1 - a_t) ** 0.5 * eps
            e = a_t ** 0.5 * eps + (1 - a_t) ** 0.5 * x
        else:
            raise ValueError(self.prediction_type)
        return a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * e


class PNDM:
    """diffusers 0.21.2 `PNDMScheduler` with `skip_prk_steps=True` (pure PLMS), the scheduler the reference's own call
    `pipeline(prompt, num_inference_steps=20)` (data_generation.py:59) runs for an SD-1.4 checkpoint [upstream-knowledge].
    Restated method by method (`set_timesteps`, `step_plms`, `_get_prev_sample`) with the same state variables."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1, set_alpha_to_one=False):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.T, self.steps_offset, self.init_noise_sigma = num_train_timesteps, steps_offset, 1.0

    def set_timesteps(self, n: int):
        self.num_inference_steps = n
        ratio = self.T // n
        t = (np.arange(0, n) * ratio).round().astype(np.int64) + self.steps_offset
        self.timesteps = np.concatenate([t[:-1], t[-2:-1], t[-1:]])[::-1].copy()      # prk_timesteps empty (skip_prk_steps)
        self.ets, self.counter, self.cur_sample = [], 0, None
        return self.timesteps

    def _get_prev_sample(self, sample, timestep, prev_timestep, model_output):
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t, b_p = 1 - a_t, 1 - a_p
        sample_coeff = (a_p / a_t) ** 0.5
        denom = a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5
        return sample_coeff * sample - (a_p - a_t) * model_output / denom

    def step(self, model_output: Tensor, timestep: int, sample: Tensor) -> Tensor:
        """`step_plms`."""
        prev_timestep = timestep - self.T // self.num_inference_steps
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_timestep = timestep
            timestep = timestep + self.T // self.num_inference_steps
        if len(self.ets) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            model_output = (model_output + self.ets[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            model_output = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            model_output = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            model_output = (1 / 24) * (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4])
        prev = self._get_prev_sample(sample, timestep, prev_timestep, model_output)
        self.counter += 1
        return prev


# ==========================================================================================
# 5. UNet2DConditionModel forward (diffusers 0.21.2 semantics)  [upstream-knowledge]
# ==========================================================================================
def timestep_embedding(t: Tensor, dim: int) -> Tensor:
    """`Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)`."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = t.float()[:, None] * torch.exp(exponent)[None]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def _gn(x, sd, pre, groups, eps):
    return F.group_norm(x, groups, sd[pre + ".weight"], sd[pre + ".bias"], eps)


def resnet_block(x: Tensor, temb: Optional[Tensor], sd, pre: str, groups: int, eps: float) -> Tensor:
    h = F.silu(_gn(x, sd, pre + "norm1", groups, eps))
    h = F.conv2d(h, sd[pre + "conv1.weight"], sd[pre + "conv1.bias"], padding=1)
    if temb is not None and (pre + "time_emb_proj.weight") in sd:
        h = h + F.linear(F.silu(temb), sd[pre + "time_emb_proj.weight"], sd[pre + "time_emb_proj.bias"])[:, :, None, None]
    h = F.silu(_gn(h, sd, pre + "norm2", groups, eps))
    h = F.conv2d(h, sd[pre + "conv2.weight"], sd[pre + "conv2.bias"], padding=1)
    if (pre + "conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[pre + "conv_shortcut.weight"], sd[pre + "conv_shortcut.bias"])
    return x + h


def transformer_2d(x: Tensor, ctx: Tensor, sd, pre: str, heads: int, groups: int, linear_proj: bool,
                   recorder=None, layer_name: str = "", intermediates: Optional[dict] = None) -> Tensor:
    b, c, hh, ww = x.shape
    res = x
    h = _gn(x, sd, pre + "norm", groups, 1e-6)
    if not linear_proj:
        h = F.conv2d(h, sd[pre + "proj_in.weight"], sd[pre + "proj_in.bias"])
        h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    else:
        h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
        h = F.linear(h, sd[pre + "proj_in.weight"], sd[pre + "proj_in.bias"])
    t = pre + "transformer_blocks.0."
    # BasicTransformerBlock: self-attn, cross-attn, GEGLU feed-forward, each pre-LN + residual
    n1 = F.layer_norm(h, (c,), sd[t + "norm1.weight"], sd[t + "norm1.bias"], 1e-5)
    h = h + explicit_attention_processor(n1, None, sd[t + "attn1.to_q.weight"], sd[t + "attn1.to_k.weight"],
                                         sd[t + "attn1.to_v.weight"], sd[t + "attn1.to_out.0.weight"],
                                         sd[t + "attn1.to_out.0.bias"], heads)
    n2 = F.layer_norm(h, (c,), sd[t + "norm2.weight"], sd[t + "norm2.bias"], 1e-5)
    rec = (lambda p, nh: recorder(p, nh, layer_name)) if recorder is not None else None
    h = h + explicit_attention_processor(n2, ctx, sd[t + "attn2.to_q.weight"], sd[t + "attn2.to_k.weight"],
                                         sd[t + "attn2.to_v.weight"], sd[t + "attn2.to_out.0.weight"],
                                         sd[t + "attn2.to_out.0.bias"], heads, recorder=rec)
    n3 = F.layer_norm(h, (c,), sd[t + "norm3.weight"], sd[t + "norm3.bias"], 1e-5)
    ff = F.linear(n3, sd[t + "ff.net.0.proj.weight"], sd[t + "ff.net.0.proj.bias"])
    val, gate = ff.chunk(2, dim=-1)                              # GEGLU: hidden * gelu(gate), exact erf gelu
    h = h + F.linear(val * F.gelu(gate), sd[t + "ff.net.2.weight"], sd[t + "ff.net.2.bias"])
    if not linear_proj:
        h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
        h = F.conv2d(h, sd[pre + "proj_out.weight"], sd[pre + "proj_out.bias"])
    else:
        h = F.linear(h, sd[pre + "proj_out.weight"], sd[pre + "proj_out.bias"])
        h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
    return h + res


def unet_forward(sd: Dict[str, Tensor], ucfg, x: Tensor, t: Tensor, ctx: Tensor, recorder=None,
                 taps: Optional[dict] = None) -> Tensor:
    """`unet(latent_model_input, t, encoder_hidden_states=ctx).sample`.  x [B,4,L,L] NCHW fp32,
    t [B] (or scalar), ctx [B,T,ctx_dim].  `recorder(p, heads, layer_name)` receives every
    cross-attention probability tensor (the processor seam).  `taps` (optional dict) collects
    named intermediates for per-layer parity tests."""
    boc = ucfg.block_out_channels
    g = ucfg.norm_num_groups
    if t.ndim == 0:
        t = t[None].expand(x.shape[0])
    temb = timestep_embedding(t, boc[0])
    temb = F.linear(temb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    temb = F.linear(F.silu(temb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    h = F.conv2d(x, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    if taps is not None:
        taps["temb"] = temb
        taps["conv_in"] = h
    skips = [h]
    nlev = len(boc)
    for i in range(nlev):
        for j in range(ucfg.layers_per_block):
            h = resnet_block(h, temb, sd, f"down_blocks.{i}.resnets.{j}.", g, 1e-5)
            if taps is not None:
                taps[f"down{i}.res{j}"] = h
            if ucfg.down_cross[i]:
                nm = f"down_blocks.{i}.attentions.{j}."
                h = transformer_2d(h, ctx, sd, nm, ucfg.num_heads[i], g, ucfg.use_linear_projection,
                                   recorder, nm + "transformer_blocks.0.attn2")
                if taps is not None:
                    taps[f"down{i}.attn{j}"] = h
            skips.append(h)
        if i != nlev - 1:
            h = F.conv2d(h, sd[f"down_blocks.{i}.downsamplers.0.conv.weight"],
                         sd[f"down_blocks.{i}.downsamplers.0.conv.bias"], stride=2, padding=1)
            skips.append(h)
    h = resnet_block(h, temb, sd, "mid_block.resnets.0.", g, 1e-5)
    h = transformer_2d(h, ctx, sd, "mid_block.attentions.0.", ucfg.num_heads[-1], g, ucfg.use_linear_projection,
                       recorder, "mid_block.attentions.0.transformer_blocks.0.attn2")
    h = resnet_block(h, temb, sd, "mid_block.resnets.1.", g, 1e-5)
    if taps is not None:
        taps["mid"] = h
    up_cross = tuple(reversed(ucfg.down_cross))
    rev_heads = tuple(reversed(ucfg.num_heads))
    for i in range(nlev):
        for j in range(ucfg.layers_per_block + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet_block(h, temb, sd, f"up_blocks.{i}.resnets.{j}.", g, 1e-5)
            if up_cross[i]:
                nm = f"up_blocks.{i}.attentions.{j}."
                h = transformer_2d(h, ctx, sd, nm, rev_heads[i], g, ucfg.use_linear_projection,
                                   recorder, nm + "transformer_blocks.0.attn2")
            if taps is not None:
                taps[f"up{i}.{j}"] = h
        if i != nlev - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[f"up_blocks.{i}.upsamplers.0.conv.weight"],
                         sd[f"up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    h = F.silu(_gn(h, sd, "conv_norm_out", g, 1e-5))
    return F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


# ==========================================================================================
# 6. AutoencoderKL.decode  [upstream-knowledge]
# ==========================================================================================
def vae_attention(x: Tensor, sd, pre: str, groups: int) -> Tensor:
    b, c, hh, ww = x.shape
    h = _gn(x, sd, pre + "group_norm", groups, 1e-6).reshape(b, c, hh * ww).transpose(1, 2)
    o = explicit_attention_processor(h, None, sd[pre + "to_q.weight"], sd[pre + "to_k.weight"],
                                     sd[pre + "to_v.weight"], sd[pre + "to_out.0.weight"], sd[pre + "to_out.0.bias"],
                                     1, bq=sd[pre + "to_q.bias"], bk=sd[pre + "to_k.bias"], bv=sd[pre + "to_v.bias"])
    return o.transpose(1, 2).reshape(b, c, hh, ww) + x


def vae_decode(sd: Dict[str, Tensor], vcfg, z: Tensor, taps: Optional[dict] = None) -> Tensor:
    """`vae.decode(z).sample`; caller divides latents by `scaling_factor` first."""
    g = vcfg.norm_num_groups
    h = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    h = F.conv2d(h, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    h = resnet_block(h, None, sd, "decoder.mid_block.resnets.0.", g, 1e-6)
    h = vae_attention(h, sd, "decoder.mid_block.attentions.0.", g)
    h = resnet_block(h, None, sd, "decoder.mid_block.resnets.1.", g, 1e-6)
    if taps is not None:
        taps["vae.mid"] = h
    n = len(vcfg.block_out_channels)
    for i in range(n):
        for j in range(vcfg.layers_per_block + 1):
            h = resnet_block(h, None, sd, f"decoder.up_blocks.{i}.resnets.{j}.", g, 1e-6)
        if i != n - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"],
                         sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
        if taps is not None:
            taps[f"vae.up{i}"] = h
    h = F.silu(_gn(h, sd, "decoder.conv_norm_out", g, 1e-6))
    return F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


# ==========================================================================================
# 7. txt2img loop  (reference data_generation.py:56-64 -> StableDiffusionPipeline.__call__)
# ==========================================================================================
def generate(unet_sd, vae_sd, cfg, ctx: Tensor, latents: Tensor, num_inference_steps: int,
             guidance_scale: float = 7.5, recorder=None, decode: bool = True, eps_out: Optional[list] = None, scheduler: str = "ddim"):
    """ctx [2B,T,D] ordered [uncond x B, cond x B]; latents [B,4,L,L] (explicit, CPU generator).
    `scheduler`: "ddim" (BASELINE's metric) or "pndm" (what data_generation.py:59 runs for an SD-1.4 checkpoint).
    Returns (uint8 images [B,H,W,3] or None, final latents)."""
    if scheduler == "pndm":
        sch = PNDM(cfg.sched.num_train_timesteps, cfg.sched.beta_start, cfg.sched.beta_end, cfg.sched.steps_offset,
                   cfg.sched.set_alpha_to_one)
    else:
        sch = DDIM(cfg.sched.num_train_timesteps, cfg.sched.beta_start, cfg.sched.beta_end, cfg.sched.steps_offset,
                   cfg.sched.set_alpha_to_one, cfg.sched.prediction_type)
    ts = sch.set_timesteps(num_inference_steps)
    x = latents.clone().float() * sch.init_noise_sigma
    with torch.no_grad():
        for t in ts:
            xin = torch.cat([x, x], 0)                            # CFG batch: [uncond, cond]
            eps = unet_forward(unet_sd, cfg.unet, xin, torch.tensor(int(t)), ctx, recorder)
            eu, ec = eps.chunk(2)
            e = eu + guidance_scale * (ec - eu)
            if eps_out is not None:
                eps_out.append(e.clone())
            x = sch.step(e, int(t), x)
        img = None
        if decode:
            img = postprocess_image(vae_decode(vae_sd, cfg.vae, x / cfg.vae.scaling_factor))
    return img, x


# ==========================================================================================
# 8. PIL `Image.resize` (default BICUBIC) on uint8, restated  (reference data_generation.py:60,85)
#    Pillow's Resample.c: double-precision coefficients (bicubic a=-0.5, support scaled by the
#    downscale factor), normalised, converted to 22-bit fixed point, horizontal pass then vertical
#    pass, each rounding through clip8.  Pinned bit-exactly against PIL itself in the tests.
# ==========================================================================================
PIL_PRECISION_BITS = 32 - 8 - 2


def _pil_bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_resample_coeffs(in_size: int, out_size: int):
    """-> (bounds [out,2] int32 (xmin, count), kk [out, ksize] int32 fixed-point weights)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_pil_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PIL_PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PIL_PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def pil_resize_u8(img: np.ndarray, out_hw: Tuple[int, int]) -> np.ndarray:
    """img uint8 [H,W] or [H,W,C] -> uint8 [oh,ow(,C)], bit-exact with `PIL.Image.resize((ow,oh))`."""
    squeeze = img.ndim == 2
    x = img[..., None] if squeeze else img
    H, W, C = x.shape
    oh, ow = out_hw

    def one_pass(src, in_size, out_size, axis):
        if in_size == out_size:
            return src
        b, kk = pil_resample_coeffs(in_size, out_size)
        src = np.moveaxis(src, axis, 0).astype(np.int64)
        out = np.empty((out_size,) + src.shape[1:], np.int64)
        for o in range(out_size):
            xmin, n = b[o]
            acc = np.tensordot(kk[o, :n].astype(np.int64), src[xmin:xmin + n], axes=(0, 0)) + (1 << (PIL_PRECISION_BITS - 1))
            out[o] = np.clip(acc >> PIL_PRECISION_BITS, 0, 255)
        return np.moveaxis(out, 0, axis).astype(np.uint8)

    y = one_pass(x, W, ow, 1)           # horizontal first
    y = one_pass(y, H, oh, 0)           # then vertical
    return y[..., 0] if squeeze else y


# ==========================================================================================
# 9. img2img front end -- PARITY UNPINNED: the reference has no img2img call site
#    (data_generation.py:59 is txt2img); semantics follow diffusers' StableDiffusionImg2ImgPipeline /
#    AutoencoderKL.encode [upstream-knowledge]; the only reference anchor is the training-time
#    `vae.encode(...).latent_dist.sample() * scaling_factor` (finetune_sd.py:764-765).
# ==========================================================================================
def vae_encode_moments(sd: Dict[str, Tensor], vcfg, x: Tensor):
    """`vae.encode(x).latent_dist` -> (mean, logvar); x [B,3,S,S] in [-1,1]."""
    g = vcfg.norm_num_groups
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    n = len(vcfg.block_out_channels)
    for i in range(n):
        for j in range(vcfg.layers_per_block):
            h = resnet_block(h, None, sd, f"encoder.down_blocks.{i}.resnets.{j}.", g, 1e-6)
        if i != n - 1:      # Downsample2D(padding=0): F.pad (0,1,0,1) then stride-2 conv
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"],
                         sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"], stride=2)
    h = resnet_block(h, None, sd, "encoder.mid_block.resnets.0.", g, 1e-6)
    h = vae_attention(h, sd, "encoder.mid_block.attentions.0.", g)
    h = resnet_block(h, None, sd, "encoder.mid_block.resnets.1.", g, 1e-6)
    h = F.silu(_gn(h, sd, "encoder.conv_norm_out", g, 1e-6))
    h = F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    m = F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])
    mean, logvar = m.chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)


def img2img(unet_sd, vae_sd, cfg, ctx: Tensor, image: Tensor, noise_enc: Tensor, noise: Tensor, num_inference_steps: int,
            strength: float = 0.8, guidance_scale: float = 7.5, recorder=None, decode: bool = True):
    """image [B,3,S,S] in [-1,1]; noise_enc / noise: explicit N(0,1) draws for the posterior sample and add_noise."""
    sch = DDIM(cfg.sched.num_train_timesteps, cfg.sched.beta_start, cfg.sched.beta_end, cfg.sched.steps_offset,
               cfg.sched.set_alpha_to_one, cfg.sched.prediction_type)
    ts = sch.set_timesteps(num_inference_steps)
    init = min(int(num_inference_steps * strength), num_inference_steps)
    ts = ts[max(num_inference_steps - init, 0):]
    with torch.no_grad():
        mean, logvar = vae_encode_moments(vae_sd, cfg.vae, image)
        x0 = (mean + torch.exp(0.5 * logvar) * noise_enc) * cfg.vae.scaling_factor
        a = float(sch.alphas_cumprod[int(ts[0])])
        x = a ** 0.5 * x0 + (1 - a) ** 0.5 * noise
        for t in ts:
            eps = unet_forward(unet_sd, cfg.unet, torch.cat([x, x], 0), torch.tensor(int(t)), ctx, recorder)
            eu, ec = eps.chunk(2)
            x = sch.step(eu + guidance_scale * (ec - eu), int(t), x)
        img = postprocess_image(vae_decode(vae_sd, cfg.vae, x / cfg.vae.scaling_factor)) if decode else None
    return img, x, (mean, logvar)
