"""Single-op bindings (fp32 torch tensors on the GPU -> HIP kernels -> fp32).  These mirror the
torch.nn.functional calls that diffusers makes on the reference path so each HIP kernel can be
parity-tested in isolation through the C ABI (`agd_op_*` in include/agenda_hip.h)."""
from __future__ import annotations

import torch

from . import _lib


def _f32c(t):
    assert t.is_cuda, "ops run on the GPU only (no CPU fallback)"
    return t.detach().to(torch.float32).contiguous()


_P8 = {0: 0, 1: 2, 2: 4, 3: 8}      # igemm8p.h: 1 = where the launcher would pick it, 2 / 3 = force the 256- / 160-wide tile


def conv2d(x, w, bias=None, stride=1, padding=None, upsample=False, halo=False, p8=0, smap=False, phases=False, pc=0, xcd_block=False):
    lib = _lib.load()
    x, w = _f32c(x), _f32c(w)
    b = _f32c(bias) if bias is not None else None
    B, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    pad = (1 if k == 3 else 0) if padding is None else padding
    up = 2 if upsample else 1
    Ho = (H * up + 2 * pad - k) // stride + 1
    Wo = (W * up + 2 * pad - k) // stride + 1
    y = torch.empty(B, Cout, Ho, Wo, device=x.device, dtype=torch.float32)
    _lib.check(lib.agd_op_conv2d_ex(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), B, Cin, H, W, Cout, k, stride,
                                    pad, int(upsample), (1 if halo else 0) | _P8[p8] | (16 if smap else 0) | (64 if phases else 0) | ((int(pc) & 15) << 7) | ((int(pc) & 48) << 8) | (2048 if xcd_block else 0), _lib.current_stream_ptr()), None, "agd_op_conv2d")
    return y


def linear(x, w, bias=None, residual=None, geglu=False, p8=0, wreg=False, kgroups=False, pc=0, xcd_block=False):
    lib = _lib.load()
    x2 = _f32c(x).reshape(-1, x.shape[-1])
    w = _f32c(w)
    M, K = x2.shape
    N = w.shape[0]
    Nout = N // 2 if geglu else N
    b = _f32c(bias) if bias is not None else None
    r = _f32c(residual).reshape(M, Nout) if residual is not None else None
    y = torch.empty(M, Nout, device=x.device, dtype=torch.float32)
    _lib.check(lib.agd_op_linear(_lib.ptr(x2), _lib.ptr(w), _lib.ptr(b), _lib.ptr(r), _lib.ptr(y), M, K, N, int(geglu) | _P8[p8] | (16 if wreg else 0) | (32 if kgroups else 0) | (64 if (wreg and kgroups) else 0) | ((int(pc) & 15) << 7) | ((int(pc) & 48) << 8) | (2048 if xcd_block else 0),
                                 _lib.current_stream_ptr()), None, "agd_op_linear")
    return y.reshape(*x.shape[:-1], Nout)


def group_norm(x, groups, gamma, beta, eps=1e-5, silu=False):
    lib = _lib.load()
    x = _f32c(x)
    B, Cc = x.shape[:2]
    HW = x[0, 0].numel()
    y = torch.empty_like(x)
    _lib.check(lib.agd_op_groupnorm(_lib.ptr(x), _lib.ptr(_f32c(gamma)), _lib.ptr(_f32c(beta)), _lib.ptr(y), B, Cc, HW,
                                    groups, float(eps), int(silu), _lib.current_stream_ptr()), None, "agd_op_groupnorm")
    return y


def layer_norm(x, gamma, beta, eps=1e-5):
    lib = _lib.load()
    x = _f32c(x)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    y = torch.empty_like(x)
    _lib.check(lib.agd_op_layernorm(_lib.ptr(x), _lib.ptr(_f32c(gamma)), _lib.ptr(_f32c(beta)), _lib.ptr(y), rows, Cc,
                                    float(eps), _lib.current_stream_ptr()), None, "agd_op_layernorm")
    return y


def ff_fused(x, gamma, beta, w1, b1, w2, b2, eps=1e-5):
    """x + ff.net.2(GEGLU(ff.net.0(LayerNorm(x)))) as ONE launch (tblock.hip); x [..., C], w1 [8C, C] (value rows then gate rows), w2 [C, 4C]."""
    lib = _lib.load()
    Cc = x.shape[-1]
    x2 = _f32c(x).reshape(-1, Cc)
    y = torch.empty_like(x2)
    _lib.check(lib.agd_op_ff_fused(_lib.ptr(x2), _lib.ptr(_f32c(gamma)), _lib.ptr(_f32c(beta)), _lib.ptr(_f32c(w1)), _lib.ptr(_f32c(b1)),
                                   _lib.ptr(_f32c(w2)), _lib.ptr(_f32c(b2)), _lib.ptr(y), x2.shape[0], Cc, float(eps),
                                   _lib.current_stream_ptr()), None, "agd_op_ff_fused")
    return y.reshape(x.shape)


def attn_chain(x, gamma, beta, wq, kv, wo, bo, heads=8, eps=1e-5, return_probs=False, rows32=False):
    """x + to_out(attention(to_q(LayerNorm(x)), k, v)) as ONE launch (tblock.hip); x [B, HW, C], kv [B, T, 2C] (K columns then V columns).
    With return_probs: also the probabilities summed over the heads, [B, T, HW]."""
    lib = _lib.load()
    x, kv = _f32c(x), _f32c(kv)
    B, HW, Cc = x.shape
    T = kv.shape[1]
    y = torch.empty_like(x)
    pr = torch.empty(B, T, HW, device=x.device, dtype=torch.float32) if return_probs else None
    _lib.check(lib.agd_op_attn_chain(_lib.ptr(x), _lib.ptr(_f32c(gamma)), _lib.ptr(_f32c(beta)), _lib.ptr(_f32c(wq)), _lib.ptr(kv), _lib.ptr(_f32c(wo)),
                                     _lib.ptr(_f32c(bo)), _lib.ptr(y), _lib.ptr(pr), B, HW, T, Cc, heads | (256 if rows32 else 0), float(eps), _lib.current_stream_ptr()),
               None, "agd_op_attn_chain")
    return (y, pr) if return_probs else y


def xattn_premul(x, gamma, beta, wq, kv, wo, bo, heads=8, eps=1e-5, return_probs=False):
    """x + to_out(attention(to_q(LayerNorm(x)), k, v)) through the pre-multiplied context matrices of the C = 1280 blocks (xattn_pre.hip): x [B, HW, C],
    kv [B, T, 2C] (K columns then V columns).  With return_probs: also every head's probabilities, [B, heads, T, HW]."""
    lib = _lib.load()
    x, kv = _f32c(x), _f32c(kv)
    B, HW, Cc = x.shape
    T = kv.shape[1]
    y = torch.empty_like(x)
    pr = torch.empty(B, heads, T, HW, device=x.device, dtype=torch.float32) if return_probs else None
    _lib.check(lib.agd_op_xattn_premul(_lib.ptr(x), _lib.ptr(_f32c(gamma)), _lib.ptr(_f32c(beta)), _lib.ptr(_f32c(wq)), _lib.ptr(kv), _lib.ptr(_f32c(wo)),
                                       _lib.ptr(_f32c(bo)), _lib.ptr(y), _lib.ptr(pr), B, HW, T, Cc, heads, float(eps), _lib.current_stream_ptr()),
               None, "agd_op_xattn_premul")
    return (y, pr) if return_probs else y


def attention(q, k, v, heads, scale=None, return_probs=False):
    """q [B,Nq,H*D], k/v [B,Nk,H*D] -> o [B,Nq,H*D] (+ probs [B,H,Nk,Nq] token-major if asked, Nk<=96)."""
    lib = _lib.load()
    q, k, v = _f32c(q), _f32c(k), _f32c(v)
    B, Nq, Cc = q.shape
    Nk = k.shape[1]
    D = Cc // heads
    scale = D ** -0.5 if scale is None else scale
    o = torch.empty_like(q)
    probs = torch.empty(B, heads, Nk, Nq, device=q.device, dtype=torch.float32) if return_probs else None
    _lib.check(lib.agd_op_attention(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(o), B, heads, D, Nq, Nk, float(scale),
                                    _lib.ptr(probs), _lib.current_stream_ptr()), None, "agd_op_attention")
    return (o, probs) if return_probs else o


def attention_headsum(q, k, v, heads, scale=None):
    """As `attention(..., return_probs=True)` with the probabilities summed over the heads: probs [B,Nk,Nq] (Nk<=96) -- the form
    the daam recorder keeps for the layers at latent resolution."""
    lib = _lib.load()
    q, k, v = _f32c(q), _f32c(k), _f32c(v)
    B, Nq, Cc = q.shape
    Nk = k.shape[1]
    D = Cc // heads
    scale = D ** -0.5 if scale is None else scale
    o = torch.empty_like(q)
    probs = torch.empty(B, Nk, Nq, device=q.device, dtype=torch.float32)
    _lib.check(lib.agd_op_attention_headsum(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(o), B, heads, D, Nq, Nk, float(scale),
                                            _lib.ptr(probs), _lib.current_stream_ptr()), None, "agd_op_attention_headsum")
    return o, probs


def bicubic_clamp_mean(maps, out_side):
    """maps [n_maps, T, side, side] -> mean_n clamp(bicubic(maps[n]), 0) : [T, S, S]"""
    lib = _lib.load()
    maps = _f32c(maps)
    n, T, side, _ = maps.shape
    out = torch.empty(T, out_side, out_side, device=maps.device, dtype=torch.float32)
    _lib.check(lib.agd_op_bicubic_clamp_mean(_lib.ptr(maps), n, T, side, out_side, _lib.ptr(out),
                                             _lib.current_stream_ptr()), None, "agd_op_bicubic_clamp_mean")
    return out


def attn_reg_loss(attn_map: torch.Tensor, obj_idx, fg_idx, bg_idx, coef: float, want_grad: bool = True):
    """finetune_sd_token.py:1046-1066 on one recorded map [B, T, h, w] (cuda fp32): returns (loss [B, 2] = {bg, fg} per sample,
    d_map [B, T, h, w] or None).  `*_idx`: int sequences / tensors of length B; obj < 0 skips the sample."""
    lib = _lib.load()
    m = attn_map.detach().to(torch.float32).contiguous()
    assert m.is_cuda and m.ndim == 4
    B, T = m.shape[:2]
    P = m.shape[2] * m.shape[3]
    host = [torch.as_tensor(list(map(int, i)) if not torch.is_tensor(i) else i.detach().cpu(), dtype=torch.int32) for i in (obj_idx, fg_idx, bg_idx)]
    has = host[0] >= 0
    # the reference indexes the map's token rows directly (finetune_sd_token.py:1049-1060): a row the map does not have is an
    # IndexError there, never a silently skipped sample (only obj < 0 means "no object in this sample")
    if bool(has.any()):
        worst = max(int(h[has].max()) for h in host)
        if worst >= T:
            raise IndexError(f"index {worst} is out of bounds for dimension 0 with size {T}")
    idx = [h.to(m.device).contiguous() for h in host]
    loss = torch.empty(B, 2, device=m.device, dtype=torch.float32)
    dmap = torch.empty_like(m) if want_grad else None
    _lib.check(lib.agd_op_attn_reg_loss(_lib.ptr(m), B, T, P, _lib.ptr(idx[0]), _lib.ptr(idx[1]), _lib.ptr(idx[2]), float(coef), _lib.ptr(loss),
                                        _lib.ptr(dmap), _lib.current_stream_ptr()), None, "agd_op_attn_reg_loss")
    return loss, dmap


def conv_groupnorm(x, w, bias, gamma, beta, groups=32, eps=1e-5, silu=True, fused=True, p8=0):
    """conv3x3 -> GroupNorm(+SiLU) chained like the graph walk; `fused`: GroupNorm statistics from the conv launch's epilogue."""
    lib = _lib.load()
    x, w = _f32c(x), _f32c(w)
    b = _f32c(bias) if bias is not None else None
    ga, be = _f32c(gamma), _f32c(beta)
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    y = torch.empty(B, Cout, H, W, device=x.device, dtype=torch.float32)
    _lib.check(lib.agd_op_conv_groupnorm(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(ga), _lib.ptr(be), _lib.ptr(y), B, Cin, H, W, Cout,
                                         groups, float(eps), int(silu), int(fused) | _P8[p8], _lib.current_stream_ptr()), None, "agd_op_conv_groupnorm")
    return y
