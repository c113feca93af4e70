"""Device-side export of images and DAAM heat maps, bit-exact with the reference's
host path: numpy min-max/`astype(uint8)` (data_generation.py:82-84), `PIL.Image.resize` default BICUBIC
(data_generation.py:60,85) and the `[obj, fg, 255-bg]` stack (postprocess_heatmap.py:44-48).  One D2H copy of the
finished uint8 buffers replaces the per-word fp32 copy + host PIL work of the reference loop."""
from __future__ import annotations

import torch

from . import _lib


def heatmaps_to_u8(hm: torch.Tensor) -> torch.Tensor:
    """fp32 [..., S, S] (cuda) -> uint8 same shape: per-map (x-min)/(max-min+1e-8)*255, truncated."""
    lib = _lib.load()
    assert hm.is_cuda
    x = hm.detach().to(torch.float32).contiguous()
    npix = x.shape[-1] * x.shape[-2]
    n = x.numel() // npix
    out = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    if x.numel() == 0:                      # images-only runs (no --word_token_heatmaps): nothing to convert
        return out
    _lib.check(lib.agd_op_heatmap_u8(_lib.ptr(x), n, npix, _lib.ptr(out), _lib.current_stream_ptr()), None, "agd_op_heatmap_u8")
    return out


def resize_u8(img: torch.Tensor, out_hw) -> torch.Tensor:
    """uint8 [n,H,W] or [n,H,W,C] (cuda) -> [n,oh,ow(,C)], identical to PIL Image.resize((ow,oh)) per image."""
    lib = _lib.load()
    assert img.is_cuda and img.dtype == torch.uint8
    squeeze = img.ndim == 3
    x = (img[..., None] if squeeze else img).contiguous()
    n, H, W, Cc = x.shape
    oh, ow = out_hw
    out = torch.empty(n, oh, ow, Cc, device=x.device, dtype=torch.uint8)
    if n == 0:
        return out[..., 0] if squeeze else out
    _lib.check(lib.agd_op_resize_u8_pil(_lib.ptr(x), n, H, W, Cc, oh, ow, _lib.ptr(out), _lib.current_stream_ptr()), None,
               "agd_op_resize_u8_pil")
    return out[..., 0] if squeeze else out


def stack_heatmaps(obj: torch.Tensor, fg: torch.Tensor, bg: torch.Tensor):
    """uint8 [..., H, W] x3 -> (rgb [..., H, W, 3] = [obj, fg, 255-bg], inv = 255-bg)."""
    lib = _lib.load()
    obj, fg, bg = (t.contiguous() for t in (obj, fg, bg))
    rgb = torch.empty(*obj.shape, 3, device=obj.device, dtype=torch.uint8)
    inv = torch.empty_like(obj)
    _lib.check(lib.agd_op_stack_heatmaps(_lib.ptr(obj), _lib.ptr(fg), _lib.ptr(bg), obj.numel(), _lib.ptr(rgb), _lib.ptr(inv),
                                         _lib.current_stream_ptr()), None, "agd_op_stack_heatmaps")
    return rgb, inv


def export_batch(images_u8: torch.Tensor, heatmaps: torch.Tensor, image_size: int):
    """images uint8 [B,H,W,3], heat maps fp32 [B,nw,S,S] (cuda) -> (uint8 [B,size,size,3], uint8 [B,nw,size,size]):
    exactly the PNG payloads `data_generation.py:60,76-86` writes."""
    small = resize_u8(images_u8, (image_size, image_size))
    B, nw = heatmaps.shape[:2]
    hm = resize_u8(heatmaps_to_u8(heatmaps).reshape(B * nw, *heatmaps.shape[2:]), (image_size, image_size))
    return small, hm.reshape(B, nw, image_size, image_size)
