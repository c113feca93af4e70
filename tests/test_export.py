"""Export path: the oracle's Pillow restatement vs PIL itself (CPU), and the HIP kernels vs the
reference's literal host code (numpy + PIL) bit for bit (GPU)."""
import numpy as np
import pytest
import torch
from PIL import Image


@pytest.mark.parametrize("h,w,oh,ow,c", [(64, 64, 112, 112, 1), (512, 512, 112, 112, 3), (64, 64, 64, 64, 1),
                                         (33, 47, 112, 80, 1), (128, 96, 100, 37, 3), (16, 16, 112, 112, 1)])
def test_oracle_pil_resize_is_bit_exact_with_pil(h, w, oh, ow, c):
    from oracle import sd_oracle as O
    rng = np.random.default_rng(h * 7 + ow)
    a = rng.integers(0, 256, size=(h, w, c) if c > 1 else (h, w), dtype=np.uint8)
    np.testing.assert_array_equal(O.pil_resize_u8(a, (oh, ow)), np.asarray(Image.fromarray(a).resize((ow, oh))))


def _ref_heatmap_png(hm: np.ndarray, size: int) -> np.ndarray:
    """reference data_generation.py:82-85, literally"""
    x = (hm - hm.min()) / (hm.max() - hm.min() + 1e-8) * 255
    return np.asarray(Image.fromarray(x.astype(np.uint8)).resize((size, size)))


@pytest.mark.gpu
def test_device_heatmap_export_matches_reference_host_code():
    from agenda_amd import export
    g = torch.Generator().manual_seed(0)
    hm = torch.rand(3, 4, 64, 64, generator=g) ** 3 * 7.3          # skewed, like attention maps
    hm[0, 0] = 0.25                                                 # constant map: max == min -> all zeros
    hm[1, 2, 5, 5] = 1e4                                            # spike
    u8 = export.heatmaps_to_u8(hm.cuda()).cpu().numpy()
    out = export.resize_u8(torch.from_numpy(u8).reshape(12, 64, 64).cuda(), (112, 112)).cpu().numpy().reshape(3, 4, 112, 112)
    for b in range(3):
        for w in range(4):
            x = hm[b, w].numpy()
            ref_u8 = ((x - x.min()) / (x.max() - x.min() + 1e-8) * 255).astype(np.uint8)
            np.testing.assert_array_equal(u8[b, w], ref_u8)
            np.testing.assert_array_equal(out[b, w], _ref_heatmap_png(x, 112))


@pytest.mark.gpu
@pytest.mark.parametrize("n,h,w,c,oh,ow", [(2, 512, 512, 3, 112, 112), (3, 64, 64, 1, 112, 112), (1, 100, 37, 3, 64, 200),
                                           (2, 64, 64, 1, 64, 64)])
def test_device_resize_is_bit_exact_with_pil(n, h, w, c, oh, ow):
    from agenda_amd import export
    rng = np.random.default_rng(n + h + ow)
    a = rng.integers(0, 256, size=(n, h, w, c), dtype=np.uint8)
    t = torch.from_numpy(a if c > 1 else a[..., 0]).cuda()
    got = export.resize_u8(t, (oh, ow)).cpu().numpy()
    for i in range(n):
        ref = np.asarray(Image.fromarray(a[i] if c > 1 else a[i, :, :, 0]).resize((ow, oh)))
        np.testing.assert_array_equal(got[i], ref)


@pytest.mark.gpu
def test_device_stack_matches_postprocess_heatmap():
    from agenda_amd import export
    from agenda_amd.generation import stack_heatmaps
    rng = np.random.default_rng(5)
    obj, fg, bg = (rng.integers(0, 256, size=(2, 112, 112), dtype=np.uint8) for _ in range(3))
    rgb, inv = export.stack_heatmaps(*(torch.from_numpy(x).cuda() for x in (obj, fg, bg)))
    for i in range(2):
        r, v = stack_heatmaps(obj[i], fg[i], bg[i])
        np.testing.assert_array_equal(rgb[i].cpu().numpy(), r)
        np.testing.assert_array_equal(inv[i].cpu().numpy(), v)
