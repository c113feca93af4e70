"""Pin the oracle's hook.py restatement against golden vectors produced by the reference's own
hook.py (tests/golden/make_golden_hook.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import sd_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("case,heads", [(0, 8), (1, 4), (2, 2)])
@pytest.mark.parametrize("is_train", [True, False])
def test_unravel_attn_matches_reference(golden_dir, case, heads, is_train):
    z = _load(golden_dir, "hook_unravel.npz")
    p = torch.from_numpy(z[f"unravel{case}_in"])
    want = z[f"unravel{case}_out_train{int(is_train)}"]
    got = O.hooker_unravel_attn(p, heads, is_train).numpy()
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-7)


def test_global_heat_map_matches_reference(golden_dir):
    z = _load(golden_dir, "hook_global.npz")
    maps = [torch.from_numpy(z[f"global_in{i}"]) for i in range(6)]
    got = O.hooker_global_heat_map(maps, 64).numpy()
    np.testing.assert_allclose(got, z["global_out"], rtol=0, atol=1e-6)
    assert float(z["global_min_unclamped"]) < 0 <= got.min()      # the clamp is exercised
    maps2 = [torch.from_numpy(z[f"global_b2_in{i}"]) for i in range(3)]
    got2 = O.hooker_global_heat_map(maps2, 32).numpy()
    np.testing.assert_allclose(got2, z["global_b2_out"], rtol=0, atol=1e-6)


def test_global_heat_map_empty_raises(golden_dir):
    z = _load(golden_dir, "hook_global.npz")
    assert int(z["global_empty_raises"]) == 1
    with pytest.raises(RuntimeError, match=str(z["global_empty_msg"])):
        O.hooker_global_heat_map([], 64)


@pytest.mark.parametrize("name", ["call_hw64", "call_hw144"])
@pytest.mark.parametrize("is_train", [True, False])
def test_processor_call_matches_reference(golden_dir, name, is_train):
    z = _load(golden_dir, "hook_call.npz")
    x, ctx = torch.from_numpy(z[name + "_x"]), torch.from_numpy(z[name + "_ctx"])
    rec = O.HookRecorder(is_train=is_train, latent_hw=64)
    w = {k: torch.from_numpy(z[f"{name}_cross_{k}"]) for k in ("wq", "wk", "wv", "wo", "bo")}
    y = O.explicit_attention_processor(x, ctx, w["wq"], w["wk"], w["wv"], w["wo"], w["bo"], 4, recorder=rec)
    np.testing.assert_allclose(y.numpy(), z[f"{name}_cross_y_train{int(is_train)}"], rtol=1e-5, atol=2e-6)
    assert len(rec.cross_attn_maps) == 1
    np.testing.assert_allclose(rec.cross_attn_maps[0].numpy(), z[f"{name}_map_train{int(is_train)}"],
                               rtol=1e-5, atol=1e-7)
    w = {k: torch.from_numpy(z[f"{name}_self_{k}"]) for k in ("wq", "wk", "wv", "wo", "bo")}
    ys = O.explicit_attention_processor(x, None, w["wq"], w["wk"], w["wv"], w["wo"], w["bo"], 4, recorder=rec)
    np.testing.assert_allclose(ys.numpy(), z[f"{name}_self_y_train{int(is_train)}"], rtol=1e-5, atol=2e-6)
    # self-attention records nothing (hook.py:110)
    assert len(rec.cross_attn_maps) == 1
    assert list(z[f"{name}_nmaps_train{int(is_train)}"]) == [1, 1]


@pytest.mark.parametrize("tag", ["nomask", "mask"])
@pytest.mark.parametrize("is_train", [True, False])
def test_processor_seam_fixture_matches_reference(golden_dir, tag, is_train):
    """The reference's `__call__` at the shapes of a `tiny` UNet layer, with and without an additive attention mask
    (hook.py:92,108) -- the fixture tests/test_golden_gpu.py drives through the C-ABI seam."""
    z = _load(golden_dir, "hook_seam.npz")
    x, ctx = torch.from_numpy(z["seam_x"]), torch.from_numpy(z["seam_ctx"])
    mc = torch.from_numpy(z["seam_mask_cross"]) if tag == "mask" else None
    ms = torch.from_numpy(z["seam_mask_self"]) if tag == "mask" else None
    rec = O.HookRecorder(is_train=is_train, latent_hw=16)
    w = {k: torch.from_numpy(z[f"seam_cross_{k}"]) for k in ("wq", "wk", "wv", "wo", "bo")}
    y = O.explicit_attention_processor(x, ctx, w["wq"], w["wk"], w["wv"], w["wo"], w["bo"], 2, recorder=rec, attention_mask=mc)
    np.testing.assert_allclose(y.numpy(), z[f"seam_{tag}_cross_y_train{int(is_train)}"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(rec.cross_attn_maps[0].numpy(), z[f"seam_{tag}_map_train{int(is_train)}"], rtol=1e-5, atol=1e-7)
    w = {k: torch.from_numpy(z[f"seam_self_{k}"]) for k in ("wq", "wk", "wv", "wo", "bo")}
    ys = O.explicit_attention_processor(x, None, w["wq"], w["wk"], w["wv"], w["wo"], w["bo"], 2, recorder=rec, attention_mask=ms)
    np.testing.assert_allclose(ys.numpy(), z[f"seam_{tag}_self_y_train{int(is_train)}"], rtol=1e-5, atol=2e-6)
    assert len(rec.cross_attn_maps) == 1


def _bf16bits(a):
    """uint16 bf16 bit patterns -> the fp32 values they denote"""
    return torch.from_numpy((a.astype(np.uint32) << 16).view(np.float32).copy())


@pytest.mark.parametrize("name,C", [("chain320", 320), ("chain640", 640), ("chain1280", 1280)])
def test_processor_call_at_chain_kernel_shapes_matches_reference(golden_dir, name, C):
    """hook.py:83-122 at the shapes of SD-1.5's 64 x 64 / 32 x 32 / 16 x 16 blocks (8 heads of 40 / 80 / 160, hw = 1024 / 256 / 256), called behind
    diffusers' norm2 (F.layer_norm of the raw residual stream): the fixture tests/test_golden_gpu.py drives the fused attn2 chain
    kernel with.  The oracle's restatement on the same inputs."""
    z = _load(golden_dir, "hook_chain1280.npz" if C == 1280 else "hook_chain.npz")
    x = _bf16bits(z[name + "_x_bf16bits"])
    w = {k: _bf16bits(z[f"{name}_{k}_bf16bits"]) for k in ("wq", "wk", "wv", "wo")}
    ctx, ga, be, bo = (torch.from_numpy(z[f"{name}_{k}"]) for k in ("ctx", "gamma", "beta", "bo"))
    assert x.shape[-1] == C and w["wq"].shape == (C, C)
    rec = O.HookRecorder(is_train=False, latent_hw=64)
    y = O.explicit_attention_processor(torch.nn.functional.layer_norm(x, (C,), ga, be, 1e-5), ctx, w["wq"], w["wk"], w["wv"], w["wo"], bo, 8, recorder=rec)
    np.testing.assert_allclose(y[1].numpy(), z[name + "_y1_f16"].astype(np.float32), rtol=2e-3, atol=2e-3)      # the fixture keeps fp16
    np.testing.assert_allclose(rec.cross_attn_maps[0].numpy(), z[name + "_map"], rtol=1e-5, atol=1e-7)
