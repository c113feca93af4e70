"""Pin the HIP attention + recorder kernel DIRECTLY to the reference's own outputs: golden vectors
produced by running /root/reference/data_generation/hook.py (tests/golden/make_golden_hook.py).
The projections (tiny Linear layers of the fixture, C=160) run in torch; the attention core
(hook.py:104-115) and the `_unravel_attn` map (hook.py:28-56) come from the HIP kernel."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,N", [("call_hw64", 64), ("call_hw144", 144)])
@pytest.mark.parametrize("is_train", [True, False])
def test_attention_kernel_matches_reference_hook_call(golden_dir, name, N, is_train):
    from agenda_amd import ops
    z = np.load(os.path.join(golden_dir, "hook_call.npz"))
    heads = 4
    x, ctx = torch.from_numpy(z[name + "_x"]), torch.from_numpy(z[name + "_ctx"])
    w = {k: torch.from_numpy(z[f"{name}_cross_{k}"]) for k in ("wq", "wk", "wv", "wo", "bo")}
    q, k, v = F.linear(x, w["wq"]), F.linear(ctx, w["wk"]), F.linear(ctx, w["wv"])
    o, probs = ops.attention(q.cuda(), k.cuda(), v.cuda(), heads, return_probs=True)       # probs [B,H,T,N]
    y = F.linear(o.cpu(), w["wo"], w["bo"])
    want_y = torch.from_numpy(z[f"{name}_cross_y_train{int(is_train)}"])
    assert float((y - want_y).abs().max() / want_y.abs().max()) < 2.0 ** -6
    # hook.py:28-56: keep the conditional half in inference mode, mean over heads, [B', T, h, w]
    side = int(N ** 0.5)
    p = probs.cpu()
    if not is_train:
        p = p[p.shape[0] // 2:]
    maps = p.mean(1).reshape(p.shape[0], p.shape[2], side, side)
    want_m = torch.from_numpy(z[f"{name}_map_train{int(is_train)}"])
    assert maps.shape == want_m.shape
    assert float((maps - want_m).abs().max()) < 2e-3
    # self-attention path (hook.py:95-99 with encoder_hidden_states None): output only
    ws = {k: torch.from_numpy(z[f"{name}_self_{k}"]) for k in ("wq", "wk", "wv", "wo", "bo")}
    qs, ks, vs = F.linear(x, ws["wq"]), F.linear(x, ws["wk"]), F.linear(x, ws["wv"])
    ys = F.linear(ops.attention(qs.cuda(), ks.cuda(), vs.cuda(), heads).cpu(), ws["wo"], ws["bo"])
    want_ys = torch.from_numpy(z[f"{name}_self_y_train{int(is_train)}"])
    assert float((ys - want_ys).abs().max() / want_ys.abs().max()) < 2.0 ** -6


def test_bicubic_clamp_mean_matches_reference_global_heat_map(golden_dir):
    """hook.py:59-81 on the reference's own fixture (mixed resolutions, clamp exercised)."""
    from agenda_amd import ops
    z = np.load(os.path.join(golden_dir, "hook_global.npz"))
    want = torch.from_numpy(z["global_out"])[0]                      # [T, 64, 64]
    acc = None
    for i in range(6):
        m = torch.from_numpy(z[f"global_in{i}"])                     # [1, T, r, r]
        part = ops.bicubic_clamp_mean(m.cuda(), 64).cpu()            # n_maps = 1
        acc = part if acc is None else acc + part
    got = acc / 6
    assert float((got - want).abs().max()) < 2e-5


def test_hooker_seam_records_per_call_maps():
    """Direct processor call through the C-ABI seam (agd_cross_attn): output + appended map (hook.py:110-112)."""
    from agenda_amd import StableDiffusionPipeline, UNetCrossAttentionHooker, config, synthetic
    from oracle import sd_oracle as O
    cfg = config.tiny()
    u, v = synthetic.make_unet_weights(cfg, 11), synthetic.make_vae_weights(cfg, 12)
    pipe = StableDiffusionPipeline(cfg, u, v, workspace_bytes=1 << 30)
    name = "down_blocks.0.attentions.0.transformer_blocks.0.attn2"
    C, heads, L = 64, 2, 16
    g = torch.Generator().manual_seed(1)
    hidden = torch.randn(2, L * L, C, generator=g).to(torch.bfloat16).float()
    ctx = synthetic.make_context(cfg, 1, seed=2)
    hk = UNetCrossAttentionHooker(is_train=False, latent_hw=L)
    pipe.unet.set_attn_processor(hk)
    y = hk(pipe.unet.attn2(name), hidden, ctx)
    rec = O.HookRecorder(is_train=False, latent_hw=L)
    t = name.rsplit("attn2", 1)[0]
    want = O.explicit_attention_processor(hidden, ctx, u[t + "attn2.to_q.weight"], u[t + "attn2.to_k.weight"],
                                          u[t + "attn2.to_v.weight"], u[t + "attn2.to_out.0.weight"],
                                          u[t + "attn2.to_out.0.bias"], heads, recorder=rec)
    assert float((y.cpu() - want).abs().max() / want.abs().max()) < 2.0 ** -6
    assert len(hk.cross_attn_maps) == 1 and hk.cross_attn_maps[0].shape == rec.cross_attn_maps[0].shape
    assert float((hk.cross_attn_maps[0].cpu() - rec.cross_attn_maps[0]).abs().max()) < 2e-3
    got = hk.compute_global_heat_map()
    assert float((got.cpu() - rec.compute_global_heat_map()).abs().max()) < 2e-3
    hk.clear()
    assert hk.cross_attn_maps == []
    pipe.engine.close()


@pytest.mark.parametrize("tag", ["nomask", "mask"])
@pytest.mark.parametrize("is_train", [True, False])
def test_processor_seam_through_c_abi_matches_reference_fixture(golden_dir, tag, is_train):
    """hook.py:83-122 end to end THROUGH THE C ABI (`agd_attn_processor`): the reference's own outputs for one cross- and
    one self-attention call (with / without an additive attention mask) vs the HIP seam, the fixture's weights loaded
    into a `tiny` UNet layer.  Covers hook.py:92 (mask), :95-99 (self path records nothing), :110-112 (map)."""
    from agenda_amd import StableDiffusionPipeline, UNetCrossAttentionHooker, config, synthetic
    z = np.load(os.path.join(golden_dir, "hook_seam.npz"))
    cfg = config.tiny()
    u, v = synthetic.make_unet_weights(cfg, 11), synthetic.make_vae_weights(cfg, 12)
    t = "down_blocks.0.attentions.0.transformer_blocks.0."
    for mod, key in (("attn2", "cross"), ("attn1", "self")):
        for wn, dst in (("wq", "to_q.weight"), ("wk", "to_k.weight"), ("wv", "to_v.weight"), ("wo", "to_out.0.weight"), ("bo", "to_out.0.bias")):
            src = torch.from_numpy(z[f"seam_{key}_{wn}"])
            assert u[t + mod + "." + dst].shape == src.shape
            u[t + mod + "." + dst] = src
    pipe = StableDiffusionPipeline(cfg, u, v, workspace_bytes=1 << 30)
    x, ctx = torch.from_numpy(z["seam_x"]), torch.from_numpy(z["seam_ctx"])
    mc = torch.from_numpy(z["seam_mask_cross"]) if tag == "mask" else None
    ms = torch.from_numpy(z["seam_mask_self"]) if tag == "mask" else None
    hk = UNetCrossAttentionHooker(is_train=is_train, latent_hw=16)
    pipe.unet.set_attn_processor(hk)
    assert len(pipe.unet.attn_processors) == 32 and all(p is hk for p in pipe.unet.attn_processors.values())
    yc = hk(pipe.unet.attn(t + "attn2"), x, encoder_hidden_states=ctx, attention_mask=mc)
    ys = hk(pipe.unet.attn(t + "attn1"), x, attention_mask=ms)
    assert len(hk.cross_attn_maps) == 1                                     # the self call recorded nothing
    want_c, want_s = (torch.from_numpy(z[f"seam_{tag}_{k}_y_train{int(is_train)}"]) for k in ("cross", "self"))
    # bf16 operands / fp32 accumulate vs the reference's fp32: 2^-6 of the output scale
    assert float((yc.cpu() - want_c).abs().max() / want_c.abs().max()) < 2.0 ** -6
    assert float((ys.cpu() - want_s).abs().max() / want_s.abs().max()) < 2.0 ** -6
    want_m = torch.from_numpy(z[f"seam_{tag}_map_train{int(is_train)}"])
    assert hk.cross_attn_maps[0].shape == want_m.shape
    assert float((hk.cross_attn_maps[0].cpu() - want_m).abs().max()) < 2e-3
    # a cross-attention module cannot serve a self-attention call, a non-square token count cannot be unravelled
    with pytest.raises(ValueError):
        hk(pipe.unet.attn(t + "attn2"), x)
    with pytest.raises(RuntimeError):
        hk(pipe.unet.attn(t + "attn2"), x[:, :250], encoder_hidden_states=ctx)
    pipe.engine.close()


def _bf16bits(a):
    """uint16 bf16 bit patterns -> the fp32 values they denote"""
    return torch.from_numpy((a.astype(np.uint32) << 16).view(np.float32).copy())


@pytest.mark.parametrize("name,C,r32", [("chain320", 320, 0), ("chain640", 640, 0), ("chain640", 640, 1)])      # r32: the 32-row panel form of the C = 640 kernel
def test_attn_chain_kernel_matches_reference_hook_call_at_production_shapes(golden_dir, name, C, r32):
    """since round 4 the attn2 layers of the 64 x 64 / 32 x 32 maps run `attn_chain_kernel<320>` / `<640>` (tblock.hip), not the
    kernel the older fixtures drive.  This pins THAT kernel to the reference's own outputs: hook.py:83-122 (`UNetCrossAttentionHooker.__call__`,
    inference mode) run on F.layer_norm(x_raw) at C = 320 / 8 heads of 40 / hw = 1024 and C = 640 / 8 heads of 80 / hw = 256
    (tests/golden/make_golden_hook.py part 5).  The kernel takes the raw rows, applies norm2, to_q, the attention with the head-summed
    probability side output, to_out + bias + residual: out - x_raw must be the reference's output, probs / heads its recorded map
    (hook.py:48-55: conditional half, mean over heads).  k / v are the fixture's to_k / to_v applied in torch (as in the test above)."""
    from agenda_amd import ops
    from _report import report
    z = np.load(os.path.join(golden_dir, "hook_chain.npz"))
    H = 8
    x = _bf16bits(z[name + "_x_bf16bits"])
    w = {k: _bf16bits(z[f"{name}_{k}_bf16bits"]) for k in ("wq", "wk", "wv", "wo")}
    ctx, ga, be, bo = (torch.from_numpy(z[f"{name}_{k}"]) for k in ("ctx", "gamma", "beta", "bo"))
    kv = torch.cat([F.linear(ctx, w["wk"]), F.linear(ctx, w["wv"])], dim=-1)               # [B, T, 2C]
    cu = lambda t: t.cuda()
    got, pr = ops.attn_chain(cu(x), cu(ga), cu(be), cu(w["wq"]), cu(kv), cu(w["wo"]), cu(bo), heads=H, return_probs=True, rows32=bool(r32))
    y1 = (got.cpu() - x)[1]
    want_y1 = torch.from_numpy(z[name + "_y1_f16"].astype(np.float32))
    e_y = float((y1 - want_y1).abs().max() / want_y1.abs().max())
    want_m = torch.from_numpy(z[name + "_map"])                                            # [1, T, side, side]
    side = want_m.shape[-1]
    m = (pr.cpu()[1:] / H).reshape(1, pr.shape[1], side, side)
    e_m = float((m - want_m).abs().max())
    print(f"attn_chain_kernel<{C}> vs the reference's __call__: output max rel {e_y:.5f}, map max abs {e_m:.6f}")
    report(f"golden_attn_chain_kernel[C={C},rows32={r32}]", out_max_rel=e_y, map_max_abs=e_m)
    assert e_y < 2.0 ** -6, e_y          # bf16 operands / fp32 accumulate + a bf16 residual stream vs the reference's fp32
    assert e_m < 2e-3, e_m


def test_premultiplied_attn2_matches_reference_hook_call_at_the_16x16_block_shape(golden_dir):
    """The attn2 form the C = 1280 blocks run since round 5 (csrc/xattn_pre.hip: per-image pre-multiplied context matrices, two GEMMs) pinned to the
    reference's own outputs: hook.py:83-122 (inference mode) on F.layer_norm(x_raw) at C = 1280 / 8 heads of 160 / hw = 256
    (tests/golden/make_golden_hook.py part 5, `chain1280`).  out - x_raw must be the reference's output, the per-head probabilities' mean its
    recorded map (hook.py:48-55: conditional half, mean over heads)."""
    from agenda_amd import ops
    from _report import report
    z = np.load(os.path.join(golden_dir, "hook_chain1280.npz"))
    name, H = "chain1280", 8
    x = _bf16bits(z[name + "_x_bf16bits"])
    w = {k: _bf16bits(z[f"{name}_{k}_bf16bits"]) for k in ("wq", "wk", "wv", "wo")}
    ctx, ga, be, bo = (torch.from_numpy(z[f"{name}_{k}"]) for k in ("ctx", "gamma", "beta", "bo"))
    kv = torch.cat([F.linear(ctx, w["wk"]), F.linear(ctx, w["wv"])], dim=-1)               # [B, T, 2C]
    cu = lambda t: t.cuda()
    got, pr = ops.xattn_premul(cu(x), cu(ga), cu(be), cu(w["wq"]), cu(kv), cu(w["wo"]), cu(bo), heads=H, return_probs=True)
    y1 = (got.cpu() - x)[1]
    want_y1 = torch.from_numpy(z[name + "_y1_f16"].astype(np.float32))
    e_y = float((y1 - want_y1).abs().max() / want_y1.abs().max())
    want_m = torch.from_numpy(z[name + "_map"])                                            # [1, T, side, side]
    side = want_m.shape[-1]
    m = pr.cpu()[1:].mean(1).reshape(1, pr.shape[2], side, side)
    e_m = float((m - want_m).abs().max())
    print(f"pre-multiplied attn2 (C = 1280) vs the reference's __call__: output max rel {e_y:.5f}, map max abs {e_m:.6f}")
    report("golden_xattn_premul[C=1280]", out_max_rel=e_y, map_max_abs=e_m)
    assert e_y < 2.0 ** -6, e_y
    assert e_m < 2e-3, e_m
