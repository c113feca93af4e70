"""Training-mode side of the seam: what reference data_generation/finetune_sd_token.py does with
hook.py's recorder (:755-757 install, :1024 clear, :1027 unet call, :1040-1069 attention regulariser, :1089 backward).
HIP kernels vs torch autograd of the oracle's restatements."""
import numpy as np
import pytest
import torch
from _report import report

pytestmark = pytest.mark.gpu


def _rel(got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    return float((got - want).abs().max() / (want.abs().max() + 1e-20))


def test_attention_regulariser_loss_and_gradient_match_autograd():
    """agd_op_attn_reg_loss vs torch autograd through the oracle's line-by-line restatement of finetune_sd_token.py:1046-1066:
    both loss terms per map and d loss / d map (min/max paths, coinciding token rows, samples without an object)."""
    from agenda_amd import ops
    from oracle import sd_oracle as O
    g = torch.Generator().manual_seed(11)
    for (B, T, side, n_obj_emb, idx) in [(3, 77, 16, 1, [[5, 9, 12], [-1, 3, 7], [2, -1, -1]]),
                                         (2, 20, 8, 0, [[4, 6, 8], [1, 2, -1]]),
                                         (2, 77, 64, 2, [[3, 8, 11], [6, 9, 15]])]:
        maps = [torch.rand(B, T, side, side, generator=g).softmax(1) for _ in range(2)]
        idx_t = torch.tensor(idx)
        leaf = [m.clone().requires_grad_(True) for m in maps]
        attn, bg, fg = O.attention_regulariser(leaf, idx_t, n_obj_emb, 0.5)
        attn.backward()
        has = idx_t[:, 0] > 0
        cnt = int(has.sum())
        obj = [int(r[0]) + n_obj_emb if r[0] > 0 else -1 for r in idx_t]
        fgi = [int(r[0]) if r[0] > 0 else -1 for r in idx_t]
        bgi = [int(r[r > -1][-1]) if r[0] > 0 else -1 for r in idx_t]
        tot_bg = tot_fg = 0.0
        for m, lf in zip(maps, leaf):
            loss, dmap = ops.attn_reg_loss(m.cuda(), obj, fgi, bgi, 0.5 / cnt)
            tot_bg += float(loss[:, 0].sum()); tot_fg += float(loss[:, 1].sum())
            want = lf.grad * len(maps)                              # the oracle divided by len(maps) (:1069)
            assert _rel(dmap, want) < 2e-3, (B, T, side, _rel(dmap, want))
            assert float(dmap.cpu()[~has].abs().max() if (~has).any() else 0.0) == 0.0
        assert tot_bg == pytest.approx(float(bg.detach()), rel=1e-4) and tot_fg == pytest.approx(float(fg.detach()), rel=1e-4)
        assert (tot_bg + tot_fg) / len(maps) == pytest.approx(float(attn.detach()), rel=1e-4)


@pytest.fixture(scope="module")
def tiny():
    from agenda_amd import StableDiffusionPipeline, config, synthetic
    cfg = config.tiny()
    u, v = synthetic.make_unet_weights(cfg, 11, bias_std=0.05), synthetic.make_vae_weights(cfg, 12)
    pipe = StableDiffusionPipeline(cfg, u, v, workspace_bytes=1 << 30)
    yield pipe, cfg, u
    pipe.engine.close()


def seam_backward_check(pipe, cfg, u, name, heads, side, is_train, B2=4, tol=0.03):
    """One cross-attention call through the seam with autograd: gradients of (a linear functional of the output + the
    attention regulariser on the recorded map) w.r.t. hidden_states and encoder_hidden_states -- HIP backward
    (recompute P, dS, dQ / dK / dV, input-gradient GEMMs) vs torch autograd of hook.py's restatement (hook.py:91-120)."""
    from agenda_amd import UNetCrossAttentionHooker
    from oracle import sd_oracle as O
    t = name.rsplit("attn2", 1)[0]
    C, T = u[t + "attn2.to_q.weight"].shape[0], 77
    g = torch.Generator().manual_seed(5 + int(is_train) + side)
    hidden = torch.randn(B2, side * side, C, generator=g).to(torch.bfloat16).float()
    ctx = torch.randn(B2, T, cfg.unet.cross_attention_dim, generator=g).to(torch.bfloat16).float()
    wout = torch.randn(B2, side * side, C, generator=g) * 0.01
    idx = torch.tensor([[4, 7, 9], [2, 5, -1], [-1, 3, 4], [6, 8, 12]])[: (B2 if is_train else B2 // 2)]

    def loss_of(y, maps):
        attn, _, _ = O.attention_regulariser(maps, idx, 1, 0.5)
        return (y * wout.to(y.device)).sum() + 100.0 * attn.to(y.device)

    # oracle: torch autograd end to end
    h0, c0 = hidden.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
    rec = O.HookRecorder(is_train=is_train, latent_hw=side)
    y0 = O.explicit_attention_processor(h0, c0, u[t + "attn2.to_q.weight"], u[t + "attn2.to_k.weight"], u[t + "attn2.to_v.weight"],
                                        u[t + "attn2.to_out.0.weight"], u[t + "attn2.to_out.0.bias"], heads, recorder=rec)
    loss_of(y0, rec.cross_attn_maps).backward()
    # HIP: the same loss on the seam's autograd-connected outputs
    hk = UNetCrossAttentionHooker(is_train=is_train, latent_hw=side)
    pipe.unet.set_attn_processor(hk)
    try:
        h1, c1 = hidden.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
        y1 = hk(pipe.unet.attn2(name), h1, c1)
        assert len(hk.cross_attn_maps) == 1 and hk.cross_attn_maps[0].requires_grad
        assert _rel(y1, y0) < 2.0 ** -6 and float((hk.cross_attn_maps[0].detach().cpu() - rec.cross_attn_maps[0].detach()).abs().max()) < 2e-3
        loss_of(y1, [m.cpu() for m in hk.cross_attn_maps]).backward()
    finally:
        pipe.unet.set_attn_processor(pipe.unet._default)
    e_h, e_c = _rel(h1.grad, h0.grad), _rel(c1.grad, c0.grad)
    print(f"seam backward {name} (C={C}, heads={heads}, N={side * side}, is_train={is_train}): d_hidden rel {e_h:.4f}, d_ctx rel {e_c:.4f}")
    report(f"seam_backward[{name},is_train={is_train}]", d_hidden_rel=e_h, d_ctx_rel=e_c)
    # bf16 operands (Q, K, V, dO, dQ, dK, dV) / fp32 arithmetic vs fp32 autograd
    assert e_h < tol and e_c < tol, (e_h, e_c)
    if not is_train:                                           # hook.py:48-49: the unconditional half never reaches the map
        assert float(h1.grad[: B2 // 2].abs().max()) > 0       # ... but still receives the gradient through the output


@pytest.mark.parametrize("is_train", [True, False])
def test_seam_call_backward_matches_torch_autograd(tiny, is_train):
    pipe, cfg, u = tiny
    seam_backward_check(pipe, cfg, u, "up_blocks.2.attentions.1.transformer_blocks.0.attn2", 2, 16, is_train)


def test_train_mode_fused_walk_keeps_the_sixteen_maps(tiny):
    """finetune_sd_token.py:1024-1069 shape of use: clear(), one `unet(x, t, ctx)` call WITHOUT CFG (any batch, here 3), then
    read `cross_attn_maps` (16 per-call maps [B, T, h, w], call order, hook.py:110-112) and the attention regulariser."""
    from agenda_amd import UNetCrossAttentionHooker, synthetic
    from oracle import sd_oracle as O
    pipe, cfg, u = tiny
    B, L = 3, 16
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, 4, L, L, generator=g).to(torch.bfloat16).float()
    ctx = torch.randn(B, 77, cfg.unet.cross_attention_dim, generator=g).to(torch.bfloat16).float()
    hk = UNetCrossAttentionHooker(is_train=True, latent_hw=L)
    pipe.unet.set_attn_processor(hk)
    hk.clear()
    eps = pipe.unet(x, 321, ctx).sample
    rec = O.HookRecorder(is_train=True, latent_hw=L)
    with torch.no_grad():
        want = O.unet_forward(u, cfg.unet, x, torch.tensor(321), ctx, rec)
    assert float(((eps.cpu() - want) ** 2).mean().sqrt() / (want ** 2).mean().sqrt()) < 2.0 ** -6
    maps = hk.cross_attn_maps
    assert len(maps) == len(rec.cross_attn_maps) == 16
    for m, w in zip(maps, rec.cross_attn_maps):
        # after up to ~60 stacked bf16 layers the probabilities carry the activations' rounding: 3 % of the map's peak
        assert m.shape == w.shape and float((m.cpu() - w).abs().max()) < 0.03 * float(w.max()), float((m.cpu() - w).abs().max())
    idx = torch.tensor([[4, 7, 9], [-1, 2, 3], [5, 6, -1]])
    attn, bg, fg, grads = hk.attention_regulariser(idx, 1, 0.5, want_grads=True)
    w_attn, w_bg, w_fg = O.attention_regulariser(rec.cross_attn_maps, idx, 1, 0.5)
    assert float(attn) == pytest.approx(float(w_attn), rel=0.05) and float(bg) == pytest.approx(float(w_bg), rel=0.05)
    assert len(grads) == 16 and grads[0].shape == maps[0].shape and float(grads[0][1].abs().max()) == 0.0   # sample 1 has no object
    # a second forward appends 16 more (the reference's list keeps growing until clear(), :1024)
    pipe.unet(x, 300, ctx)
    assert len(hk.cross_attn_maps) == 32
    hk.clear()
    assert len(hk.cross_attn_maps) == 0
    with pytest.raises(RuntimeError, match="No heat maps"):
        hk.attention_regulariser(idx, 1, 0.5)
    pipe.unet.set_attn_processor(pipe.unet._default)


def test_hook_mode_recording_is_reproducible(tiny):
    """The head mean of hook.py:55 is an ordered sum now (no float atomics across head workgroups): two runs give the same
    global heat map bit for bit."""
    from agenda_amd import UNetCrossAttentionHooker, synthetic
    pipe, cfg, u = tiny
    ctx = synthetic.make_context(cfg, 2, seed=4)
    lat = synthetic.make_latents(cfg, [1, 2], 16)
    hk = UNetCrossAttentionHooker(is_train=False, latent_hw=16)
    pipe.unet.set_attn_processor(hk)
    outs = []
    for _ in range(2):
        pipe(prompt_embeds=ctx, latents=lat, num_inference_steps=2, output_type="latent")
        outs.append(hk.compute_global_heat_map().clone())
    assert torch.equal(outs[0], outs[1])
    pipe.unet.set_attn_processor(pipe.unet._default)


def test_unet_training_call_signature_with_per_sample_timesteps(tiny):
    """finetune_sd_token.py:1027: `unet(noisy_latents, timesteps, encoder_hidden_states, class_labels=None, return_dict=False)[0]`
    with a [bsz] timestep tensor (train_batch_size 4, one random timestep per sample): every image gets its own time-embedding row."""
    from oracle import sd_oracle as O
    pipe, cfg, u = tiny
    B, L = 4, 16
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, 4, L, L, generator=g).to(torch.bfloat16).float()
    ctx = torch.randn(B, 77, cfg.unet.cross_attention_dim, generator=g).to(torch.bfloat16).float()
    ts = torch.tensor([981, 12, 500, 333])
    got = pipe.unet(x, ts, ctx, class_labels=None, return_dict=False)
    assert isinstance(got, tuple) and len(got) == 1 and got[0].shape == x.shape
    with torch.no_grad():
        want = O.unet_forward(u, cfg.unet, x, ts, ctx)
    rel = float(((got[0].cpu() - want) ** 2).mean().sqrt() / (want ** 2).mean().sqrt())
    assert rel < 2.0 ** -6, rel
    # the rows really differ per image: with one shared timestep the other images come out different
    same = pipe.unet(x, 981, ctx).sample
    assert float((same[0] - got[0][0]).abs().max()) == 0.0 and float((same[1] - got[0][1]).abs().max()) > 1e-3
    with pytest.raises(ValueError):
        pipe.unet(x, torch.tensor([1, 2, 3]), ctx)
    with pytest.raises(NotImplementedError):
        pipe.unet(x, ts, ctx, class_labels=torch.zeros(B))


def test_attention_regulariser_rejects_out_of_range_token_rows(tiny):
    """finetune_sd_token.py:1049-1060 index the map rows directly: a token index beyond the map's rows is an IndexError, not a
    silently disabled loss."""
    from agenda_amd import UNetCrossAttentionHooker
    pipe, cfg, u = tiny
    B, L = 2, 16
    g = torch.Generator().manual_seed(23)
    x = torch.randn(B, 4, L, L, generator=g)
    ctx = torch.randn(B, 77, cfg.unet.cross_attention_dim, generator=g)
    hk = UNetCrossAttentionHooker(is_train=True, latent_hw=L)
    pipe.unet.set_attn_processor(hk)
    try:
        hk.clear()
        pipe.unet(x, 5, ctx)
        hk.attention_regulariser(torch.tensor([[4, 7, 9], [-1, 2, 3]]), 1, 0.5)          # fine
        with pytest.raises(IndexError):
            hk.attention_regulariser(torch.tensor([[76, 7, 9], [-1, 2, 3]]), 1, 0.5)     # obj row 77 of 77
        with pytest.raises(IndexError):
            hk.attention_regulariser(torch.tensor([[4, 7, 80], [-1, 2, 3]]), 1, 0.5)
    finally:
        pipe.unet.set_attn_processor(pipe.unet._default)
