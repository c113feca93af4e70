"""Full-size (SD-1.5 shapes) GPU tests.
 * config 1 (256 px) UNet forward + DAAM step vs the CPU oracle;
 * config 2 (512 px, batch 4) through size-independent properties: cross-attention probability mass is
   conserved in the DAAM maps, runs are bitwise deterministic, an image does not depend on its batch."""
import numpy as np
import pytest
import torch

from _report import report

pytestmark = pytest.mark.gpu
_TB = 7935          # the engine's default "tblock_fuse" mask (model.hip: bits 0 - 7, 9, 10 and round 6's bits 11 = 64-row block-head panels, 12 = its new schedule)


def _rms_rel(got, want):
    got = got.detach().float().cpu()
    return float(((got - want) ** 2).mean().sqrt() / ((want ** 2).mean().sqrt() + 1e-12))


def _norm_map_diff_255(got, want):
    lo, hi = got.amin((-1, -2), keepdim=True), got.amax((-1, -2), keepdim=True)
    wlo, whi = want.amin((-1, -2), keepdim=True), want.amax((-1, -2), keepdim=True)
    return ((got - lo) / (hi - lo + 1e-8) - (want - wlo) / (whi - wlo + 1e-8)).abs() * 255


def _norm_map_err_255(got, want):
    d = _norm_map_diff_255(got, want)
    return float(d.max()), float(d.mean())


def _norm_map_stats_255(got, want):
    """max, 99.9th percentile and mean of the error of the min-max-normalised maps (what data_generation.py:82-84 exports), in 1/255.  The max is ONE pixel of a
    normalised low-contrast row and moves between 6 and 12 with any change of summation order (profiles/r06_bisect_config1.txt); the percentile and the mean
    do not -- they are the robust figures."""
    d = _norm_map_diff_255(got, want)
    return float(d.max()), float(d.flatten().float().quantile(0.999)), float(d.mean())


@pytest.fixture(scope="module")
def sd15_cuda():
    from agenda_amd import StableDiffusionPipeline
    return StableDiffusionPipeline.from_synthetic("sd15", seed=1234, weights_device="cuda", workspace_bytes=12 << 30)


def test_sd15_unet_forward_256px_matches_oracle():
    """BASELINE config 1 shapes: SD-1.5, 1x256x256 (latent 32), CFG batch 2, one UNet forward + DAAM record."""
    from agenda_amd import StableDiffusionPipeline, config, synthetic, trace
    from oracle import sd_oracle as O
    cfg = config.sd15()
    u = synthetic.make_unet_weights(cfg, 1234)
    v = synthetic.make_vae_weights(cfg, 1235)
    pipe = StableDiffusionPipeline(cfg, u, v, workspace_bytes=4 << 30)
    ctx = synthetic.make_context(cfg, 1, seed=7)
    lat = synthetic.make_latents(cfg, [0], 32)
    x = torch.cat([lat, lat]).to(torch.bfloat16).float()
    rec = O.DaamRecorder(32 * 32, 77)
    with torch.no_grad():
        want = O.unet_forward(u, cfg.unet, x, torch.tensor(981), ctx, rec)
    pipe.engine.set_context(ctx)
    pipe.engine.record_config(1, False, 77)
    pipe.engine.record_reset(1, 32)
    got = pipe.engine.unet_forward(x, 981.0)
    hm = pipe.engine.daam_global(0, 77, 32).cpu()
    whm = rec.compute_global_heat_map()[0]
    report("config1_forward_256px", rms_rel=_rms_rel(got, want), heat_map_rel=float((hm - whm).abs().max() / whm.abs().max()),
           norm_map_max_255=_norm_map_err_255(hm, whm)[0])
    assert _rms_rel(got, want) < 2.0 ** -6, _rms_rel(got, want)
    assert len(rec.acc) == 15 * 8                                  # 15 recorded attn2 layers x 8 heads
    assert float((hm - whm).abs().max() / whm.abs().max()) < 0.02
    pipe.engine.record_config(0)
    pipe.engine.close()


def test_sd15_512px_batch4_properties(sd15_cuda):
    from agenda_amd import synthetic, trace
    pipe = sd15_cuda
    cfg = pipe.cfg
    B, L, steps = 4, 64, 3
    ctx = synthetic.make_context(cfg, B, seed=7)
    lat = synthetic.make_latents(cfg, [10, 11, 12, 13], L)

    def run(ctx_, lat_):
        with trace(pipe) as trc:
            out = pipe(prompt_embeds=ctx_, latents=lat_, num_inference_steps=steps, output_type="pt")
            maps = torch.stack([trc.compute_global_heat_map(image_index=i).heat_maps for i in range(lat_.shape[0])])
        return out.images.clone(), out.latents.clone(), maps

    img, latents, maps = run(ctx, lat)
    assert img.shape == (B, 512, 512, 3) and img.dtype == torch.uint8
    assert torch.isfinite(latents).all() and torch.isfinite(maps).all()
    assert maps.shape == (B, 77, 64, 64) and float(maps.min()) >= 0.0
    # every attn2 softmax row sums to 1, bicubic is a partition of unity, so summed over the 77 tokens the
    # global map equals the number of denoise steps (clamping of undershoot can only add a little)
    tot = maps.sum(1)
    assert float((tot - steps).abs().max()) < 0.02 * steps, float((tot - steps).abs().max())
    # bitwise deterministic (no atomics on the DAAM path, split-K reduce is ordered)
    img2, latents2, maps2 = run(ctx, lat)
    assert torch.equal(img, img2) and torch.equal(latents, latents2) and torch.equal(maps, maps2)
    # batch invariance: image 2 alone == image 2 in the batch of 4 (tile/split choices may differ -> tolerance)
    c1 = torch.cat([ctx[2:3], ctx[B + 2:B + 3]])
    img1, lat1, maps1 = run(c1, lat[2:3])
    assert _rms_rel(lat1[0], latents[2].cpu()) < 0.05
    d = (img1[0].float() - img[2].float()).abs()
    assert float(d.mean()) < 4.0, float(d.mean())      # fp summation order differs (split-K / tile choice), measured 2.2
    assert float((maps1[0] - maps[2]).abs().max() / maps[2].abs().max()) < 0.05


def _psnr_u8(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    mse = ((a - b) ** 2).mean()
    return float("inf") if mse == 0 else float(10 * np.log10(255.0 ** 2 / mse))


@pytest.fixture(scope="module")
def sd15_host_weights():
    """The same seeded SD-1.5-shaped weights on the host (for the CPU oracle) -- incl. the VAE encoder (config 3)."""
    from agenda_amd import config, synthetic
    cfg = config.sd15()
    return cfg, synthetic.make_unet_weights(cfg, 1234), synthetic.make_vae_weights(cfg, 1235, with_encoder=True)


@pytest.fixture(scope="module")
def sd15_pipe(sd15_host_weights):
    from agenda_amd import StableDiffusionPipeline
    cfg, u, v = sd15_host_weights
    pipe = StableDiffusionPipeline(cfg, u, v, workspace_bytes=24 << 30)      # room for config 3's per-GPU share (UNet batch 16 at 512 px)
    yield pipe
    pipe.engine.close()


def test_cold_weight_warmup_options_do_not_change_results(sd15_cuda):
    """`weight_warm` (in-kernel streaming of the weight matrix ahead of the main loops) and `weight_touch` (a read-only launch in
    front of a weight-heavy launch) only move bytes through the caches: the UNet output must be bit-identical with them off."""
    from agenda_amd import synthetic
    pipe = sd15_cuda
    cfg = pipe.cfg
    ctx = synthetic.make_context(cfg, 1, seed=11)
    lat = synthetic.make_latents(cfg, [3], 64)
    x = torch.cat([lat, lat]).to(torch.bfloat16).float()
    pipe.engine.set_context(ctx)
    outs = []
    for warm, touch in ((3, 3), (0, 0), (1, 3), (0, 3)):
        pipe.engine.set_option("weight_warm", warm)
        pipe.engine.set_option("weight_touch", touch)
        outs.append(pipe.engine.unet_forward(x, 500.0).clone())
    pipe.engine.set_option("weight_warm", 3)
    pipe.engine.set_option("weight_touch", 3)
    assert torch.isfinite(outs[0]).all()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


_ORACLE_CACHE = {}


@pytest.mark.parametrize("p8", [1, 4])
def test_sd15_unet_forward_512px_matches_oracle(sd15_host_weights, sd15_pipe, p8):
    """BASELINE config 2 at ITS OWN size: SD-1.5, 512 px (latent 64, 4096 tokens), CFG batch 2, one UNet forward with the
    DAAM recorder on -- the forward bench.py times -- against the fp32 CPU oracle (data_generation.py:57-64 semantics).
    p8 = 4: every launch the 256-row 8-wave / 8-phase igemm (igemm8p.h) can legally take goes through it (concat and stride-2
    convs, time-embedding row add, LayerNorm-fold producers / consumers, GroupNorm partial sums per 256-row tile, GEGLU), not
    only the ones the launcher would pick at this batch size."""
    from agenda_amd import synthetic
    from oracle import sd_oracle as O
    cfg, u, v = sd15_host_weights
    pipe, L = sd15_pipe, 64
    ctx = synthetic.make_context(cfg, 1, seed=7)
    lat = synthetic.make_latents(cfg, [0], L)
    x = torch.cat([lat, lat]).to(torch.bfloat16).float()
    if "unet512" not in _ORACLE_CACHE:
        rec = O.DaamRecorder(L * L, 77)
        with torch.no_grad():
            want = O.unet_forward(u, cfg.unet, x, torch.tensor(981), ctx, rec)
        _ORACLE_CACHE["unet512"] = (want, rec.compute_global_heat_map()[0], len(rec.acc))
    want, whm, nacc = _ORACLE_CACHE["unet512"]
    pipe.engine.set_option("igemm8p", p8)
    try:
        pipe.engine.set_context(ctx)
        pipe.engine.record_config(1, False, 77)
        pipe.engine.record_reset(1, L)
        got = pipe.engine.unet_forward(x, 981.0)
        hm = pipe.engine.daam_global(0, 77, L).cpu()
    finally:
        pipe.engine.set_option("igemm8p", 1)
        pipe.engine.record_config(0)
    err = _rms_rel(got, want)
    hm_err = float((hm - whm).abs().max() / whm.abs().max())
    print(f"config2 forward (512 px, igemm8p={p8}): rms rel {err:.5f}, heat map rel {hm_err:.4f}")
    report(f"config2_forward_512px_cfg_pair[igemm8p={p8}]", rms_rel=err, heat_map_rel=hm_err, norm_map_max_255=_norm_map_err_255(hm, whm)[0])
    assert nacc == 15 * 8                                          # 15 recorded attn2 layers x 8 heads
    assert err < 2.0 ** -6, err                                    # bf16 storage / fp32 accumulate vs the fp32 oracle
    assert hm_err < 0.02, hm_err


def test_fused_transformer_block_kernels_match_the_kernel_chain_512px(sd15_host_weights, sd15_pipe):
    """opt "tblock_fuse" (tblock.hip, the C = 320 blocks of the 64 x 64 maps): bit 0 norm3 -> GEGLU -> ff.net.2 + residual in one launch, bit 1
    norm2 -> to_q -> cross-attention (+ DAAM record) -> to_out + residual in one launch, bit 2 that launch starting at attn1.to_out, bit 3 the
    feed-forward launch ending with proj_out (+ the next GroupNorm's partial sums), bit 4 proj_in -> norm1 -> q / k / v in one launch, bit 5 the attn2 chain for the C = 640 blocks (32 x 32 maps) too, bit 7 the transformer's GroupNorm applied inside the bit-4 launch (no fold launch), bit 9 ff.net.2 / proj_out pre-multiplied inside the feed-forward launch; opt "reduce_gn": split-K slab sum + GroupNorm in one launch.
    Every combination against the fp32 oracle (UNet output and DAAM heat maps) and against the unfused kernel chain: the same fp32 function
    with bf16 roundings at different points, so within bf16 noise of each other, each within the oracle bound."""
    from agenda_amd import synthetic
    from oracle import sd_oracle as O
    cfg, u, v = sd15_host_weights
    pipe, L = sd15_pipe, 64
    ctx = synthetic.make_context(cfg, 1, seed=7)
    lat = synthetic.make_latents(cfg, [0], L)
    x = torch.cat([lat, lat]).to(torch.bfloat16).float()
    if "unet512" not in _ORACLE_CACHE:
        rec = O.DaamRecorder(L * L, 77)
        with torch.no_grad():
            want = O.unet_forward(u, cfg.unet, x, torch.tensor(981), ctx, rec)
        _ORACLE_CACHE["unet512"] = (want, rec.compute_global_heat_map()[0], len(rec.acc))
    want, whm, _ = _ORACLE_CACHE["unet512"]
    outs = {}
    try:
        for fuse, rg in ((0, 0), (1, 1), (2, 1), (6, 1), (9, 1), (16, 1), (31, 1), (34, 1), (63, 1), (63, 0), (144, 1), (176, 1), (255, 1), (767, 1), (1791, 1), (1791 | 2048, 1),
                         (1791 | 4096, 1), (_TB, 1)):
            pipe.engine.set_option("tblock_fuse", fuse)
            pipe.engine.set_option("reduce_gn", rg)
            pipe.engine.set_context(ctx)
            pipe.engine.record_config(1, False, 77)
            pipe.engine.record_reset(1, L)
            got = pipe.engine.unet_forward(x, 981.0).clone()
            hm = pipe.engine.daam_global(0, 77, L).cpu()
            outs[(fuse, rg)] = (got, hm)
            if (fuse, rg) == (63, 1):                                  # run-to-run identical (no atomics anywhere on the fused paths)
                pipe.engine.record_reset(1, L)
                again = pipe.engine.unet_forward(x, 981.0)
                assert torch.equal(got, again) and torch.equal(hm, pipe.engine.daam_global(0, 77, L).cpu())
    finally:
        pipe.engine.set_option("tblock_fuse", _TB)
        pipe.engine.set_option("reduce_gn", 1)
        pipe.engine.record_config(0)
    # round 6: the block head on 64-row panels (bit 11) and on its new schedule (bit 12) moves no arithmetic: bit-identical
    for k in ((1791 | 2048, 1), (1791 | 4096, 1), (_TB, 1)):
        assert torch.equal(outs[k][0], outs[(1791, 1)][0]) and torch.equal(outs[k][1], outs[(1791, 1)][1]), k
    base, bhm = outs[(0, 0)]
    for key, (got, hm) in outs.items():
        e_o, e_b = _rms_rel(got, want), _rms_rel(got, base.cpu())
        h_o = float((hm - whm).abs().max() / whm.abs().max())
        print(f"tblock_fuse={key[0]:2d} reduce_gn={key[1]}: vs oracle {e_o:.5f}, vs kernel chain {e_b:.5f}, heat map vs oracle {h_o:.4f}")
        report(f"config2_forward_512px_fused_kernels[tblock_fuse={key[0]},reduce_gn={key[1]}]", rms_rel=e_o, vs_kernel_chain=e_b, heat_map_rel=h_o)
        assert e_o < 2.0 ** -6, (key, e_o)
        assert e_b < 2.0 ** -5, (key, e_b)
        assert h_o < 0.02, (key, h_o)


@pytest.mark.parametrize("side,B", [(384, 3), (640, 1), (256, 5)])
def test_merged_launches_at_odd_sizes_match_the_unmerged_walk(sd15_host_weights, sd15_pipe, side, B):
    """Shapes the bench never runs -- 48 / 80 / 32-pixel latent maps, batches 3 / 1 / 5 -- through every round-4 merge at its default (fused row-panel kernels, slab pass + GroupNorm,
    shortcut / ff-proj / upsampling merges, K groups, weight-streaming kernel) against the same two denoise steps with all of them off: each merge decides from the launch's
    shape whether it applies (row-halo geometry, tiles inside one image or phase, partial sums present), so an odd size must fall back cleanly, never fault or diverge.  Two valid
    realisations of the same steps: classifier-free guidance multiplies their bf16 noise (cf. the shared-prefix test), the bound is calibrated there."""
    from agenda_amd import synthetic
    pipe = sd15_pipe
    cfg = pipe.cfg
    L = side // 8
    ctx = synthetic.make_context(cfg, B, seed=side)
    lat = synthetic.make_latents(cfg, list(range(B)), L)
    off = {"tblock_fuse": 0, "reduce_gn": 0, "shortcut_fuse": 0, "ff_proj_fuse": 0, "upsample_phases": 0, "igemm_kgroups": 0, "wreg_mask": 0, "conv_smap": 0, "attn2_premul": 0,
           "igemm_pc": 0, "xcd_block": 0}
    on = {"tblock_fuse": _TB, "reduce_gn": 1, "shortcut_fuse": 3, "ff_proj_fuse": 1, "upsample_phases": 7, "igemm_kgroups": 1, "wreg_mask": 3, "conv_smap": 1, "attn2_premul": 1,
          "igemm_pc": 49, "xcd_block": 1}

    def run():
        return pipe(prompt_embeds=ctx, latents=lat, num_inference_steps=2, height=side, width=side, output_type="latent").latents.clone()

    try:
        a = run()
        for k, v in off.items():
            pipe.engine.set_option(k, v)
        b = run()
    finally:
        for k, v in on.items():
            pipe.engine.set_option(k, v)
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    e = _rms_rel(a, b.cpu())
    print(f"{side} px, batch {B}: merged vs unmerged walk, two steps: latents rms rel {e:.5f}")
    report(f"odd_size_merged_vs_unmerged[{side}px,B={B}]", latents_rms_rel=e)
    assert e < 0.08, e
    if True:
        # ... and "is right", not only "agrees with itself" (every size now): the same two CFG steps through the fp32 oracle (batch 5 at
        # 256 px is seconds of host work, batch 3 at 384 px and batch 1 at 640 px under a minute each); both walks within the two-step bound of the CFG-pair tests
        # (classifier-free guidance multiplies the bf16 noise of eps)
        from oracle import sd_oracle as O
        cfg, u, v = sd15_host_weights
        _, want = O.generate(u, v, cfg, ctx, lat, 2, 7.5, decode=False)
        e_a, e_b = _rms_rel(a, want), _rms_rel(b, want)
        print(f"  vs the oracle's two steps: merged {e_a:.5f}, unmerged {e_b:.5f}")
        report(f"odd_size_merged_vs_unmerged[{side}px,B={B}]", merged_vs_oracle=e_a, unmerged_vs_oracle=e_b)
        assert e_a < 0.05 and e_b < 0.05, (e_a, e_b)


def test_cfg_shared_prefix_inside_the_fused_kernels_512px(sd15_pipe):
    """`cfg_shared_prefix` with the fused block kernels behind it (tblock_fuse bit 6): the B' shared rows are not copied -- the attn2 chain reads input
    row m % M' (and starts at attn1.to_out), the feed-forward's proj_out stage adds block-input row m % M'.  Two denoise steps at 512 px against the
    same run with explicit copies (bit 6 off) and against the run that shares nothing: every row sees the same inputs either way, the kernels that
    produce them differ (fused vs separate to_out), so agreement is to bf16 noise (amplified by the guidance scale: the bound is on two steps'
    latents, calibrated against the distance between the two older forms), and the lazy form is run-to-run identical."""
    from agenda_amd import synthetic, trace
    pipe = sd15_pipe
    cfg = pipe.cfg
    ctx = synthetic.make_context(cfg, 2, seed=5)
    lat = synthetic.make_latents(cfg, [3, 4], 64)

    def run():
        with trace(pipe) as trc:
            out = pipe(prompt_embeds=ctx, latents=lat, num_inference_steps=2, output_type="latent")
            maps = torch.stack([trc.compute_global_heat_map(image_index=i).heat_maps for i in range(2)])
        return out.latents.clone(), maps.clone()

    try:
        a = run(); a2 = run()
        pipe.engine.set_option("tblock_fuse", _TB & ~64)
        b = run()
        pipe.engine.set_option("cfg_shared_prefix", 0)
        c0 = run()
    finally:
        pipe.engine.set_option("tblock_fuse", _TB)
        pipe.engine.set_option("cfg_shared_prefix", 1)
    assert torch.equal(a[0], a2[0]) and torch.equal(a[1], a2[1])
    # calibration: how far two VALID realisations of the same two steps drift apart (classifier-free guidance multiplies the bf16 noise of eps by ~10)
    base = _rms_rel(b[0], c0[0].cpu())
    for other, name in ((b, "copies"), (c0, "unshared")):
        e_l = _rms_rel(a[0], other[0].cpu()); e_m = float((a[1] - other[1]).abs().max() / other[1].abs().max())
        print(f"lazy shared prefix vs {name}: latents rms rel {e_l:.5f} (copies vs unshared: {base:.5f}), heat maps {e_m:.4f}")
        report(f"cfg_shared_prefix_lazy_vs_{name}", latents_rms_rel=e_l, heat_map_rel=e_m)
        assert e_l < 0.08 and e_m < 0.02, (name, e_l, e_m)
    # sharing nothing runs the very same kernels on every row (attn1.to_out inside the chain either way): the lazy form is bit-identical to it
    assert torch.equal(a[0], c0[0]) and torch.equal(a[1], c0[1])


def test_sd15_unet_forward_512px_batch4_matches_oracle(sd15_host_weights, sd15_pipe):
    """BASELINE config 2 at the bench's OWN batch: four images (UNet batch 8, M = 32768 at 64 x 64), one forward with the DAAM
    recorder on, against the fp32 CPU oracle.  At this batch the launcher's choices differ from the CFG-pair test above (the
    8-phase igemm on qkv / GEGLU / the upsampling convs, other tile and split-K choices), so this is the oracle check of the
    kernels the benchmark actually runs."""
    from agenda_amd import synthetic
    from oracle import sd_oracle as O
    cfg, u, v = sd15_host_weights
    pipe, L, B = sd15_pipe, 64, 4
    ctx = synthetic.make_context(cfg, B, seed=21)
    lat = synthetic.make_latents(cfg, [30, 31, 32, 33], L)
    x = torch.cat([lat, lat]).to(torch.bfloat16).float()
    rec = O.DaamRecorder(L * L, 77)
    with torch.no_grad():
        want = O.unet_forward(u, cfg.unet, x, torch.tensor(601), ctx, rec)
    whm = rec.compute_global_heat_map()
    pipe.engine.set_context(ctx)
    pipe.engine.record_config(1, False, 77)
    pipe.engine.record_reset(B, L)
    try:
        got = pipe.engine.unet_forward(x, 601.0).clone()
        hm = torch.stack([pipe.engine.daam_global(i, 77, L).cpu() for i in range(B)])
        # tblock_fuse bit 8 (off by default: measured slower): proj_in -> norm1 -> q / k / v with the GroupNorm inside for the C = 640 blocks too
        # (only at this batch do the 32 x 32 convs leave the partial sums the kernel needs)
        pipe.engine.set_option("tblock_fuse", _TB | 256)
        pipe.engine.record_reset(B, L)
        got8 = pipe.engine.unet_forward(x, 601.0).clone()
        hm8 = torch.stack([pipe.engine.daam_global(i, 77, L).cpu() for i in range(B)])
        # shortcut_fuse off: the resnets' 1x1 conv_shortcut as its own launch (its bf16-rounded output added as conv2's residual) instead of extra K of conv2
        pipe.engine.set_option("tblock_fuse", _TB)
        pipe.engine.set_option("shortcut_fuse", 0)
        pipe.engine.record_reset(B, L)
        got_s = pipe.engine.unet_forward(x, 601.0).clone()
        # ff_proj_fuse off: ff.net.2 (+ residual) and proj_out (+ block residual) as two launches instead of one GEMM with the pre-multiplied matrix [Wp W2 | Wp]
        pipe.engine.set_option("shortcut_fuse", 3)
        pipe.engine.set_option("ff_proj_fuse", 0)
        pipe.engine.record_reset(B, L)
        got_f = pipe.engine.unet_forward(x, 601.0).clone()
        # upsample_phases off: the nearest-2x upsampling convs as 3x3 convs on the upsampled map (the gather folds the upsample) instead of four 2x2 phase convs on the source map
        pipe.engine.set_option("ff_proj_fuse", 1)
        pipe.engine.set_option("upsample_phases", 0)
        pipe.engine.record_reset(B, L)
        got_u = pipe.engine.unet_forward(x, 601.0).clone()
        pipe.engine.set_option("upsample_phases", 7)
        # round 5: attn2_premul off -- attn2 of the C = 1280 blocks as to_q, the attention kernel and to_out instead of two GEMMs against per-image pre-multiplied context matrices
        pipe.engine.set_option("attn2_premul", 0)
        pipe.engine.set_context(ctx)                               # (the products are built in agd_set_context)
        pipe.engine.record_reset(B, L)
        got_p = pipe.engine.unet_forward(x, 601.0).clone()
        hm_p = torch.stack([pipe.engine.daam_global(i, 77, L).cpu() for i in range(B)])
        pipe.engine.set_option("attn2_premul", 1)
        pipe.engine.set_context(ctx)
        # round 5: the producer / consumer kernels (igemm_pc.h, igemm_pch.h) and the XCD tile blocks compute the same tiles with the same summation order: bit-identical outputs
        pipe.engine.set_option("igemm_pc", 0)
        pipe.engine.set_option("xcd_block", 0)
        pipe.engine.record_reset(B, L)
        got_x = pipe.engine.unet_forward(x, 601.0).clone()
    finally:
        pipe.engine.set_option("igemm_pc", 49)
        pipe.engine.set_option("xcd_block", 1)
        pipe.engine.set_option("attn2_premul", 1)
        pipe.engine.set_option("upsample_phases", 7)
        pipe.engine.set_option("tblock_fuse", _TB)
        pipe.engine.set_option("shortcut_fuse", 3)
        pipe.engine.set_option("ff_proj_fuse", 1)
        pipe.engine.record_config(0)
    err = _rms_rel(got, want)
    hm_err = float((hm - whm).abs().max() / whm.abs().max())
    err8, hm_err8, d8 = _rms_rel(got8, want), float((hm8 - whm).abs().max() / whm.abs().max()), _rms_rel(got8, got.cpu())
    print(f"config2 forward at batch 4 (UNet batch 8, 512 px): rms rel {err:.5f}, heat map rel {hm_err:.4f}; with the C = 640 qkv chain: {err8:.5f}, {hm_err8:.4f} (vs default {d8:.5f})")
    report("config2_forward_512px_batch4", rms_rel=err, heat_map_rel=hm_err, norm_map_max_255=_norm_map_err_255(hm, whm)[0], qkv640_rms_rel=err8, qkv640_heat_map_rel=hm_err8)
    assert err < 2.0 ** -6, err
    assert hm_err < 0.02, hm_err
    assert err8 < 2.0 ** -6 and hm_err8 < 0.02 and 0 < d8 < 2.0 ** -5, (err8, hm_err8, d8)
    err_s, d_s = _rms_rel(got_s, want), _rms_rel(got_s, got.cpu())
    print(f"  conv_shortcut as its own launch: {err_s:.5f} vs the oracle, {d_s:.5f} vs the fused form")
    report("config2_forward_512px_batch4", shortcut_unfused_rms_rel=err_s, shortcut_unfused_vs_default=d_s)
    assert err_s < 2.0 ** -6 and 0 < d_s < 2.0 ** -5, (err_s, d_s)
    err_f, d_f = _rms_rel(got_f, want), _rms_rel(got_f, got.cpu())
    print(f"  ff.net.2 and proj_out as two launches: {err_f:.5f} vs the oracle, {d_f:.5f} vs the pre-multiplied form")
    report("config2_forward_512px_batch4", ffproj_unfused_rms_rel=err_f, ffproj_unfused_vs_default=d_f)
    assert err_f < 2.0 ** -6 and 0 < d_f < 2.0 ** -5, (err_f, d_f)
    err_u, d_u = _rms_rel(got_u, want), _rms_rel(got_u, got.cpu())
    print(f"  upsampling convs on the upsampled map: {err_u:.5f} vs the oracle, {d_u:.5f} vs the phase form")
    report("config2_forward_512px_batch4", upsample_3x3_rms_rel=err_u, upsample_3x3_vs_default=d_u)
    assert err_u < 2.0 ** -6 and 0 < d_u < 2.0 ** -5, (err_u, d_u)
    err_p, d_p = _rms_rel(got_p, want), _rms_rel(got_p, got.cpu())
    hm_p_err = float((hm_p - whm).abs().max() / whm.abs().max())
    print(f"  attn2 of the C = 1280 blocks as the kernel chain: {err_p:.5f} vs the oracle (heat map {hm_p_err:.4f}), {d_p:.5f} vs the pre-multiplied form")
    report("config2_forward_512px_batch4", attn2_chain_rms_rel=err_p, attn2_chain_heat_map_rel=hm_p_err, attn2_chain_vs_default=d_p)
    assert err_p < 2.0 ** -6 and hm_p_err < 0.02 and 0 < d_p < 2.0 ** -5, (err_p, hm_p_err, d_p)
    assert torch.equal(got_x, got)                                 # igemm_pc / xcd_block off: the same bits


def _oracle_cfg_pairs(u, ucfg, x, t, ctx, L, tokens=77):
    """The fp32 oracle's forward of a CFG batch [uncond..., cond...], one (uncond, cond) pair at a time -- images are independent in the UNet
    (per-sample GroupNorm, per-sample attention), and a pair keeps the explicit [B H, N, N] softmax of hook.py:108 at a few GB of host memory.
    Returns the outputs in the batch's own order and one daam global heat map per image."""
    from oracle import sd_oracle as O
    B = x.shape[0] // 2
    outs, hms = [None] * (2 * B), []
    for i in range(B):
        rec = O.DaamRecorder(L * L, tokens)
        with torch.no_grad():
            w = O.unet_forward(u, ucfg, x[[i, B + i]], torch.tensor(t), ctx[[i, B + i]], rec)
        outs[i], outs[B + i] = w[0], w[1]
        hms.append(rec.compute_global_heat_map()[0])
    return torch.stack(outs), torch.stack(hms)


def test_config3_share_unet_batch16_512px_matches_oracle(sd15_host_weights, sd15_pipe):
    """BASELINE config 3's per-GPU share (VERDICT r4 missing #2): 8 images per GPU -> UNet batch 16 at 512 px (M = 65536 rows at 64 x 64), the batch
    tools/bench_configs.py times.  The launcher's tile / split-K / kernel-family choices depend on the batch (512 / 256 / 128 / 32 row tiles per
    level instead of 256 / 64 / 16 / 4), so this is the oracle check of the kernels THAT batch runs -- one forward with the DAAM recorder on."""
    from agenda_amd import synthetic
    cfg, u, v = sd15_host_weights
    pipe, L, B = sd15_pipe, 64, 8
    ctx = synthetic.make_context(cfg, B, seed=33)
    lat = synthetic.make_latents(cfg, list(range(40, 40 + B)), L)
    x = torch.cat([lat, lat]).to(torch.bfloat16).float()
    want, whm = _oracle_cfg_pairs(u, cfg.unet, x, 441, ctx, L)
    pipe.engine.set_context(ctx)
    pipe.engine.record_config(1, False, 77)
    pipe.engine.record_reset(B, L)
    try:
        got = pipe.engine.unet_forward(x, 441.0).clone()
        hm = torch.stack([pipe.engine.daam_global(i, 77, L).cpu() for i in range(B)])
    finally:
        pipe.engine.record_config(0)
    err = _rms_rel(got, want)
    worst = max(_rms_rel(got[i], want[i]) for i in range(2 * B))
    hm_err = float((hm - whm).abs().max() / whm.abs().max())
    print(f"config 3 share (UNet batch 16, 512 px): rms rel {err:.5f} (worst image {worst:.5f}), heat map rel {hm_err:.4f}")
    report("config3_share_forward_512px_unet_batch16", rms_rel=err, worst_image_rms_rel=worst, heat_map_rel=hm_err, norm_map_max_255=_norm_map_err_255(hm, whm)[0])
    assert err < 2.0 ** -6 and worst < 2.0 ** -6, (err, worst)
    assert hm_err < 0.02, hm_err


@pytest.mark.parametrize("p8", [1, 4])
def test_sd15_vae_decode_512_matches_oracle(sd15_host_weights, sd15_pipe, p8):
    """`vae.decode` at 512 px (the decode bench.py times) against the CPU oracle, plus the uint8 post-process rule
    (p8 = 4: every conv the 8-phase igemm can take goes through it, see above)."""
    from agenda_amd import synthetic
    from oracle import sd_oracle as O
    cfg, u, v = sd15_host_weights
    z = (synthetic.make_latents(cfg, [1], 64) * 0.18215).to(torch.bfloat16).float()
    if "vae512" not in _ORACLE_CACHE:
        with torch.no_grad():
            _ORACLE_CACHE["vae512"] = O.vae_decode(v, cfg.vae, z / cfg.vae.scaling_factor)
    want = _ORACLE_CACHE["vae512"]
    sd15_pipe.engine.set_option("igemm8p", p8)
    try:
        u8, f32 = sd15_pipe.engine.vae_decode(z, want_f32=True)
    finally:
        sd15_pipe.engine.set_option("igemm8p", 1)
    assert u8.shape == (1, 512, 512, 3) and torch.isfinite(f32).all()
    got = f32.permute(0, 3, 1, 2)
    err = _rms_rel(got, want)
    psnr = _psnr_u8(u8.cpu().numpy(), O.postprocess_image(want))
    print(f"vae decode 512 px: rms rel {err:.5f}, PSNR {psnr:.1f} dB")
    report(f"vae_decode_512px[igemm8p={p8}]", rms_rel=err, psnr_db=psnr)
    assert err < 2.0 ** -6, err
    assert psnr > 40.0, psnr
    want_u8 = ((f32 / 2 + 0.5).clamp(0, 1) * 255).round().to(torch.uint8)
    assert torch.equal(u8, want_u8)                                # post-process = round-half-even(255 x)


def test_config3_sd15_vae_encode_512_and_img2img_match_oracle(sd15_host_weights, sd15_pipe):
    """BASELINE config 3 at SD-1.5 shapes on one GPU's shard: `vae.encode` at 512 px (128..512-channel stride-2 convs with
    the asymmetric (0,1,0,1) padding, 4096-token mid attention) and a 3-step img2img tail, against the oracle's
    restatement of diffusers' img2img semantics (parity-unpinned: the reference has no img2img call site)."""
    from agenda_amd import synthetic, trace
    from oracle import sd_oracle as O
    cfg, u, v = sd15_host_weights
    pipe = sd15_pipe
    g = torch.Generator().manual_seed(40)
    S, steps, strength = 512, 5, 0.6                                   # int(5 * 0.6) = 3 denoise steps run
    image = (torch.rand(1, 3, S, S, generator=g) * 2 - 1).to(torch.bfloat16).float()
    ctx = synthetic.make_context(cfg, 1, seed=7)
    ne, nz = torch.randn(1, 4, 64, 64, generator=g), torch.randn(1, 4, 64, 64, generator=g)
    rec = O.DaamRecorder(64 * 64, 77)
    want_img, want_lat, (wm, wl) = O.img2img(u, v, cfg, ctx, image, ne, nz, steps, strength, recorder=rec)
    mean, logvar = pipe.engine.vae_encode(image)
    e_m, e_l = _rms_rel(mean, wm), _rms_rel(logvar, wl)
    with trace(pipe) as trc:
        out = pipe.img2img(prompt_embeds=ctx, image=image, strength=strength, num_inference_steps=steps, noise_enc=ne, noise=nz,
                           output_type="np")
        hm = trc.compute_global_heat_map(image_index=0).heat_maps.cpu()
    whm = rec.compute_global_heat_map()[0]
    lat_err, psnr = _rms_rel(out.latents, want_lat), _psnr_u8(out.images, want_img)
    hm_err = float((hm - whm).abs().max() / whm.abs().max())
    print(f"config3 (512 px): moments rms rel {e_m:.5f}/{e_l:.5f}, latents rms rel {lat_err:.4f}, PSNR {psnr:.1f} dB, heat map rel {hm_err:.4f}")
    report("config3_vae_encode_img2img_512px", moments_mean_rms_rel=e_m, moments_logvar_rms_rel=e_l, latents_rms_rel=lat_err, psnr_db=psnr, heat_map_rel=hm_err,
           norm_map_max_255=_norm_map_err_255(hm, whm)[0])
    assert e_m < 2.0 ** -6 and e_l < 2.0 ** -6, (e_m, e_l)
    assert lat_err < 0.05, lat_err
    assert psnr > 30.0, psnr
    # config 3's per-GPU share of the encoder: eight DIFFERENT images in one `vae_encode` call against the oracle's moments of each
    # (the decoder and the UNet were already checked at that batch; the encoder had only seen batch 1)
    imgs8 = (torch.rand(8, 3, S, S, generator=g) * 2 - 1).to(torch.bfloat16).float()
    m8, l8 = pipe.engine.vae_encode(imgs8)
    with torch.no_grad():
        mo = [O.vae_encode_moments(v, cfg.vae, imgs8[i:i + 1]) for i in range(8)]        # image by image: bounded host memory
    wm8, wl8 = torch.cat([a for a, _ in mo]), torch.cat([b for _, b in mo])
    per_m = [_rms_rel(m8[i], wm8[i]) for i in range(8)]
    per_l = [_rms_rel(l8[i], wl8[i]) for i in range(8)]
    print(f"config3 share, vae_encode at batch 8: moments rms rel worst image {max(per_m):.5f}/{max(per_l):.5f}")
    report("config3_share_vae_encode_512px_batch8", moments_mean_rms_rel_worst=max(per_m), moments_logvar_rms_rel_worst=max(per_l))
    assert max(per_m) < 2.0 ** -6 and max(per_l) < 2.0 ** -6, (per_m, per_l)
    assert hm_err < 0.03, hm_err
    assert float(hm.sum(0).mean()) == pytest.approx(3, rel=0.02)       # 3 of the 5 steps ran


def test_config1_sd15_256px_10_steps_end_to_end_vs_oracle():
    """BASELINE config 1 in full: SD-1.5 shapes, 1 x 256 x 256, 10 DDIM steps, CFG 7.5, DAAM on -- the HIP path
    against the fp32 CPU oracle run end to end on the same seeded weights / context / latents."""
    from agenda_amd import StableDiffusionPipeline, config, synthetic, trace
    from oracle import sd_oracle as O
    cfg = config.sd15()
    u = synthetic.make_unet_weights(cfg, 1234)
    v = synthetic.make_vae_weights(cfg, 1235)
    pipe = StableDiffusionPipeline(cfg, u, v, workspace_bytes=4 << 30)
    L, steps = 32, 10
    ctx = synthetic.make_context(cfg, 1, seed=7)
    lat = synthetic.make_latents(cfg, [0], L)
    rec = O.DaamRecorder(L * L, context_size=77)
    want_img, want_lat = O.generate(u, v, cfg, ctx, lat, steps, 7.5, recorder=rec)
    with trace(pipe) as trc:
        out = pipe(prompt_embeds=ctx, latents=lat, height=256, width=256, num_inference_steps=steps, output_type="np")
        got = trc.compute_global_heat_map(prompt=None, image_index=0).heat_maps.cpu()
    with pytest.raises(ValueError):                        # latents that do not match height/width are refused
        pipe(prompt_embeds=ctx, latents=lat, num_inference_steps=1)
    want = rec.compute_global_heat_map()[0]
    lat_err = _rms_rel(out.latents, want_lat)
    psnr = _psnr_u8(out.images, want_img)
    hm_err = float((got - want).abs().max() / want.abs().max())
    norm_err, norm_p999, norm_mean = _norm_map_stats_255(got, want)
    print(f"config1: latents rms rel {lat_err:.4f}, image PSNR {psnr:.1f} dB, heat map rel {hm_err:.4f}, "
          f"normalised-map err max {norm_err:.1f}/255, 99.9th percentile {norm_p999:.2f}/255, mean {norm_mean:.2f}/255")
    report("config1_256px_10_steps_end_to_end", latents_rms_rel=lat_err, psnr_db=psnr, heat_map_rel=hm_err, norm_map_max_255=norm_err,
           norm_map_p999_255=norm_p999, norm_map_mean_255=norm_mean)
    # measured on MI355X (bf16 storage / fp32 accumulate vs the fp32 oracle, 10 steps): latents 2.1 %, PSNR 43.7 dB, heat map 0.8 %; min-max-normalised
    # maps: max 12.3/255, 99.9th percentile 6.0, mean 1.0.  The max sits on one pixel of a row whose range is 0.066 around a mean of 0.128; across twenty
    # builds / option sets of round 6's bisect (profiles/r06_bisect_config1.txt) it lands anywhere in 6.0 .. 12.3 while the percentile
    # stays in 4.2 .. 6.0 and the mean in 0.91 .. 1.05: the robust figures carry the tight bounds, the max keeps 1.5x over the worst value seen
    # (SURVEY 8c's 2/255 aspiration is not met by any of them on random synthetic weights; PSNR >= 30 dB is, with 13 dB to spare)
    assert lat_err < 0.05, lat_err
    assert psnr > 36.0, psnr
    assert hm_err < 0.02, hm_err
    assert norm_p999 < 9.0, norm_p999
    assert norm_mean < 1.6, norm_mean
    assert norm_err < 19.0, norm_err
    pipe.engine.close()


_C2 = {"steps": 50, "chunk": 10}


@pytest.mark.parametrize("part", range(_C2["steps"] // _C2["chunk"]))
def test_config2_oracle_50_steps_chunk(sd15_host_weights, part):
    """The fp32 CPU oracle's side of the 50-step config-2 run (below), advanced 10 DDIM steps per test so that the suite keeps
    reporting progress (50 oracle steps at 512 px are ~4 minutes of host work): state carried in a module cache."""
    import time
    from agenda_amd import synthetic
    from oracle import sd_oracle as O
    cfg, u, v = sd15_host_weights
    L, steps, chunk = 64, _C2["steps"], _C2["chunk"]
    if part == 0:
        sc = cfg.sched
        sch = O.DDIM(sc.num_train_timesteps, sc.beta_start, sc.beta_end, sc.steps_offset, sc.set_alpha_to_one, sc.prediction_type)
        _C2.update(sch=sch, ts=list(sch.set_timesteps(steps)), ctx=synthetic.make_context(cfg, 1, seed=7), lat=synthetic.make_latents(cfg, [0], L),
                   rec=O.DaamRecorder(L * L, context_size=77), t=0.0)
        _C2["x"] = _C2["lat"].clone().float() * sch.init_noise_sigma
    assert "x" in _C2 and len(_C2["ts"]) == steps
    t0 = time.time()
    x = _C2["x"]
    with torch.no_grad():                                          # O.generate's loop body (data_generation.py:59 semantics, DDIM eta 0, CFG 7.5)
        for t in _C2["ts"][part * chunk:(part + 1) * chunk]:
            eps = O.unet_forward(u, cfg.unet, torch.cat([x, x], 0), torch.tensor(int(t)), _C2["ctx"], _C2["rec"])
            eu, ec = eps.chunk(2)
            x = _C2["sch"].step(eu + 7.5 * (ec - eu), int(t), x)
    _C2["x"] = x
    _C2["t"] += time.time() - t0
    assert torch.isfinite(x).all()


def test_config2_sd15_512px_50_steps_end_to_end_vs_oracle(sd15_host_weights, sd15_pipe):
    """BASELINE config 2 at the metric's OWN length: SD-1.5 shapes, one 512 x 512 image, 50 DDIM steps,
    CFG 7.5, DAAM on, VAE decode -- the loop of data_generation.py:56-64 that bench.py times -- HIP path against the fp32 CPU
    oracle on the same seeded weights / context / latents.  What the bf16 residual stream accumulates over 50 steps is measured
    here (latents, image, heat maps, and the min-max-normalised maps data_generation.py:82-84 exports, in /255)."""
    from agenda_amd import trace
    from oracle import sd_oracle as O
    cfg, u, v = sd15_host_weights
    pipe, steps = sd15_pipe, _C2["steps"]
    if "x" not in _C2 or "rec" not in _C2:
        pytest.skip("the oracle chunks did not run (selected alone?)")
    ctx, lat, rec, want_lat = _C2["ctx"], _C2["lat"], _C2["rec"], _C2["x"]
    with torch.no_grad():
        want_img = O.postprocess_image(O.vae_decode(v, cfg.vae, want_lat / cfg.vae.scaling_factor))
    want = rec.compute_global_heat_map()[0]
    with trace(pipe) as trc:
        out = pipe(prompt_embeds=ctx, latents=lat, num_inference_steps=steps, output_type="np")
        got = trc.compute_global_heat_map(prompt=None, image_index=0).heat_maps.cpu()
    lat_err = _rms_rel(out.latents, want_lat)
    psnr = _psnr_u8(out.images, want_img)
    hm_err = float((got - want).abs().max() / want.abs().max())
    norm_max, norm_p999, norm_mean = _norm_map_stats_255(got, want)
    print(f"config2 50 steps (512 px, oracle {_C2['t']:.0f} s): latents rms rel {lat_err:.4f}, image PSNR {psnr:.1f} dB, "
          f"heat map rel {hm_err:.4f}, normalised-map err max {norm_max:.1f}/255, 99.9th percentile {norm_p999:.2f}/255, mean {norm_mean:.2f}/255")
    report("config2_512px_50_steps_end_to_end", latents_rms_rel=lat_err, psnr_db=psnr, heat_map_rel=hm_err, norm_map_max_255=norm_max, norm_map_p999_255=norm_p999,
           norm_map_mean_255=norm_mean, oracle_seconds=_C2["t"])
    assert len(rec.acc) == 15 * 8
    assert float(got.sum(0).mean()) == pytest.approx(steps, rel=0.02)      # every step recorded, probability mass conserved
    # bounds: see DESIGN section 2 (measured on MI355X, random synthetic weights)
    assert lat_err < 0.10, lat_err
    assert psnr > 30.0, psnr
    assert hm_err < 0.03, hm_err
    assert norm_max < 13.0, norm_max                                       # measured 4.0 (one pixel; config 1's test explains the statistic)
    assert norm_mean < 1.0, norm_mean                                      # measured 0.47
    assert norm_p999 < 6.0, norm_p999
    for k in ("x", "rec", "ctx", "lat"):
        _C2.pop(k, None)


@pytest.mark.parametrize("name,heads,side", [("down_blocks.0.attentions.0.transformer_blocks.0.attn2", 8, 64),     # C = 320, d = 40, N = 4096
                                             ("mid_block.attentions.0.transformer_blocks.0.attn2", 8, 8)])          # C = 1280, d = 160, N = 64
def test_seam_backward_at_sd15_shapes(sd15_host_weights, sd15_pipe, name, heads, side):
    """SURVEY 8f-4 at SD-1.5's own shapes: the seam's backward (attn_bwd_kernel + the input-gradient GEMMs) with head
    dims 40 / 160 and 4096 / 64 queries, against torch autograd of hook.py's restatement."""
    from test_train_gpu import seam_backward_check
    cfg, u, v = sd15_host_weights
    seam_backward_check(sd15_pipe, cfg, u, name, heads, side, True, B2=2)


@pytest.fixture(scope="module")
def sd21():
    from agenda_amd import StableDiffusionPipeline, config, synthetic
    cfg = config.sd21()
    u = synthetic.make_unet_weights(cfg, 2100)
    v = synthetic.make_vae_weights(cfg, 2101)
    pipe = StableDiffusionPipeline(cfg, u, v, workspace_bytes=32 << 30)      # room for config 5's per-GPU share (UNet batch 8 at 768 px)
    yield cfg, u, v, pipe
    pipe.engine.close()


def test_config5_sd21_768px_vae_decode_vs_oracle(sd21):
    """BASELINE config 5's decode: 768 px (latent 96 -> 9216-token mid-block attention, 768^2 convs) against the CPU oracle."""
    from agenda_amd import synthetic
    from oracle import sd_oracle as O
    cfg, u, v, pipe = sd21
    z = (synthetic.make_latents(cfg, [1], 96) * 0.18215).to(torch.bfloat16).float()
    with torch.no_grad():
        want = O.vae_decode(v, cfg.vae, z / cfg.vae.scaling_factor)
    u8, f32 = pipe.engine.vae_decode(z, want_f32=True)
    assert u8.shape == (1, 768, 768, 3) and torch.isfinite(f32).all()
    err = _rms_rel(f32.permute(0, 3, 1, 2), want)
    psnr = _psnr_u8(u8.cpu().numpy(), O.postprocess_image(want))
    print(f"config5 vae decode 768 px: rms rel {err:.5f}, PSNR {psnr:.1f} dB")
    report("config5_vae_decode_768px", rms_rel=err, psnr_db=psnr)
    assert err < 2.0 ** -6, err
    assert psnr > 40.0, psnr


def test_config5_sd21_768px_v_prediction_denoise_vs_oracle(sd21):
    """BASELINE config 5's loop at size: three DDIM steps with v-prediction (SD-2.1's scheduler config) at 768 px, CFG 7.5, DAAM
    recording on, against the oracle's loop (DDIM.step with prediction_type = v_prediction)."""
    from agenda_amd import synthetic, trace
    from oracle import sd_oracle as O
    cfg, u, v, pipe = sd21
    assert cfg.sched.prediction_type == "v_prediction"
    L, steps = 96, 3
    ctx = synthetic.make_context(cfg, 1, seed=9)
    lat = synthetic.make_latents(cfg, [4], L)
    rec = O.DaamRecorder(L * L, 77)
    _, want = O.generate(u, v, cfg, ctx, lat, steps, 7.5, recorder=rec, decode=False)
    with trace(pipe) as trc:
        out = pipe(prompt_embeds=ctx, latents=lat, num_inference_steps=steps, height=768, width=768, output_type="latent")
        hm = trc.compute_global_heat_map(image_index=0).heat_maps.cpu()
    err = _rms_rel(out.latents, want)
    whm = rec.compute_global_heat_map()[0]
    hm_err = float((hm - whm).abs().max() / whm.abs().max())
    print(f"config5 v-prediction denoise (768 px, {steps} steps): latents rms rel {err:.5f}, heat map rel {hm_err:.4f}")
    report("config5_vpred_3_steps_768px", latents_rms_rel=err, heat_map_rel=hm_err, norm_map_max_255=_norm_map_err_255(hm, whm)[0])
    assert err < 0.05, err
    assert hm_err < 0.03, hm_err


def test_config5_sd21_768px_unet_forward_vs_oracle(sd21):
    """BASELINE config 5 shapes: SD-2.1 (heads 5/10/20/20 -> d = 64, ctx 1024, linear proj_in/out), 768 px
    (latent 96, 9216 tokens), CFG batch 2: one UNet forward + DAAM record against the CPU oracle."""
    from agenda_amd import synthetic
    from oracle import sd_oracle as O
    cfg, u, v, pipe = sd21
    L = 96
    ctx = synthetic.make_context(cfg, 1, seed=7)
    lat = synthetic.make_latents(cfg, [0], L)
    x = torch.cat([lat, lat]).to(torch.bfloat16).float()
    rec = O.DaamRecorder(L * L, 77)
    with torch.no_grad():
        want = O.unet_forward(u, cfg.unet, x, torch.tensor(981), ctx, rec)
    pipe.engine.set_context(ctx)
    pipe.engine.record_config(1, False, 77)
    pipe.engine.record_reset(1, L)
    got = pipe.engine.unet_forward(x, 981.0)
    err = _rms_rel(got, want)
    hm = pipe.engine.daam_global(0, 77, L).cpu()
    whm = rec.compute_global_heat_map()[0]
    hm_err = float((hm - whm).abs().max() / whm.abs().max())
    print(f"config5 forward: rms rel {err:.5f}, heat map rel {hm_err:.4f}")
    report("config5_forward_768px_cfg_pair", rms_rel=err, heat_map_rel=hm_err, norm_map_max_255=_norm_map_err_255(hm, whm)[0])
    assert err < 0.02, err              # measured 0.0121 (the SD-1.5 256 px forward: < 2^-6); 9216-token softmax rows
    assert hm_err < 0.01, hm_err        # measured 0.0036
    pipe.engine.record_config(0)


def test_config5_share_unet_batch8_768px_matches_oracle(sd21):
    """BASELINE config 5's per-GPU share (VERDICT r4 missing #2): 4 images per GPU -> UNet batch 8 at 768 px (SD-2.1 shapes: d = 64 heads, 1024-wide
    context, linear proj_in / proj_out, 9216 tokens), the batch tools/bench_configs.py times -- one forward + DAAM record against the oracle."""
    from agenda_amd import synthetic
    cfg, u, v, pipe = sd21
    L, B = 96, 4
    ctx = synthetic.make_context(cfg, B, seed=55)
    lat = synthetic.make_latents(cfg, list(range(60, 60 + B)), L)
    x = torch.cat([lat, lat]).to(torch.bfloat16).float()
    want, whm = _oracle_cfg_pairs(u, cfg.unet, x, 721, ctx, L)
    pipe.engine.set_context(ctx)
    pipe.engine.record_config(1, False, 77)
    pipe.engine.record_reset(B, L)
    try:
        got = pipe.engine.unet_forward(x, 721.0).clone()
        hm = torch.stack([pipe.engine.daam_global(i, 77, L).cpu() for i in range(B)])
    finally:
        pipe.engine.record_config(0)
    err = _rms_rel(got, want)
    worst = max(_rms_rel(got[i], want[i]) for i in range(2 * B))
    hm_err = float((hm - whm).abs().max() / whm.abs().max())
    print(f"config 5 share (UNet batch 8, 768 px): rms rel {err:.5f} (worst image {worst:.5f}), heat map rel {hm_err:.4f}")
    report("config5_share_forward_768px_unet_batch8", rms_rel=err, worst_image_rms_rel=worst, heat_map_rel=hm_err, norm_map_max_255=_norm_map_err_255(hm, whm)[0])
    assert err < 0.02 and worst < 0.02, (err, worst)      # the bound of the CFG-pair test above (9216-token softmax rows)
    assert hm_err < 0.01, hm_err
