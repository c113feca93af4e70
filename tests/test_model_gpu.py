"""End-to-end parity of the HIP UNet / VAE / denoise loop / heat maps vs the CPU oracle, through
the C ABI.  Tolerances: single forward rel-err <= 2^-5 of the output scale after
~60 stacked bf16 layers; multi-step images by PSNR; heat maps by max-abs on the min-max
normalised map (<= 2/255)."""
import math

import numpy as np
import pytest
import torch
from _report import report

pytestmark = pytest.mark.gpu


def _rel(got, want):
    got = got.detach().float().cpu()
    return float((got - want).abs().max() / (want.abs().max() + 1e-12))


def _rms_rel(got, want):
    got = got.detach().float().cpu()
    return float(((got - want) ** 2).mean().sqrt() / ((want ** 2).mean().sqrt() + 1e-12))


def _psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10 * math.log10(255.0 ** 2 / mse)


@pytest.fixture(scope="module")
def tiny_pipe():
    from agenda_amd import StableDiffusionPipeline, config, synthetic
    cfg = config.tiny()
    u = synthetic.make_unet_weights(cfg, 11, bias_std=0.05, perturb_norm=0.1)
    v = synthetic.make_vae_weights(cfg, 12, bias_std=0.05, perturb_norm=0.1)
    pipe = StableDiffusionPipeline(cfg, u, v, workspace_bytes=1 << 30)
    return pipe, cfg, u, v


@pytest.mark.parametrize("cfgname,L", [("tiny", 16), ("tiny40", 16)])
def test_unet_forward_matches_oracle(cfgname, L):
    from agenda_amd import StableDiffusionPipeline, config, synthetic
    from oracle import sd_oracle as O
    cfg = config.CONFIGS[cfgname]()
    u = synthetic.make_unet_weights(cfg, 21, bias_std=0.05, perturb_norm=0.1)
    v = synthetic.make_vae_weights(cfg, 22)
    pipe = StableDiffusionPipeline(cfg, u, v, workspace_bytes=1 << 30)
    ctx = synthetic.make_context(cfg, 2, seed=3)
    x = synthetic.make_latents(cfg, [0, 1, 2, 3], L)
    x = x.to(torch.bfloat16).float()
    with torch.no_grad():
        want = O.unet_forward(u, cfg.unet, x, torch.tensor(501), ctx)
    pipe.engine.set_context(ctx)
    got = pipe.engine.unet_forward(x, 501.0)
    assert got.shape == want.shape
    assert _rms_rel(got, want) < 2.0 ** -6, _rms_rel(got, want)
    assert _rel(got, want) < 2.0 ** -4, _rel(got, want)
    pipe.engine.close()


def test_vae_decode_matches_oracle(tiny_pipe):
    from agenda_amd import synthetic
    from oracle import sd_oracle as O
    pipe, cfg, u, v = tiny_pipe
    z = synthetic.make_latents(cfg, [5, 6], 16).to(torch.bfloat16).float()
    with torch.no_grad():
        want = O.vae_decode(v, cfg.vae, z / cfg.vae.scaling_factor)          # [B,3,H,W]
        want_u8 = O.postprocess_image(want)
    u8, f32 = pipe.engine.vae_decode(z, want_f32=True)
    got = f32.permute(0, 3, 1, 2)
    assert _rms_rel(got, want) < 2.0 ** -6, _rms_rel(got, want)
    assert _psnr(u8.cpu().numpy(), want_u8) > 35.0


def _norm_maps(m):
    lo, hi = m.amin((-1, -2), keepdim=True), m.amax((-1, -2), keepdim=True)
    return (m - lo) / (hi - lo + 1e-8)


@pytest.mark.parametrize("cfgname,steps,tol_rel,tol_norm", [
    # tolerances are the measured bf16-vs-fp32 levels on synthetic weights with ~2x headroom
    # (tests/diag_parity.py: 1 step 0.9% / 5.9 per 255; 4 steps 2.0% / 11.6 per 255); DESIGN.md §parity
    ("tiny", 1, 0.02, 12 / 255), ("tiny", 4, 0.05, 24 / 255), ("tiny40", 2, 0.05, 24 / 255),
])
def test_generate_with_daam_matches_oracle(cfgname, steps, tol_rel, tol_norm):
    from agenda_amd import StableDiffusionPipeline, config, synthetic, trace
    from oracle import sd_oracle as O
    cfg = config.CONFIGS[cfgname]()
    u = synthetic.make_unet_weights(cfg, 11, bias_std=0.05, perturb_norm=0.1)
    v = synthetic.make_vae_weights(cfg, 12, bias_std=0.05, perturb_norm=0.1)
    pipe = StableDiffusionPipeline(cfg, u, v, workspace_bytes=1 << 30)
    B, L = 2, 16
    ctx = synthetic.make_context(cfg, B, seed=9)
    lat = synthetic.make_latents(cfg, [100, 101], L)
    rec = O.DaamRecorder(L * L, context_size=cfg.max_tokens)
    want_img, want_lat = O.generate(u, v, cfg, ctx, lat, steps, 7.5, recorder=rec)
    with trace(pipe) as trc:
        out = pipe(prompt_embeds=ctx, latents=lat, num_inference_steps=steps, output_type="np")
        got = torch.stack([trc.compute_global_heat_map(prompt=None, image_index=i).heat_maps.cpu() for i in range(B)])
    assert _rms_rel(out.latents, want_lat) < 0.06, _rms_rel(out.latents, want_lat)
    assert _psnr(out.images, want_img) > 30.0, _psnr(out.images, want_img)
    want = rec.compute_global_heat_map()            # [B, T, S, S]
    assert len(rec.acc) > 0 and got.shape == want.shape
    assert _rel(got, want) < tol_rel, _rel(got, want)
    # per-token min-max normalised maps: what data_generation.py:82 exports
    e = (_norm_maps(got) - _norm_maps(want)).abs().amax((-1, -2))
    assert float(e.max()) < tol_norm, float(e.max()) * 255
    pipe.engine.close()


def test_trace_without_generation_raises(tiny_pipe):
    from agenda_amd import trace
    pipe = tiny_pipe[0]
    with trace(pipe) as trc:
        with pytest.raises(RuntimeError, match="No heat maps found"):
            trc.compute_global_heat_map()


def test_hook_mode_matches_oracle(tiny_pipe):
    """hook.py semantics inside the fused UNet walk: head-mean per call, every attn2 (mid included),
    conditional half only (is_train=False), mean over all calls of clamp(bicubic)."""
    from agenda_amd import synthetic, UNetCrossAttentionHooker
    from oracle import sd_oracle as O
    pipe, cfg, u, v = tiny_pipe
    B, L, steps = 2, 16, 2
    ctx = synthetic.make_context(cfg, B, seed=19)
    lat = synthetic.make_latents(cfg, [7, 8], L)
    hk = UNetCrossAttentionHooker(is_train=False, latent_hw=L)
    with pytest.raises(RuntimeError, match="No heat maps found."):
        hk.compute_global_heat_map()
    pipe.unet.set_attn_processor(hk)
    try:
        out = pipe(prompt_embeds=ctx, latents=lat, num_inference_steps=steps, output_type="latent")
        got = hk.compute_global_heat_map()
        n_calls = hk.num_recorded
    finally:
        pipe.unet.set_attn_processor("default")
    rec = O.HookRecorder(is_train=False, latent_hw=L)
    O.generate(u, v, cfg, ctx, lat, steps, 7.5, recorder=rec, decode=False)
    want = rec.compute_global_heat_map()
    assert n_calls == len(rec.cross_attn_maps)
    assert got.shape == want.shape
    assert _rel(got, want) < 0.03, _rel(got, want)


def test_sd21_style_config_ragged_sizes():
    """Linear proj_in/out, head dim 64, v-prediction, latent 24 -> token counts 576/144/36/9 (768-px-like tails)."""
    from agenda_amd import StableDiffusionPipeline, config, synthetic, trace
    from oracle import sd_oracle as O
    cfg = config.tiny21()
    u = synthetic.make_unet_weights(cfg, 31, bias_std=0.05, perturb_norm=0.1)
    v = synthetic.make_vae_weights(cfg, 32, bias_std=0.05, perturb_norm=0.1)
    pipe = StableDiffusionPipeline(cfg, u, v, workspace_bytes=2 << 30)
    B, L, steps = 1, 24, 2
    ctx = synthetic.make_context(cfg, B, seed=5)
    lat = synthetic.make_latents(cfg, [42], L)
    rec = O.DaamRecorder(L * L, cfg.max_tokens)
    want_img, want_lat = O.generate(u, v, cfg, ctx, lat, steps, 7.5, recorder=rec)
    with trace(pipe) as trc:
        out = pipe(prompt_embeds=ctx, latents=lat, num_inference_steps=steps, height=L * 8, width=L * 8, output_type="np")
        hm = trc.compute_global_heat_map(image_index=0).heat_maps.cpu()
    assert out.images.shape == want_img.shape == (1, 192, 192, 3)
    assert _rms_rel(out.latents, want_lat) < 0.06, _rms_rel(out.latents, want_lat)
    assert _psnr(out.images, want_img) > 30.0
    assert _rel(hm, rec.compute_global_heat_map()[0]) < 0.05
    pipe.engine.close()


def test_from_pretrained_roundtrip(tmp_path, tiny_pipe):
    """diffusers on-disk layout (what `save_pretrained` writes, reference finetune_sd_token.py:164-187)."""
    import json
    from safetensors.torch import save_file
    from agenda_amd import StableDiffusionPipeline, synthetic
    from _util import write_tiny_clip_tokenizer
    pipe, cfg, u, v = tiny_pipe
    v = dict(v)
    v.update({k: t for k, t in synthetic.make_vae_weights(cfg, 12, bias_std=0.05, perturb_norm=0.1, with_encoder=True).items()
              if k.startswith(("encoder.", "quant_conv."))})          # save_pretrained writes the whole AutoencoderKL
    (tmp_path / "unet").mkdir(); (tmp_path / "vae").mkdir(); (tmp_path / "scheduler").mkdir()
    n_vocab = write_tiny_clip_tokenizer(str(tmp_path / "tokenizer"))
    uc = {"in_channels": 4, "out_channels": 4, "block_out_channels": list(cfg.unet.block_out_channels),
          "down_block_types": ["CrossAttnDownBlock2D" if c else "DownBlock2D" for c in cfg.unet.down_cross],
          "layers_per_block": cfg.unet.layers_per_block, "attention_head_dim": list(cfg.unet.num_heads),
          "cross_attention_dim": cfg.unet.cross_attention_dim, "use_linear_projection": False, "norm_num_groups": 32,
          "sample_size": cfg.default_sample_size}
    vc = {"latent_channels": 4, "out_channels": 3, "block_out_channels": list(cfg.vae.block_out_channels),
          "layers_per_block": cfg.vae.layers_per_block, "norm_num_groups": 32, "scaling_factor": cfg.vae.scaling_factor}
    json.dump(uc, open(tmp_path / "unet" / "config.json", "w"))
    json.dump(vc, open(tmp_path / "vae" / "config.json", "w"))
    json.dump({"_class_name": "PNDMScheduler", "num_train_timesteps": 1000, "beta_start": 0.00085, "beta_end": 0.012, "steps_offset": 1,
               "set_alpha_to_one": False, "prediction_type": "epsilon", "skip_prk_steps": True},
              open(tmp_path / "scheduler" / "scheduler_config.json", "w"))
    save_file({k: t.contiguous() for k, t in u.items()}, str(tmp_path / "unet" / "diffusion_pytorch_model.safetensors"))
    # store the VAE attention with the pre-0.18 names to exercise the renaming (decoder AND encoder mid blocks)
    old = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    vsd = {}
    for k, t in v.items():
        if ".attentions." in k:
            for new, o in old.items():
                k = k.replace("." + new + ".", "." + o + ".")
        vsd[k] = t.contiguous()
    save_file(vsd, str(tmp_path / "vae" / "diffusion_pytorch_model.safetensors"))
    # text_encoder/ in transformers-4.x naming (text_model. prefix + position_ids buffer) -> device CLIP encoder
    from agenda_amd import config as _config
    from agenda_amd.text import HipCLIPTextEncoder
    tcfg = _config.tiny(); tcfg.text = _config.TextConfig(hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=128, vocab_size=n_vocab)
    (tmp_path / "text_encoder").mkdir()
    json.dump({"hidden_size": 64, "num_hidden_layers": 1, "num_attention_heads": 1, "intermediate_size": 128, "vocab_size": n_vocab,
               "max_position_embeddings": 77, "hidden_act": "quick_gelu", "layer_norm_eps": 1e-5}, open(tmp_path / "text_encoder" / "config.json", "w"))
    tsd = {"text_model." + k: t.contiguous() for k, t in synthetic.make_text_weights(tcfg, 3).items()}
    tsd["text_model.embeddings.position_ids"] = torch.arange(77)[None].float()
    save_file(tsd, str(tmp_path / "text_encoder" / "model.safetensors"))
    from agenda_amd.scheduler import PNDMScheduler, DDIMScheduler
    p0 = StableDiffusionPipeline.from_pretrained(str(tmp_path), workspace_bytes=1 << 30)
    assert isinstance(p0.scheduler, PNDMScheduler)         # the checkpoint's own scheduler, as the reference's from_pretrained gives it
    p0.engine.close()
    p2 = StableDiffusionPipeline.from_pretrained(str(tmp_path), workspace_bytes=1 << 30, scheduler="DDIMScheduler")
    assert isinstance(p2.scheduler, DDIMScheduler)
    assert isinstance(p2.text_encoder, HipCLIPTextEncoder) and p2.cfg.text.hidden_size == 64
    e = p2.text_encoder(["an aerial view image with cars"])
    assert e.shape == (1, 77, 64) and torch.isfinite(e).all()
    ctx = synthetic.make_context(cfg, 1, seed=3)
    lat = synthetic.make_latents(cfg, [9], 16)
    a = pipe(prompt_embeds=ctx, latents=lat, num_inference_steps=2, output_type="pt")
    b = p2(prompt_embeds=ctx, latents=lat, num_inference_steps=2, output_type="pt")
    assert torch.equal(a.images, b.images)
    # the prompt side runs on the real tokenizer class: learned-token injection (data_generation.py:45-52) reaches the device
    from agenda_amd.generation import inject_learned_tokens
    base = p2.text_encoder(["an aerial view image with cars"])
    ids = inject_learned_tokens(p2, {"new_token_v0": torch.randn(64)}, ["new_token_v0"])
    assert ids == [n_vocab]
    e2 = p2.text_encoder(["an aerial view image with new_token_v0 cars"])
    assert not torch.allclose(e2, base)
    # from_pretrained -> img2img: the encoder half of the VAE was loaded too (vae.encode, finetune_sd.py:764-765)
    img = (torch.rand(1, 3, 128, 128, generator=torch.Generator().manual_seed(0)) * 2 - 1)
    ne, nz = torch.randn(1, 4, 16, 16), torch.randn(1, 4, 16, 16)
    pw = StableDiffusionPipeline(cfg, u, v, workspace_bytes=1 << 30)
    o1 = pw.img2img(prompt_embeds=ctx, image=img, strength=0.5, num_inference_steps=4, noise_enc=ne, noise=nz, output_type="pt")
    o2 = p2.img2img(prompt_embeds=ctx, image=img, strength=0.5, num_inference_steps=4, noise_enc=ne, noise=nz, output_type="pt")
    assert torch.equal(o1.images, o2.images)
    pw.engine.close()
    # save_pretrained (finetune_sd_token.py:164-187 layout): the injected token survives the round trip -- tokenizer with the added
    # token, text encoder with the resized embedding table -- and the reloaded pipeline produces the same context and images
    prompt = "an aerial view image with new_token_v0 cars"
    p2.save_pretrained(str(tmp_path / "resaved"))
    p3 = StableDiffusionPipeline.from_pretrained(str(tmp_path / "resaved"), workspace_bytes=1 << 30, scheduler="DDIMScheduler")
    assert len(p3.tokenizer) == n_vocab + 1 and p3.tokenizer.convert_tokens_to_ids("new_token_v0") == n_vocab
    assert torch.equal(p3.text_encoder([prompt]), e2)
    lat0 = synthetic.make_latents(cfg, [5], 16)
    i2 = p2([prompt], latents=lat0, num_inference_steps=2, output_type="pt").images
    i3 = p3([prompt], latents=lat0, num_inference_steps=2, output_type="pt").images
    assert torch.equal(i2, i3)
    # the safety-checker slot: flagged images come back black (and the generation driver then skips them, data_generation.py:61-62)
    p3.safety_checker = lambda imgs: [True] * imgs.shape[0]
    o = p3([prompt], latents=lat0, num_inference_steps=1, output_type="np")
    assert o.nsfw_content_detected == [True] and int(o.images.max()) == 0
    p3.engine.close()
    p2.engine.close()
    with pytest.raises(ValueError, match="in-memory weights"):
        StableDiffusionPipeline(cfg, u, v, workspace_bytes=1 << 28).save_pretrained(str(tmp_path / "nope"))
    # real CLIP weights without a loadable tokenizer must NOT fall back to the word-level stand-in
    import shutil
    from agenda_amd import _lib
    shutil.rmtree(tmp_path / "tokenizer")
    with pytest.raises(_lib.AgendaHipError, match="tokenizer"):
        StableDiffusionPipeline.from_pretrained(str(tmp_path), workspace_bytes=1 << 30)


def test_generation_driver_cli_layout_and_device_export(tmp_path):
    """`python -m agenda_amd.generation` equivalent: directory layout of data_generation.py:66-86, and the device
    export path writes the same bytes as the reference's host code on the same tensors."""
    import os
    from PIL import Image
    from agenda_amd import generation, StableDiffusionPipeline
    from agenda_amd.generation import generate_batch, save_outputs
    out = tmp_path / "run"
    generation.main(["--save-dir", str(out), "--num-images", "3", "--batch-size", "2", "--num-inference-steps", "2",
                     "--synthetic-config", "tiny", "--word_token_heatmaps", "cars", "utah", "view", "--image-size", "56",
                     "--stack", "cars", "utah", "view"])
    assert sorted(os.listdir(out / "images")) == ["0.png", "1.png", "2.png"]
    for d in ("daam_cars_heatmaps", "daam_utah_heatmaps", "daam_view_heatmaps", "daam_stack_heatmaps", "daam_inv_heatmaps"):
        assert sorted(os.listdir(out / d)) == ["0.png", "1.png", "2.png"], d
    im = np.asarray(Image.open(out / "images" / "1.png"))
    assert im.shape == (56, 56, 3) and im.dtype == np.uint8
    st = np.asarray(Image.open(out / "daam_stack_heatmaps" / "1.png"))
    o, f, b = (np.asarray(Image.open(out / f"daam_{w}_heatmaps" / "1.png")) for w in ("cars", "utah", "view"))
    np.testing.assert_array_equal(st, np.stack([o, f, 255 - b], -1))
    # device export == host export on identical tensors
    pipe = StableDiffusionPipeline.from_synthetic("tiny", workspace_bytes=1 << 30)
    imgs, hms = generate_batch(pipe, [0, 1], ["cars"], prompt="An aerial view image with cars in Utah", num_inference_steps=2)
    save_outputs(str(tmp_path / "dev"), [0, 1], imgs, hms, ["cars"], 56)
    save_outputs(str(tmp_path / "host"), [0, 1], imgs.cpu().numpy(), hms.cpu().numpy(), ["cars"], 56)
    for sub in ("images", "daam_cars_heatmaps"):
        for n in ("0.png", "1.png"):
            np.testing.assert_array_equal(np.asarray(Image.open(tmp_path / "dev" / sub / n)), np.asarray(Image.open(tmp_path / "host" / sub / n)))
    pipe.engine.close()


def test_vae_encode_and_img2img_match_oracle():
    """img2img front end: encoder moments (asymmetric-pad stride-2 convs) and the strength-truncated
    schedule vs the oracle's restatement of diffusers' semantics (parity-unpinned: no reference call site)."""
    from agenda_amd import StableDiffusionPipeline, config, synthetic, trace
    from oracle import sd_oracle as O
    cfg = config.tiny()
    u = synthetic.make_unet_weights(cfg, 11, bias_std=0.05, perturb_norm=0.1)
    v = synthetic.make_vae_weights(cfg, 12, bias_std=0.05, perturb_norm=0.1, with_encoder=True)
    pipe = StableDiffusionPipeline(cfg, u, v, workspace_bytes=2 << 30)
    g = torch.Generator().manual_seed(4)
    B, S, steps, strength = 2, 128, 5, 0.6
    image = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(torch.bfloat16).float()
    ctx = synthetic.make_context(cfg, B, seed=2)
    ne, nz = torch.randn(B, 4, 16, 16, generator=g), torch.randn(B, 4, 16, 16, generator=g)
    want_img, want_lat, (wm, wl) = O.img2img(u, v, cfg, ctx, image, ne, nz, steps, strength)
    mean, logvar = pipe.engine.vae_encode(image)
    assert _rms_rel(mean, wm) < 2.0 ** -6 and _rms_rel(logvar, wl) < 2.0 ** -6
    with trace(pipe) as trc:
        out = pipe.img2img(prompt_embeds=ctx, image=image, strength=strength, num_inference_steps=steps, noise_enc=ne, noise=nz,
                           output_type="np")
        hm = trc.compute_global_heat_map(image_index=1).heat_maps
    assert out.images.shape == want_img.shape
    assert _rms_rel(out.latents, want_lat) < 0.06
    assert _psnr(out.images, want_img) > 30.0
    assert float(hm.sum(0).mean()) == pytest.approx(int(steps * strength), rel=0.02)     # 3 of the 5 steps ran
    pipe.engine.close()


def test_cfg_shared_prefix_is_bit_identical(tiny_pipe):
    """agd_denoise shares the layers ahead of the first cross-attention between the (identical) unconditional and
    conditional halves; the result must equal running both halves, bit for bit (every shared op is row/image-local)."""
    from agenda_amd import synthetic, trace
    pipe, cfg = tiny_pipe[0], tiny_pipe[1]
    B, L = 2, 16
    ctx = synthetic.make_context(cfg, B, seed=21)
    lat = synthetic.make_latents(cfg, [5, 6], L)

    def run():
        with trace(pipe) as trc:
            out = pipe(prompt_embeds=ctx, latents=lat, num_inference_steps=3, output_type="pt")
            maps = torch.stack([trc.compute_global_heat_map(image_index=i).heat_maps for i in range(B)])
        return out.images.clone(), out.latents.clone(), maps.clone()

    a = run()
    pipe.engine.set_option("cfg_shared_prefix", 0)
    b = run()
    pipe.engine.set_option("cfg_shared_prefix", 1)
    with pytest.raises(Exception, match="unknown option"):
        pipe.engine.set_option("no_such_option", 1)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_generation_cli_images_only(tmp_path):
    """The reference's images-only mode (no --word_token_heatmaps, no learned tokens): zero word maps per image must not
    launch zero-sized grids on the device export path."""
    import os
    from agenda_amd import generation
    out = tmp_path / "run"
    generation.main(["--save-dir", str(out), "--num-images", "3", "--batch-size", "2", "--num-inference-steps", "1",
                     "--synthetic-config", "tiny", "--image-size", "56"])
    assert sorted(os.listdir(out)) == ["images"]
    assert sorted(os.listdir(out / "images")) == ["0.png", "1.png", "2.png"]


@pytest.mark.parametrize("cfgname", ["tiny", "tiny21"])
def test_cfg_ddim_step_matches_oracle(cfgname):
    """`agd_cfg_ddim_step` (the call-by-call `scheduler.step`): CFG combine + DDIM eta-0 update vs the oracle's DDIM.step,
    epsilon- and v-prediction, NCHW eps in / in-place NCHW latents out."""
    from agenda_amd import StableDiffusionPipeline, config
    from oracle import sd_oracle as O
    cfg = config.CONFIGS[cfgname]()
    pipe = StableDiffusionPipeline.from_synthetic(cfg, workspace_bytes=1 << 28)
    sc = cfg.sched
    sch = O.DDIM(sc.num_train_timesteps, sc.beta_start, sc.beta_end, sc.steps_offset, sc.set_alpha_to_one, sc.prediction_type)
    ts = sch.set_timesteps(10)
    g = torch.Generator().manual_seed(5)
    B, L, s = 3, 24, 7.5
    x = torch.randn(B, 4, L, L, generator=g)
    eps = torch.randn(2 * B, 4, L, L, generator=g)
    for t in (int(ts[0]), int(ts[4]), int(ts[-1])):
        eu, ec = eps.chunk(2)
        want = sch.step(eu + s * (ec - eu), t, x)
        a_t, a_p = sch.coeffs(t)
        lat = x.clone().cuda()
        pipe.engine.cfg_ddim_step(eps, lat, s, a_t, a_p)
        assert float((lat.cpu() - want).abs().max()) < 2e-5 * float(want.abs().max()), (cfgname, t)
    with pytest.raises(ValueError):
        pipe.engine.cfg_ddim_step(eps[:3], x.clone().cuda(), s, 0.5, 0.6)
    pipe.engine.close()


def test_manual_loop_matches_fused_denoise(tiny_pipe):
    """`unet(...)` + `agd_cfg_ddim_step` call by call == the fused `agd_denoise` loop (same kernels, same order)."""
    from agenda_amd import synthetic
    pipe, cfg = tiny_pipe[0], tiny_pipe[1]
    B, L, steps = 2, 16, 3
    ctx = synthetic.make_context(cfg, B, seed=31)
    lat0 = synthetic.make_latents(cfg, [7, 8], L)
    fused = pipe(prompt_embeds=ctx, latents=lat0, num_inference_steps=steps, output_type="latent").latents
    ts = pipe.scheduler.set_timesteps(steps)
    a_t, a_p = pipe.scheduler.step_coeffs()
    lat = lat0.clone().cuda()
    pipe.engine.set_context(ctx)
    for i, t in enumerate(ts):
        eps = pipe.engine.unet_forward(torch.cat([lat, lat]), float(t))
        pipe.engine.cfg_ddim_step(eps, lat, 7.5, a_t[i], a_p[i])
    assert _rms_rel(lat, fused.cpu()) < 1e-5


def test_recorder_buffers_follow_token_count_and_batch(tiny_pipe):
    """accumulators are capacity-tracked.  trace(rec_tokens=14) then trace() with all 77 rows (5.5x larger),
    then a larger batch, on ONE pipeline: results must equal a fresh pipeline's and nothing may be corrupted."""
    from agenda_amd import StableDiffusionPipeline, synthetic, trace
    pipe, cfg, u, v = tiny_pipe
    L, steps = 16, 2
    ctx1, lat1 = synthetic.make_context(cfg, 1, seed=41), synthetic.make_latents(cfg, [3], L)
    ctx3, lat3 = synthetic.make_context(cfg, 3, seed=42), synthetic.make_latents(cfg, [4, 5, 6], L)

    def run(p, ctx, lat, rec_tokens):
        with trace(p, rec_tokens=rec_tokens) as trc:
            out = p(prompt_embeds=ctx, latents=lat, num_inference_steps=steps, output_type="pt")
            maps = torch.stack([trc.compute_global_heat_map(image_index=i).heat_maps for i in range(lat.shape[0])])
        return out.images.clone(), maps.clone()

    a14 = run(pipe, ctx1, lat1, 14)
    a77 = run(pipe, ctx1, lat1, None)
    a3 = run(pipe, ctx3, lat3, None)
    b14 = run(pipe, ctx1, lat1, 14)                                  # shrinking again reuses the larger blocks
    fresh = StableDiffusionPipeline(cfg, u, v, workspace_bytes=1 << 30)
    f77, f3 = run(fresh, ctx1, lat1, None), run(fresh, ctx3, lat3, None)
    fresh.engine.close()
    assert a14[1].shape == (1, 14, L, L) and a77[1].shape == (1, 77, L, L) and a3[1].shape == (3, 77, L, L)
    assert torch.equal(a77[0], f77[0]) and torch.equal(a77[1], f77[1])
    assert torch.equal(a3[0], f3[0]) and torch.equal(a3[1], f3[1])
    assert torch.equal(a14[1], a77[1][:, :14]) and torch.equal(a14[1], b14[1]) and torch.equal(a14[0], b14[0])
    # weights next to the accumulators in device memory are intact: the UNet still answers as before
    x = torch.cat([lat1, lat1])
    pipe.engine.set_context(ctx1)
    e1 = pipe.engine.unet_forward(x, 500.0)
    fresh = StableDiffusionPipeline(cfg, u, v, workspace_bytes=1 << 30)
    fresh.engine.set_context(ctx1)
    assert torch.equal(e1, fresh.engine.unet_forward(x, 500.0))
    fresh.engine.close()


def test_hooker_context_length_change_resizes_the_recorder():
    """the hooker sees the token count change between direct seam calls (20 -> 77 tokens): the recorder is reset
    and re-sized, maps come out right for both lengths."""
    from agenda_amd import StableDiffusionPipeline, UNetCrossAttentionHooker, config, synthetic
    from oracle import sd_oracle as O
    cfg = config.tiny()
    u, v = synthetic.make_unet_weights(cfg, 11), synthetic.make_vae_weights(cfg, 12)
    pipe = StableDiffusionPipeline(cfg, u, v, workspace_bytes=1 << 30)
    name = "up_blocks.1.attentions.0.transformer_blocks.0.attn2"
    t = name.rsplit("attn2", 1)[0]
    C = u[t + "attn2.to_q.weight"].shape[0]
    hk = UNetCrossAttentionHooker(is_train=True, latent_hw=16)
    pipe.unet.set_attn_processor(hk)
    g = torch.Generator().manual_seed(3)
    for T in (20, 77, 20):
        hidden = torch.randn(2, 64, C, generator=g).to(torch.bfloat16).float()
        ctx = torch.randn(2, T, cfg.unet.cross_attention_dim, generator=g).to(torch.bfloat16).float()
        hk.clear()
        y = hk(pipe.unet.attn2(name), hidden, ctx)
        rec = O.HookRecorder(is_train=True, latent_hw=16)
        want = O.explicit_attention_processor(hidden, ctx, u[t + "attn2.to_q.weight"], u[t + "attn2.to_k.weight"], u[t + "attn2.to_v.weight"],
                                              u[t + "attn2.to_out.0.weight"], u[t + "attn2.to_out.0.bias"], 2, recorder=rec)
        assert float((y.cpu() - want).abs().max() / want.abs().max()) < 2.0 ** -6
        assert hk.cross_attn_maps[-1].shape == (2, T, 8, 8)
        assert float((hk.cross_attn_maps[-1].cpu() - rec.cross_attn_maps[0]).abs().max()) < 2e-3
        assert float((hk.compute_global_heat_map().cpu() - rec.compute_global_heat_map()).abs().max()) < 2e-3
    pipe.engine.close()


def test_pndm_generation_matches_oracle(tiny_pipe, tmp_path):
    """The reference's own scheduler (data_generation.py:59: checkpoint default = PNDM/PLMS x 20): the fused device loop
    (`agd_denoise_plms`, n + 1 model evaluations) vs the oracle's method-by-method restatement, DAAM recording on."""
    from agenda_amd import StableDiffusionPipeline, synthetic, trace
    from agenda_amd.scheduler import PNDMScheduler
    from oracle import sd_oracle as O
    _, cfg, u, v = tiny_pipe
    pipe = StableDiffusionPipeline(cfg, u, v, workspace_bytes=1 << 30, scheduler="PNDMScheduler")
    assert isinstance(pipe.scheduler, PNDMScheduler)
    B, L, steps = 2, 16, 6
    ctx = synthetic.make_context(cfg, B, seed=51)
    lat = synthetic.make_latents(cfg, [1, 2], L)
    rec = O.DaamRecorder(L * L, context_size=cfg.max_tokens)
    want_img, want_lat = O.generate(u, v, cfg, ctx, lat, steps, 7.5, recorder=rec, scheduler="pndm")
    with trace(pipe) as trc:
        out = pipe(prompt_embeds=ctx, latents=lat, num_inference_steps=steps, output_type="np")
        hm = torch.stack([trc.compute_global_heat_map(image_index=i).heat_maps for i in range(B)]).cpu()
    assert _rms_rel(out.latents, want_lat) < 0.06, _rms_rel(out.latents, want_lat)
    assert _psnr(out.images, want_img) > 30.0
    whm = rec.compute_global_heat_map()
    assert float(hm.sum(1).mean()) == pytest.approx(steps + 1, rel=0.02)          # steps + 1 UNet evaluations were recorded
    assert _rel(hm, whm) < 0.06
    # img2img on a PNDM pipeline (what from_pretrained builds for an SD-1.x checkpoint) runs the strength-truncated DDIM schedule on a
    # DDIM scheduler made from the same config: identical to a DDIM pipeline's img2img
    ve = synthetic.make_vae_weights(cfg, 12, bias_std=0.05, perturb_norm=0.1, with_encoder=True)
    p_pndm = StableDiffusionPipeline(cfg, u, ve, workspace_bytes=1 << 30, scheduler="PNDMScheduler")
    p_ddim = StableDiffusionPipeline(cfg, u, ve, workspace_bytes=1 << 30, scheduler="DDIMScheduler")
    g = torch.Generator().manual_seed(3)
    img = torch.rand(B, 3, 128, 128, generator=g) * 2 - 1
    ne, nz = torch.randn(B, 4, 16, 16, generator=g), torch.randn(B, 4, 16, 16, generator=g)
    a = p_pndm.img2img(prompt_embeds=ctx, image=img, num_inference_steps=4, noise_enc=ne, noise=nz, output_type="latent").latents
    b = p_ddim.img2img(prompt_embeds=ctx, image=img, num_inference_steps=4, noise_enc=ne, noise=nz, output_type="latent").latents
    assert torch.isfinite(a).all() and torch.equal(a, b)
    p_pndm.engine.close(); p_ddim.engine.close()
    with pytest.raises(ValueError, match="not implemented"):
        StableDiffusionPipeline(cfg, u, v, workspace_bytes=1 << 28, scheduler="EulerDiscreteScheduler")
    pipe.engine.close()


@pytest.mark.parametrize("cfgname", ["tiny", "tiny40"])
def test_layernorm_fold_matches_the_separate_layernorm_kernels(cfgname):
    """opt "ln_fold" (default on): LayerNorm folded into the GEMMs around it (row statistics from the producing GEMM's
    epilogue, W diag(gamma) + colsum correction in the consuming GEMM) vs the LayerNorm kernels + plain GEMMs, and both vs
    the oracle.  The folded path skips one bf16 rounding (of the normalised rows), so the two agree to bf16 noise, not bitwise."""
    from agenda_amd import StableDiffusionPipeline, config, synthetic
    from oracle import sd_oracle as O
    cfg = config.CONFIGS[cfgname]()
    u = synthetic.make_unet_weights(cfg, 21, bias_std=0.05, perturb_norm=0.1)
    v = synthetic.make_vae_weights(cfg, 22)
    pipe = StableDiffusionPipeline(cfg, u, v, workspace_bytes=1 << 30)
    ctx = synthetic.make_context(cfg, 2, seed=3)
    x = synthetic.make_latents(cfg, [0, 1, 2, 3], 16).to(torch.bfloat16).float()
    with torch.no_grad():
        want = O.unet_forward(u, cfg.unet, x, torch.tensor(501), ctx)
    pipe.engine.set_context(ctx)
    a = pipe.engine.unet_forward(x, 501.0).clone()
    a2 = pipe.engine.unet_forward(x, 501.0).clone()
    pipe.engine.set_option("ln_fold", 0)
    b = pipe.engine.unet_forward(x, 501.0).clone()
    pipe.engine.set_option("ln_fold", 1)
    assert torch.equal(a, a2)                                  # the fold is deterministic (ordered partial sums, no atomics)
    e_a, e_b, e_ab = _rms_rel(a, want), _rms_rel(b, want), _rms_rel(a, b.cpu())
    print(f"ln_fold {cfgname}: folded vs oracle {e_a:.5f}, unfolded vs oracle {e_b:.5f}, folded vs unfolded {e_ab:.5f}")
    report(f"ln_fold[{cfgname}]", folded_rms_rel=e_a, unfolded_rms_rel=e_b, folded_vs_unfolded=e_ab)
    # two independent bf16-noise realisations of the same fp32 function: each within 2^-6 of the oracle, ~sqrt(2) x that apart
    assert e_a < 2.0 ** -6 and e_b < 2.0 ** -6 and e_ab < 2.0 ** -5
    pipe.engine.close()


@pytest.mark.parametrize("cfgname", ["tiny", "tiny40"])
def test_groupnorm_fold_into_proj_in_matches_the_groupnorm_kernel(cfgname):
    """opt "gn_proj_fold" (default on): the transformer's GroupNorm (no activation behind it) folded into per-image proj_in matrices
    (norm.hip gn_fold_weight_kernel: Wb = W diag(rstd gamma), the mean and beta terms in a per-image row added in the epilogue)
    vs the GroupNorm kernel + the plain proj_in, and both vs the oracle.  What is rounded to bf16 differs (W gamma rstd instead of the
    normalised activation), so the two agree to bf16 noise, not bitwise; both are deterministic.  perturb_norm: gamma / beta are not
    1 / 0, bias_std: every bias term of the identity is exercised."""
    from agenda_amd import StableDiffusionPipeline, config, synthetic
    from oracle import sd_oracle as O
    cfg = config.CONFIGS[cfgname]()
    u = synthetic.make_unet_weights(cfg, 31, bias_std=0.05, perturb_norm=0.1)
    v = synthetic.make_vae_weights(cfg, 32)
    pipe = StableDiffusionPipeline(cfg, u, v, workspace_bytes=1 << 30)
    ctx = synthetic.make_context(cfg, 2, seed=3)
    x = synthetic.make_latents(cfg, [0, 1, 2, 3], 32).to(torch.bfloat16).float() + 0.3      # 32x32 and 16x16 maps fold (HW % 128 == 0); a mean to subtract
    with torch.no_grad():
        want = O.unet_forward(u, cfg.unet, x, torch.tensor(501), ctx)
    pipe.engine.set_context(ctx)
    a = pipe.engine.unet_forward(x, 501.0).clone()
    a2 = pipe.engine.unet_forward(x, 501.0).clone()
    pipe.engine.set_option("gn_proj_fold", 0)
    b = pipe.engine.unet_forward(x, 501.0).clone()
    pipe.engine.set_option("gn_proj_fold", 1)
    assert torch.equal(a, a2)
    assert not torch.equal(a, b)                               # the option really switches the path
    e_a, e_b, e_ab = _rms_rel(a, want), _rms_rel(b, want), _rms_rel(a, b.cpu())
    print(f"gn_proj_fold {cfgname}: folded vs oracle {e_a:.5f}, unfolded vs oracle {e_b:.5f}, folded vs unfolded {e_ab:.5f}")
    report(f"gn_proj_fold[{cfgname}]", folded_rms_rel=e_a, unfolded_rms_rel=e_b, folded_vs_unfolded=e_ab)
    assert e_a < 2.0 ** -6 and e_b < 2.0 ** -6 and e_ab < 2.0 ** -5
    pipe.engine.close()


def test_groupnorm_statistics_from_producer_epilogues(tiny_pipe):
    """opt "gn_fused_stats" (default on): the igemm launch that writes an activation also leaves per-(M tile, channel) partial
    sums, and the GroupNorm that reads it skips its statistics pass (one kernel instead of two).  The statistics are sums of
    the same bf16 values in a different (fixed) order: results agree to fp32 rounding of the group mean / rstd, runs are
    bitwise reproducible, and the VAE decode (512-channel, many-tile images) agrees too."""
    from agenda_amd import synthetic
    pipe, cfg = tiny_pipe[0], tiny_pipe[1]
    ctx = synthetic.make_context(cfg, 2, seed=3)
    x = synthetic.make_latents(cfg, [0, 1, 2, 3], 16).to(torch.bfloat16).float()
    pipe.engine.set_context(ctx)
    a = pipe.engine.unet_forward(x, 501.0).clone()
    a2 = pipe.engine.unet_forward(x, 501.0).clone()
    z = synthetic.make_latents(cfg, [5, 6], 32).to(torch.bfloat16).float() * 0.18215
    va = pipe.engine.vae_decode(z, want_f32=True)[1].clone()
    pipe.engine.set_option("gn_fused_stats", 0)
    b = pipe.engine.unet_forward(x, 501.0).clone()
    vb = pipe.engine.vae_decode(z, want_f32=True)[1].clone()
    pipe.engine.set_option("gn_fused_stats", 1)
    assert torch.equal(a, a2)
    e, ev = _rms_rel(a, b.cpu()), _rms_rel(va, vb.cpu())
    print(f"gn_fused_stats: unet fused vs separate rms rel {e:.6f}, vae {ev:.6f}")
    report("gn_fused_stats", unet_fused_vs_separate=e, vae_fused_vs_separate=ev)
    # a last-bit change of a group's mean / rstd flips isolated bf16 roundings, which the next ~60 layers amplify to the bf16
    # noise floor: the two paths are two realisations of that noise (exactness of the statistics: tests/test_ops_gpu.py)
    assert e < 2.0 ** -5 and ev < 2.0 ** -5
