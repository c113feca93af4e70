import sys, os, torch, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from agenda_amd import StableDiffusionPipeline, config, synthetic, trace
from oracle import sd_oracle as O
cfgname = sys.argv[1] if len(sys.argv) > 1 else "tiny"
cfg = config.CONFIGS[cfgname]()
u = synthetic.make_unet_weights(cfg, 11, bias_std=0.05, perturb_norm=0.1)
v = synthetic.make_vae_weights(cfg, 12, bias_std=0.05, perturb_norm=0.1)
pipe = StableDiffusionPipeline(cfg, u, v, workspace_bytes=2 << 30)
B, L = 2, 16
ctx = synthetic.make_context(cfg, B, seed=9)
lat = synthetic.make_latents(cfg, [100, 101], L)
for steps in (1, 2, 4, 8):
    rec = O.DaamRecorder(L * L, context_size=cfg.max_tokens)
    wi, wl = O.generate(u, v, cfg, ctx, lat, steps, 7.5, recorder=rec, decode=False)
    with trace(pipe) as trc:
        out = pipe(prompt_embeds=ctx, latents=lat, num_inference_steps=steps, output_type="latent")
        gm = torch.stack([trc.compute_global_heat_map(image_index=i).heat_maps.cpu() for i in range(B)])
    wm = rec.compute_global_heat_map()
    lat_rms = float(((out.latents.cpu() - wl) ** 2).mean().sqrt() / (wl ** 2).mean().sqrt())
    gn = (gm - gm.amin((-1, -2), keepdim=True)) / (gm.amax((-1, -2), keepdim=True) - gm.amin((-1, -2), keepdim=True) + 1e-8)
    wn = (wm - wm.amin((-1, -2), keepdim=True)) / (wm.amax((-1, -2), keepdim=True) - wm.amin((-1, -2), keepdim=True) + 1e-8)
    e = (gn - wn).abs().amax((-1, -2))
    print(f"steps={steps}: latents rms-rel {lat_rms:.4f}; heat rel-max {float((gm-wm).abs().max()/wm.abs().max()):.4f}; "
          f"normalised-map max-abs: worst {float(e.max())*255:.1f}/255, median {float(e.median())*255:.1f}/255")
