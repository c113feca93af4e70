#!/usr/bin/env python3
"""Per-GPU shares of the other BASELINE.json configs on ONE MI355X (reported in README/DESIGN, not bench lines):
   config 3: SD-1.5 img2img, 512 px, batch 8 per GPU (64 over 8), 50-step schedule at strength 0.8 (40 steps run), DAAM on
   config 5: SD-2.1 shapes, 768 px, batch 4 per GPU (32 over 8), 50 DDIM steps, DAAM on
   config 2 at batch 8 for comparison.
Usage (GPU box): python tools/bench_configs.py [c2b8] [c3] [c5]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from agenda_amd import StableDiffusionPipeline, synthetic, trace  # noqa: E402
from agenda_amd.generation import generate_batch  # noqa: E402

what = sys.argv[1:] or ["c2b8", "c3", "c5"]


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


if "c2b8" in what or "c3" in what:
    from agenda_amd import config
    cfg15 = config.sd15()
    pipe = StableDiffusionPipeline(cfg15, synthetic.make_unet_weights(cfg15, 1234, device="cuda"),
                                   synthetic.make_vae_weights(cfg15, 1235, device="cuda", with_encoder=True), workspace_bytes=24 << 30)
    B = 8
    ctx = synthetic.make_context(pipe.cfg, B, seed=7)
    if "c2b8" in what:
        dt = timed(lambda: generate_batch(pipe, list(range(B)), [], prompt_embeds=ctx, num_inference_steps=50, word_rows=[[5], [8, 9]]))
        print(f"config 2 at batch 8: {B / dt:.3f} images/s ({dt * 1e3:.0f} ms per batch; 82.84 TFLOP/image -> {B / dt * 82.84 / 2500:.3f} of MFMA peak)", flush=True)
    if "c3" in what:
        g = torch.Generator().manual_seed(3)
        img = torch.rand(B, 3, 512, 512, generator=g) * 2 - 1

        def run():
            with trace(pipe) as trc:
                out = pipe.img2img(image=img, strength=0.8, num_inference_steps=50, prompt_embeds=ctx, generator=torch.Generator().manual_seed(0), output_type="pt")
                return out, [trc.compute_global_heat_map(image_index=i).heat_maps[[5, 8]] for i in range(B)]
        dt = timed(run)
        print(f"config 3 share (img2img 512 px, batch 8, strength 0.8 = 40 of 50 steps, DAAM on, VAE encode+decode): "
              f"{B / dt:.3f} images/s ({dt * 1e3:.0f} ms per batch)", flush=True)
    pipe.engine.close()
    del pipe
    torch.cuda.empty_cache()

if "c5" in what:
    pipe = StableDiffusionPipeline.from_synthetic("sd21", seed=2100, weights_device="cuda", workspace_bytes=40 << 30)
    B = 4
    ctx = synthetic.make_context(pipe.cfg, B, seed=7)
    dt = timed(lambda: generate_batch(pipe, list(range(B)), [], prompt_embeds=ctx, num_inference_steps=50, height=768, word_rows=[[5], [8, 9]]), reps=1)
    print(f"config 5 share (SD-2.1 shapes 768 px, batch 4, 50 DDIM steps, DAAM on): {B / dt:.3f} images/s ({dt * 1e3:.0f} ms per batch; "
          f"220.66 TFLOP/image -> {B / dt * 220.66 / 2500:.3f} of MFMA peak)", flush=True)
