#!/usr/bin/env python3
"""Experiment: two engine instances, each denoising HALF the batch on its own stream from its own host thread (the whole denoise loop is one ctypes call, so the GIL is free),
against one engine with the whole batch.  If launches of the two streams overlap on the chip, one stream's ramp / store tail runs under the other's main loop.
python tools/two_streams_try.py [rounds]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from agenda_amd import StableDiffusionPipeline, synthetic
from agenda_amd.generation import generate_batch
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = 4
pipe = StableDiffusionPipeline.from_synthetic("sd15", seed=1234, weights_device="cuda", workspace_bytes=12 << 30)
ctx = synthetic.make_context(pipe.cfg, B, seed=7)
def whole(i): generate_batch(pipe, [B * i + k for k in range(B)], [], prompt_embeds=ctx, num_inference_steps=50, word_rows=[[5], [8, 9]])
whole(0)
halves = [StableDiffusionPipeline.from_synthetic("sd15", seed=1234, weights_device="cuda", workspace_bytes=8 << 30) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
full = synthetic.make_context(pipe.cfg, B, seed=7)
def half(h, i):
    with torch.cuda.stream(streams[h]):
        # the context tensor holds [uncond | cond] rows per image pair as make_context lays them out: take this half's images from both parts
        n = full.shape[0] // 2
        sel = torch.cat([full[:n][h * B // 2:(h + 1) * B // 2], full[n:][h * B // 2:(h + 1) * B // 2]])
        generate_batch(halves[h], [B * i + h * B // 2 + k for k in range(B // 2)], [], prompt_embeds=sel, num_inference_steps=50, word_rows=[[5], [8, 9]])
def both(i):
    ts = [threading.Thread(target=half, args=(h, i)) for h in range(2)]
    for t in ts: t.start()
    for t in ts: t.join()
both(0)
res = {"one engine, batch 4": [], "two engines x batch 2, two streams": []}
for r in range(rounds):
    torch.cuda.synchronize(); t0 = time.perf_counter(); whole(r + 1); torch.cuda.synchronize(); res["one engine, batch 4"].append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize(); t0 = time.perf_counter(); both(r + 1); torch.cuda.synchronize(); res["two engines x batch 2, two streams"].append((time.perf_counter() - t0) * 1e3)
for k, v in res.items(): print(f"{k}: " + " ".join(f"{t:.1f}" for t in v) + f"  median {sorted(v)[len(v) // 2]:.1f} ms per 4 images")
