#!/usr/bin/env python3
"""Do two values of a ctx option give BIT-IDENTICAL results?  One 512 px UNet forward at UNet batch 8 with the DAAM recorder on, per value:
python tools/eq_option.py <option> v0,v1  -> torch.equal of eps and of the aggregated heat maps (same tiles / same summation order claims)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from agenda_amd import StableDiffusionPipeline, synthetic

opt = sys.argv[1]
values = [int(v) for v in sys.argv[2].split(",")]
pipe = StableDiffusionPipeline.from_synthetic("sd15", seed=1234, weights_device="cuda", workspace_bytes=12 << 30)
cfg = pipe.cfg
B = 4
ctx = synthetic.make_context(cfg, B, seed=7)
lat = synthetic.make_latents(cfg, list(range(B)), 64)
x = torch.cat([lat, lat]).to(torch.bfloat16).float()
outs = []
for v in values:
    pipe.engine.set_option(opt, v)
    pipe.engine.set_context(ctx)
    pipe.engine.record_config(1, False, 77)
    pipe.engine.record_reset(B, 64)
    eps = pipe.engine.unet_forward(x, 981.0).clone()
    hm = torch.stack([pipe.engine.daam_global(i, 77, 64) for i in range(B)]).clone()
    outs.append((eps, hm))
    pipe.engine.record_config(0)
for v, (eps, hm) in zip(values[1:], outs[1:]):
    d = float((eps - outs[0][0]).abs().max())
    print(f"{opt}={v} vs {values[0]}: eps equal {bool(torch.equal(eps, outs[0][0]))} (max abs diff {d:.3e}), heat maps equal {bool(torch.equal(hm, outs[0][1]))}")
