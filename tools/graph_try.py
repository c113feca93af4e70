#!/usr/bin/env python3
"""Does replaying the 50-step denoise as ONE captured HIP graph (torch.cuda.CUDAGraph around Engine.denoise) run faster than the
eager launch sequence?  python tools/graph_try.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from agenda_amd import StableDiffusionPipeline, synthetic
from agenda_amd.trace import trace

pipe = StableDiffusionPipeline.from_synthetic("sd15", seed=1234, device=0, weights_device="cuda", workspace_bytes=12 << 30)
cfg = pipe.cfg
B, L, steps = 4, 64, int(os.environ.get("STEPS", "50"))
ctx = synthetic.make_context(cfg, B, seed=7)
lat0 = torch.randn(B, 4, L, L, device="cuda")
lat = lat0.clone()
with trace(pipe) as trc:
    pipe.engine.set_context(ctx); pipe._apply_record_mode(); pipe.engine.record_reset(B, L)
    def run(): pipe._denoise(lat, steps, 7.5)
    for _ in range(2):
        lat.copy_(lat0); run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        lat.copy_(lat0); torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("eager  ms:", [round(1e3 * t, 1) for t in ts], flush=True)
    ref = lat.clone()
    g = torch.cuda.CUDAGraph()
    lat.copy_(lat0)
    try:
        with torch.cuda.graph(g):
            run()
    except Exception as e:
        print("capture failed:", repr(e)[:400]); sys.exit(0)
    ts = []
    for _ in range(3):
        lat.copy_(lat0); torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("graph  ms:", [round(1e3 * t, 1) for t in ts])
    print("graph result equals eager:", bool(torch.equal(lat, ref)), float((lat - ref).abs().max()))
