#!/usr/bin/env python3
"""Same-process, interleaved A/B of a ctx option on the bench workload (SD-1.5 512 px, batch 4, 50 DDIM steps, DAAM on):
python tools/ab_option.py <option> [rounds] [v0,v1,..]   -> ms per batch with the option at each value, alternating (cdna guide rule 24)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from agenda_amd import StableDiffusionPipeline, synthetic
from agenda_amd.generation import generate_batch

opt = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
values = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1]
pipe = StableDiffusionPipeline.from_synthetic("sd15", seed=1234, weights_device="cuda", workspace_bytes=12 << 30)
ctx = synthetic.make_context(pipe.cfg, 4, seed=7)


def batch(i):
    generate_batch(pipe, [4 * i + k for k in range(4)], [], prompt_embeds=ctx, num_inference_steps=50, word_rows=[[5], [8, 9]])


batch(0)
res = {v: [] for v in values}
for r in range(rounds):
    for v in values:
        pipe.engine.set_option(opt, v)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        batch(r + 1)
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) * 1e3)
for v in values:
    print(f"{opt}={v}: " + " ".join(f"{t:.1f}" for t in res[v]) + f"  median {sorted(res[v])[len(res[v]) // 2]:.1f}  min {min(res[v]):.1f} ms per batch")
