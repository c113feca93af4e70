#!/usr/bin/env python3
"""A/B of option COMBINATIONS: python tools/ab_combo.py rounds "a=1,b=2" "a=0,b=5" ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from agenda_amd import StableDiffusionPipeline, synthetic
from agenda_amd.generation import generate_batch
rounds = int(sys.argv[1]); combos = sys.argv[2:]
pipe = StableDiffusionPipeline.from_synthetic("sd15", seed=1234, weights_device="cuda", workspace_bytes=12 << 30)
ctx = synthetic.make_context(pipe.cfg, 4, seed=7)
def batch(i): generate_batch(pipe, [4 * i + k for k in range(4)], [], prompt_embeds=ctx, num_inference_steps=50, word_rows=[[5], [8, 9]])
batch(0)
res = {c: [] for c in combos}
for r in range(rounds):
    for c in combos:
        for kv in c.split(","):
            k, v = kv.split("="); pipe.engine.set_option(k, int(v))
        batch(100 + r)       # one untimed batch in the new configuration (order learning, caches)
        torch.cuda.synchronize(); t0 = time.perf_counter(); batch(r + 1); torch.cuda.synchronize()
        res[c].append((time.perf_counter() - t0) * 1e3)
for c in combos: print(f"{c}: " + " ".join(f"{t:.1f}" for t in res[c]) + f"  median {sorted(res[c])[len(res[c]) // 2]:.1f} ms per batch")
