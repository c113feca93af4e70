import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from agenda_amd import StableDiffusionPipeline, synthetic
pipe = StableDiffusionPipeline.from_synthetic("sd15", seed=1234, weights_device="cuda", workspace_bytes=12 << 30)
cfg = pipe.cfg
for B, L in ((1, 32), (4, 64)):
    ctx = synthetic.make_context(cfg, B, seed=7)
    lat = synthetic.make_latents(cfg, list(range(B)), L)
    x = torch.cat([lat, lat]).cuda()
    pipe.engine.set_context(ctx)
    e0 = pipe.engine.unet_forward(x, 981.0).clone()
    bad = 0
    for i in range(3):
        e = pipe.engine.unet_forward(x, 981.0)
        d = (e0 - e).abs()
        bad += int((d > 0).sum())
        if i == 0 and bad:
            nz = (d > 0).nonzero()
            print("   first diffs (b,c,y,x):", nz[:5].tolist(), "count", len(nz), "of", d.numel(), "by image", [(int((d[b] > 0).sum())) for b in range(2 * B)])
    print(f"B={B} L={L}: nondeterministic elements {bad}", flush=True)
