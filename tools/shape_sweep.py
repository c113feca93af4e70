import torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from agenda_amd import StableDiffusionPipeline, synthetic, trace
from agenda_amd.generation import generate_batch
pipe = StableDiffusionPipeline.from_synthetic("sd15", seed=1234, weights_device="cuda", workspace_bytes=24 << 30)
for B, side in ((1, 512), (3, 512), (5, 512), (2, 256), (1, 768), (7, 384)):
    ctx = synthetic.make_context(pipe.cfg, B, seed=7)
    imgs, hms = generate_batch(pipe, list(range(B)), [], prompt_embeds=ctx, num_inference_steps=2, height=side, word_rows=[[5], [8, 9]])
    torch.cuda.synchronize()
    print(B, side, tuple(imgs.shape), tuple(hms.shape), bool(torch.isfinite(hms).all()), float(hms.sum(1).mean()), flush=True)
