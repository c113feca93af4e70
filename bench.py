#!/usr/bin/env python3
"""bench.py -- 512x512 images/sec with per-token DAAM heat maps, 50 DDIM steps, SD-1.5 shapes,
bf16 compute, synthetic weights/context (BASELINE.json metric; workload = configs[1]: batch 4).

A "step" = one pass of the hot path over one batch on every rank: 50 x (CFG-batched UNet forward
+ fused DAAM accumulation + DDIM update), VAE decode, heat-map aggregation, and (N>1) the RCCL
all_gather of images + heat maps.  One process per GPU; rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TFLOP_PER_IMAGE = {50: 82.84}          # 2*50*803.3 GF + 2514.5 GF
UNET_GF, VAE_GF = 803.3, 2514.5
MFMA_PEAK_TF = 2500.0                  # bf16 dense, MI355X_MICROARCH.md


def pmc_traffic_mb():
    """HBM-side MB per launch of the dominant kernel class (3x3 igemm), from the committed rocprofv3 --pmc
    passes (profiles/r01_pmc_traffic_summary.csv: FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE);
    PMC counters cannot be sampled from inside this process, so this is the offline figure or None."""
    p = os.path.join(ROOT, "profiles", "r01_pmc_traffic_summary.csv")
    try:
        n = mb = 0.0
        for line in open(p).read().splitlines()[1:]:
            f = line.rsplit(",", 5)
            if "igemm_kernel" in f[0] and ", 3, 2, 0," in f[0]:
                n += float(f[1]); mb += float(f[1]) * (float(f[3]) + float(f[4]))
        return {"MB_per_launch": round(mb / n, 1), "source": "profiles/r01_pmc_traffic_summary.csv"} if n else None
    except Exception:
        return None


def cpu_baseline(cfg, usd, vsd, ctx, threads):
    """Oracle (fp32 PyTorch CPU restatement, kind 'port') on a bounded sample of the same workload:
    one CFG denoise step (UNet batch 2 at 512 px, DAAM recording on) + one VAE decode for ONE
    image; images/s extrapolated to 50 steps."""
    import torch
    from oracle import sd_oracle as O
    from agenda_amd import synthetic
    torch.set_num_threads(threads)
    lat = synthetic.make_latents(cfg, [0], 64)
    rec = O.DaamRecorder(64 * 64, cfg.max_tokens)
    c1 = torch.cat([ctx[:1], ctx[ctx.shape[0] // 2: ctx.shape[0] // 2 + 1]]).cpu()
    t0 = time.time()
    with torch.no_grad():
        O.unet_forward(usd, cfg.unet, torch.cat([lat, lat]), torch.tensor(981), c1, rec)
    t_step = time.time() - t0
    t0 = time.time()
    with torch.no_grad():
        O.vae_decode(vsd, cfg.vae, lat / cfg.vae.scaling_factor)
    t_vae = time.time() - t0
    per_img = 50 * t_step + t_vae
    return {"value": 1.0 / per_img, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"1 CFG denoise step (UNet batch 2, 512px, DAAM on) {t_step:.1f}s + 1 VAE decode {t_vae:.1f}s on the host CPU, "
                      f"extrapolated x50 steps; fp32 PyTorch restatement of the reference path (diffusers absent)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus and world > 1:
        raise SystemExit(f"WORLD_SIZE {world} != --gpus {args.gpus}")
    # AGD_FORCE_DEVICE / AGD_DIST_BACKEND exist only to rehearse the multi-rank path on a 1-GPU box
    # (ranks share cuda:0, gloo instead of RCCL); the driver's real runs use one GPU per rank + nccl.
    if "AGD_FORCE_DEVICE" in os.environ:
        local = int(os.environ["AGD_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("AGD_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)

    from agenda_amd import StableDiffusionPipeline, synthetic
    from agenda_amd.generation import generate_batch, gather_outputs
    B = args.batch
    pipe = StableDiffusionPipeline.from_synthetic("sd15", seed=1234, device=local, weights_device="cuda", keep_weights=True,
                                                  workspace_bytes=12 << 30)
    cfg = pipe.cfg
    ctx = synthetic.make_context(cfg, B, seed=7)
    n_rows = 14                                    # T' = len(tokens)+2 rows that daam reads
    word_rows = [[5], [8, 9]]                      # two "words" (one single-token, one two-token)

    def one_step(step_idx, gather=True):
        seeds = [(step_idx * world + rank) * B + i for i in range(B)]
        imgs, hms = generate_batch(pipe, seeds, [], prompt_embeds=ctx, num_inference_steps=args.ddim_steps,
                                   word_rows=word_rows)
        return gather_outputs(imgs, hms) if gather else (imgs, hms)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        one_step(-1 - w)
    sync()
    t0 = time.perf_counter()
    for s in range(args.steps):
        imgs, hms = one_step(s)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    n_img = world * B * args.steps
    value = n_img / dt

    roof = None
    classes = None
    if rank == 0 and not args.no_profile:
        # dominant kernel = implicit-GEMM conv3x3: live HIP-event timing on the launch stream
        pipe.engine.profile_begin()
        one_step(10 ** 6, gather=False)              # rank 0 only: no collective in here
        classes = pipe.engine.profile_end()
        conv = classes["igemm_conv3x3"]
        ach = conv["flops"] / (conv["ms"] * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "igemm_kernel<3x3>", "achieved": round(ach, 1), "peak": MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_PEAK_TF, 4), "traffic": pmc_traffic_mb(),
                "launches": conv["launches"], "avg_launch_us": round(conv["ms"] * 1e3 / max(conv["launches"], 1), 1),
                "end_to_end_frac": round(value / world * TFLOP_PER_IMAGE.get(args.ddim_steps, (2 * args.ddim_steps * UNET_GF + VAE_GF) / 1e3) / MFMA_PEAK_TF, 4)}
    daam = None
    if classes and classes.get("attn_cross_daam", {}).get("ms"):
        # accumulator read+write = 132.5 MB per image per denoise step (15 layers x 8 heads x 77 rows, fp32);
        # the accumulation is fused into the cross-attention kernels, so their class time is the time spent on it
        gb = 132.5e-3 * args.ddim_steps * B
        gbs = gb / (classes["attn_cross_daam"]["ms"] * 1e-3)
        daam = {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                "algorithmic_GB_per_batch": round(gb, 2), "kernel": "attn_kernel<RECORD> (cross-attention + fused accumulate)"}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        usd, vsd = pipe.synthetic_weights
        usd = {k: v.cpu() for k, v in usd.items()}
        vsd = {k: v.cpu() for k, v in vsd.items()}
        # a 1-GPU box grants 16 host cores; never oversubscribe past that share
        threads = int(os.environ.get("AGD_CPU_THREADS", min(os.cpu_count() or 1, 16)))
        cpu = cpu_baseline(cfg, usd, vsd, ctx, threads)
    if world > 1:
        dist.barrier()
    if rank == 0:
        line = {"metric": f"512x512 images/sec + DAAM heatmaps, {args.ddim_steps} DDIM steps, SD-1.5", "value": round(value, 4), "unit": "images/sec",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": f"SD-1.5 512x512 batch={B}/GPU, {args.ddim_steps} DDIM steps (eta 0, CFG 7.5), DAAM heat maps on (77 rows recorded, 2 word maps), VAE decode",
                           "global_batch": world * B, "ddim_steps": args.ddim_steps, "parallelism": f"seed-sharded x{world} + all_gather"},
                "roofline": roof, "cpu_baseline": cpu}
        if daam:
            line["daam_accumulate"] = daam
        if classes:
            line["kernel_classes_ms"] = {k: round(v["ms"], 2) for k, v in classes.items() if v["launches"]}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
