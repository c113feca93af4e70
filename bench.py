#!/usr/bin/env python3
"""bench.py -- 512x512 images/sec with per-token DAAM heat maps, 50 DDIM steps, SD-1.5 shapes,
bf16 compute, synthetic weights/context (BASELINE.json metric; workload = configs[1]: batch 4).

A "step" = one pass of the hot path over one batch on every rank: 50 x (CFG-batched UNet forward
+ fused DAAM accumulation + DDIM update), VAE decode, heat-map aggregation, and (N>1) the RCCL
all_gather of images + heat maps.  One process per GPU; rank 0 prints ONE JSON line.

`python bench.py --gpus N` with N > 1 and no launcher environment starts the N ranks itself
(`python -m torch.distributed.run ...` as a child process, before this process touches the GPU);
under a launcher (RANK / WORLD_SIZE set, as the driver runs it) it is one rank of the job.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TFLOP_PER_IMAGE = {50: 82.84}          # 2*50*803.3 GF + 2514.5 GF
UNET_GF, VAE_GF = 803.3, 2514.5
MFMA_PEAK_TF = 2500.0                  # bf16 dense, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0                  # HBM3E spec, MI355X_MICROARCH.md (6.3 TB/s is what a copy achieves)


def _latest_profile(suffix):
    """the newest committed profiles/rNN_<suffix> (the round's own rocprofv3 summaries of this command)"""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return os.path.relpath(found[-1], ROOT) if found else os.path.join("profiles", "r06_" + suffix)


PMC_CSV = _latest_profile("pmc_traffic_summary.csv")
STATS_CSV = _latest_profile("bench_kernel_stats.csv")

# the kernel instantiations behind each class, as rocprofv3 names them: igemm_kernel<BM, BN, WM, WN, KS, ...>,
# igemm_halo_kernel<BN, SPLITK, BST> / igemm_pch_kernel<BN, SPLITK> (3x3 only), igemm8p_kernel<WM, WN, MI, NI0, NI1, KS, GEGLU>
# round 4: the fused row-panel kernels of the C = 320 blocks (tblock.hip) are booked where the launches they replace were: ff_fused / qkv_chain
# under the linears, attn_chain under cross-attention; igemm_smap_kernel (8 x 8 maps) and the split-K slab passes under the convs
CLASS_REP = {"igemm_conv3x3": "igemm_pch_kernel", "igemm_linear_1x1": "igemm_kernel", "attn_self_flash": "attn_kernel", "attn_cross_daam": "attn_kernel", "groupnorm": "gn_apply_part_kernel"}


def _targs(name, prefix):
    """template arguments of kernel `prefix<...>` in a rocprof kernel name, or None"""
    i = name.find(prefix + "<")
    if i < 0:
        return None
    j = name.find(">", i)
    return [a.strip() for a in name[i + len(prefix) + 1:j].split(",")]


def _is(name, cls):
    """does the rocprofv3 kernel name belong to the bench's kernel class?  (igemm_kernel<BM, BN, WM, WN, KS, STAGES, GEGLU, SPLITK, KG>,
    igemm8p_kernel<WM, WN, MI, NI0, NI1, KS, GEGLU>, attn_kernel<D, KB, QB, RECORD, AMASK>)"""
    ig, i8, at = _targs(name, "igemm_kernel"), _targs(name, "igemm8p_kernel"), _targs(name, "attn_kernel")
    if cls == "igemm_conv3x3":
        # (KS = 2: the phase convs of the upsampling convs -- booked under the convs by the walk's profile scopes too)
        pc = _targs(name, "igemm_pc_kernel")                        # igemm_pc_kernel<BM, BN, NLW, STAGES, KS, GEGLU, SPLITK>
        return (ig is not None and ig[4] in ("2", "3")) or (i8 is not None and i8[5] in ("2", "3")) or "igemm_halo_kernel<" in name or "igemm_pch_kernel<" in name or "igemm_smap_kernel<" in name or (pc is not None and pc[4] == "3")
    if cls == "igemm_linear_1x1":
        pc = _targs(name, "igemm_pc_kernel")
        return (ig is not None and ig[4] == "1") or (i8 is not None and i8[5] == "1") or (pc is not None and pc[4] == "1") or any(k in name for k in ("ff_fused_kernel<", "qkv_chain_kernel<", "igemm_wreg_kernel<"))
    if cls == "attn_self_flash":
        return at is not None and at[3] == "0"
    if cls == "attn_cross_daam":
        return (at is not None and at[3] != "0") or "attn_chain_kernel<" in name or "xattn_s_kernel" in name or "premul_" in name
    if cls == "groupnorm":
        return "gn_" in name and "splitk_reduce" not in name
    if cls == "layernorm":
        return "layernorm_kernel" in name
    return False


def self_launch_command(gpus, argv, n_devices):
    """None = run in this process (single rank, or already one rank of a launcher's job); else the child command that
    starts `gpus` ranks.  Raises SystemExit on an impossible request.  Never touches the GPU."""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:                                  # under torchrun / the driver's launcher
        if int(world_env) != gpus:
            raise SystemExit(f"WORLD_SIZE {world_env} != --gpus {gpus}")
        return None
    if gpus <= 1:
        return None
    if n_devices < gpus:
        raise SystemExit(f"--gpus {gpus} requested but only {n_devices} visible GPU(s): refusing to report a {gpus}-GPU number")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def pmc_traffic(cls):
    """HBM-side bytes per launch of a kernel class from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 per the gfx950
    correction + WRITE_SIZE; tools/pmc_traffic.py writes the CSV).  PMC counters cannot be sampled from inside this
    process, so this is the offline figure of the same command, or None."""
    try:
        n = mb = 0.0
        for line in open(os.path.join(ROOT, PMC_CSV)).read().splitlines()[1:]:
            f = line.rsplit(",", 5)                            # kernel, launches, avg_us, fetch_MB, write_MB, total_MB
            if _is(f[0], cls):
                n += float(f[1]); mb += float(f[1]) * float(f[5])
        # (a committed rocprofv3 --pmc pass of the same command, NOT a measurement of this run: the counters cannot be sampled in-process)
        return {"MB_per_launch": round(mb / n, 2), "launches_profiled": int(n), "source": PMC_CSV, "measured_in_this_run": False} if n else None
    except Exception:
        return None


def heaviest_instantiation(cls):
    """The single kernel instantiation of a class with the most total time in the committed `rocprofv3 --kernel-trace --stats` summary of
    this command (so that the class-level fraction can be re-derived from one row of that file), or None."""
    try:
        best = None
        import csv
        with open(os.path.join(ROOT, STATS_CSV)) as f:
            for r in csv.DictReader(f):
                name = r.get("Name") or r.get("KernelName") or ""
                if _is(name, cls):
                    tot = float(r.get("TotalDurationNs") or 0.0)
                    if best is None or tot > best[0]:
                        best = (tot, name, int(float(r.get("Calls") or 0)), float(r.get("AverageNs") or 0.0))
        if best is None:
            return None
        return {"kernel": best[1].split("(")[0][:120], "calls": best[2], "avg_us": round(best[3] / 1e3, 1), "total_ms": round(best[0] / 1e6, 2),
                "source": STATS_CSV, "measured_in_this_run": False}
    except Exception:
        return None


def cpu_baseline(cfg, usd, vsd, ctx, threads):
    """Oracle (fp32 PyTorch CPU restatement, kind 'port') on a bounded sample of the same workload, as BASELINE.md section 3 states it:
    two CFG denoise steps (UNet batch 2 at 512 px, DAAM recording on) + one VAE decode for ONE image; images/s extrapolated
    (the two steps x 25) to 50 steps."""
    import torch
    from oracle import sd_oracle as O
    from agenda_amd import synthetic
    torch.set_num_threads(threads)
    lat = synthetic.make_latents(cfg, [0], 64)
    rec = O.DaamRecorder(64 * 64, cfg.max_tokens)
    c1 = torch.cat([ctx[:1], ctx[ctx.shape[0] // 2: ctx.shape[0] // 2 + 1]]).cpu()
    t0 = time.time()
    with torch.no_grad():
        for t in (981, 961):
            O.unet_forward(usd, cfg.unet, torch.cat([lat, lat]), torch.tensor(t), c1, rec)
    t_steps = time.time() - t0
    t0 = time.time()
    with torch.no_grad():
        O.vae_decode(vsd, cfg.vae, lat / cfg.vae.scaling_factor)
    t_vae = time.time() - t0
    per_img = 25 * t_steps + t_vae
    return {"value": 1.0 / per_img, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"2 CFG denoise steps (UNet batch 2, 512px, DAAM on) {t_steps:.1f}s + 1 VAE decode {t_vae:.1f}s on the host CPU, "
                      f"the steps extrapolated x25 to 50; fp32 PyTorch restatement of the reference path (diffusers absent)"}


def rccl_version():
    """RCCL's version as torch reports it (torch.cuda.nccl is RCCL on ROCm); never lets a reporting detail fail a multi-GPU run."""
    try:
        import torch
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:                                      # pragma: no cover
        return f"unavailable ({type(e).__name__})"


def class_table(classes, step_ms):
    """Per kernel class: time, achieved TFLOP/s and GB/s (algorithmic work / time), the fraction of each roof, and `frac`
    against the BINDING roof of each launch (max(flop / 2.5 PF, bytes / 8 TB/s) summed over the class's launches).
    Every launch of the profiled batch is bracketed by its own HIP-event pair, which stretches the batch by ~5 %: the class
    times are scaled by (un-profiled ms per step) / (sum of the bracketed times) so that the table adds up to the step the
    headline value was measured on (`ms_events` keeps the raw bracketed time; launch gaps, ~2 %, are spread over the classes)."""
    out = {}
    tot = sum(v["ms"] for v in classes.values() if v["launches"] and v["ms"] > 0)
    scale = step_ms / tot if tot > 0 else 1.0
    for k, v in classes.items():
        if not v["launches"] or v["ms"] <= 0:
            continue
        v = dict(v); v["ms_events"] = v["ms"]; v["ms"] = v["ms"] * scale
        classes[k]["ms_scaled"] = v["ms"]
        s = v["ms"] * 1e-3
        tf, gbs = v["flops"] / s / 1e12, v["bytes"] / s / 1e9
        out[k] = {"ms": round(v["ms"], 2), "ms_events": round(v["ms_events"], 2), "launches": v["launches"], "TFLOPs": round(tf, 1), "GBs": round(gbs, 1),
                  "frac_mfma": round(tf / MFMA_PEAK_TF, 4), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4),
                  "frac": round(v["roof_ms"] / v["ms"], 4),
                  "hbm_bound_share": round(v["roof_ms_hbm_bound"] / v["roof_ms"], 3) if v["roof_ms"] > 0 else None}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="engine option (agd_set_option), for A/B runs")
    args = ap.parse_args()

    import torch                                             # device_count() does not initialise the GPU on this image
    cmd = self_launch_command(args.gpus, sys.argv[1:], torch.cuda.device_count())
    if cmd is not None:                                      # start the N ranks as a CHILD job and hand its exit code on
        raise SystemExit(subprocess.run(cmd).returncode)

    import torch.distributed as dist
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    # AGD_FORCE_DEVICE / AGD_DIST_BACKEND exist only to rehearse the multi-rank path on a 1-GPU box
    # (ranks share cuda:0, gloo instead of RCCL); the driver's real runs use one GPU per rank + nccl.
    if "AGD_FORCE_DEVICE" in os.environ:
        local = int(os.environ["AGD_FORCE_DEVICE"])
    elif world > 1 and torch.cuda.device_count() <= local:
        raise SystemExit(f"rank {rank}: local rank {local} has no GPU ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("AGD_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == world == args.gpus

    from agenda_amd import StableDiffusionPipeline, synthetic
    from agenda_amd.generation import generate_batch, gather_outputs
    B = args.batch
    pipe = StableDiffusionPipeline.from_synthetic("sd15", seed=1234, device=local, weights_device="cuda", keep_weights=True,
                                                  workspace_bytes=12 << 30)
    for kv in args.option:
        name, _, val = kv.partition("=")
        pipe.engine.set_option(name, int(val))
    cfg = pipe.cfg
    ctx = synthetic.make_context(cfg, B, seed=7)
    word_rows = [[5], [8, 9]]                      # two "words" (one single-token, one two-token)

    def one_step(step_idx, gather=True):
        """One batch on this rank; with gather: the final exchange is POSTED (async) and its handle returned -- the caller waits for
        it after it has enqueued the next batch, so the collective (<= 7 MB per rank) runs beside the next batch's first kernels."""
        seeds = [(step_idx * world + rank) * B + i for i in range(B)]
        imgs, hms = generate_batch(pipe, seeds, [], prompt_embeds=ctx, num_inference_steps=args.ddim_steps,
                                   word_rows=word_rows)
        if gather and backend == "gloo":           # rehearsal only: gloo moves host tensors
            imgs, hms = imgs.cpu(), hms.cpu()
        return gather_outputs(imgs, hms, async_op=True) if gather else (imgs, hms)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        one_step(-1 - w).wait()
    sync()
    t0 = time.perf_counter()
    pending = None
    for s in range(args.steps):
        nxt = one_step(s)
        if pending is not None:
            imgs, hms = pending.wait()             # the previous batch's gather, overlapped with this batch's launch
        pending = nxt
    imgs, hms = pending.wait()                     # every gather completes inside the timed region
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        assert imgs.shape[0] == world * B and hms.shape[0] == world * B      # the gather really delivered every rank's rows
        tt = torch.tensor([dt], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    n_img = world * B * args.steps
    value = n_img / dt

    roof = None
    table = None
    if rank == 0 and not args.no_profile:
        # live HIP-event timing of every launch on the launch stream, one more batch on rank 0 (no collective in here)
        pipe.engine.profile_begin()
        one_step(10 ** 6, gather=False)
        classes = pipe.engine.profile_end(MFMA_PEAK_TF * 1e12, HBM_PEAK_GBS * 1e9)
        table = class_table(classes, dt / args.steps * 1e3)
        # the same batch once more with the recorder OFF (no daam.trace around the pipeline call): the difference of the cross-attention class is the
        # time the fused accumulation itself costs (SURVEY 8d "time-in-accumulate-kernels")
        lat_off = synthetic.make_latents(cfg, [10 ** 6 * B + i for i in range(B)], 64)
        pipe.engine.profile_begin()
        pipe(prompt_embeds=ctx, num_inference_steps=args.ddim_steps, guidance_scale=7.5, latents=lat_off, height=512, width=512, output_type="pt")
        classes_off = pipe.engine.profile_end(MFMA_PEAK_TF * 1e12, HBM_PEAK_GBS * 1e9)
        dom = max(table, key=lambda k: table[k]["ms"])                         # dominant = the class with the most time
        d, raw = table[dom], classes[dom]
        mfma_bound = d["frac_mfma"] >= d["frac_hbm"]
        ig_ms = sum(table[k]["ms"] for k in table if k.startswith("igemm"))
        ig_fl = sum(classes[k]["flops"] for k in table if k.startswith("igemm"))
        heavy = heaviest_instantiation(dom)
        # the class is named after the instantiation that carries most of its time in the committed --stats summary (e.g. igemm_halo_kernel for the 3x3 class)
        rep_kernel = heavy["kernel"].replace("void ", "").split("<")[0].strip() if heavy else CLASS_REP.get(dom, dom)
        roof = {"bound": "mfma" if mfma_bound else "hbm", "kernel": f"{rep_kernel} [{dom}]",
                "achieved": d["TFLOPs"] if mfma_bound else d["GBs"], "peak": MFMA_PEAK_TF if mfma_bound else HBM_PEAK_GBS,
                "unit": "TFLOP/s" if mfma_bound else "GB/s", "frac": d["frac_mfma"] if mfma_bound else d["frac_hbm"],
                "frac_binding_roof": d["frac"], "traffic": pmc_traffic(dom), "heaviest_instantiation": heavy,
                "launches": d["launches"], "avg_launch_us": round(d["ms"] * 1e3 / max(raw["launches"], 1), 1),
                "algorithmic_per_launch": {"GFLOP": round(raw["flops"] / raw["launches"] / 1e9, 2), "MB": round(raw["bytes"] / raw["launches"] / 1e6, 2)},
                "igemm_all_frac_mfma": round(ig_fl / (ig_ms * 1e-3) / 1e12 / MFMA_PEAK_TF, 4) if ig_ms else None,
                "end_to_end_frac": round(value / world * TFLOP_PER_IMAGE.get(args.ddim_steps, (2 * args.ddim_steps * UNET_GF + VAE_GF) / 1e3) / MFMA_PEAK_TF, 4)}
        # the nominal per-image figure is the reference formulation's work: both CFG halves in full, 3x3 convs on the upsampled maps.  Not executed here: the shared CFG
        # prefix (conv_in .. first self-attention on B rows instead of 2B) and 5/9 of the upsampling convs' MACs (four 2x2 phase convs on the un-upsampled map, exact);
        # executed = the flop of every launch of the profiled batch as launched
        nominal = TFLOP_PER_IMAGE.get(args.ddim_steps, (2 * args.ddim_steps * UNET_GF + VAE_GF) / 1e3)
        executed = sum(v["flops"] for v in classes.values()) / B / 1e12
        roof["executed_TFLOP_per_image"] = round(executed, 2)
        roof["flops_not_executed"] = round(1.0 - executed / nominal, 4)        # CFG shared prefix (2.7 %) + phase-decomposed upsampling convs (4.5 %)
        roof["end_to_end_frac_executed"] = round(value / world * executed / MFMA_PEAK_TF, 4)
    daam = None
    if table and "attn_cross_daam" in table:
        # accumulator read+write = 132.5 MB per image per denoise step (15 layers x 8 heads x 77 rows, fp32);
        # the accumulation is fused into the cross-attention kernels, so their class time is the time spent on it
        gb = 132.5e-3 * args.ddim_steps * B
        cls_s = table["attn_cross_daam"]["ms"] * 1e-3
        gbs = gb / cls_s
        # the 5 layers at latent resolution keep ONE head-summed accumulator per image (the aggregation is linear there):
        # 5 x 77 x 4096 + (5 x 1024 + 5 x 256) x 8 x 77 fp32, read + written = 44.1 MB per image and step actually moved
        moved = 44.1e-3 * args.ddim_steps * B
        daam = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "algorithmic_GB_per_batch": round(gb, 2), "kernel": "attn_chain_kernel / attn_kernel<RECORD> / xattn_pre kernels (cross-attention + fused accumulate)",
                "accumulator_rmw_GB_per_batch": round(moved, 2), "achieved_on_moved_bytes_GBs": round(moved / cls_s, 1),
                "note": "the class time also carries to_q / to_out (and attn1.to_out) of the fused chain kernels and the attention math of both CFG halves: "
                        "a lower bound on the accumulate rate, not an HBM efficiency of the read-modify-write itself",
                "traffic": pmc_traffic("attn_cross_daam")}
        # isolated: the class's raw event time with the recorder on minus off (same kernels, same shapes; only the read-modify-write of the accumulators differs)
        on_ms, off_ms = classes["attn_cross_daam"]["ms"], classes_off.get("attn_cross_daam", {}).get("ms", 0.0)
        delta = on_ms - off_ms
        daam["record_on_ms"], daam["record_off_ms"], daam["delta_ms"] = round(on_ms, 2), round(off_ms, 2), round(delta, 2)
        if delta > 0:
            daam["accumulate_GBs_on_delta"] = round(moved / (delta * 1e-3), 1)                 # 44.1 MB x steps x B / delta
            daam["accumulate_frac_hbm_on_delta"] = round(moved / (delta * 1e-3) / HBM_PEAK_GBS, 4)
            daam["delta_note"] = ("difference of two ~37 ms event sums (one batch each): +- 0.3 ms; the 88 MB of accumulators of a 4-image batch fit the "
                                  "Infinity Cache, so this is the memory system's read-modify-write rate, not necessarily DRAM traffic")
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
                           "global_batch": world * B, "ddim_steps": args.ddim_steps, "parallelism": f"seed-sharded x{world} + all_gather",
                           "collective_backend": backend, "world_size": dist.get_world_size() if world > 1 else 1,
                           "rccl_version": rccl_version() if backend == "nccl" else None},
                "roofline": roof, "cpu_baseline": cpu}
        if daam:
            line["daam_accumulate"] = daam
        if table:
            line["kernel_classes"] = table
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
