"""Bisect of config 1's normalised-map error: one oracle run (BASELINE config 1: SD-1.5 shapes, 1 x 256 x 256,
10 DDIM steps, CFG 7.5, DAAM on), then the HIP path of (a) the source trees of earlier commits built under build_bisect/<sha>/ and
(b) this tree with one option off at a time.  For every leg: max / 99.9th percentile / mean of the min-max-normalised map error in
1/255, the token row and pixel that carry the max, and that row's dynamic range.  Diagnostic, not collected by pytest.

    python tests/diag_config1_bisect.py            parent: oracle + all legs, table to gpurun_out/bisect_config1.txt
    python tests/diag_config1_bisect.py child OUT [k=v,...]   one HIP run of the tree named by AGD_TREE"""
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L, STEPS = 32, 10


def child(out_path, opts):
    tree = os.environ.get("AGD_TREE", HERE)
    sys.path.insert(0, tree)
    from agenda_amd import StableDiffusionPipeline, config, synthetic, trace
    cfg = config.sd15()
    u = synthetic.make_unet_weights(cfg, 1234)
    v = synthetic.make_vae_weights(cfg, 1235)
    pipe = StableDiffusionPipeline(cfg, u, v, workspace_bytes=4 << 30)
    for kv in filter(None, opts.split(",")):
        k, val = kv.split("=")
        pipe.engine.set_option(k, int(val))
    ctx = synthetic.make_context(cfg, 1, seed=7)
    lat = synthetic.make_latents(cfg, [0], L)
    with trace(pipe) as trc:
        out = pipe(prompt_embeds=ctx, latents=lat, height=256, width=256, num_inference_steps=STEPS, output_type="np")
        got = trc.compute_global_heat_map(prompt=None, image_index=0).heat_maps.cpu()
    torch.save({"map": got, "latents": out.latents.cpu()}, out_path)


def norm(m):
    lo, hi = m.amin((-1, -2), keepdim=True), m.amax((-1, -2), keepdim=True)
    return (m - lo) / (hi - lo + 1e-8)


def line(name, got, want, want_lat):
    d = (norm(got["map"]) - norm(want)).abs() * 255
    flat = int(d.argmax())
    t, rem = divmod(flat, d.shape[-1] * d.shape[-2])
    y, x = divmod(rem, d.shape[-1])
    wr = want[t]
    rng = float(wr.max() - wr.min())
    rel = float(((got["latents"] - want_lat) ** 2).mean().sqrt() / (want_lat ** 2).mean().sqrt())
    per_row = d.amax((-1, -2))
    return (f"{name:44s} max {float(d.max()):6.2f}  p99.9 {float(d.flatten().quantile(0.999)):5.2f}  mean {float(d.mean()):5.3f}  "
            f"row {t:2d} px ({y:2d},{x:2d})  row range {rng:.4e} (row mean {float(wr.mean()):.4e})  rows>6: {int((per_row > 6).sum())}  latents {rel:.4f}")


def main():
    sys.path.insert(0, HERE)
    out_dir = os.path.join(HERE, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    legs = []
    bdir = os.path.join(HERE, "build_bisect")
    for sha in ("45de735", "d500953", "58baa16", "35d123f", "3db1057"):
        tree = os.path.join(bdir, sha)
        if os.path.exists(os.path.join(tree, "agenda_amd", "libagenda_hip.so")):
            legs.append((f"commit {sha}", tree, ""))
    legs.append(("this tree, defaults", HERE, ""))
    for opt in sys.argv[1:] or ["igemm_pc=0", "xcd_block=0", "conv_smap=0", "tblock_fuse=0", "attn2_premul=0", "conv_halo=0", "igemm8p=0",
                                "reduce_gn=0", "shortcut_fuse=0", "ff_proj_fuse=0", "upsample_phases=0", "ln_fold=0", "gn_fused_stats=0",
                                "igemm_pc=0,xcd_block=0,conv_smap=0,tblock_fuse=0,attn2_premul=0,conv_halo=0,igemm8p=0,reduce_gn=0,shortcut_fuse=0,ff_proj_fuse=0,upsample_phases=0"]:
        legs.append((f"this tree, {opt if len(opt) < 30 else 'all merges off'}", HERE, opt))
    procs = []
    for i, (name, tree, opt) in enumerate(legs):                       # HIP legs run one after another (one process on the card at a time)
        path = os.path.join(out_dir, f"bisect_leg{i}.pt")
        env = dict(os.environ, AGD_TREE=tree)
        env.pop("AGD_LIB", None)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", path, opt], env=env, cwd=tree, capture_output=True, text=True)
        procs.append((name, path, r))
        print(name, "rc", r.returncode, flush=True)
    from agenda_amd import config, synthetic
    from oracle import sd_oracle as O
    cfg = config.sd15()
    u = synthetic.make_unet_weights(cfg, 1234)
    v = synthetic.make_vae_weights(cfg, 1235)
    ctx = synthetic.make_context(cfg, 1, seed=7)
    lat = synthetic.make_latents(cfg, [0], L)
    rec = O.DaamRecorder(L * L, context_size=77)
    _, want_lat = O.generate(u, v, cfg, ctx, lat, STEPS, 7.5, recorder=rec, decode=False)
    want = rec.compute_global_heat_map()[0]
    torch.save({"map": want, "latents": want_lat}, os.path.join(out_dir, "bisect_want.pt"))
    lines = []
    base = None
    for name, path, r in procs:
        if r.returncode != 0:
            lines.append(f"{name:44s} FAILED: {r.stderr.strip().splitlines()[-1] if r.stderr.strip() else r.returncode}")
            continue
        got = torch.load(path)
        s = line(name, got, want, want_lat)
        if name == "this tree, defaults":
            base = got
        elif base is not None:
            s += f"  | vs defaults: map bit-identical {bool(torch.equal(got['map'], base['map']))}"
        lines.append(s)
        os.remove(path)
    with open(os.path.join(out_dir, "bisect_config1.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
    else:
        main()
