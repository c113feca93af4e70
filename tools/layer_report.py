#!/usr/bin/env python3
"""Join the per-launch igemm shape log (experiments library, AGD_IGEMM_LOG=1, stderr) with a rocprofv3 kernel trace: per-layer in-situ
time and TFLOP/s for ONE UNet forward.  Usage: layer_report.py <stderr log> <kernel_trace.csv> [forward index]"""
import collections
import csv
import re
import sys

log, trace = sys.argv[1], sys.argv[2]
fwd = int(sys.argv[3]) if len(sys.argv) > 3 else 24
shapes = []
for ln in open(log, errors="ignore"):
    if ln.startswith("IGEMM "):
        shapes.append({k: v for k, v in (f.split("=") for f in ln.split()[1:])})
rows = []
with open(trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# every igemm launch logs one line, whichever kernel family it dispatches to (4-wave general / row-halo / 8-phase); the log and the
# trace are both in launch order on the one stream
ig = [(i, r) for i, r in enumerate(rows) if r[2].startswith(("void igemm_kernel", "void igemm_halo_kernel", "void igemm8p_kernel", "void igemm_smap_kernel", "void igemm_wreg_kernel", "void igemm_pc_kernel", "void igemm_pch_kernel"))]
assert len(ig) == len(shapes), (len(ig), len(shapes))
fam = {"void igemm_kernel": "4w", "void igemm_halo_kernel": "halo", "void igemm8p_kernel": "8p", "void igemm_smap_kernel": "smap", "void igemm_wreg_kernel": "wreg", "void igemm_pc_kernel": "pc", "void igemm_pch_kernel": "pch"}
marks = [i for i, r in enumerate(rows) if "timestep_embed" in r[2] or "prep_latents" in r[2]]
prep = [i for i, r in enumerate(rows) if "prep_latents" in r[2]]
a, b = prep[fwd], prep[fwd + 1]
agg = collections.OrderedDict()
for (i, r), sh in zip(ig, shapes):
    if not (a <= i < b):
        continue
    us = (r[1] - r[0]) / 1e3
    if int(sh["splits"]) > 1 and i + 1 < len(rows) and "splitk_reduce" in rows[i + 1][2]:
        us += (rows[i + 1][1] - rows[i + 1][0]) / 1e3
    kind = next(v for k, v in fam.items() if r[2].startswith(k))
    key = (sh["ks"], sh["M"], sh["K"], sh["N"], sh["stride"], sh["up"], sh["geglu"], sh["res"], sh["tile"] + "/" + kind, sh["splits"])
    e = agg.setdefault(key, [0, 0.0])
    e[0] += 1
    e[1] += us
tot = 0.0
print(f"{'ks':>2} {'M':>6} {'K':>6} {'N':>6} s u g r {'tile/kernel':>12} sp   n   us/launch   TF/s   total_us")
for key, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    ks, M, K, N, st, up, gg, res, tile, sp = key
    fl = 2.0 * int(M) * int(K) * int(N)
    tot += us
    print(f"{ks:>2} {M:>6} {K:>6} {N:>6} {st} {up} {gg} {res} {tile:>12} {sp:>2} {n:>3} {us / n:>10.1f} {fl * n / us / 1e6:>7.0f} {us:>10.1f}")
print(f"igemm total {tot:.0f} us per forward")
# everything else of the same forward (round 4: the fused row-panel kernels of tblock.hip are not igemm launches), by kernel
other = collections.OrderedDict()
igset = {i for i, _ in ig}
for i in range(a, b):
    if i in igset or "splitk_reduce" in rows[i][2]:
        continue
    nm = re.sub(r"\(.*", "", rows[i][2]).replace("void ", "")[:60]
    e = other.setdefault(nm, [0, 0.0]); e[0] += 1; e[1] += (rows[i][1] - rows[i][0]) / 1e3
ot = 0.0
print(f"\n{'kernel':60s}   n   us/launch   total_us")
for nm, (n, us) in sorted(other.items(), key=lambda kv: -kv[1][1]):
    ot += us
    print(f"{nm:60s} {n:>3} {us / n:>10.1f} {us:>10.1f}")
print(f"other kernels total {ot:.0f} us per forward; forward span {(rows[b - 1][1] - rows[a][0]) / 1e3:.0f} us")
