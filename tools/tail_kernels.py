#!/usr/bin/env python3
"""What runs between the last DDIM step of a batch and the first UNet kernel of the next (VAE decode, heat maps, export, latents upload):
python tools/tail_kernels.py <kernel_trace.csv>  -> per-kernel totals of the LAST such window of the trace"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith("cfg_ddim_kernel")]
# windows: after a cfg_ddim whose next cfg_ddim is > 20 ms away
wins = [(idx[k], idx[k + 1]) for k in range(len(idx) - 1) if rows[idx[k + 1]][0] - rows[idx[k]][1] > 20e6]
if not wins: sys.exit("no batch boundary found")
a, b = wins[-1]
def totals(i0, i1):
    tot = collections.Counter(); cnt = collections.Counter()
    for s_, t_, n in rows[i0 + 1:i1]:
        k = n.split("(")[0][:60]; tot[k] += t_ - s_; cnt[k] += 1
    return tot, cnt
# a normal step of the same batch (between the two cfg_ddim launches before the boundary) is subtracted kernel by kernel
ka = idx.index(a)
ref, refc = totals(idx[ka - 1], a)
tot, cnt = totals(a, b)
extra = {k: tot[k] - ref.get(k, 0) for k in tot}
span = (rows[b][0] - rows[a][1]) / 1e3; span_ref = (rows[a][0] - rows[idx[ka - 1]][1]) / 1e3
print(f"boundary step {span:.0f} us, ordinary step {span_ref:.0f} us -> {span - span_ref:.0f} us of per-batch work; kernels beyond an ordinary step:")
for k, v in sorted(extra.items(), key=lambda kv: -kv[1])[:25]:
    if v > 5e3: print(f"  {v / 1e3:9.1f} us  x{cnt[k] - refc.get(k, 0):<4d} {k}")
