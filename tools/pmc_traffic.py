#!/usr/bin/env python3
"""Merge two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, MI355X_MICROARCH.md 'rocprofv3 PMC slots') and,
optionally, a --kernel-trace pass of the same command into per-kernel HBM-side traffic per launch:
  fetch_MB = FETCH_SIZE [KB] * 1024 * 2 / 1e6   (gfx950: FETCH_SIZE reports half of a wide streaming read's bytes)
  write_MB = WRITE_SIZE [KB] * 1024 / 1e6
Usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [kernel_trace.csv] > profiles/rNN_pmc_traffic_summary.csv
Columns (bench.py reads the last five): kernel, launches, avg_us, fetch_MB_per_launch, write_MB_per_launch, total_MB_per_launch"""
import collections
import csv
import re
import sys


def short(name):
    return re.sub(r"\(.*", "", name).strip()[:110]


def pmc(path, counter):
    tot, cnt, seen = collections.Counter(), collections.Counter(), set()
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            if (r["Dispatch_Id"], k) not in seen:
                seen.add((r["Dispatch_Id"], k)); cnt[k] += 1
    return tot, cnt


fetch, nf = pmc(sys.argv[1], "FETCH_SIZE")
write, nw = pmc(sys.argv[2], "WRITE_SIZE")
dur, nd = collections.Counter(), collections.Counter()
if len(sys.argv) > 3:
    with open(sys.argv[3]) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            nd[k] += 1
print("kernel,launches,avg_us,fetch_MB_per_launch(FETCH_SIZE*1024*2),write_MB_per_launch(WRITE_SIZE*1024),total_MB_per_launch")
rows = []
for k in fetch:
    n = nf[k]
    fm = fetch[k] * 1024 * 2 / 1e6 / n
    wm = write.get(k, 0.0) * 1024 / 1e6 / max(nw.get(k, n), 1)
    us = dur[k] / nd[k] if nd.get(k) else 0.0
    rows.append((fm * n + wm * n, k, n, us, fm, wm))
for _, k, n, us, fm, wm in sorted(rows, reverse=True):
    print(f"{k},{n},{us:.1f},{fm:.2f},{wm:.2f},{fm + wm:.2f}")
