#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass per kernel:
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 1024 SIMDs)  (busy matrix-pipe cycles over
available SIMD cycles; GRBM_GUI_ACTIVE is reported summed over the 8 XCDs, MI355X_MICROARCH.md 'DVFS give-back').
Usage: pmc_mfma.py <counter_collection.csv> > profiles/rNN_pmc_mfma_util.csv"""
import collections
import csv
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(.*", "", r["Kernel_Name"])[:80]
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], name)
        if key not in seen:
            seen.add(key); cnt[name] += 1
print("kernel,launches,mfma_busy_cycles_per_launch,gui_active_cycles_per_launch_per_xcd,mfma_util")
rows = []
for name, c in acc.items():
    busy, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if busy <= 0 or gui <= 0:
        continue
    rows.append((busy, name, cnt[name], busy / cnt[name], gui / cnt[name], busy / (gui * 1024.0)))
for busy, name, n, b1, g1, u in sorted(rows, reverse=True):
    print(f"\"{name}\",{n},{b1:.0f},{g1:.0f},{u:.3f}")
