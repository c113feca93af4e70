#!/usr/bin/env python3
"""Per-kernel sums of arbitrary rocprofv3 PMC counters: pmc_generic.py <counter_collection.csv>"""
import collections, csv, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set(); names = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] not in names: names.append(r["Counter_Name"])
        if (r["Dispatch_Id"], k) not in seen: seen.add((r["Dispatch_Id"], k)); cnt[k] += 1
print("kernel,launches," + ",".join(n + "_per_launch" for n in names))
for k in sorted(acc, key=lambda k: -acc[k].get(names[0], 0)):
    print(f"\"{k}\",{cnt[k]}," + ",".join(f"{acc[k].get(n, 0) / cnt[k]:.0f}" for n in names))
