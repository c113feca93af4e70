#!/usr/bin/env python3
"""Idle gaps of the GPU in a rocprofv3 kernel trace: python tools/gaps.py <kernel_trace.csv> [min_us]  -> the gaps above min_us, total idle, busy span."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50]))
rows.sort()
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
end = rows[0][1]; idle = 0; big = []
for i in range(1, len(rows)):
    g = rows[i][0] - end
    if g > 0:
        idle += g
        if g / 1e3 > thr: big.append((g / 1e3, rows[i - 1][2], rows[i][2], (rows[i][0] - rows[0][0]) / 1e6))
    end = max(end, rows[i][1])
span = (end - rows[0][0]) / 1e6
print(f"kernels {len(rows)}, span {span:.1f} ms, idle {idle / 1e6:.2f} ms ({100 * idle / 1e6 / span:.2f} %)")
for g, a, b, t in big: print(f"  gap {g:9.1f} us at {t:9.1f} ms  after [{a}] before [{b}]")
# steady state: the last 40 % of the trace -- gap histogram
n = len(rows); st = int(n * 0.6)
import collections
h = collections.Counter(); tot = 0; end = rows[st][1]; t0 = rows[st][0]
for i in range(st + 1, n):
    g = rows[i][0] - end
    if g > 0:
        tot += g
        h["<1us" if g < 1000 else "1-2us" if g < 2000 else "2-5us" if g < 5000 else "5-20us" if g < 20000 else "20-300us" if g < 300000 else ">300us"] += 1
    end = max(end, rows[i][1])
print(f"steady state: {n - st} kernels over {(end - t0) / 1e6:.1f} ms, idle {tot / 1e6:.2f} ms ({100 * tot / (end - t0):.2f} %), gap histogram {dict(h)}")
