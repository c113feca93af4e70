#!/usr/bin/env python3
"""Self-attention variants (builds of attention.hip with different occupancy / key-tile flags, build_exp/lib_<tag>.so) interleaved in
ONE process (cdna guide rule 24): us per launch, median of the rounds.  python tools/kb_attn_variants.py base lb3 kb4lb2 ..."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tags = sys.argv[1:] or ["base"]
libs = {}
for t in tags:
    path = os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so") if t == "base" else os.path.join(ROOT, "build_exp", f"lib_{t}.so")
    l = C.CDLL(path); l.agd_bench_attention.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_double)]; libs[t] = l
shapes = [(8, 8, 40, 4096), (8, 8, 80, 1024), (8, 8, 160, 256), (8, 5, 64, 9216), (8, 10, 64, 2304), (4, 8, 40, 4096)]
print(f"{'shape':26s}" + "".join(f"{t:>10s}" for t in tags))
for (B, H, D, N) in shapes:
    res = {t: [] for t in tags}
    for r in range(4):
        for t in tags:
            ms = C.c_double(); libs[t].agd_bench_attention(B, H, D, N, N, 0, 5 if N > 8000 else 20, C.byref(ms)); res[t].append(ms.value * 1e3)
    print(f"B{B} H{H} d{D} N{N:5d}        " + "".join(f"{sorted(res[t])[1]:10.1f}" for t in tags), flush=True)
