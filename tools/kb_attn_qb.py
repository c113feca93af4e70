#!/usr/bin/env python3
"""Self-attention: 32 against 64 queries per wave (agd_set_attn_qb, experiments library), interleaved.  python tools/kb_attn_qb.py"""
import ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_attention.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_double)]
for (B, H, D, N) in ((8, 8, 40, 4096), (8, 8, 80, 1024), (8, 5, 64, 9216)):
    for rnd in range(3):
        r = []
        for qb in (1, 2):
            lib.agd_set_attn_qb(qb)
            ms = C.c_double(); lib.agd_bench_attention(B, H, D, N, N, 0, 10, C.byref(ms)); r.append(ms.value * 1e3)
        print(f"B{B} H{H} d{D} N{N}: 32 queries per wave {r[0]:.1f} us   64 queries per wave {r[1]:.1f} us")
