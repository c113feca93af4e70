#!/usr/bin/env python3
"""GEGLU feed-forward of the C = 320 transformer blocks (M = 32768 rows at UNet batch 8): fused kernel against the two launches."""
import ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_ff.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_double)]
for M in (32768, 16384, 4096):
    row = []
    for fused in (0, 1):
        ms = C.c_double(); rc = lib.agd_bench_ff(M, 320, fused, 20, C.byref(ms)); row.append(ms.value * 1e3 if rc == 0 else float("nan"))
    fl = 2.0 * M * 320 * 12 * 320
    print(f"M={M:6d} C=320: two launches {row[0]:7.1f} us ({fl / row[0] / 1e6:5.0f} TF/s)   fused {row[1]:7.1f} us ({fl / row[1] / 1e6:5.0f} TF/s)")
