#!/usr/bin/env python3
"""The 8 x 8 maps' whole-images 3x3 kernel (igemm_smap.h) at UNet batch 8 (M = 512): us per launch including the slab pass, hot operands (experiments library)."""
import ctypes as C
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
for rnd in range(3):
    row = []
    for C0, C1 in ((1280, 0), (1280, 1280), (640, 0), (2560, 0)):
        ms = C.c_double()
        lib.agd_bench_conv(8, 8, 8, C0, C1, 1280, 3, 1, 1, 8 | 256, 0, 50, C.byref(ms))
        row.append(f"{C0}+{C1}->1280 {ms.value * 1e3:6.1f}")
    print("   ".join(row), flush=True)
