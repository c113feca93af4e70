#!/usr/bin/env python3
"""3x3 convs of the 8 x 8 maps (UNet batch 8): whole-images-resident kernel (igemm_smap.h) against the launcher's choice, hot and
cold-ish (the bench rewrites nothing between iterations: hot).  python tools/kb_smap.py"""
import ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
for C0, C1, Cout, res, n in ((1280, 0, 1280, 1, 7), (1280, 0, 1280, 0, 5), (1280, 1280, 1280, 0, 3)):
    row = []
    for mode in (0, 256):
        ms = C.c_double()
        rc = lib.agd_bench_conv(8, 8, 8, C0, C1, Cout, 3, 1, 1, mode, res, 50, C.byref(ms))
        row.append(ms.value * 1e3 if rc == 0 else float("nan"))
    fl = 2.0 * 512 * Cout * 9 * (C0 + C1)
    print(f"8x8 {C0}+{C1}->{Cout} res{res} x{n}: launcher {row[0]:6.1f} us ({fl / row[0] / 1e6:5.0f} TF/s)   smap {row[1]:6.1f} us ({fl / row[1] / 1e6:5.0f} TF/s)", flush=True)
