#!/usr/bin/env python3
"""Where do the M = 512 (8 x 8 maps, UNet batch 8) 1x1 launches spend their time?  Timing variants of the experiments library (tools/kb_lin.py):
cfg 0 production; 16 A loads dropped; 32 B (weight) loads dropped; 48 both; 64 no DMA instructions; 512 dispatch only; 1024 no epilogue.  Hot operands."""
import ctypes as C
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]


def conv(B, H, C0, Cout, k=1, geglu=0, res=0, iters=50):
    ms = C.c_double()
    lib.agd_bench_conv(B, H, H, C0, 0, Cout, k, 1, 1, geglu, res, iters, C.byref(ms))
    return ms.value * 1e3


shapes = [("M512  K1280 N1280 +res", (8, 8, 1280, 1280, 1, 0, 1)), ("M512  K1280 N1280", (8, 8, 1280, 1280, 1, 0, 0)), ("M512  K2560 N1280", (8, 8, 2560, 1280, 1, 0, 0)),
          ("M512  K1280 N3840", (8, 8, 1280, 3840, 1, 0, 0)), ("M2048 K1280 N1280 +res", (8, 16, 1280, 1280, 1, 0, 1)), ("M2048 K640  N1280", (8, 16, 640, 1280, 1, 0, 0))]
cfgs = [0, 16, 32, 48, 64, 512, 1024]
print(f"{'shape':26s}" + "".join(f"{('cfg' + str(c)):>9s}" for c in cfgs) + "   | two K groups per workgroup: cfg0, cfg64")
for name, a in shapes:
    row = []
    for c in cfgs:
        lib.agd_set_igemm_cfg(c)
        row.append(conv(*a))
    a2 = a[:5] + (a[5] | 512,) + a[6:]
    kg = []
    for c in (0, 64):
        lib.agd_set_igemm_cfg(c)
        kg.append(conv(*a2))
    lib.agd_set_igemm_cfg(0)
    print(f"{name:26s}" + "".join(f"{t:9.1f}" for t in row) + f"   | {kg[0]:9.1f} {kg[1]:9.1f}", flush=True)
