#!/usr/bin/env python3
"""Where does the 8 x 8 maps' whole-images 3x3 kernel (igemm_smap.h) spend a launch?  Timing variants of the experiments library at UNet batch 8 (M = 512; us per launch):
cfg 0 production; 16 no weight DMA; 32 no image DMA; 48 neither; 64 no MFMAs; 128 no fragment reads; 192 neither; 240 barriers only.
hot: operands left in the caches between launches (agd_bench_conv); cold: 1 GiB written between launches (agd_bench_conv_cold, as inside the UNet walk)."""
import ctypes as C
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
lib.agd_bench_conv_cold.argtypes = [C.c_int] * 10 + [C.POINTER(C.c_double)]


def hot(C0, Cout, iters=50):
    ms = C.c_double()
    lib.agd_bench_conv(8, 8, 8, C0, 0, Cout, 3, 1, 1, 8 | 256, 0, iters, C.byref(ms))
    return ms.value * 1e3


def cold(C0, Cout, warm=0, iters=12):
    ms = C.c_double()
    lib.agd_bench_conv_cold(8, 8, 8, C0, Cout, 3, 0, 0, warm, iters, C.byref(ms))
    return ms.value * 1e3


cfgs = [0, 16, 32, 48, 64, 128, 192, 240]
print(f"{'':22s}" + "".join(f"{('cfg' + str(c)):>9s}" for c in cfgs))
for C0 in (1280, 2560):
    for name, f in (("hot", hot), ("cold", cold), ("cold, W left hot", lambda a, b: cold(a, b, 2))):
        row = []
        for c in cfgs:
            lib.agd_set_igemm_cfg(c)
            row.append(f(C0, 1280))
        lib.agd_set_igemm_cfg(0)
        print(f"{C0:4d}->1280 {name:10s}" + "".join(f"{t:9.1f}" for t in row), flush=True)
