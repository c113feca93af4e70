#!/usr/bin/env python3
"""The split-K 3x3 conv of the 16 x 16 maps (1280 -> 1280, UNet batch 8: M = 2048, K = 11520, N = 1280) in three forms, for counter
passes:  KB_FORM=halo (production: igemm_halo_kernel<160,1,4>, A-major walk) | gen (general kernel, A-major) |
gen + AGD_IGEMM_WMAJOR=1 (general kernel, W-major walk).  rocprofv3 --pmc <counters> -- python3 tools/kb_conv16.py"""
import ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
ms = C.c_double()
mode = 8 if os.environ.get("KB_FORM", "halo") == "halo" else 0
lib.agd_bench_conv(8, 16, 16, 1280, 0, 1280, 3, 1, 1, mode, 1, 10, C.byref(ms))
print(os.environ.get("KB_FORM", "halo"), os.environ.get("AGD_IGEMM_WMAJOR", "-"), f"{ms.value * 1e3:.1f} us")
