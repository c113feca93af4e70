#!/usr/bin/env python3
"""A few launches of three igemm shapes (3x3 L0, C->C L0, GEGLU L0) and the L0 self-attention, for counter passes:
rocprofv3 --pmc <counters> -d <dir> -- python3 tools/kb_one.py"""
import ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
ms = C.c_double()
for sh in ((8, 64, 640, 320, 3, 0, 0), (8, 64, 320, 320, 1, 0, 1), (8, 64, 320, 2560, 1, 1, 0)):
    B, H, C0, Cout, k, geglu, res = sh
    lib.agd_bench_conv(B, H, H, C0, 0, Cout, k, 1, 1, geglu, res, 5, C.byref(ms))
    print(sh, f"{ms.value * 1e3:.1f} us")
lib.agd_bench_attention.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_double)]
lib.agd_bench_attention(8, 8, 40, 4096, 4096, 0, 5, C.byref(ms))      # L0 self-attention
print("self-attention d=40 N=4096", f"{ms.value * 1e3:.1f} us")
