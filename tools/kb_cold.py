#!/usr/bin/env python3
"""Weight-heavy launches with cold weights (as inside the UNet walk) against hot ones, and with a streaming warm-up of the weight
matrix in front: python tools/kb_cold.py"""
import ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_conv_cold.argtypes = [C.c_int] * 10 + [C.POINTER(C.c_double)]
shapes = [("L2 C->C  M2048 K1280 N1280 +res", (8, 16, 16, 1280, 1280, 1, 0, 1)), ("L2 qkv   N3840", (8, 16, 16, 1280, 3840, 1, 0, 0)),
          ("L2 geglu N10240", (8, 16, 16, 1280, 10240, 1, 1, 0)), ("L2 ff2   K5120 +res", (8, 16, 16, 5120, 1280, 1, 0, 1)),
          ("L1 C->C  M8192 K640 N640 +res", (8, 32, 32, 640, 640, 1, 0, 1)), ("L1 geglu N5120", (8, 32, 32, 640, 5120, 1, 1, 0)),
          ("L1 ff2   K2560 +res", (8, 32, 32, 2560, 640, 1, 0, 1)),
          ("conv3 L2 1280->1280 +res", (8, 16, 16, 1280, 1280, 3, 0, 1)), ("conv3 L3 M512 1280->1280 +res", (8, 8, 8, 1280, 1280, 3, 0, 1)),
          ("L3 C->C M512 +res", (8, 8, 8, 1280, 1280, 1, 0, 1)), ("L3 geglu", (8, 8, 8, 1280, 10240, 1, 1, 0)),
          ("conv3 L0 320->320 +res", (8, 64, 64, 320, 320, 3, 0, 1))]
print(f"{'shape':36s}{'cold':>9s}{'touch+run':>11s}{'hot':>9s}{'in-kernel':>11s}{'hot+1':>8s}{'hot+2':>8s}{'hot+4':>8s}  (us)")
for name, a in shapes:
    row = []
    for warm in (0, 1, 2, 3, 4, 5, 6):
        ms = C.c_double()
        rc = lib.agd_bench_conv_cold(*a, warm, 10, C.byref(ms))
        row.append(ms.value * 1e3 if rc == 0 else float("nan"))
    print(f"{name:36s}{row[0]:9.1f}{row[1]:11.1f}{row[2]:9.1f}{row[3]:11.1f}{row[4]:8.1f}{row[5]:8.1f}{row[6]:8.1f}", flush=True)
