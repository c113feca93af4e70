#!/usr/bin/env python3
"""Weight-streaming 1x1 kernel (igemm_wreg.h) against the launcher's choice, per 1x1 shape of the 32x32 / 16x16 / 8x8 maps (UNet batch 8),
hot operands.  python tools/kb_wreg.py"""
import ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
lib.agd_last_error.restype = C.c_char_p; lib.agd_last_error.argtypes = [C.c_void_p]
B = 8


def lin(H, K, N, mode, res):
    ms = C.c_double()
    if lib.agd_bench_conv(B, H, H, K, 0, N, 1, 1, 1, mode, res, 50, C.byref(ms)):
        return float("nan")
    return ms.value * 1e3


# (side, K, N, geglu, residual, launches per forward, what)
shapes = [(32, 640, 640, 0, 1, 15, "L1 C->C +res"), (32, 640, 640, 0, 0, 10, "L1 C->C"), (32, 640, 1920, 0, 0, 5, "L1 qkv"), (32, 640, 5120, 1, 0, 5, "L1 GEGLU"),
          (32, 2560, 640, 0, 1, 5, "L1 ff.net.2"), (16, 1280, 1280, 0, 1, 15, "L2 C->C +res"), (16, 1280, 1280, 0, 0, 10, "L2 C->C"), (16, 1280, 3840, 0, 0, 5, "L2 qkv"),
          (16, 1280, 10240, 1, 0, 5, "L2 GEGLU"), (16, 5120, 1280, 0, 1, 5, "L2 ff.net.2"), (8, 1280, 1280, 0, 1, 3, "L3 C->C +res"), (8, 1280, 3840, 0, 0, 1, "L3 qkv"),
          (8, 1280, 10240, 1, 0, 1, "L3 GEGLU"), (8, 5120, 1280, 0, 1, 1, "L3 ff.net.2")]
tot = [0.0, 0.0]
for side, K, N, g, r, n, what in shapes:
    base = lin(side, K, N, g | 16, r)                 # the launcher's choice (8-phase kernel allowed, as in the walk)
    wr = lin(side, K, N, g | 128, r)
    wr2 = lin(side, K, N, g | 128 | 1024, r)            # two K groups of waves (plain launches of <= 256 tiles only; else the same kernel)
    fl = 2.0 * B * side * side * K * N
    tot[0] += base * n; tot[1] += min(base, wr) * n
    print(f"{what:14s} M={B*side*side:5d} K={K:5d} N={N:5d}: launcher {base:6.1f} us ({fl / base / 1e6:5.0f} TF/s)   wreg {wr:6.1f} us ({fl / wr / 1e6:5.0f} TF/s)  wreg, 2 K groups {wr2:6.1f}  x{n}", flush=True)
print(f"per forward: launcher {tot[0]:.0f} us, best of both {tot[1]:.0f} us")
