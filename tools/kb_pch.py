#!/usr/bin/env python3
"""Row-halo producer / consumer kernel (igemm_pch.h, IgemmP::pc bit 4 = mode bit 16 of agd_bench_conv) against the row-halo kernel on the 3x3 convs of the 32 x 32 and
16 x 16 maps at UNet batch 8 (hot operands, us per launch; experiments library).  mode 8 = row-halo kernel allowed, bit 15 = XCD-aware tile blocks."""
import ctypes as C
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
lib.agd_last_error.restype = C.c_char_p
lib.agd_last_error.argtypes = [C.c_void_p]


def conv(B, H, C0, C1, Cout, mode=0, iters=50):
    ms = C.c_double()
    rc = lib.agd_bench_conv(B, H, H, C0, C1, Cout, 3, 1, 1, mode, 0, iters, C.byref(ms))
    if rc:
        print("ERR", lib.agd_last_error(None)); return float("nan")
    return ms.value * 1e3


shapes = [("3x3 M8192 640->640", (8, 32, 640, 0, 640)), ("3x3 M8192 320->640", (8, 32, 320, 0, 640)), ("3x3 M8192 1280+640->640", (8, 32, 1280, 640, 640)),
          ("3x3 M8192 640+640->640", (8, 32, 640, 640, 640)), ("3x3 M8192 640+320->640", (8, 32, 640, 320, 640)),
          ("3x3 M2048 1280->1280", (8, 16, 1280, 0, 1280)), ("3x3 M2048 640->1280", (8, 16, 640, 0, 1280)), ("3x3 M2048 1280+1280->1280", (8, 16, 1280, 1280, 1280)),
          ("3x3 M2048 1280+640->1280", (8, 16, 1280, 640, 1280)), ("3x3 M4096 640->640 (batch 4)", (4, 32, 640, 0, 640)), ("3x3 M16384 640->640 (batch 16)", (16, 32, 640, 0, 640))]
XB, PCH = 1 << 15, 1 << 16
print(f"{'shape':36s} {'halo':>9s} {'+ blocks':>9s} {'pch':>9s} {'pch+blocks':>10s}   TF/s (best)")
for name, a in shapes:
    ts = [conv(*a, mode=8 | 256 | m) for m in (0, XB, PCH, PCH | XB)]
    fl = 2.0 * a[0] * a[1] * a[1] * a[4] * 9 * (a[2] + a[3])
    print(f"{name:36s} {ts[0]:9.1f} {ts[1]:9.1f} {ts[2]:9.1f} {ts[3]:10.1f}   {fl / min(ts) / 1e6:7.0f}", flush=True)
