#!/usr/bin/env python3
"""Producer / consumer igemm (igemm_pc.h) against the 4-wave kernel on the one-workgroup-per-CU launches of the 16 x 16 maps (UNet batch 8, hot operands, us per launch;
experiments library).  mode bits of agd_bench_conv: 11..14 = IgemmP::pc mask (1: 1x1 64 x 160, 2: 3x3 64 x 160 unsplit, 4: 3x3 128 x 160 x 2 K slices, 8: 8 x 8 maps 64 x 160 x 4 K slices; a mask that does
not apply to the shape falls back to the launcher's choice), 8 = row-halo kernel allowed, 256 = the 8 x 8 whole-images kernel allowed."""
import ctypes as C
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
lib.agd_last_error.restype = C.c_char_p
lib.agd_last_error.argtypes = [C.c_void_p]


def conv(B, H, C0, C1, Cout, k=1, mode=0, res=0, iters=50):
    ms = C.c_double()
    rc = lib.agd_bench_conv(B, H, H, C0, C1, Cout, k, 1, 1, mode, res, iters, C.byref(ms))
    if rc:
        print("ERR", lib.agd_last_error(None)); return float("nan")
    return ms.value * 1e3


shapes = [("1x1 M2048 K1280 N1280 +res", (8, 16, 1280, 0, 1280, 1), 1), ("1x1 M2048 K1280 N1280", (8, 16, 1280, 0, 1280, 1), 0), ("1x1 M2048 K640 N1280 +res", (8, 16, 640, 0, 1280, 1), 1),
          ("1x1 M2048 K5120+1280 N1280 +res", (8, 16, 5120, 1280, 1280, 1), 1), ("1x1 M2048 K2560 N1280", (8, 16, 2560, 0, 1280, 1), 0),
          ("3x3 M2048 1280->1280", (8, 16, 1280, 0, 1280, 3), 0), ("3x3 M2048 2560->1280", (8, 16, 2560, 0, 1280, 3), 0), ("3x3 M2048 640->1280", (8, 16, 640, 0, 1280, 3), 0),
          ("3x3 M512 1280->1280", (8, 8, 1280, 0, 1280, 3), 0), ("3x3 M512 2560->1280", (8, 8, 2560, 0, 1280, 3), 0),
          ("1x1 M2048 K1280 N3840 (qkv)", (8, 16, 1280, 0, 3840, 1), 0), ("1x1 M512 K1280 N1280 +res", (8, 8, 1280, 0, 1280, 1), 1), ("1x1 M8192 K640 N640 +res", (8, 32, 640, 0, 640, 1), 1),
          ("1x1 M8192 K2560+640 N640 +res", (8, 32, 2560, 640, 640, 1), 1), ("1x1 M8192 K640 N1920", (8, 32, 640, 0, 1920, 1), 0),
          ("3x3 M8192 640->640", (8, 32, 640, 0, 640, 3), 0), ("3x3 M8192 1280->640", (8, 32, 1280, 0, 640, 3), 0), ("3x3 M32768 320->320", (8, 64, 320, 0, 320, 3), 0),
          ("3x3 M32768 640->320", (8, 64, 640, 0, 320, 3), 0)]
XB = 1 << 15      # XCD-aware tile blocks (igemm.hip pick_xcd_block)
print(f"{'shape':36s} {'launcher':>9s} {'+ blocks':>9s} {'pc':>9s} {'pc+blocks':>10s}   TF/s (best)")
for name, a, res in shapes:
    base_mode = 8 | 256 if a[5] == 3 else 0
    pcm = (1 if a[5] == 1 else (2 if a[1] == 16 else 8)) << 11
    ts = [conv(*a[:5], k=a[5], mode=base_mode | m, res=res) for m in (0, XB, pcm, pcm | XB)]
    fl = 2.0 * a[0] * a[1] * a[1] * a[4] * a[5] * a[5] * (a[2] + a[3])
    print(f"{name:36s} {ts[0]:9.1f} {ts[1]:9.1f} {ts[2]:9.1f} {ts[3]:10.1f}   {fl / min(ts) / 1e6:7.0f}", flush=True)
