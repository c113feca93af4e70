#!/usr/bin/env python3
"""3x3 stride-1 convs: general kernel against the row-halo kernel (igemm_halo.h), per shape.  python tools/kb_halo.py"""
import ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
shapes = [("L0 320->320 +res", (8, 64, 320, 0, 320, 1)), ("L0 640->320 (concat)", (8, 64, 320, 320, 320, 0)), ("L0 960->320 (concat)", (8, 64, 640, 320, 320, 0)),
          ("L1 640->640 +res", (8, 32, 640, 0, 640, 1)), ("L1 1280->640 (concat)", (8, 32, 640, 640, 640, 0)), ("L1 320->640", (8, 32, 320, 0, 640, 0)),
          ("L2 1280->1280 +res", (8, 16, 1280, 0, 1280, 1)), ("L2 2560->1280 (concat)", (8, 16, 1280, 1280, 1280, 0)),
          ("L3 1280->1280 +res", (8, 8, 1280, 0, 1280, 1)),
          ("VAE 512px 128->128", (4, 512, 128, 0, 128, 0)), ("VAE 256px 256->256", (4, 256, 256, 0, 256, 0)), ("VAE 128px 512->512", (4, 128, 512, 0, 512, 0))]
shapes += [("UNet up 32->64 640->640", (8, 32, 640, 0, 640, 0, 2)), ("UNet up 16->32 1280->1280", (8, 16, 1280, 0, 1280, 0, 2)),
           ("UNet up 8->16 1280->1280", (8, 8, 1280, 0, 1280, 0, 2)), ("VAE up 256->512 256->256", (4, 256, 256, 0, 256, 0, 2))]
for name, sh in shapes:
    B, H, C0, C1, Cout, res = sh[:6]
    up = sh[6] if len(sh) > 6 else 1
    row = []
    for mode in (0, 8):
        ms = C.c_double()
        for it in (3, 20):
            rc = lib.agd_bench_conv(B, H, H, C0, C1, Cout, 3, 1, up, mode, res, it, C.byref(ms))
        row.append(ms.value * 1e3 if rc == 0 else float("nan"))
    fl = 2.0 * B * H * H * up * up * Cout * 9 * (C0 + C1)
    print(f"{name:26s} general {row[0]:7.1f} us ({fl / row[0] / 1e6:5.0f} TF/s)   halo {row[1]:7.1f} us ({fl / row[1] / 1e6:5.0f} TF/s)", flush=True)
