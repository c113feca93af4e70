#!/usr/bin/env python3
"""4-wave igemm kernels vs the 256-row 8-wave / 8-phase kernel (igemm8p.h) on the SD-1.5 shapes at UNet batch 8 and the VAE
decoder's convs at batch 4: us per launch and TFLOP/s, hot operands (agd_bench_conv; mode bits 16 = launcher's choice,
32 / 64 = force the 256- / 160-wide tile).  Usage (GPU box): python tools/kb_8p.py"""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
lib.agd_last_error.restype = C.c_char_p
lib.agd_last_error.argtypes = [C.c_void_p]


def conv(B, H, C0, C1, Cout, k, mode, res, stride=1, up=1, iters=30):
    ms = C.c_double()
    rc = lib.agd_bench_conv(B, H, H, C0, C1, Cout, k, stride, up, mode, res, iters, C.byref(ms))
    return ms.value * 1e3 if rc == 0 else float("nan")


# name, (B, H, C0, C1, Cout, k, geglu, res, stride, up)
shapes = [("L0 conv3 320->320 +res", (8, 64, 320, 0, 320, 3, 0, 1, 1, 1)),
          ("L0 conv3 640->320 cat", (8, 64, 320, 320, 320, 3, 0, 0, 1, 1)),
          ("L0 conv3 960->320 cat", (8, 64, 640, 320, 320, 3, 0, 0, 1, 1)),
          ("L0 conv3 s2 320->320", (8, 64, 320, 0, 320, 3, 0, 0, 2, 1)),
          ("L0 up conv3 640->640 @64", (8, 32, 640, 0, 640, 3, 0, 0, 1, 2)),
          ("L1 conv3 640->640 +res", (8, 32, 640, 0, 640, 3, 0, 1, 1, 1)),
          ("L0 C->C +res", (8, 64, 320, 0, 320, 1, 0, 1, 1, 1)),
          ("L0 qkv N960", (8, 64, 320, 0, 960, 1, 0, 0, 1, 1)),
          ("L0 geglu N2560", (8, 64, 320, 0, 2560, 1, 1, 0, 1, 1)),
          ("L0 ff2 K1280 +res", (8, 64, 1280, 0, 320, 1, 0, 1, 1, 1)),
          ("L0 shortcut 960->320 cat", (8, 64, 640, 320, 320, 1, 0, 0, 1, 1)),
          ("L1 geglu N5120", (8, 32, 640, 0, 5120, 1, 1, 0, 1, 1)),
          ("L1 qkv N1920", (8, 32, 640, 0, 1920, 1, 0, 0, 1, 1)),
          ("L1 ff2 K2560 +res", (8, 32, 2560, 0, 640, 1, 0, 1, 1, 1)),
          ("L2 geglu N10240", (8, 16, 1280, 0, 10240, 1, 1, 0, 1, 1)),
          ("L2 qkv N3840", (8, 16, 1280, 0, 3840, 1, 0, 0, 1, 1)),
          ("VAE 512px conv3 128->128", (4, 512, 128, 0, 128, 3, 0, 1, 1, 1)),
          ("VAE 512px conv3 256->128", (4, 512, 256, 0, 128, 3, 0, 0, 1, 1)),
          ("VAE 256px conv3 256->256", (4, 256, 256, 0, 256, 3, 0, 1, 1, 1)),
          ("VAE 256->512 up conv3 256", (4, 256, 256, 0, 256, 3, 0, 0, 1, 2)),
          ("VAE 128px conv3 512->512", (4, 128, 512, 0, 512, 3, 0, 1, 1, 1)),
          ("VAE 64px conv3 512->512", (4, 64, 512, 0, 512, 3, 0, 1, 1, 1))]
print(f"{'shape':28s} {'4-wave':>9s} {'+halo':>9s} {'8p auto':>9s} {'8p 256':>9s} {'8p 160':>9s}   TF/s: 4w-best / 8p-best")
for name, (B, H, C0, C1, Cout, k, geglu, res, stride, up) in shapes:
    it = 10 if H >= 256 else 30
    t = [conv(B, H, C0, C1, Cout, k, geglu | m, res, stride, up, it) for m in (0, 8, 16 | 8, 32, 64)]
    Ho = H * up // stride
    fl = 2.0 * B * Ho * Ho * Cout * k * k * (C0 + C1)
    b4 = min(t[0], t[1]); b8 = min(x for x in t[3:] if x == x) if any(x == x for x in t[3:]) else float("nan")
    print(f"{name:28s}" + "".join(f"{x:9.1f}" for x in t) + f"   {fl / b4 / 1e6:6.0f} / {fl / b8 / 1e6:6.0f}", flush=True)
