#!/usr/bin/env python3
"""GroupNorm(+SiLU) launches of one SD-1.5 UNet forward (UNet batch 8): two-kernel path against the fused-statistics path, per shape,
with the streaming rate (one read + one write of the activation).  python tools/kb_gn.py"""
import ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_groupnorm_ex.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_double)]
B = 8
# (HW, C0, C1, launches per forward): resnet norm1 / norm2, transformer norm, conv_norm_out
shapes = [(4096, 320, 0, 12), (4096, 320, 320, 2), (4096, 640, 320, 1), (1024, 320, 0, 1), (1024, 640, 0, 11), (1024, 640, 640, 1),
          (1024, 1280, 640, 1), (1024, 640, 320, 1), (256, 640, 0, 1), (256, 1280, 0, 11), (256, 1280, 1280, 2), (256, 1280, 640, 1),
          (64, 1280, 0, 9), (64, 1280, 1280, 3)]
tot = [0.0, 0.0]
for HW, C0, C1, n in shapes:
    row = []
    for fused in (0, 1):
        ms = C.c_double()
        rc = lib.agd_bench_groupnorm_ex(B, HW, C0, C1, fused if HW % 128 == 0 else 0, 30, C.byref(ms))
        row.append(ms.value * 1e3 if rc == 0 else float("nan"))
        tot[fused] += row[-1] * n
    by = 4.0 * B * HW * (C0 + C1)
    print(f"HW={HW:5d} C={C0:4d}+{C1:4d} x{n:2d}: two-pass {row[0]:6.1f} us   fused-stats {row[1]:6.1f} us ({by / row[1] / 1e3:6.0f} GB/s)", flush=True)
print(f"per forward: two-pass {tot[0]:.0f} us, fused-stats {tot[1]:.0f} us")
