#!/usr/bin/env python3
"""Where does a K step of the row-halo producer / consumer kernel (igemm_pch.h) go?  Timing variants of the experiments library on the 128 x 160-tile 3x3 convs of the 32 x 32 maps
(UNet batch 8: 256 workgroups, hot operands, us per launch): cfg 0 production; 16 the loaders issue no LDS-DMA; 32 the consumers read no fragments (MFMAs on register constants);
48 neither (barriers + MFMAs + epilogue); 64 fragment reads but no MFMAs; 128 every workgroup computes tile (0, 0); 160 that, and no fragment reads.
Cin = 640 and Cin = 1280: the difference is 90 K steps (taps x 64-channel chunks)."""
import ctypes as C
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]


def conv(B, H, C0, Cout, mode, iters=50):
    ms = C.c_double()
    lib.agd_bench_conv(B, H, H, C0, 0, Cout, 3, 1, 1, mode, 0, iters, C.byref(ms))
    return ms.value * 1e3


cfgs = [0, 16, 32, 48, 64, 128, 160]
print(f"{'kernel':10s}" + "".join(f"{('cfg' + str(c)):>17s}" for c in cfgs) + "    (us at Cin = 640 / us at Cin = 1280 / us per K step)")
for name, mode in (("pch", 8 | 256 | (1 << 16) | (1 << 15)), ("row-halo", 8 | 256 | (1 << 15))):
    row = []
    for c in cfgs:
        lib.agd_set_igemm_cfg(c)
        t1, t2 = conv(8, 32, 640, 640, mode), conv(8, 32, 1280, 640, mode)
        row.append(f"{t1:5.1f}/{t2:5.1f}/{(t2 - t1) / 90:5.3f}")
    lib.agd_set_igemm_cfg(0)
    print(f"{name:10s}" + "".join(f"{r:>17s}" for r in row), flush=True)
