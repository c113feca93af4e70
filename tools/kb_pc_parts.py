#!/usr/bin/env python3
"""Where does a K step of the producer / consumer igemm (igemm_pc.h) go?  Timing variants of the experiments library on the 64 x 160-tile 1x1 launches of the 16 x 16 maps
(hot operands, us per launch): cfg 0 production; 16 the loaders issue no LDS-DMA; 32 the consumers read no fragments (MFMAs on register constants); 48 no DMA and no fragment reads
(barriers + MFMAs + epilogue); 128 every workgroup computes tile (0, 0) (one A and one W panel for the whole chip: the memory side at its best); 160 that, and no fragment reads.  K = 1280 and K = 2560: the difference is 20 K steps."""
import ctypes as C
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]


def conv(B, H, C0, Cout, mode, iters=50):
    ms = C.c_double()
    lib.agd_bench_conv(B, H, H, C0, 0, Cout, 1, 1, 1, mode, 0, iters, C.byref(ms))
    return ms.value * 1e3


cfgs = [0, 16, 32, 48, 128, 160]
print(f"{'kernel':10s}" + "".join(f"{('cfg' + str(c)):>16s}" for c in cfgs) + "    (us at K = 1280 / us at K = 2560 / us per K step)")
for name, mode in (("pc", 1 << 11), ("4-wave", 0)):
    row = []
    for c in cfgs:
        lib.agd_set_igemm_cfg(c)
        t1, t2 = conv(8, 16, 1280, 1280, mode), conv(8, 16, 2560, 1280, mode)
        row.append(f"{t1:5.1f}/{t2:5.1f}/{(t2 - t1) / 20:5.3f}")
    lib.agd_set_igemm_cfg(0)
    print(f"{name:10s}" + "".join(f"{r:>16s}" for r in row), flush=True)
