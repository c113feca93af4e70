"""Launch list of tools/kb_fetch.sh: four 3x3 shapes x (row-halo, + XCD blocks, producer / consumer + XCD blocks), 10 launches each (experiments library)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so"))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
XB, PCH = 1 << 15, 1 << 16
shapes = [(8, 16, 1280, 0, 1280), (8, 16, 1280, 1280, 1280), (8, 32, 640, 0, 640), (8, 32, 1280, 640, 640)]
for s in shapes:
    for m in (0, XB, PCH | XB):
        ms = C.c_double()
        lib.agd_bench_conv(s[0], s[1], s[1], s[2], s[3], s[4], 3, 1, 1, 8 | 256 | m, 0, 8, C.byref(ms))
        print(s, m, ms.value * 1e3, flush=True)
