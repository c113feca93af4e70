import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so"))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
cfg = int(sys.argv[1]); C0 = int(sys.argv[2])
lib.agd_set_igemm_cfg(cfg)
ms = C.c_double()
lib.agd_bench_conv(8, 8, 8, C0, 0, 1280, 3, 1, 1, 8 | 256, 0, 50, C.byref(ms))
print(cfg, C0, ms.value * 1e3)
