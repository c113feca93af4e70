import ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
import sys
lib.agd_set_igemm_cfg(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
def conv(B, H, C0, Cout, k=1, geglu=0, res=0, iters=20):
    ms = C.c_double(); lib.agd_bench_conv(B, H, H, C0, 0, Cout, k, 1, 1, geglu, res, iters, C.byref(ms)); return ms.value * 1e3
for geglu, N, res in ((1, 2560, 0), (0, 1280, 0), (0, 1280, 1), (0, 320, 0), (0, 320, 1)):
    row = []
    for K in (64, 128, 320, 640, 1280):
        conv(8, 64, K, N, geglu=geglu, res=res, iters=3)
        row.append(f"K={K}: {conv(8, 64, K, N, geglu=geglu, res=res):6.1f}")
    print(f"M=32768 N={N} geglu={geglu} res={res}:  " + "  ".join(row))
