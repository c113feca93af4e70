import ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_attention.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_double)]
ms = C.c_double(); lib.agd_bench_attention(8, 8, 40, 4096, 4096, 0, 6, C.byref(ms)); print("d40 N4096", ms.value * 1e3, "us")
