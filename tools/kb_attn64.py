import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_attention.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_double)]
def attn(B, H, D, Nq, Nk, record=0, iters=10):
    ms = C.c_double(); lib.agd_bench_attention(B, H, D, Nq, Nk, record, iters, C.byref(ms))
    print(f"attn B{B} H{H} D{D} N{Nq}: {ms.value*1e3:8.1f} us {4.0*B*H*Nq*Nk*D/ms.value/1e9:7.1f} TF/s")
for _ in range(2):
    attn(8, 8, 40, 4096, 4096); attn(8, 5, 64, 9216, 9216, iters=5); attn(8, 10, 64, 2304, 2304); attn(8, 20, 64, 576, 576); attn(8, 8, 80, 1024, 1024)
