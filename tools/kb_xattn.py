import ctypes as C, os
lib = C.CDLL(os.environ.get("AGD_LIB", "agenda_amd/libagenda_hip_exp.so"))
lib.agd_bench_attention.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_double)]
for (B,H,D,Nq) in ((8,8,40,4096),(8,8,80,1024),(8,8,160,256)):
    for rec in (0,1,2,3,4):
        ms = C.c_double(); lib.agd_bench_attention(B,H,D,Nq,77,rec,20,C.byref(ms))
        print(B,H,D,Nq,"record",rec,f"{ms.value*1e3:.1f} us")
