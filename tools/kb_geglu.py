import ctypes as C, os
ROOT = "/root/repo" if os.path.exists("/root/repo/agenda_amd") else os.environ.get("GRAFT_REPO_ROOT", ".")
lib = C.CDLL(os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so"))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
def conv(B, H, C0, Cout, mode, iters=50):
    ms = C.c_double(); rc = lib.agd_bench_conv(B, H, H, C0, 0, Cout, 1, 1, 1, mode, 0, iters, C.byref(ms))
    return ms.value * 1e3 if rc == 0 else float("nan")
for name, a in (("L2 GEGLU M2048 K1280 N10240", (8, 16, 1280, 10240)), ("L1 GEGLU M8192 K640 N5120", (8, 32, 640, 5120)), ("L3 GEGLU M512", (8, 8, 1280, 10240))):
    row = {nm: conv(*a, mode=1 | 4 | m) for nm, m in (("4-wave", 0), ("8p auto", 16), ("8p 256 forced", 32), ("wreg", 128), ("wreg+blocks", 128 | (1 << 15)), ("4-wave+blocks", 1 << 15), ("8p forced+blocks", 32 | (1 << 15)))}
    print(name, " ".join(f"{k}: {v:.1f}" for k, v in row.items()), flush=True)
