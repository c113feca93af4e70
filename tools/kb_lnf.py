import ctypes as C, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
lib = C.CDLL(os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so"))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
def conv(B, H, C0, C1, Cout, k, mode, res, stride=1, up=1, iters=30):
    ms = C.c_double()
    rc = lib.agd_bench_conv(B, H, H, C0, C1, Cout, k, stride, up, mode, res, iters, C.byref(ms))
    return ms.value * 1e3 if rc == 0 else float("nan")
shapes = [("L0 qkv N960", (8, 64, 320, 0, 960, 1, 0, 0)), ("L0 geglu N2560", (8, 64, 320, 0, 2560, 1, 1, 0)),
          ("L1 qkv N1920", (8, 32, 640, 0, 1920, 1, 0, 0)), ("L1 geglu N5120", (8, 32, 640, 0, 5120, 1, 1, 0)), ("L0 toq N320", (8, 64, 320, 0, 320, 1, 0, 0))]
print(f"{'shape':20s} {'4w':>8s} {'4w+lnf':>8s} {'8p':>8s} {'8p+lnf':>8s}")
for name, (B, H, C0, C1, Cout, k, g, res) in shapes:
    t = [conv(B, H, C0, C1, Cout, k, g | m, res) for m in (0, 4, 32, 32 | 4)]
    print(f"{name:20s}" + "".join(f"{x:8.1f}" for x in t), flush=True)
