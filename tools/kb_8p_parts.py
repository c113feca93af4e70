#!/usr/bin/env python3
"""Where does a short-K 8-phase launch spend its time?  production | no epilogue | dispatch only (experiments library knobs)."""
import ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
def conv(B, H, C0, Cout, k, mode, res, iters=30):
    ms = C.c_double(); lib.agd_bench_conv(B, H, H, C0, 0, Cout, k, 1, 1, mode, res, iters, C.byref(ms)); return ms.value * 1e3
shapes = [("L0 geglu N2560 8p+lnf", (8, 64, 320, 2560, 1, 1 | 4 | 32, 0)), ("L0 qkv N960 8p+lnf", (8, 64, 320, 960, 1, 4 | 32, 0)),
          ("L1 geglu N5120 8p+lnf", (8, 32, 640, 5120, 1, 1 | 4 | 32, 0)), ("VAE 128px 512->512 8p", (4, 128, 512, 512, 3, 32, 1)),
          ("L0 geglu 4w+lnf", (8, 64, 320, 2560, 1, 1 | 4, 0))]
print(f"{'shape':26s} {'prod':>8s} {'no-epi':>8s} {'dispatch':>8s}")
for name, a in shapes:
    row = []
    for cfg in (0, 1024, 512):
        lib.agd_set_igemm_cfg(cfg); row.append(conv(*a))
    print(f"{name:26s}" + "".join(f"{t:8.1f}" for t in row), flush=True)
