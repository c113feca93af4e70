#!/usr/bin/env python3
"""Where does a short-K linear launch spend its time?  Times the SD-1.5 linear shapes (UNet batch 8) under the timing
experiments of an EXPERIMENTS build (make EXTRA=-DAGD_EXPERIMENTS; AGD_LIB=<path to that .so>):
  cfg 0 = production path; 16 = A loads dropped; 32 = B loads dropped; 48 = both; 64 = no DMA instructions;
  512 = dispatch only; 1024 = no epilogue.  Usage (GPU box): AGD_LIB=/tmp/exp/libagenda_hip.so python tools/kb_lin.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
has_cfg = hasattr(lib, "agd_set_igemm_cfg")


def conv(B, H, C0, Cout, k=1, geglu=0, res=0, iters=30):
    ms = C.c_double()
    lib.agd_bench_conv(B, H, H, C0, 0, Cout, k, 1, 1, geglu, res, iters, C.byref(ms))
    return ms.value * 1e3


shapes = [("L0 C->C   M32768 K320  N320 ", (8, 64, 320, 320, 1, 0, 1)),
          ("L0 qkv    M32768 K320  N960 ", (8, 64, 320, 960, 1, 0, 0)),
          ("L0 geglu  M32768 K320  N2560", (8, 64, 320, 2560, 1, 1, 0)),
          ("L0 ff2    M32768 K1280 N320 ", (8, 64, 1280, 320, 1, 0, 1)),
          ("L1 C->C   M8192  K640  N640 ", (8, 32, 640, 640, 1, 0, 1)),
          ("L1 geglu  M8192  K640  N5120", (8, 32, 640, 5120, 1, 1, 0)),
          ("L1 ff2    M8192  K2560 N640 ", (8, 32, 2560, 640, 1, 0, 1)),
          ("L2 C->C   M2048  K1280 N1280", (8, 16, 1280, 1280, 1, 0, 1)),
          ("L2 qkv    M2048  K1280 N3840", (8, 16, 1280, 3840, 1, 0, 0)),
          ("L2 geglu  M2048  K1280 N10240", (8, 16, 1280, 10240, 1, 1, 0)),
          ("L2 ff2    M2048  K5120 N1280", (8, 16, 5120, 1280, 1, 0, 1)),
          ("conv3 L0  M32768 K2880 N320 ", (8, 64, 320, 320, 3, 0, 1)),
          ("conv3 L1  M8192  K5760 N640 ", (8, 32, 640, 640, 3, 0, 1)),
          ("conv3 L2  M2048  K11520 N1280", (8, 16, 1280, 1280, 3, 0, 1))]
if os.environ.get("KB_LN"):      # LayerNorm-fold overheads: producers (C->C with residual) +2, consumers (qkv / geglu) +4
    shapes = [(n + " plain", a) for n, a in shapes[:11]] + [(n + " +stat", a[:5] + (a[5] | 2,) + a[6:]) for n, a in shapes[:11] if "C->C" in n] + \
             [(n + " +lnf ", a[:5] + (a[5] | 4,) + a[6:]) for n, a in shapes[:11] if "qkv" in n or "geglu" in n]
cfgs = [int(x) for x in os.environ.get("KB_CFGS", "0,16,32,48,64,512,1024").split(",")] if has_cfg else [0]
print(f"{'shape':34s}" + "".join(f"{('cfg' + str(c)):>9s}" for c in cfgs) + "   TF/s(cfg0)")
for name, a in shapes:
    row = []
    for c in cfgs:
        if has_cfg:
            lib.agd_set_igemm_cfg(c)
        row.append(conv(*a))
    B, H, C0, Cout, k = a[:5]
    fl = 2.0 * B * H * H * Cout * k * k * C0
    print(f"{name:34s}" + "".join(f"{t:9.1f}" for t in row) + f"   {fl / row[0] / 1e6:7.0f}", flush=True)
