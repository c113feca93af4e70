#!/usr/bin/env python3
"""Per-shape kernel micro-benchmarks over the SD-1.5 layer shapes at UNet batch 8 (config 2).
Usage (GPU box): python tools/kbench.py [conv|lin|attn|gn|all]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
lib.agd_bench_attention.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_double)]
lib.agd_bench_groupnorm.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_double)]
lib.agd_last_error.restype = C.c_char_p
lib.agd_last_error.argtypes = [C.c_void_p]


def conv(B, H, C0, C1, Cout, k=3, stride=1, up=1, geglu=0, res=0, iters=20, cnt=1):
    ms = C.c_double()
    rc = lib.agd_bench_conv(B, H, H, C0, C1, Cout, k, stride, up, geglu, res, iters, C.byref(ms))
    if rc:
        print("ERR", lib.agd_last_error(None)); return 0
    Ho = H * up // stride
    fl = 2.0 * B * Ho * Ho * Cout * k * k * (C0 + C1)
    print(f"  conv{k}x{k} B{B} {H}x{H} {C0}+{C1}->{Cout} s{stride} up{up} geglu{geglu}: {ms.value*1e3:8.1f} us  {fl/ms.value/1e9:7.1f} TF/s  (x{cnt}/fwd -> {ms.value*cnt:.3f} ms)")
    return ms.value * cnt


def attn(B, H, D, Nq, Nk, record=0, iters=20, cnt=1):
    ms = C.c_double()
    rc = lib.agd_bench_attention(B, H, D, Nq, Nk, record, iters, C.byref(ms))
    if rc:
        print("ERR", lib.agd_last_error(None)); return 0
    fl = 4.0 * B * H * Nq * Nk * D
    extra = f"  rec {B//2*H*Nk*Nq*8/ms.value/1e9:6.2f} TB/s" if record else ""
    print(f"  attn B{B} H{H} D{D} Nq{Nq} Nk{Nk} rec{record}: {ms.value*1e3:8.1f} us  {fl/ms.value/1e9:7.1f} TF/s{extra} (x{cnt} -> {ms.value*cnt:.3f} ms)")
    return ms.value * cnt


def gn(B, HW, Cc, iters=20, cnt=1):
    ms = C.c_double()
    lib.agd_bench_groupnorm(B, HW, Cc, iters, C.byref(ms))
    by = B * HW * Cc * 2 * 3
    print(f"  gn B{B} HW{HW} C{Cc}: {ms.value*1e3:8.1f} us  {by/ms.value/1e9:6.2f} TB/s (3-pass bytes) (x{cnt} -> {ms.value*cnt:.3f} ms)")
    return ms.value * cnt


what = sys.argv[1] if len(sys.argv) > 1 else "all"
if len(sys.argv) > 2:            # experiment knob: only in builds made with `make EXTRA=-DAGD_EXPERIMENTS`
    lib.agd_set_igemm_cfg(int(sys.argv[2]))
    print("igemm cfg", sys.argv[2])
B = 8
tot = 0
if what in ("conv", "all"):
    print("== 3x3 convs (UNet batch 8)")
    t = 0
    t += conv(B, 64, 320, 0, 320, cnt=4 + 3 + 0)           # down0 x4, up3 conv2 x3
    t += conv(B, 64, 960, 0, 320); t += conv(B, 64, 640, 0, 320, cnt=2)
    t += conv(B, 32, 640, 0, 640, up=2)                    # up2 upsampler @64
    t += conv(B, 64, 320, 0, 320, stride=2)
    t += conv(B, 32, 320, 0, 640); t += conv(B, 32, 640, 0, 640, cnt=3 + 3)
    t += conv(B, 32, 1920, 0, 640); t += conv(B, 32, 1280, 0, 640); t += conv(B, 32, 960, 0, 640)
    t += conv(B, 16, 1280, 0, 1280, up=2); t += conv(B, 32, 640, 0, 640, stride=2)
    t += conv(B, 16, 640, 0, 1280); t += conv(B, 16, 1280, 0, 1280, cnt=3 + 3)
    t += conv(B, 16, 2560, 0, 1280, cnt=2); t += conv(B, 16, 1920, 0, 1280)
    t += conv(B, 8, 1280, 0, 1280, up=2); t += conv(B, 16, 1280, 0, 1280, stride=2)
    t += conv(B, 8, 1280, 0, 1280, cnt=4 + 4 + 3); t += conv(B, 8, 2560, 0, 1280, cnt=3)
    print(f"  -> conv3x3 total per UNet forward: {t:.3f} ms"); tot += t
if what in ("lin", "all"):
    print("== linears / 1x1 (tokens = 8*HW)")
    t = 0
    for (hw, Cc, n) in ((64, 320, 5), (32, 640, 5), (16, 1280, 5), (8, 1280, 1)):
        t += conv(B, hw, Cc, 0, Cc, k=1, res=1, cnt=5 * n)             # proj_in/out, attn out x2, to_q
        t += conv(B, hw, Cc, 0, 3 * Cc, k=1, cnt=n)                    # qkv
        t += conv(B, hw, Cc, 0, 8 * Cc, k=1, geglu=1, cnt=n)           # ff1 geglu
        t += conv(B, hw, 4 * Cc, 0, Cc, k=1, res=1, cnt=n)             # ff2
    t += conv(B, 64, 640, 320, 320, k=1); t += conv(B, 32, 1280, 640, 640, k=1); t += conv(B, 16, 1280, 1280, 1280, k=1)  # shortcuts (sample)
    print(f"  -> linear total per UNet forward: {t:.3f} ms"); tot += t
if what in ("attn", "all"):
    print("== attention")
    for qb in ((2, 1) if hasattr(lib, "agd_set_attn_qb") else (1,)):
        if hasattr(lib, "agd_set_attn_qb"):
            lib.agd_set_attn_qb(qb)
        print(f" [qb={qb}]")
        attn(B, 8, 40, 4096, 4096); attn(B, 8, 80, 1024, 1024); attn(B, 8, 160, 256, 256)
        attn(B, 5, 64, 9216, 9216, iters=5); attn(B, 10, 64, 2304, 2304); attn(B, 20, 64, 576, 576)   # SD-2.1 768 px
    t = 0
    t += attn(B, 8, 40, 4096, 4096, cnt=5); t += attn(B, 8, 80, 1024, 1024, cnt=5); t += attn(B, 8, 160, 256, 256, cnt=5)
    t += attn(B, 8, 160, 64, 64, cnt=1)
    t += attn(B, 8, 40, 4096, 77, record=1, cnt=5); t += attn(B, 8, 80, 1024, 77, record=1, cnt=5)
    t += attn(B, 8, 160, 256, 77, record=1, cnt=5); t += attn(B, 8, 160, 64, 77, record=0, cnt=1)
    print(f"  -> attention total per UNet forward: {t:.3f} ms"); tot += t
if what in ("gn", "all"):
    print("== groupnorm")
    t = 0
    t += gn(B, 4096, 320, cnt=12); t += gn(B, 4096, 960); t += gn(B, 4096, 640, cnt=2)
    t += gn(B, 1024, 640, cnt=12); t += gn(B, 1024, 1920); t += gn(B, 1024, 1280); t += gn(B, 1024, 960); t += gn(B, 1024, 320)
    t += gn(B, 256, 1280, cnt=12); t += gn(B, 256, 2560, cnt=2); t += gn(B, 256, 1920); t += gn(B, 256, 640)
    t += gn(B, 64, 1280, cnt=9); t += gn(B, 64, 2560, cnt=3)
    print(f"  -> groupnorm total per UNet forward: {t:.3f} ms"); tot += t
print(f"TOTAL (listed kernels) per UNet forward: {tot:.3f} ms -> x50 = {tot*50:.1f} ms per 4-image batch")
