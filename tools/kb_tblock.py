#!/usr/bin/env python3
"""Fused row-panel kernels of the C = 320 transformer blocks (tblock.hip) at UNet batch 8 x 64 x 64, stand-alone (hot weights):
the feed-forward kernel against the GEGLU + ff.net.2 pair it replaces, the attn2 chain against to_q + attention + to_out.
python tools/kb_tblock.py [B]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
lib.agd_bench_tblock.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_double)]
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
lib.agd_bench_attention.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_double)]
lib.agd_last_error.restype = C.c_char_p; lib.agd_last_error.argtypes = [C.c_void_p]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
HW, Cc = 4096, 320


def tb(kind):
    ms = C.c_double()
    if lib.agd_bench_tblock(kind, B, HW, 50, C.byref(ms)):
        print("ERR", lib.agd_last_error(None)); return float("nan")
    return ms.value * 1e3


def lin(K, N, geglu=0, res=0):
    ms = C.c_double()
    if lib.agd_bench_conv(B, 64, 64, K, 0, N, 1, 1, 1, geglu, res, 50, C.byref(ms)):
        print("ERR", lib.agd_last_error(None)); return float("nan")
    return ms.value * 1e3


if hasattr(lib, "agd_set_tb_variant"):          # where the feed-forward kernel's time goes (timing variants; outputs are garbage)
    names = {0: "production", 1: "no gelu arithmetic", 2: "weights loaded once", 4: "activation fragments read once", 6: "no operand traffic in the loops",
             7: "6 + no gelu", 8: "no GEGLU epilogue", 10: "8 + weights once", 14: "MFMAs + barriers only"}
    for v, nm in names.items():
        lib.agd_set_tb_variant(v)
        print(f"ff_fused variant {v:2d} ({nm}): {tb(0):7.1f} us", flush=True)
    lib.agd_set_tb_variant(0)
for rnd in range(3):
    ff = tb(0)
    g = lin(Cc, 8 * Cc, geglu=1 | 4 | 16)        # GEGLU, LayerNorm-fold consumer, 8-phase kernel allowed (as in the walk)
    f2 = lin(4 * Cc, Cc, res=1)
    fl = 2.0 * B * HW * 12 * Cc * Cc
    print(f"[{rnd}] ff_fused {ff:7.1f} us ({fl / ff / 1e6:6.0f} TF/s)   vs GEGLU {g:6.1f} + ff.net.2 {f2:6.1f} = {g + f2:6.1f} us", flush=True)
    ch = tb(1)
    tq = lin(Cc, Cc, geglu=4)                    # to_q as a LayerNorm-fold consumer
    to = lin(Cc, Cc, geglu=2, res=1)             # to_out + residual as a row-statistics producer
    ms = C.c_double(); lib.agd_bench_attention(B, 8, 40, HW, 77, 2, 50, C.byref(ms)); at = ms.value * 1e3
    print(f"[{rnd}] attn_chain {ch:7.1f} us   vs to_q {tq:6.1f} + attention(record) {at:6.1f} + to_out {to:6.1f} = {tq + at + to:6.1f} us", flush=True)
    po = lin(Cc, Cc, res=1)
    print(f"[{rnd}] ff_fused + proj_out {tb(2):7.1f} us (vs {ff:6.1f} + C->C launch {po:5.1f});  chain from attn1.to_out {tb(3):7.1f} us (vs {ch:6.1f} + C->C launch {to:5.1f})", flush=True)

# the attn2 chain at C = 640 (32 x 32 maps, 64-row panels on M / 64 = 128 workgroups) against the launches it replaces
def tb640(kind):
    ms = C.c_double()
    if lib.agd_bench_tblock(kind, B, 1024, 50, C.byref(ms)):
        print("ERR", lib.agd_last_error(None)); return float("nan")
    return ms.value * 1e3


def lin32(K, N, geglu=0, res=0):
    ms = C.c_double()
    if lib.agd_bench_conv(B, 32, 32, K, 0, N, 1, 1, 1, geglu, res, 50, C.byref(ms)):
        return float("nan")
    return ms.value * 1e3


for rnd in range(2):
    tq, to = lin32(640, 640, geglu=4), lin32(640, 640, geglu=2, res=1)
    ms = C.c_double(); lib.agd_bench_attention(B, 8, 80, 1024, 77, 1, 50, C.byref(ms)); at = ms.value * 1e3
    print(f"[{rnd}] C=640 attn_chain {tb640(4):6.1f} us (from attn1.to_out {tb640(5):6.1f})   vs to_q {tq:5.1f} + attention(record) {at:5.1f} + to_out {to:5.1f} = {tq + at + to:5.1f} (+ {to:5.1f})", flush=True)
