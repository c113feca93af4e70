#!/usr/bin/env python3
"""In-kernel time line of igemm8p_kernel (stamps library: `make -C agenda_amd/csrc stamps`, AGD_IGEMM_CFG bit 10; bit 11: wave 4) on 1x1 launches at M = 8192:
1 start | 2 prologue issued | 3 first K tile landed | 4 per K tile | 5 loop left | 6 epilogue starts | 7 epilogue issued | 8 stores drained.
python tools/kb_8p_trace.py K N [geglu]"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_stamps.so")))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
lib.agd_smap_ts.argtypes = [C.c_int, C.POINTER(C.c_ulonglong)]
K = int(sys.argv[1]) if len(sys.argv) > 1 else 640
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
geglu = int(sys.argv[3]) if len(sys.argv) > 3 else 0
mode = (1 | 4 | 16) if geglu else 16          # as the walk: GEGLU + LayerNorm-fold consumer; 8-phase kernel allowed
for bit in (0, 2048):
    for wg in (0, 100):
        lib.agd_smap_ts(wg, None)
        lib.agd_set_igemm_cfg(1024 | bit)
        ms = C.c_double()
        lib.agd_bench_conv(8, 32, 32, K, 0, N, 1, 1, 1, mode, 0, 20, C.byref(ms))
        buf = (C.c_ulonglong * 1024)()
        lib.agd_smap_ts(0, buf)
        n = int(buf[1023])
        ev = [(int(buf[i]) >> 56, int(buf[i]) & ((1 << 56) - 1)) for i in range(n)]
        t0 = ev[0][1]
        rt = (int(buf[1021]) - int(buf[1020])) * 10e-9
        print(f"wave {4 if bit else 0}, workgroup {wg}: {ms.value * 1e3:.1f} us per launch; wave lifetime {rt * 1e6:.1f} us, {(ev[-1][1] - t0) / rt / 1e9:.2f} GHz")
        print("   " + "  ".join(f"{k}:{t - t0}(+{t - p})" for (k, t), p in zip(ev, [t0] + [e[1] for e in ev[:-1]])))
