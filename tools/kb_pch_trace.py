#!/usr/bin/env python3
"""In-kernel time line of igemm_pch_kernel (stamps library: `make -C agenda_amd/csrc stamps`, AGD_IGEMM_CFG bit 10; bit 11: loader wave 0 instead of consumer wave 0) on the 3x3 conv 640 -> 640 of the
32 x 32 maps at UNet batch 8.  Consumer marks: 1 start | per step 2 at the barrier, 3 behind it | 5 loop left, 6 epilogue done.
Loader marks: 1 start | per step 2 top, 3 operands landed, 4 behind the barrier | 5 loop left, 6 drained.  Ticks of s_memtime from the start."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_stamps.so")))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
lib.agd_smap_ts.argtypes = [C.c_int, C.POINTER(C.c_ulonglong)]
C0 = int(sys.argv[1]) if len(sys.argv) > 1 else 640
H = int(os.environ.get("KB_H", "32")); N = int(os.environ.get("KB_N", "640"))      # KB_H=16 KB_N=1280: the 16 x 16 maps (2 K slices)
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0       # further AGD_IGEMM_CFG bits
brief = len(sys.argv) > 3
for who, bit in (("consumer wave 0", 0), ("loader wave 0", 2048)):
    for wg in (0, 100):
        lib.agd_smap_ts(wg, None)
        lib.agd_set_igemm_cfg(1024 | bit | extra)
        ms = C.c_double()
        lib.agd_bench_conv(8, H, H, C0, 0, N, 3, 1, 1, 8 | 256 | (1 << 16) | (1 << 15), 0, 20, C.byref(ms))
        buf = (C.c_ulonglong * 1024)()
        lib.agd_smap_ts(0, buf)
        n = int(buf[1023])
        ev = [(int(buf[i]) >> 56, int(buf[i]) & ((1 << 56) - 1)) for i in range(n)]
        t0 = ev[0][1]
        rt = (int(buf[1021]) - int(buf[1020])) * 10e-9
        print(f"   wave lifetime {rt * 1e6:.2f} us (s_memrealtime), {ev[-1][1] - t0} s_memtime ticks: {(ev[-1][1] - t0) / rt / 1e9:.3f} GHz")
        print(f"{who}, workgroup {wg}: {n} marks, {ms.value * 1e3:.1f} us per launch; ticks from the start (one line per K step):")
        line = []
        if brief: continue
        for k, t in ev:
            if k == 2 and line: print("   " + " ".join(line)); line = []
            line.append(f"{k}:{t - t0}")
        print("   " + " ".join(line))
