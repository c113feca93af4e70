#!/usr/bin/env python3
"""In-kernel time line of igemm_smap_kernel (stamps library: `make -C agenda_amd/csrc stamps`, AGD_IGEMM_CFG bit 10): wave 0 of one workgroup stores s_memtime at marks
1 start | per step: 2 top, (3 behind the chunk barrier, 4 image issued), 5 operands landed, 6 behind the step barrier, 7 weight piece issued | 8 loop left, 9 drained, 10 epilogue done.
Prints the gaps in the counter's ticks, scaled so that start -> end matches the launch's duration."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_stamps.so")))
lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
lib.agd_smap_ts.argtypes = [C.c_int, C.POINTER(C.c_ulonglong)]
C0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for wg in (0, 7, 19):
    lib.agd_smap_ts(wg, None)
    lib.agd_set_igemm_cfg(1024 | cfg)
    ms = C.c_double()
    lib.agd_bench_conv(8, 8, 8, C0, 0, 1280, 3, 1, 1, 8 | 256, 0, 20, C.byref(ms))
    buf = (C.c_ulonglong * 1024)()
    lib.agd_smap_ts(0, buf)
    n = int(buf[1023])
    ev = [(int(buf[i]) >> 56, int(buf[i]) & ((1 << 56) - 1)) for i in range(n)]
    t0 = ev[0][1]
    print(f"workgroup {wg}: {n} marks, {ms.value * 1e3:.1f} us per launch (with the slab pass); ticks from the start:")
    line = []
    for k, t in ev:
        if k == 2 and line: print("   " + " ".join(line)); line = []
        line.append(f"{k}:{t - t0}")
    print("   " + " ".join(line))
