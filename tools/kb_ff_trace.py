#!/usr/bin/env python3
"""In-kernel time line of ff_fused_kernel (stamps library: `make -C agenda_amd/csrc stamps`): 1 start | 2 panel + statistics | 3 first two intervals | per steady interval: 10 GEMM2 done, 11 GEMM1 done,
12 GEGLU done, 13 behind the barrier | 4 steady loops left | 5 last two intervals | 99 epilogue + stores drained.  python tools/kb_ff_trace.py [kind]   (kind 2: with the proj_out stage, the production form)"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_stamps.so")))
lib.agd_bench_tblock.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_double)]
lib.agd_tb_ts.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for wg, wave in ((0, 0), (0, 5), (100, 2), (100, 7)):
    lib.agd_tb_ts(wg, wave, None)
    ms = C.c_double()
    lib.agd_bench_tblock(kind, 8, 4096, 20, C.byref(ms))
    buf = (C.c_ulonglong * 256)()
    lib.agd_tb_ts(0, 0, buf)
    n = int(buf[255])
    ev = [(int(buf[i]) >> 56, int(buf[i]) & ((1 << 56) - 1)) for i in range(n)]
    rt = (int(buf[251]) - int(buf[250])) * 10e-9
    t0 = ev[0][1]
    print(f"workgroup {wg} wave {wave}: {ms.value * 1e3:.1f} us per launch; wave lifetime {rt * 1e6:.1f} us, {(ev[-1][1] - t0) / rt / 1e9:.2f} GHz; mark:ticks (delta)")
    print("   " + "  ".join(f"{k}:{t - t0}(+{t - p})" for (k, t), p in zip(ev, [t0] + [e[1] for e in ev[:-1]])))
