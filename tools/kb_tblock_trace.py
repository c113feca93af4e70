#!/usr/bin/env python3
"""In-kernel time line of attn_chain_kernel (stamps library: `make -C agenda_amd/csrc stamps`): one wave of one workgroup stores s_memtime at the phase marks
1 start | 2 panel + K/V requested, 3 landed | (PRE: 4 to_out(attn1) GEMM done, 5 h1 stored + statistics) | 6 norm2 in the panel | 7 .. 8 to_q GEMM | 9 Q in the panel |
10 + h head h's attention done, 20 + h next head's K / V staged | 30 recorder flushed | 31 | 32 to_out GEMM done | 99 stores drained.
python tools/kb_tblock_trace.py [kind]   (kind 3: C = 320 from attn1.to_out, the production form; 1: plain; 5 / 4: C = 640)"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_stamps.so")))
lib.agd_bench_tblock.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_double)]
lib.agd_tb_ts.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 3
HW = 4096 if kind < 4 else 1024
for wg, wave in ((0, 0), (0, 5), (100, 0), (100, 7)):
    lib.agd_tb_ts(wg, wave, None)
    ms = C.c_double()
    lib.agd_bench_tblock(kind, 8, HW, 20, C.byref(ms))
    buf = (C.c_ulonglong * 256)()
    lib.agd_tb_ts(0, 0, buf)
    n = int(buf[255])
    ev = [(int(buf[i]) >> 56, int(buf[i]) & ((1 << 56) - 1)) for i in range(n)]
    rt = (int(buf[251]) - int(buf[250])) * 10e-9
    t0 = ev[0][1]
    print(f"workgroup {wg} wave {wave}: {ms.value * 1e3:.1f} us per launch; wave lifetime {rt * 1e6:.1f} us, {(ev[-1][1] - t0) / rt / 1e9:.2f} GHz; mark:ticks (delta)")
    prev = t0
    print("   " + "  ".join(f"{k}:{t - t0}(+{t - p})" for (k, t), p in zip(ev, [t0] + [e[1] for e in ev[:-1]])))
