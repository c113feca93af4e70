#!/usr/bin/env python3
"""In-kernel time line of qkv_chain_kernel<320> (stamps library: `make -C agenda_amd/csrc stamps`): one wave of one workgroup stores s_memtime at
1 start | 2 panel DMA + first weight fragments requested | 3 GroupNorm statistics reduced | 4 panel landed | 5 GroupNorm applied in the panel | 6 proj_in GEMM done |
7 h stored + row sums staged | 8 barrier | 9 norm1 in the panel | 10 + s GEMM of q / k / v done | 20 + s its stores issued | 99 stores drained.
python tools/kb_qkv_trace.py [kinds]   (6 = one 128-row workgroup per CU, 7 = two co-resident 64-row workgroups; 8 / 9: the same on qkv_chain2_kernel, round 6's schedule)"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_stamps.so")))
lib.agd_bench_tblock.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_double)]
lib.agd_tb_ts.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
for kind in ((6, 7, 8, 9) if len(sys.argv) < 2 else [int(a) for a in sys.argv[1:]]):
    for wg, wave in ((0, 0), (1, 0), (100, 0), (100, 3)):
        lib.agd_tb_ts(wg, wave, None)
        ms = C.c_double()
        if lib.agd_bench_tblock(kind, 8, 4096, 20, C.byref(ms)) != 0:
            raise SystemExit("agd_bench_tblock failed")
        buf = (C.c_ulonglong * 256)()
        lib.agd_tb_ts(0, 0, buf)
        n = int(buf[255])
        ev = [(int(buf[i]) >> 56, int(buf[i]) & ((1 << 56) - 1)) for i in range(n)]
        rt = (int(buf[251]) - int(buf[250])) * 10e-9
        t0 = ev[0][1]
        print(f"kind {kind} workgroup {wg} wave {wave}: {ms.value * 1e3:.1f} us per launch; wave lifetime {rt * 1e6:.1f} us, {(ev[-1][1] - t0) / rt / 1e9:.2f} GHz; mark:ticks (delta)")
        print("   " + "  ".join(f"{k}:{t - t0}(+{t - p})" for (k, t), p in zip(ev, [t0] + [e[1] for e in ev[:-1]])))
