"""Explore igemm configurations for one shape: python tools/kb_force.py M_side Cin Cout ksize [res]   (B=8)
Needs an experiments build of the library: make -C agenda_amd/csrc clean all EXTRA=-DAGD_EXPERIMENTS (AGD_IGEMM_FORCE)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
    lib.agd_bench_conv.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_double)]
    H, C0, Cout, k, res = map(int, sys.argv[2:7])
    ms = C.c_double()
    for it in (3, 20):
        lib.agd_bench_conv(8, H, H, C0, 0, Cout, k, 1, 1, 0, res, it, C.byref(ms))
    print(f"{ms.value * 1e3:8.1f} us  {2.0 * 8 * H * H * Cout * k * k * C0 / ms.value / 1e9:7.1f} TF/s")
else:
    shapes = [(8, 1280, 1280, 3, 1), (8, 2560, 1280, 3, 0), (16, 1280, 1280, 3, 1), (16, 1280, 1280, 1, 1), (32, 640, 640, 1, 1), (64, 320, 320, 1, 1), (16, 5120, 1280, 1, 1), (32, 2560, 640, 1, 1), (16, 1280, 3840, 1, 0), (64, 320, 960, 1, 0)]
    for sh in shapes:
        print("shape H=%d Cin=%d Cout=%d k=%d res=%d" % sh, flush=True)
        for force in (sys.argv[1:] or ("", "128:1:2", "128:1:4", "160:1:2", "160:1:4", "64:1:2", "64:1:4", "128:2:4", "160:2:4", "128:2:2", "64:2:4", "1264:1:2", "1264:1:3")):
            env = dict(os.environ); 
            if force: env["AGD_IGEMM_FORCE"] = force
            r = subprocess.run([sys.executable, __file__, "--one"] + [str(x) for x in sh], env=env, capture_output=True, text=True)
            print(f"   {force or 'default':>10}: {r.stdout.strip()}", flush=True)
