"""Tile configuration sweep with COLD weights + the in-kernel warm-up (what a launch meets inside the UNet walk) for the 32x32 / 16x16 / 8x8
shapes at UNet batch 8: the launcher's choice against forced configurations (experiments library, AGD_IGEMM_FORCE=<tile>:<K slices>:<ring>;
tile 128 / 160 / 64 / 1064 (= 64x160) / 2064 (= 64x128)).  python tools/sweep_cold.py [quick]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    lib = C.CDLL(os.environ.get("AGD_LIB", os.path.join(ROOT, "agenda_amd", "libagenda_hip_exp.so")))
    lib.agd_bench_conv_cold.argtypes = [C.c_int] * 10 + [C.POINTER(C.c_double)]
    H, C0, Cout, k, res = map(int, sys.argv[2:7])
    geglu = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    ms = C.c_double()
    rc = lib.agd_bench_conv_cold(8, H, H, C0, Cout, k, geglu, res, 3, 8, C.byref(ms))
    print(f"{ms.value * 1e3:.1f}" if rc == 0 else "nan")
    sys.exit(0)
shapes = [(32, 640, 640, 1, 1), (32, 640, 1920, 1, 0), (32, 2560, 640, 1, 1), (32, 1280, 640, 1, 0), (32, 1920, 640, 1, 0), (32, 960, 640, 1, 0),
          (32, 640, 640, 3, 1), (32, 1280, 640, 3, 0), (32, 1920, 640, 3, 0), (32, 960, 640, 3, 0), (32, 320, 640, 3, 0),
          (16, 1280, 1280, 1, 1), (16, 1280, 3840, 1, 0), (16, 5120, 1280, 1, 1), (16, 2560, 1280, 1, 0), (16, 1920, 1280, 1, 0), (16, 640, 1280, 1, 0),
          (16, 1280, 1280, 3, 1), (16, 2560, 1280, 3, 0), (16, 1920, 1280, 3, 0), (16, 640, 1280, 3, 0),
          (8, 1280, 1280, 1, 1), (8, 1280, 3840, 1, 0), (8, 5120, 1280, 1, 1), (8, 2560, 1280, 1, 0), (8, 1280, 1280, 3, 1), (8, 2560, 1280, 3, 0)]
if len(sys.argv) > 1 and sys.argv[1] == "l0":      # the 64x64 maps (M = 32768): no K slices
    shapes = [(64, 320, 320, 1, 1), (64, 320, 320, 1, 0), (64, 320, 960, 1, 0), (64, 1280, 320, 1, 1), (64, 960, 320, 1, 0), (64, 640, 320, 1, 0),
              (64, 320, 320, 3, 1), (64, 640, 320, 3, 0), (64, 960, 320, 3, 0)]
if len(sys.argv) > 1 and sys.argv[1] == "geglu":   # GEGLU launches (sixth field 1): 128x128 on the 2- / 4-stage ring; default = the launcher (8-phase kernel where it applies)
    shapes = [(64, 320, 2560, 1, 0, 1), (32, 640, 5120, 1, 0, 1), (16, 1280, 10240, 1, 0, 1), (8, 1280, 10240, 1, 0, 1)]
mode = sys.argv[1] if len(sys.argv) > 1 else ""
cfgs = ["", "128:1:2", "128:1:4", "160:1:2", "160:1:4", "64:1:4", "1064:1:4", "1064:1:2", "2064:1:4", "128:2:4", "160:2:4", "1064:2:4", "128:4:4", "160:4:4", "1064:4:4", "160:8:4", "128:8:4"]
if mode == "l0": cfgs = ["", "128:1:2", "128:1:4", "160:1:2", "160:1:4", "64:1:2", "64:1:4", "1064:1:2", "1064:1:4", "2064:1:2", "2064:1:4"]
if mode == "geglu": cfgs = ["", "128:1:2", "128:1:4"]
print(f"{'shape (H Cin Cout k res)':28s}" + "".join(f"{(c or 'default'):>10s}" for c in cfgs), flush=True)
for sh in shapes:
    row = []
    for force in cfgs:
        env = dict(os.environ)
        if force: env["AGD_IGEMM_FORCE"] = force
        r = subprocess.run([sys.executable, __file__, "--one"] + [str(x) for x in sh], env=env, capture_output=True, text=True)
        try: row.append(float(r.stdout.strip().split()[-1]))
        except Exception: row.append(float("nan"))
    best = min(x for x in row if x == x)
    print(f"{str(sh):28s}" + "".join(f"{x:10.1f}" for x in row) + f"   best {cfgs[row.index(best)] or 'default'} ({100 * (row[0] - best) / row[0]:.0f} %)", flush=True)
