"""64-row tile configurations (64x160, 64x128) against the production choice on the small-M shapes (experiments library, AGD_IGEMM_FORCE).
python tools/kb_force64.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = [(16, 1280, 1280, 1, 1), (16, 1280, 1280, 1, 0), (16, 1280, 3840, 1, 0), (16, 5120, 1280, 1, 1), (16, 2560, 1280, 1, 0), (32, 640, 640, 1, 1), (32, 2560, 640, 1, 1),
          (8, 1280, 1280, 1, 1), (8, 5120, 1280, 1, 1), (8, 1280, 3840, 1, 0), (8, 1280, 1280, 3, 1), (16, 1280, 1280, 3, 1)]
for sh in shapes:
    print("shape H=%d Cin=%d Cout=%d k=%d res=%d" % sh, flush=True)
    for force in ("", "1064:1:2", "1064:1:4", "2064:1:2", "2064:1:4", "1064:2:4", "1064:4:4", "2064:2:4"):
        env = dict(os.environ)
        if force: env["AGD_IGEMM_FORCE"] = force
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kb_force.py"), "--one"] + [str(x) for x in sh], env=env, capture_output=True, text=True)
        print(f"   {force or 'default':>10}: {r.stdout.strip()}", flush=True)
