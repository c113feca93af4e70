"""The N > 1 path on hardware.  On a box with >= 2 GPUs: `bench.py --gpus 2` self-launches two ranks over RCCL.  On the
1-GPU test box: the same two-rank code paths rehearsed with both ranks on cuda:0 and gloo as the transport (the RCCL
collective itself needs two devices), for bench.py and for the seed-sharded generation driver with a ragged last round."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _torchrun(nproc, args, extra_env=None, timeout=900):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL between two devices)")
def test_bench_self_launches_two_rccl_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--ddim-steps", "2",
                        "--no-cpu-baseline", "--no-profile"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8 and line["config"]["collective_backend"] == "nccl"


def test_bench_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(n + 1), "--steps", "1"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and f"only {n} visible" in (r.stderr + r.stdout)


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """Two ranks of bench.py (launcher environment as the driver provides it) sharing cuda:0, gloo transport: barrier +
    max-over-ranks timing + the all_gather of images and heat maps run, and the line reports n_gpus = 2."""
    r = _torchrun(2, ["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--ddim-steps", "2", "--batch", "1",
                      "--no-cpu-baseline", "--no-profile"], {"AGD_FORCE_DEVICE": "0", "AGD_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 2 and line["scaling"] == "weak"


def test_generation_driver_two_ranks_ragged_gather(tmp_path):
    """`torchrun -m agenda_amd.generation` with 2 ranks and 5 images at batch 2: rank 0 holds seeds 0,2,4, rank 1 holds 1,3;
    the second round is ragged (1 image vs none).  The final gather delivers every seed to rank 0, which writes the same
    files a single-rank run writes (data_generation.py:66-86 layout)."""
    from PIL import Image
    common = ["--num-images", "5", "--batch-size", "2", "--num-inference-steps", "1", "--synthetic-config", "tiny",
              "--word_token_heatmaps", "cars", "view", "--image-size", "56"]
    r = _torchrun(2, ["-m", "agenda_amd.generation", "--save-dir", str(tmp_path / "two")] + common,
                  {"AGD_FORCE_DEVICE": "0", "AGD_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    from agenda_amd import generation
    generation.main(["--save-dir", str(tmp_path / "one")] + common)
    want = ["0.png", "1.png", "2.png", "3.png", "4.png"]
    for sub in ("images", "daam_cars_heatmaps", "daam_view_heatmaps"):
        assert sorted(os.listdir(tmp_path / "two" / sub)) == want, sub
        for n in want:      # batches are composed differently (tile / split-K choices follow M): equal up to rounding
            a = np.asarray(Image.open(tmp_path / "two" / sub / n)).astype(np.float64)
            b = np.asarray(Image.open(tmp_path / "one" / sub / n)).astype(np.float64)
            assert a.shape == b.shape and float(np.abs(a - b).mean()) < 1.5, (sub, n, float(np.abs(a - b).mean()))
