"""world_size-2 gloo test of the only exchange step on the path: the final all_gather of images +
heat maps after seed-sharded generation.  Runs on CPU tensors."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from agenda_amd.generation import shard_seeds, gather_outputs
    seeds = shard_seeds(6, rank, world)                       # rank 0: 0,2,4 ; rank 1: 1,3,5
    imgs = torch.stack([torch.full((4, 4, 3), s, dtype=torch.uint8) for s in seeds])
    hms = torch.stack([torch.full((2, 8, 8), float(s)) for s in seeds])
    gi, gh = gather_outputs(imgs, hms)
    np.save(os.path.join(out_dir, f"i{rank}.npy"), gi.numpy())
    np.save(os.path.join(out_dir, f"h{rank}.npy"), gh.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gather_restores_global_seed_order(tmp_path):
    world, port = 2, _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    for r in range(world):
        gi = np.load(tmp_path / f"i{r}.npy"); gh = np.load(tmp_path / f"h{r}.npy")
        assert gi.shape == (6, 4, 4, 3) and gh.shape == (6, 2, 8, 8)
        assert [int(x[0, 0, 0]) for x in gi] == [0, 1, 2, 3, 4, 5]       # interleaved back to seed order
        assert [float(x[0, 0, 0]) for x in gh] == [0, 1, 2, 3, 4, 5]


def _ragged_worker(rank, world, port, out_dir, num_images, batch):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from agenda_amd.generation import shard_seeds, gather_outputs
    os.environ["AGD_GATHER_CHECK"] = "1"                      # the sync-free path verifies the caller's global_seeds against the gathered ids
    seeds = shard_seeds(num_images, rank, world)
    per_rank = (num_images + world - 1) // world
    rounds = (per_rank + batch - 1) // batch                  # every rank joins every round (generation.main)
    got_s, got_i, got_h = [], [], []
    calls = {"n": 0}
    real_all_gather = dist.all_gather

    def counted(*a, **k):
        calls["n"] += 1
        return real_all_gather(*a, **k)
    dist.all_gather = counted                                 # gloo takes the list form; RCCL all_gather_into_tensor
    for r in range(rounds):
        chunk = seeds[r * batch:(r + 1) * batch]              # may be shorter than `batch`, or empty
        imgs = torch.stack([torch.full((4, 4, 3), s, dtype=torch.uint8) for s in chunk]) if chunk else torch.zeros(0, 4, 4, 3, dtype=torch.uint8)
        hms = torch.stack([torch.full((2, 8, 8), float(s)) for s in chunk]) if chunk else torch.zeros(0, 2, 8, 8)
        if r % 2:                                             # both forms: blocking, and posted-then-waited (bench.py's overlap)
            s_, i_, h_ = gather_outputs(imgs, hms, seeds=chunk, max_batch=batch, async_op=True).wait()
        else:                                                 # generation.main's form: the round's seeds are known on the host -> no sync
            gseeds = sorted(x for rk in range(world) for x in shard_seeds(num_images, rk, world)[r * batch:(r + 1) * batch])
            s_, i_, h_ = gather_outputs(imgs, hms, seeds=chunk, max_batch=batch, global_seeds=gseeds)
        got_s += s_; got_i.append(i_); got_h.append(h_)
    dist.all_gather = real_all_gather
    assert calls["n"] == rounds, (calls, rounds)              # literally ONE collective per batch (ids + images + heat maps in one buffer)
    np.save(os.path.join(out_dir, f"s{rank}.npy"), np.array(got_s))
    np.save(os.path.join(out_dir, f"i{rank}.npy"), torch.cat(got_i).numpy())
    np.save(os.path.join(out_dir, f"h{rank}.npy"), torch.cat(got_h).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,num_images,batch", [(2, 7, 2), (3, 4, 2), (2, 1, 4), (8, 19, 2)])      # last: a full node, ragged + empty tail
def test_gather_handles_ragged_and_empty_last_batches(tmp_path, world, num_images, batch):
    """The last round of `shard_seeds` leaves ranks with fewer (or zero) images: the collective count and the tensor
    shapes must still match on every rank, and every seed must come back exactly once with its own payload."""
    port = _free_port()
    mp.start_processes(_ragged_worker, args=(world, port, str(tmp_path), num_images, batch), nprocs=world, join=True, start_method="spawn")
    for r in range(world):
        s = np.load(tmp_path / f"s{r}.npy"); gi = np.load(tmp_path / f"i{r}.npy"); gh = np.load(tmp_path / f"h{r}.npy")
        assert sorted(s.tolist()) == list(range(num_images))
        assert gi.shape == (num_images, 4, 4, 3) and gh.shape == (num_images, 2, 8, 8)
        assert [int(x[0, 0, 0]) for x in gi] == s.tolist() and [float(x[0, 0, 0]) for x in gh] == [float(v) for v in s]


def _mismatch_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), AGD_GATHER_CHECK="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from agenda_amd.generation import gather_outputs
    chunk = [rank]                                            # the ranks packed seeds 0 and 1 ...
    imgs, hms = torch.full((1, 4, 4, 3), rank, dtype=torch.uint8), torch.full((1, 2, 8, 8), float(rank))
    msg = ""
    try:
        gather_outputs(imgs, hms, seeds=chunk, max_batch=1, global_seeds=[0, 2])      # ... the caller claims 0 and 2
    except RuntimeError as e:
        msg = str(e)
    open(os.path.join(out_dir, f"m{rank}.txt"), "w").write(msg)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_debug_check_catches_a_global_seed_list_that_disagrees_with_the_ranks(tmp_path):
    """with `global_seeds` nothing synchronises with the host, so a caller's list that disagrees with what the ranks packed would
    mis-pair seeds and rows silently; AGD_GATHER_CHECK=1 turns that into an error on every rank."""
    world, port = 2, _free_port()
    mp.start_processes(_mismatch_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    for r in range(world):
        assert "disagree with the gathered ids" in open(tmp_path / f"m{r}.txt").read()


def test_bench_launcher_plans_n_ranks_before_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus N` without a launcher environment must start N ranks itself (torch.distributed.run as a
    CHILD process, before any GPU call) and must refuse when fewer than N devices are visible."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    cmd = bench.self_launch_command(4, ["--gpus", "4", "--steps", "2"], n_devices=8)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    with pytest.raises(SystemExit, match="only 1 visible"):
        bench.self_launch_command(4, ["--gpus", "4"], n_devices=1)
    assert bench.self_launch_command(1, ["--gpus", "1"], n_devices=1) is None       # single rank: run in-process
    monkeypatch.setenv("WORLD_SIZE", "4"); monkeypatch.setenv("RANK", "0")
    assert bench.self_launch_command(4, ["--gpus", "4"], n_devices=8) is None       # already under a launcher
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit, match="WORLD_SIZE 2 != --gpus 4"):
        bench.self_launch_command(4, ["--gpus", "4"], n_devices=8)
