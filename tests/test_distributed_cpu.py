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
