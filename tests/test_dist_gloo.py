"""CPU, world_size 2, gloo: the sharded-batch plumbing (shard bounds, table broadcast, result gather,
max-over-ranks timing) used by bench.py and fit_observations on multi-GPU nodes."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from hipdrt import synth
    from hipdrt.mapping import dist as hd
    from hipdrt.mapping.drtmd import shard_bounds
    r, w, _ = hd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # rank 0 "builds" the tables, everyone must end up with identical copies
    tables = [np.arange(2000.0) * (1 if rank == 0 else -1), np.linspace(0, 1, 2000) * (rank == 0)]
    got = hd.broadcast_arrays(tables, src=0)
    np.testing.assert_array_equal(got[0], np.arange(2000.0))
    np.testing.assert_array_equal(got[1], np.linspace(0, 1, 2000))
    # shard a seeded batch; every rank "fits" its block (here: a deterministic function of the data)
    freq = np.logspace(3, 0, 16)
    a, b = shard_bounds(total, world, rank)
    z = synth.zarc2_batch(freq, b - a, first_seed=a) if b > a else np.zeros((0, 16), dtype=complex)
    local = np.stack([z.real.sum(1), z.imag.sum(1)], axis=1) if b > a else np.zeros((0, 2))
    counts = [shard_bounds(total, world, q)[1] - shard_bounds(total, world, q)[0] for q in range(world)]
    full = hd.gather_rows(local, counts, dst=0)
    if rank == 0:
        zall = synth.zarc2_batch(freq, total)
        np.testing.assert_array_equal(full, np.stack([zall.real.sum(1), zall.imag.sum(1)], axis=1))
    else:
        assert full is None
    assert hd.max_over_ranks(1.0 + rank) == float(world)
    hd.barrier()
    dist.destroy_process_group()


def test_sharded_batch_plumbing_world2():
    port = _free_port()
    mp.spawn(_worker, args=(2, port, 7), nprocs=2, join=True)
