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


def _fake_fit(drt, frequencies, z_obs, tau_supergrid=None, drt_var=False, **kw):
    """stand-in for mapping.fit_observations with the same return structure: every output is a deterministic function of
    the observation's own data, so the gathered result can be checked against a single-process evaluation"""
    num, nsup = z_obs.shape[0], 12
    obs_x = np.outer(z_obs.real.sum(1), np.arange(1.0, nsup + 1))
    special = {"R_inf": z_obs.real[:, 0].copy(), "inductance": z_obs.imag[:, -1].copy()}
    res = {"obs_llh": -np.abs(z_obs).sum(1), "obs_rss": (np.abs(z_obs) ** 2).sum(1),
           "outer_iters": (np.abs(z_obs[:, 0]) * 10).astype(np.int64) % 50, "qp_iters_total": np.arange(num) * 0 + 7,
           "status": np.where(z_obs.real[:, 1] > 2.4, 0, 1)}
    if drt_var:
        res["obs_drt_var"], res["obs_drt_var_ok"] = obs_x ** 2, np.ones(num, dtype=bool)
    return obs_x, special, res


def _sharded_worker(rank, world, port, total, scheme, drt_var):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from hipdrt import synth
    from hipdrt.mapping import dist as hd
    from hipdrt.mapping.drtmd import fit_observations_sharded
    hd.init_from_env(backend="gloo")
    freq = np.logspace(3, 0, 16)
    z = synth.zarc2_batch(freq, total) if total else np.zeros((0, 16), dtype=complex)
    out = fit_observations_sharded(None, freq, z, scheme=scheme, drt_var=drt_var, fit=_fake_fit)
    if rank == 0:
        obs_x, special, res = out
        ex, es, er = _fake_fit(None, freq, z, drt_var=drt_var)
        np.testing.assert_array_equal(obs_x, ex)
        for k in es:
            np.testing.assert_array_equal(special[k], es[k])
        for k in er:
            np.testing.assert_array_equal(res[k], er[k], err_msg=k)
        np.testing.assert_array_equal(res["obs_fit_status"], er["status"] >= 0)
    else:
        assert out is None
    hd.barrier()
    dist.destroy_process_group()


def test_fit_observations_sharded_world2_every_scheme():
    """the configs[3] path itself (shard -> per-rank fit -> one gather -> original order) on two gloo ranks, with a stand-in
    for the device fit: block / interleaved / cost-balanced shards, an odd observation count, with and without the
    variance rows"""
    for scheme, drt_var, total in (("block", False, 7), ("interleave", True, 9), ("lpt", False, 10)):
        mp.spawn(_sharded_worker, args=(2, _free_port(), total, scheme, drt_var), nprocs=2, join=True)


def test_shard_indices_partition_and_balance():
    from hipdrt.mapping.drtmd import shard_indices
    rng = np.random.default_rng(3)
    for num, world in ((10000, 8), (17, 4), (5, 8), (0, 2)):
        cost = rng.lognormal(size=num)
        for scheme in ("block", "interleave", "lpt"):
            parts = [shard_indices(num, world, r, scheme, cost) for r in range(world)]
            assert sorted(np.concatenate(parts).tolist()) == list(range(num)), (num, world, scheme)
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1
        if num >= world * 8:
            lpt = [cost[shard_indices(num, world, r, "lpt", cost)].sum() for r in range(world)]
            blk = [cost[shard_indices(num, world, r, "block", cost)].sum() for r in range(world)]
            assert max(lpt) / min(lpt) < 1.02 <= max(1.02, max(blk) / min(blk))
