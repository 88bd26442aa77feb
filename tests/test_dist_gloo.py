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


def _fake_fit(drt, frequencies=None, z_obs=None, tau_supergrid=None, drt_var=False, observations=None, **kw):
    """stand-in for mapping.fit_observations with the same return structure: every output is a deterministic function of
    the observation's own data, so the gathered result can be checked against a single-process evaluation"""
    if observations is not None:
        # heterogeneous form: (chrono_data | None, (frequencies, z)); spectra of different lengths -> different tau slices
        num, nsup = len(observations), 12
        obs_x = np.zeros((num, nsup))
        ti, rinf, vb = [], np.zeros(num), np.zeros(num)
        for k, (chrono, (f, z)) in enumerate(observations):
            left = len(f) % 3
            obs_x[k, left:left + 8] = z.real.sum() * np.arange(1.0, 9.0)
            ti.append((left, left + 8))
            rinf[k] = z.real[0]
            vb[k] = 0.0 if chrono is None else float(np.sum(chrono[2]))
        zs = np.array([z.sum() for _, (f, z) in observations])
        res = {"obs_llh": -np.abs(zs), "obs_rss": np.abs(zs) ** 2,
               "outer_iters": np.array([len(f) + int(chrono is not None) for chrono, (f, z) in observations]) % 5 + 2,
               "qp_iters_total": np.arange(num) * 0 + 7, "status": np.zeros(num, dtype=np.int64), "obs_tau_indices": ti}
        # x_dop is vector valued (one column per basis_nu point), v_baseline here a two-coefficient polynomial
        xdop = np.outer(rinf, np.arange(1.0, 6.0))
        return obs_x, {"v_baseline": np.stack([vb, -vb], axis=1), "vz_offset": vb * 0.5, "R_inf": rinf, "inductance": rinf * 2,
                       "x_dop": xdop}, res
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


def _sharded_general_worker(rank, world, port, total):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from hipdrt import synth
    from hipdrt.mapping import dist as hd
    from hipdrt.mapping.drtmd import fit_observations_sharded
    hd.init_from_env(backend="gloo")
    obs = []
    for k in range(total):
        f = np.logspace(3, 0, 14 + k % 3)
        z = synth.zarc2_spectrum(f, k)
        obs.append(((np.arange(5.0), np.ones(5), np.arange(5.0) * (k + 1)) if k % 4 == 0 else None, (f, z)))
    out = fit_observations_sharded(None, observations=obs, tau_supergrid=np.logspace(-3, 1, 12), fit=_fake_fit)
    if rank == 0:
        obs_x, special, res = out
        ex, es, er = _fake_fit(None, observations=obs)
        np.testing.assert_array_equal(obs_x, ex)
        assert set(special) == set(es)
        for k in es:
            assert special[k].shape == es[k].shape, (k, special[k].shape, es[k].shape)      # x_dop (num, 5), v_baseline (num, 2)
            np.testing.assert_array_equal(special[k], es[k])
        for k in ("obs_llh", "obs_rss", "outer_iters", "qp_iters_total", "status"):
            np.testing.assert_array_equal(res[k], er[k], err_msg=k)
        assert res["obs_tau_indices"] == er["obs_tau_indices"]
    else:
        assert out is None
    hd.barrier()
    dist.destroy_process_group()


def test_fit_observations_sharded_world2_heterogeneous_observations():
    """the observation-list form (any mix of data types and grids) through the sharded driver: per-observation tau slices
    and the union of the special parameters travel through the one gather"""
    mp.spawn(_sharded_general_worker, args=(2, _free_port(), 11), nprocs=2, join=True)


def test_fit_observations_sharded_world2_every_scheme():
    """the configs[3] path itself (shard -> per-rank fit -> one gather -> original order) on two gloo ranks, with a stand-in
    for the device fit: block / interleaved / cost-balanced shards, an odd observation count, with and without the
    variance rows"""
    for scheme, drt_var, total in (("block", False, 7), ("interleave", True, 9), ("lpt", False, 10)):
        mp.spawn(_sharded_worker, args=(2, _free_port(), total, scheme, drt_var), nprocs=2, join=True)


class _TableDRT:
    """just enough of a DRT for share_lookup_tables: an epsilon, table abscissae, build / install hooks that count"""
    def __init__(self, rank):
        self.tau_epsilon, self.integrate_method = 22.19, 'interp'
        self._wt_re, self._wt_im = np.ones(2000), np.ones(2000)
        self.rank, self.built, self.installed = rank, 0, 0
        self.tables = None

    def lookup_tables(self):
        self.built += 1
        return np.arange(2000.0), np.arange(2000.0) * 2, np.arange(2000.0) * 3

    def install_lookup_tables(self, z_re, z_im, resp):
        self.installed += 1
        self.tables = (z_re, z_im, resp)


def _share_once_worker(rank, world, port):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from hipdrt.mapping import dist as hd
    from hipdrt.mapping import drtmd
    hd.init_from_env(backend="gloo")
    calls = {"broadcast": 0, "all_reduce": 0, "gather": 0}
    for name in calls:
        orig = getattr(dist, name)
        def counted(*a, _o=orig, _n=name, **k):
            calls[_n] += 1
            return _o(*a, **k)
        setattr(dist, name, counted)
    drt = _TableDRT(rank)
    freq = np.logspace(3, 0, 16)
    from hipdrt import synth
    z = synth.zarc2_batch(freq, 9)
    # `fit` is the real entry point's identity as far as the driver is concerned: wrap the stand-in so that the table
    # broadcast is not skipped
    orig_fit = drtmd.fit_observations
    drtmd.fit_observations = _fake_fit
    try:
        for call in range(3):
            out = drtmd.fit_observations_sharded(drt, freq, z, rank=rank, world=world, fit=drtmd.fit_observations)
            assert (out is None) == (rank != 0)
            # first map: one broadcast + one gather; every later map: the gather only (no all_reduce: every rank owns rows)
            assert calls == {"broadcast": 1, "all_reduce": 0, "gather": call + 1}, (call, calls)
    finally:
        drtmd.fit_observations = orig_fit
    assert drt.built == (1 if rank == 0 else 0) and drt.installed == (0 if rank == 0 else 1)
    if rank != 0:
        np.testing.assert_array_equal(drt.tables[2], np.arange(2000.0) * 3)
    hd.barrier()
    dist.destroy_process_group()


def test_second_sharded_map_issues_no_broadcast():
    """the lookup tables travel once per DRT instance, not once per map; a map whose ranks all own observations costs exactly
    one collective (the gather)"""
    mp.spawn(_share_once_worker, args=(2, _free_port()), nprocs=2, join=True)


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


def _forced_world1_worker(rank, world, port):
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from hipdrt.mapping import dist as hd
    assert hd.init_from_env(backend="gloo")[:2] == (0, 1) and not dist.is_initialized() and not hd.active(1)   # default: no group
    assert hd.init_from_env(backend="gloo", force=True)[:2] == (0, 1) and dist.is_initialized() and hd.active(1)
    calls = {"broadcast": 0, "gather": 0, "all_reduce": 0}
    for name in calls:
        def counted(*a, _f=getattr(dist, name), _n=name, **k):
            calls[_n] += 1
            return _f(*a, **k)
        setattr(dist, name, counted)
    a = np.arange(5.0)
    np.testing.assert_array_equal(hd.broadcast_arrays([a])[0], a)
    rows = np.arange(8.0).reshape(4, 2)
    np.testing.assert_array_equal(hd.gather_rows(rows, [4]), rows)
    assert hd.max_over_ranks(3.0) == 3.0
    assert calls == {"broadcast": 1, "gather": 1, "all_reduce": 1}
    hd.barrier()
    dist.destroy_process_group()


def test_forced_one_rank_group_runs_the_collectives():
    """`force` (bench.py --force-dist, HIPDRT_FORCE_DIST=1): a world of one rank creates its process group and sends
    broadcast / gather / all-reduce through the backend instead of short-cutting them -- the form the GPU suite uses to
    exercise RCCL on a one-GPU box."""
    mp.spawn(_forced_world1_worker, args=(1, _free_port()), nprocs=1, join=True)
