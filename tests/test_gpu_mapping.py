"""The batch driver on a heterogeneous map: DRTMD.fit_observations (hybdrt/mapping/drtmd.py:245-319) fits any
mix of EIS / chrono / joint observations, each on its own frequency range and its own slice of the tau supergrid.  The fixture
tests/golden/refrun_drtmd_mixed16.npz is the REFERENCE's own DRTMD run (oracle/make_golden.py: run_drtmd_mixed) on the
16-observation map rebuilt below from the same seeds."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, parity_close

pytestmark = pytest.mark.gpu


def mixed_map_observations(n_obs=16):
    """as oracle/make_golden.py: mixed_map_observations (the fixture's inputs; hipdrt.synth is seeded)"""
    from hipdrt import synth
    fa, fb = np.logspace(5, 0, 41), np.logspace(4, -1, 36)
    obs = []
    for k in range(n_obs):
        if k % 3 == 0:
            m = synth.hybrid_measurement(seed=200 + k, jitter=True, n_post=100, nf=31)
            obs.append(((m[0], m[1], m[2]), (m[3], m[4])))
        else:
            f = fa if k % 3 == 1 else fb
            obs.append((None, (f, synth.zarc2_spectrum(f, 300 + k, jitter=True))))
    return obs


def _check_against_reference(g, obs_x, obs_special, res, drt_var):
    peak = np.abs(g["obs_x"]).max(axis=1, keepdims=True)
    parity_close("mixed_map.obs_x", obs_x / peak, g["obs_x"] / peak, 1e-10, scale=1.0)        # measured 2.9e-12
    assert [tuple(t) for t in res["obs_tau_indices"]] == [tuple(t) for t in g["obs_tau_indices"].tolist()]
    assert set(obs_special) == set(str(k) for k in g["special_names"])
    for key in obs_special:
        ref = g["special_" + key].reshape(len(obs_x), -1)[:, 0]
        parity_close("mixed_map.special_" + key, np.asarray(obs_special[key]).reshape(len(obs_x), -1)[:, 0], ref, 1e-9)     # 1.9e-11
    # DRTMD's default metrics: weights='uniform', normalize=True (drtmd.py:121-134)
    parity_close("mixed_map.obs_llh", res["obs_llh"] / g["obs_llh"], np.ones(len(obs_x)), 1e-9, scale=1.0)     # measured 1.9e-11
    parity_close("mixed_map.obs_rss", res["obs_rss"] / g["obs_rss"], np.ones(len(obs_x)), 5e-9, scale=1.0)     # measured 9.8e-11
    if drt_var:
        vmax = g["obs_drt_var"].max(axis=1, keepdims=True)
        parity_close("mixed_map.obs_drt_var", res["obs_drt_var"] / vmax, g["obs_drt_var"] / vmax, 1e-9, scale=1.0)      # measured 5.0e-12


def test_mixed_map_matches_the_reference_drtmd():
    """16 observations, three groups (joint chrono + EIS; EIS on 1e5..1 Hz; EIS on 1e4..0.1 Hz), interleaved: grouped on the
    host, one device plan per group, scattered into each observation's own supergrid slice"""
    from hipdrt.mapping import fit_observations
    from hipdrt.mapping.drtmd import observation_groups
    from hipdrt.models import DRT
    g = np.load(os.path.join(GOLDEN, "refrun_drtmd_mixed16.npz"))
    obs = mixed_map_observations(int(g["n_obs"]))
    groups = observation_groups(obs)
    assert [kind for kind, _ in groups] == ["hybrid", "eis", "eis"] and [len(i) for _, i in groups] == [6, 5, 5]
    drt = DRT(tau_supergrid=g["tau_supergrid"], warn=False)
    obs_x, obs_special, res = fit_observations(drt, observations=obs, tau_supergrid=g["tau_supergrid"], drt_var=True, nonneg=True)
    assert res["obs_fit_status"].all() and len(res["groups"]) == 3
    assert len({tuple(t) for t in res["obs_tau_indices"]}) == 3          # three different slices of the supergrid
    _check_against_reference(g, obs_x, obs_special, res, drt_var=True)


def outlier_map_observations(n_obs=9):
    """as oracle/make_golden.py: outlier_map_observations (gross errors planted in some observations of the mixed map)"""
    obs = mixed_map_observations(n_obs)
    plant = {0: dict(v={100: 5e-5}, z={10: 0.3}), 1: dict(z={10: 0.3}), 4: dict(z={10: 0.3, 25: -0.25j}), 5: dict(z={20: -0.25j}),
             6: dict(v={110: -8e-5})}
    out = []
    for k, (chrono, eis) in enumerate(obs):
        p_ = plant.get(k, {})
        if chrono is not None and "v" in p_:
            v = np.array(chrono[2], dtype=float)
            for i, dv in p_["v"].items():
                v[i] += dv
            chrono = (chrono[0], chrono[1], v)
        if "z" in p_:
            z = np.array(eis[1], dtype=complex)
            for i, dz in p_["z"].items():
                z[i] += dz
            eis = (eis[0], z)
        out.append((chrono, eis))
    return out


def test_remove_outliers_in_a_map_matches_the_reference_drtmd():
    """remove_outliers=True among a map's fit keywords (drt1d.py:214-302 per observation in the reference's DRTMD loop): the
    detection pass runs as one device batch per group of like observations, every observation loses its own flagged points, and
    the refits are batched over observations that lost the same points -- against the reference's own DRTMD run with the same
    keywords (nine observations, gross errors planted in five; it removes between 0 and 5 points per observation)."""
    from hipdrt.mapping import fit_observations
    from hipdrt.mapping.drtmd import observation_groups, prefilter_observations
    from hipdrt.models import DRT
    g = np.load(os.path.join(GOLDEN, "refrun_drtmd_outliers9.npz"))
    obs = outlier_map_observations(int(g["n_obs"]))
    sup = g["tau_supergrid"]
    drt = DRT(tau_supergrid=sup, warn=False)
    kw = dict(nonneg=True, remove_outliers=True, outlier_p=0.05)
    cleaned, kw2, tags, step_times = prefilter_observations(drt, obs, kw)
    lost = [((0 if o[0] is None else len(o[0][0])) - (0 if c[0] is None else len(c[0][0])), len(o[1][0]) - len(c[1][0]))
            for o, c in zip(obs, cleaned)]
    assert lost == [tuple(r) for r in g["removed"].tolist()]                 # the same number of points per observation
    assert "remove_outliers" not in kw2 and kw2["outlier_p"] is None
    groups = observation_groups(cleaned, tags)
    assert 3 < len(groups) < len(obs)                                        # new batches: neither the three old groups nor singles
    obs_x, obs_special, res = fit_observations(drt, observations=obs, tau_supergrid=sup, **kw)
    assert res["obs_fit_status"].all() and len(res["groups"]) == len(groups)
    peak = np.abs(g["obs_x"]).max(axis=1, keepdims=True)
    parity_close("outlier_map.obs_x", obs_x / peak, g["obs_x"] / peak, 1e-9, scale=1.0)
    assert [tuple(t) for t in res["obs_tau_indices"]] == [tuple(t) for t in g["obs_tau_indices"].tolist()]
    for key in obs_special:
        ref = g["special_" + key].reshape(len(obs_x), -1)[:, 0]
        parity_close("outlier_map.special_" + key, np.asarray(obs_special[key]).reshape(len(obs_x), -1)[:, 0], ref, 1e-8)
    parity_close("outlier_map.obs_llh", res["obs_llh"] / g["obs_llh"], np.ones(len(obs_x)), 1e-8, scale=1.0)
    parity_close("outlier_map.obs_rss", res["obs_rss"] / g["obs_rss"], np.ones(len(obs_x)), 1e-8, scale=1.0)
    # remove_extremes: per observation on the host, then batches as usual (same values as the single fit with the keyword)
    ext = [(c, (e[0], np.where(np.arange(len(e[0])) == 7, e[1] + (5.0 if k == 1 else 0.0), e[1]))) for k, (c, e) in enumerate(obs[:3])]
    ox, _, rs = fit_observations(DRT(tau_supergrid=sup, warn=False), observations=ext, tau_supergrid=sup, nonneg=True, remove_extremes=True)
    single = DRT(tau_supergrid=sup, warn=False)
    single.fit_eis(ext[1][1][0], ext[1][1][1], nonneg=True, remove_extremes=True)
    l, r = rs["obs_tau_indices"][1]
    np.testing.assert_allclose(ox[1, l:r], single.fit_parameters["x"], rtol=0, atol=1e-9 * np.abs(ox[1]).max())


def test_sharded_map_downloads_only_what_a_map_records():
    """fit_observations_sharded on a shared grid collects the lean set (distribution, special parameters, counts, status -- the
    solution in scaled units, weights, rho, s vectors and q stay on the device): the same bits as the direct call returns for
    those fields, the DRT object's own setting is restored, and a later plain batch fit on it is complete again"""
    from hipdrt import synth
    from hipdrt.mapping import fit_observations, fit_observations_sharded
    from hipdrt.models import DRT
    c1 = synth.config_c1()
    z = synth.zarc2_batch(c1["freq"], 40, first_seed=900)
    direct = fit_observations(DRT(fixed_basis_tau=c1["tau"]), c1["freq"], z, drt_var=True)
    drt = DRT(fixed_basis_tau=c1["tau"])
    obs_x, obs_special, res = fit_observations_sharded(drt, c1["freq"], z, rank=0, world=1, drt_var=True)
    np.testing.assert_array_equal(obs_x, direct[0])
    for key in ("R_inf", "inductance"):
        np.testing.assert_array_equal(obs_special[key], direct[1][key], err_msg=key)
    for key in ("obs_llh", "obs_rss", "outer_iters", "qp_iters_total", "status", "obs_drt_var"):
        np.testing.assert_array_equal(res[key], direct[2][key], err_msg=key)
    assert getattr(drt, "collect_fields", None) is None
    full = drt.fit_eis_batch(c1["freq"], z[:4])
    assert {"x", "weights", "rho", "s_vectors", "q_vector", "z_sigma_tot"} <= set(full)
    lean = drt._plan.download(lean=True)
    assert "weights" not in lean and "x" not in lean
    np.testing.assert_array_equal(lean["fit_x"], full["fit_x"])


def test_map_cut_into_consecutive_batches_gives_the_same_bits():
    """a map that does not fit the device at once (here: max_batch = 20 of 60 spectra) runs as consecutive batches through the same
    plan -- identical to the one-batch result; and the library's estimate of what a staged spectrum costs (4.9 MB at 256 x 512)
    is what the default limit is made of"""
    from hipdrt import _ffi, synth
    from hipdrt.mapping import fit_observations
    from hipdrt.mapping.drtmd import max_batch_for
    from hipdrt.models import DRT
    c1 = synth.config_c1()
    z = synth.zarc2_batch(c1["freq"], 60, first_seed=1200)
    whole = fit_observations(DRT(fixed_basis_tau=c1["tau"]), c1["freq"], z, drt_var=True)
    drt = DRT(fixed_basis_tau=c1["tau"])
    cut = fit_observations(drt, c1["freq"], z, drt_var=True, max_batch=20)
    assert drt._plan.B == 20                                               # the last of three batches
    np.testing.assert_array_equal(cut[0], whole[0])
    for key in whole[1]:
        np.testing.assert_array_equal(cut[1][key], whole[1][key], err_msg=key)
    for key in ("obs_llh", "obs_rss", "outer_iters", "qp_iters_total", "status", "obs_drt_var", "fit_x", "weights"):
        np.testing.assert_array_equal(cut[2][key], whole[2][key], err_msg=key)
    c2 = synth.config_c2()
    ctx = _ffi.get_context(0)
    per = ctx.plan_bytes_per_spectrum(len(c2["freq"]), len(c2["tau"]), 2)
    assert 4.7e6 < per < 5.1e6                                             # measured 4.88 MB (tools/probe_plan_memory.py)
    limit = max_batch_for(DRT(fixed_basis_tau=c2["tau"]), c2["freq"])
    assert limit == int(0.8 * ctx.device_info()["hbm_bytes"] / per) and 30000 < limit < 80000


def test_mixed_map_through_the_sharded_driver():
    """the same map through fit_observations_sharded (world 1 = what every rank of a node runs on its shard): identical to
    the direct call"""
    from hipdrt.mapping import fit_observations, fit_observations_sharded
    from hipdrt.models import DRT
    g = np.load(os.path.join(GOLDEN, "refrun_drtmd_mixed16.npz"))
    obs = mixed_map_observations(int(g["n_obs"]))
    drt = DRT(tau_supergrid=g["tau_supergrid"], warn=False)
    direct = fit_observations(drt, observations=obs, tau_supergrid=g["tau_supergrid"], nonneg=True)
    obs_x, obs_special, res = fit_observations_sharded(DRT(tau_supergrid=g["tau_supergrid"], warn=False), observations=obs,
                                                       rank=0, world=1, tau_supergrid=g["tau_supergrid"], nonneg=True)
    np.testing.assert_array_equal(obs_x, direct[0])
    assert set(obs_special) == set(direct[1])
    for key in direct[1]:           # same shapes too: special parameters travel at their real widths
        np.testing.assert_array_equal(obs_special[key], np.asarray(direct[1][key]), err_msg=key)
    for key in ("obs_llh", "obs_rss", "outer_iters", "status"):
        np.testing.assert_array_equal(res[key], direct[2][key], err_msg=key)
    assert res["obs_tau_indices"] == direct[2]["obs_tau_indices"]
    _check_against_reference(g, obs_x, obs_special, res, drt_var=False)


def test_lookup_tables_travel_between_instances():
    """share_lookup_tables' two halves on one rank: tables taken from one DRT (lookup_tables) and installed into another
    (install_lookup_tables) give bit-identical fits, for an EIS plan and for a prepared-matrix plan"""
    from hipdrt import synth
    from hipdrt.models import DRT
    freq = np.logspace(5, 0, 41)
    z = synth.zarc2_batch(freq, 3, first_seed=40)
    a = DRT(warn=False)
    ra = a.fit_eis_batch(freq, z)
    b = DRT(warn=False)
    b.install_lookup_tables(*a.lookup_tables())
    rb = b.fit_eis_batch(freq, z)
    np.testing.assert_array_equal(ra["x"], rb["x"])
    m = synth.hybrid_measurement(seed=3, n_post=100, nf=31)
    fa = DRT(warn=False).fit_hybrid(*m)
    c = DRT(warn=False)
    c.install_lookup_tables(*a.lookup_tables())
    fc = c.fit_hybrid(*m)
    np.testing.assert_array_equal(fa["x"], fc["x"])


_WORLD_ONE_NCCL = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["HIPDRT_ROOT"])
import torch
import torch.distributed as tdist
from hipdrt.mapping import dist as hd
torch.cuda.set_device(0)
rank, world, local = hd.init_from_env(backend="nccl", device=0, force=True)      # a ONE-rank RCCL process group
assert (rank, world) == (0, 1) and tdist.is_initialized() and tdist.get_backend() == "nccl" and hd.active(world)
calls = {"broadcast": 0, "gather": 0, "all_reduce": 0}
for name in calls:
    def counted(*a, _f=getattr(tdist, name), _n=name, **k):
        calls[_n] += 1
        return _f(*a, **k)
    setattr(tdist, name, counted)
a, b = np.arange(7.0), np.linspace(0, 1, 12).reshape(3, 4)
ra, rb = hd.broadcast_arrays([a, b], src=0)
assert np.array_equal(ra, a) and np.array_equal(rb, b)
rows = np.arange(15.0).reshape(5, 3)
assert np.array_equal(hd.gather_rows(rows, [5], dst=0), rows)
assert hd.max_over_ranks(2.5) == 2.5
hd.barrier()
assert calls == {"broadcast": 1, "gather": 1, "all_reduce": 1}, calls
# the sharded driver itself on the one-rank group: lookup tables broadcast once, the map gathered with one collective
from hipdrt import synth
from hipdrt.mapping import fit_observations, fit_observations_sharded
from hipdrt.models import DRT
freq = np.logspace(5, 0, 41)
z = synth.zarc2_batch(freq, 6, first_seed=60)
drt = DRT()
obs_x, obs_special, res = fit_observations_sharded(drt, freq, z)                  # rank / world from the process group
assert calls["broadcast"] == 2 and calls["gather"] == 2, calls
ref_x, ref_special, ref = fit_observations(DRT(), freq, z)
assert np.array_equal(obs_x, ref_x) and np.array_equal(obs_special["R_inf"], ref_special["R_inf"])
assert np.array_equal(res["outer_iters"], ref["outer_iters"])
fit_observations_sharded(drt, freq, z)                                            # second map: no broadcast, one gather
assert calls["broadcast"] == 2 and calls["gather"] == 3, calls
hd.barrier()
tdist.destroy_process_group()
print("WORLD1_NCCL_OK " + json.dumps(calls))
'''


_WORLD_ONE_RCCL = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["HIPDRT_ROOT"])
from hipdrt import _ffi
from hipdrt.mapping import dist as hd
rank, world, local = hd.init_from_env(device=0, force=True)          # default backend on a GPU box: RCCL behind the C ABI
assert (rank, world) == (0, 1) and hd.backend() == "rccl" and hd.is_initialized() and hd.active(world)
assert "torch" not in sys.modules, "the native backend must not import torch"
calls = {"broadcast": 0, "gather": 0, "allreduce_max": 0}
for name in calls:
    def counted(self, *a, _f=getattr(_ffi.Comm, name), _n=name, **k):
        calls[_n] += 1
        return _f(self, *a, **k)
    setattr(_ffi.Comm, name, counted)
a, b = np.arange(7.0), np.linspace(0, 1, 12).reshape(3, 4)
ra, rb = hd.broadcast_arrays([a, b], src=0)
assert np.array_equal(ra, a) and np.array_equal(rb, b)
rows = np.arange(15.0).reshape(5, 3)
assert np.array_equal(hd.gather_rows(rows, [5], dst=0), rows)
assert hd.max_over_ranks(2.5) == 2.5
assert calls == {"broadcast": 1, "gather": 1, "allreduce_max": 1}, calls
hd.barrier()
# device buffers in, device buffers out
ctx = _ffi.Context(0)
lib, comm = _ffi.load_library(), hd._STATE["comm"]
src = np.arange(1000.0)
dptr = ctx.device_alloc(2 * src.nbytes)
import ctypes as C
from hipdrt.models import DRT
assert lib.hipdrt_comm_broadcast_dev(comm._h, C.c_void_p(dptr), src.size, 0) == 0
assert lib.hipdrt_comm_gather_dev(comm._h, C.c_void_p(dptr), src.size, C.c_void_p(dptr + src.nbytes), 0) == 0
ctx.device_free(dptr)
# the sharded driver itself on the one-rank communicator: lookup tables broadcast once, the map gathered with one collective
from hipdrt import synth
from hipdrt.mapping import fit_observations, fit_observations_sharded
freq = np.logspace(5, 0, 41)
z = synth.zarc2_batch(freq, 6, first_seed=60)
drt = DRT()
obs_x, obs_special, res = fit_observations_sharded(drt, freq, z)                  # rank / world from the communicator
assert calls["broadcast"] == 2 and calls["gather"] == 2, calls
ref_x, ref_special, ref = fit_observations(DRT(), freq, z)
assert np.array_equal(obs_x, ref_x) and np.array_equal(obs_special["R_inf"], ref_special["R_inf"])
assert np.array_equal(res["outer_iters"], ref["outer_iters"])
fit_observations_sharded(drt, freq, z)                                            # second map: no broadcast, one gather
assert calls["broadcast"] == 2 and calls["gather"] == 3, calls
hd.barrier()
hd.destroy()
assert "torch" not in sys.modules
print("WORLD1_RCCL_OK " + json.dumps(calls))
'''


def test_world_one_rccl_communicator_behind_the_c_abi():
    """RCCL behind the C ABI (hipdrt_comm_*, csrc/comm.hip; librccl loaded on first use), the default backend
    of mapping.dist on a GPU box.  A fresh child process creates a world-1 communicator on device 0 WITHOUT importing torch and
    runs broadcast / gather / all-reduce-max / barrier, the device-buffer forms of broadcast and gather, and two maps of
    fit_observations_sharded through it (collectives counted: lookup tables once per DRT instance, ONE gather per map), the
    gathered map bit-equal to the un-sharded driver's."""
    import subprocess
    import sys
    from conftest import ROOT as root
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29518",
               HIPDRT_ROOT=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("HIPDRT_DIST_BACKEND", None)
    pr = subprocess.run([sys.executable, "-c", _WORLD_ONE_RCCL], env=env, capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0 and "WORLD1_RCCL_OK" in pr.stdout, (pr.returncode, pr.stdout[-2000:], pr.stderr[-4000:])


def test_world_one_nccl_group_runs_every_collective():
    """The multi-GPU path's collectives through the REAL backend on the one GPU of the box: a fresh child process creates a
    world-1 `nccl` (= RCCL) process group bound to cuda:0 and runs broadcast_arrays / gather_rows / max_over_ranks / barrier
    and two maps of fit_observations_sharded through it (`force`: no world == 1 early-out), with the collectives counted
    and the gathered map bit-equal to the un-sharded driver's.  What this pins before an 8-GPU node exists: RCCL loads and
    initialises, `device_id` binding, the numpy -> device tensor -> collective -> host staging of mapping/dist.py."""
    import subprocess
    import sys
    from conftest import ROOT as root
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517",
               HIPDRT_ROOT=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    pr = subprocess.run([sys.executable, "-c", _WORLD_ONE_NCCL], env=env, capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0 and "WORLD1_NCCL_OK" in pr.stdout, (pr.returncode, pr.stdout[-2000:], pr.stderr[-4000:])


def test_drtmd_fit_type_pfrt_matches_the_reference():
    """mapping.fit_observations(fit_type='pfrt') against the reference's own DRTMD(fit_type='pfrt').fit_all (drtmd.py:98-100,
    1136-1158, 1338-1342) on six impedance observations on two frequency grids (refrun_drtmd_pfrt6.npz): one solution per
    factor and observation in the supergrid slots, specials per factor, obs_llh / obs_rss / obs_drt_var of the first step."""
    from hipdrt.mapping import fit_observations
    from hipdrt.models import DRT
    g = np.load(os.path.join(GOLDEN, "refrun_drtmd_pfrt6.npz"), allow_pickle=False)
    obs = [o for o in mixed_map_observations(18) if o[0] is None][:6]
    sup = g["tau_supergrid"]
    drt = DRT(tau_supergrid=sup, warn=False)
    obs_x, obs_special, res = fit_observations(drt, observations=obs, tau_supergrid=sup, fit_type='pfrt', drt_var=True)
    assert obs_x.shape == g["obs_x"].shape == (6, 11, len(sup)) and res["obs_fit_status"].all()
    # (the reference's DRTMD fits with _pfrt_fit_core's default factors, not with its own pfrt_factors attribute: see the driver)
    np.testing.assert_array_equal(res["pfrt_factors"], np.logspace(-1, 1, 11))
    assert not np.array_equal(g["pfrt_factors"], res["pfrt_factors"])
    assert [tuple(t) for t in res["obs_tau_indices"]] == [tuple(t) for t in g["obs_tau_indices"].tolist()]
    peak = np.abs(g["obs_x"]).max(axis=(1, 2), keepdims=True)
    parity_close("pfrt_map.obs_x", obs_x / peak, g["obs_x"] / peak, 5e-10, scale=1.0)            # measured 1.3e-11
    for key in ("R_inf", "inductance"):
        parity_close("pfrt_map.special_" + key, obs_special[key], g["special_" + key], 2e-10)             # 7.9e-12
    parity_close("pfrt_map.obs_llh", res["obs_llh"] / g["obs_llh"], np.ones(6), 5e-11, scale=1.0)        # 1.9e-12
    parity_close("pfrt_map.obs_rss", res["obs_rss"] / g["obs_rss"], np.ones(6), 2e-10, scale=1.0)        # 7.2e-12
    vpeak = g["obs_drt_var"].max(axis=(1, 2), keepdims=True)
    parity_close("pfrt_map.obs_drt_var", res["obs_drt_var"] / vpeak, g["obs_drt_var"] / vpeak, 5e-11, scale=1.0)   # 1.7e-12
    with pytest.raises(ValueError):
        fit_observations(drt, observations=obs, tau_supergrid=sup, fit_type='nope')


def test_reproducible_map_is_bit_identical_however_it_is_sharded():
    """fit_observations_sharded(reproducible=True): 24 spectra at the configs[2] grids as one device batch (one workgroup per
    problem) and as three shares of eight (which would each take the several-workgroups kernel: 1e-14 apart) give the same bits;
    without the flag the two maps agree to rounding only."""
    from hipdrt import synth
    from hipdrt.mapping import fit_observations_sharded
    from hipdrt.models import DRT
    c2 = synth.config_c2()
    z = synth.zarc2_batch(c2["freq"], 24, first_seed=700)
    drt = DRT(fixed_basis_tau=c2["tau"])
    whole = fit_observations_sharded(drt, c2["freq"], z, rank=0, world=1, reproducible=True)
    parts = [fit_observations_sharded(drt, c2["freq"], z[8 * r:8 * r + 8], rank=0, world=1, reproducible=True) for r in range(3)]
    np.testing.assert_array_equal(np.concatenate([p_[0] for p_ in parts]), whole[0])
    np.testing.assert_array_equal(np.concatenate([p_[1]["R_inf"] for p_ in parts]), whole[1]["R_inf"])
    np.testing.assert_array_equal(np.concatenate([p_[2]["outer_iters"] for p_ in parts]), whole[2]["outer_iters"])
    loose = [fit_observations_sharded(drt, c2["freq"], z[8 * r:8 * r + 8], rank=0, world=1) for r in range(3)]
    lx = np.concatenate([p_[0] for p_ in loose])
    assert not np.array_equal(lx, whole[0])                     # (the small shares ran on the group kernel)
    parity_close("reproducible_map.group_vs_batch_kernel", lx, whole[0], 2e-9)      # measured 1.1e-10 (a fit of 27 outer iterations, two kernels)


def test_store_fit_all_resumes_where_it_stopped_and_matches_the_reference_map():
    """mapping.DRTMD (observation store, drtmd.py:186-329): ten observations of the reference's 16-observation map are added and
    fitted, six more are added, fit_all(refit=False) sends exactly those six to the device (drtmd.py:321-329) -- and the store
    then holds the reference's own DRTMD result for all sixteen (a fit does not depend on what shares its device batch)."""
    from hipdrt.mapping import DRTMD
    g = np.load(os.path.join(GOLDEN, "refrun_drtmd_mixed16.npz"))
    obs = mixed_map_observations(int(g["n_obs"]))
    md = DRTMD(g["tau_supergrid"], warn=False)
    for k in range(10):
        md.add_observation([k], *obs[k])
    assert md.fit_all().tolist() == list(range(10)) and md.obs_fit_status.all()
    first_x = md.obs_x.copy()
    for k in range(10, 16):
        md.add_observation([k], *obs[k])
    assert md.fit_all(refit=False).tolist() == list(range(10, 16))
    assert md.last_fit_index.tolist() == list(range(10, 16)) and md.obs_fit_status.all() and not md.obs_ignore_flag.any()
    np.testing.assert_array_equal(md.obs_x[:10], first_x)                     # untouched by the second call
    assert md.fit_all(refit=False).tolist() == []
    res = dict(obs_tau_indices=md.obs_tau_indices, obs_llh=md.obs_llh, obs_rss=md.obs_rss, obs_drt_var=md.obs_drt_var)
    _check_against_reference(g, md.obs_x, md.obs_special, res, drt_var=True)
