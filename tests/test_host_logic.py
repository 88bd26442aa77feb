"""CPU: host-side grid / option logic of the product against the oracle and reference-run fixtures."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def test_basis_tau_and_epsilon_rules():
    from hipdrt import preprocessing as pp
    g = load("refrun_golden71x91.npz")
    np.testing.assert_array_equal(pp.get_basis_tau(g["freq"]), g["basis_tau"])
    assert pp.get_epsilon_from_ppd(10) == float(g["tau_epsilon"])
    sg = np.logspace(-9, 3, 121)
    np.testing.assert_array_equal(pp.get_basis_tau(g["freq"], tau_grid=sg), sg[12:104])
    assert pp.estimate_rp(None, None, None, None, None, g["z"]) == g["z"].real.max() - g["z"].real.min()
    # chrono data widens the limits to the time since the step (floored at the sample period)
    lo, hi = pp.get_tau_lim(g["freq"], times=np.array([0.0, 1.0, 2.0, 1000.0]), step_times=np.array([0.5]))
    assert lo == 1 / (2 * np.pi * g["freq"].max()) and hi == 999.5


def test_toeplitz_decisions_match_oracle():
    from hipdrt import synth
    from hipdrt.matrices import mat1d
    from oracle import drt_oracle as orc
    g = load("refrun_golden71x91.npz")
    c2 = synth.config_c2()
    cases = [(g["freq"], g["basis_tau"]), (c2["freq"], c2["tau"]), (np.logspace(4, 0, 32), np.logspace(-6, 1, 64)),
             (np.logspace(3, 0, 31), 1 / (2 * np.pi * np.logspace(3, 0, 31)))]
    for f, t in cases:
        assert mat1d.impedance_matrix_is_toeplitz(f, t) == orc.impedance_matrix_is_toeplitz(f, t)
    assert mat1d.impedance_matrix_is_toeplitz(*cases[0]) and not mat1d.impedance_matrix_is_toeplitz(*cases[1])


def test_h_constraint_and_hypers():
    from hipdrt.models import qphb
    from oracle import drt_oracle as orc
    sp = {'R_inf': {'index': 0, 'nonneg': True, 'size': 1}, 'v_baseline': {'index': 1, 'nonneg': False, 'size': 2}}
    for nonneg in (True, False):
        np.testing.assert_array_equal(qphb.make_h_constraint(None, 7, sp, nonneg), orc.make_h_constraint(7, sp, nonneg))
    h = qphb.make_h_constraint(None, 7, sp, False, neg_allowed_indices=np.array([5]))
    assert h.tolist() == [0, 1000, 1000, 0, 0, 1e5, 0]
    a, b = qphb.get_default_hypers(), orc.get_default_hypers()
    assert set(a) == set(b)
    for k in a:
        np.testing.assert_array_equal(np.asarray(a[k], dtype=object), np.asarray(b[k], dtype=object))


def test_shard_bounds_cover_everything():
    from hipdrt.mapping.drtmd import shard_bounds
    for n, w in ((10000, 8), (1024, 3), (5, 8), (0, 2)):
        spans = [shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_synth_is_seeded():
    from hipdrt import synth
    f = np.logspace(6, -1, 256)
    np.testing.assert_array_equal(synth.zarc2_spectrum(f, 3, jitter=True), synth.zarc2_batch(f, 4)[3])
    g = load("refrun_c2_256x512_s1.npz")
    np.testing.assert_array_equal(synth.zarc2_spectrum(f, 1), g["z"])
