"""CPU: the host logic of mapping.fit_observations(inflight=k) -- chunking, threads, order-preserving merge, error capture --
with a stand-in DRT whose "device" results are deterministic functions of each observation (no GPU, no library call)."""
import threading

import numpy as np
import pytest


class _FakeDRT:
    """quacks like hipdrt.models.DRT for fit_observations: fit_eis_batch, evaluate_obs_llh_rss_batch,
    estimate_distribution_var_batch; remembers which thread served it and how many observations it saw"""
    ntau = 9

    def __init__(self):
        self.calls = []
        self._z = None

    def fit_eis_batch(self, frequencies, z, **kw):
        z = np.asarray(z)
        self._z = z
        self.calls.append((threading.get_ident(), len(z), dict(kw)))
        s = z.real.sum(1)
        bad = ~np.isfinite(s)
        return dict(fit_x=np.where(bad[:, None], 0.0, np.outer(np.nan_to_num(s), np.arange(1.0, self.ntau + 1))),
                    R_inf=np.nan_to_num(z.real[:, 0]), inductance=np.nan_to_num(z.imag[:, -1]),
                    status=np.where(bad, -1, 0), outer_iters=np.full(len(z), 3), qp_iters_total=np.full(len(z), 11),
                    x=np.nan_to_num(z.real), basis_tau=np.logspace(-4, 0, self.ntau),
                    timings_ms={"qp": 1.0}, launches={"qp": 1})

    def evaluate_obs_llh_rss_batch(self, **kw):
        a = np.nan_to_num(np.abs(self._z))
        return -a.sum(1), (a ** 2).sum(1)

    def estimate_distribution_var_batch(self, tau=None, extend_var=False):
        n = len(self._z)
        return np.outer(np.arange(n) + 1.0, np.ones(len(tau))) * np.nan_to_num(self._z.real[:, :1]), np.ones(n, dtype=bool)


def _data(num, nf=9, seed=3):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((num, nf)) + 1j * rng.standard_normal((num, nf))


@pytest.mark.parametrize("num,inflight", [(23, 3), (8, 4), (10, 4), (5, 4)])
def test_inflight_merge_equals_one_batch(monkeypatch, num, inflight):
    from hipdrt.mapping import drtmd
    fakes = [_FakeDRT() for _ in range(inflight)]
    monkeypatch.setattr(drtmd, "drt_siblings", lambda drt, count: fakes[:count])
    freq = np.logspace(3, 0, 9)
    z = _data(num)
    sup = np.logspace(-5, 1, 13)
    one = drtmd.fit_observations(_FakeDRT(), freq, z, tau_supergrid=sup, drt_var=True, nonneg=True)
    many = drtmd.fit_observations(fakes[0], freq, z, tau_supergrid=sup, drt_var=True, inflight=inflight, nonneg=True)
    np.testing.assert_array_equal(one[0], many[0])
    for k in one[1]:
        np.testing.assert_array_equal(one[1][k], many[1][k])
    for k in ("obs_llh", "obs_rss", "status", "outer_iters", "x", "obs_fit_status", "fit_x"):
        np.testing.assert_array_equal(one[2][k], many[2][k], err_msg=k)
    assert many[2]["obs_tau_indices"] == one[2]["obs_tau_indices"] and len(many[2]["obs_fit_errors"]) == num
    assert many[2]["basis_tau"].shape == (9,)                     # not a per-observation array: taken once, not concatenated
    if num >= 2 * inflight:
        used = [f for f in fakes if f.calls]
        assert len(used) == inflight and sum(c[1] for f in used for c in f.calls) == num
        assert all(c[0] != threading.get_ident() for f in used for c in f.calls)      # every batch on a worker thread
        assert all(c[2] == {"nonneg": True} for f in used for c in f.calls)   # fit keywords reach every batch, `inflight` does not
    else:
        assert not any(f.calls for f in fakes[1:])                # too few observations: one batch on the caller's DRT


def test_inflight_error_capture_and_reraising(monkeypatch):
    from hipdrt.mapping import drtmd
    fakes = [_FakeDRT() for _ in range(3)]
    monkeypatch.setattr(drtmd, "drt_siblings", lambda drt, count: fakes[:count])
    freq = np.logspace(3, 0, 9)
    z = _data(12)
    z[7] = np.nan                                                  # falls into the second chunk
    obs_x, special, res = drtmd.fit_observations(fakes[0], freq, z, inflight=3, ignore_errors=True)
    assert res["obs_fit_status"].tolist() == [True] * 7 + [False] + [True] * 4
    assert isinstance(res["obs_fit_errors"][7], ValueError) and not obs_x[7].any()
    with pytest.raises(ValueError):                # the reference's default (drtmd.py:245): raise at the first failure
        drtmd.fit_observations(fakes[0], freq, z, inflight=3)

    class Boom(_FakeDRT):
        def fit_eis_batch(self, *a, **k):
            raise RuntimeError("device lost")
    monkeypatch.setattr(drtmd, "drt_siblings", lambda drt, count: [fakes[0], Boom(), fakes[2]][:count])
    with pytest.raises(RuntimeError, match="device lost"):         # a worker thread's exception surfaces in the caller
        drtmd.fit_observations(fakes[0], freq, _data(12), inflight=3)


def test_auto_inflight_rule(monkeypatch):
    from hipdrt.mapping import drtmd
    # round 6: one plan (which cuts its batch into ranges inside the library) at every size
    assert [drtmd.auto_inflight(n) for n in (1, 511, 512, 1999, 2000, 10000)] == [1] * 6
    fakes = [_FakeDRT() for _ in range(3)]
    monkeypatch.setattr(drtmd, "drt_siblings", lambda drt, count: fakes[:count])
    freq = np.logspace(3, 0, 9)
    drtmd.fit_observations(fakes[0], freq, _data(600), inflight='auto')
    assert [len(f.calls) for f in fakes] == [1, 0, 0] and fakes[0].calls[0][1] == 600
    drtmd.fit_observations(fakes[0], freq, _data(600), inflight=2)          # an explicit count still splits
    assert [len(f.calls) for f in fakes] == [2, 1, 0] and [f.calls[-1][1] for f in fakes[:2]] == [300, 300]


def test_extremes_are_dropped_per_observation_before_batches_are_formed():
    """remove_extremes among a map's fit keywords (host logic only): the quantile-range filter runs per observation, observations that
    lose different points no longer share a group, the switches leave the fit keywords; tags keep like-looking observations apart"""
    import types
    from hipdrt.mapping import drtmd
    from hipdrt.models.prepared import PreparedFitMixin
    stub = types.SimpleNamespace(warn=False)
    stub._drop_extremes = lambda meas, ekw=None: PreparedFitMixin._drop_extremes(stub, meas, ekw)
    freq = np.logspace(4, 0, 30)
    z = (1.0 / (1.0 + 1j * freq / 50.0)).astype(complex)
    z_bad = z.copy()
    z_bad[7] += 40.0
    obs = [(None, (freq, z)), (None, (freq, z_bad)), (None, (freq, z * 1.01))]
    assert [len(i) for _, i in drtmd.observation_groups(obs)] == [3]
    cleaned, kw, tags, step_times = drtmd.prefilter_observations(stub, obs, dict(nonneg=True, remove_extremes=True, extreme_kw=None))
    assert kw == dict(nonneg=True) and tags is None and step_times == [None] * 3
    assert [len(o[1][0]) for o in cleaned] == [30, 29, 30] and 7 not in np.searchsorted(-freq, -cleaned[1][1][0])
    groups = drtmd.observation_groups(cleaned)
    assert [idx for _, idx in groups] == [[0, 2], [1]]                      # the filtered one has its own frequency grid now
    assert [idx for _, idx in drtmd.observation_groups(obs, tags=["a", "b", "a"])] == [[0, 2], [1]]
    with pytest.raises(ValueError, match="outlier_p"):
        drtmd.prefilter_observations(stub, obs, dict(remove_outliers=True))


def test_a_map_larger_than_one_batch_is_fitted_in_consecutive_batches():
    """max_batch: the shared-grid form cuts a map that does not fit the device at once into consecutive batches of nearly equal
    size on the same plan; results come back in the original order (host logic with a stand-in DRT)"""
    from hipdrt.mapping import drtmd
    fake = _FakeDRT()
    freq = np.logspace(3, 0, 9)
    z = _data(25)
    obs_x, obs_special, res = drtmd.fit_observations(fake, freq, z, max_batch=10)
    assert [c[1] for c in fake.calls] == [9, 8, 8]                      # three batches, not 10 + 10 + 5
    whole = _FakeDRT()
    ref_x, ref_special, ref = drtmd.fit_observations(whole, freq, z)
    assert [c[1] for c in whole.calls] == [25]                          # (a stand-in cannot say what fits: one batch)
    np.testing.assert_array_equal(obs_x, ref_x)
    for key in ref_special:
        np.testing.assert_array_equal(obs_special[key], ref_special[key])
    np.testing.assert_array_equal(res["obs_llh"], ref["obs_llh"])
    assert len(res["obs_fit_errors"]) == 25 and res["obs_fit_status"].all()
    assert drtmd.max_batch_for(fake, freq) is None
