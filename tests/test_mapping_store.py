"""CPU: the bookkeeping of mapping.DRTMD (observation store + fit_all, hybdrt/mapping/drtmd.py:186-329) with a stand-in for the
device fit: which observations a call sends to the device, what a failure leaves behind with and without ignore_errors."""
import numpy as np
import pytest

from hipdrt.mapping import store


class _NoDevice:            # stands for the DRT object: the store only hands it on to the driver
    pass


@pytest.fixture
def fake_driver(monkeypatch):
    calls = []

    def fit_observations(drt, observations=None, tau_supergrid=None, drt_var=False, ignore_errors=False, llh_kw=None, rss_kw=None,
                         fit_type='drt', pfrt_factors=None, **fit_kw):
        calls.append([int(o[1][1][0].real) for o in observations])          # the tag every fake spectrum carries
        num, nt = len(observations), len(tau_supergrid)
        tags = np.array(calls[-1])
        ok = tags >= 0                                                       # a negative tag = a spectrum the solver breaks down on
        obs_x = np.where(ok[:, None], tags[:, None] * np.ones((num, nt)), 0.0)
        res = dict(obs_fit_status=ok, obs_fit_errors=[None if g else ValueError("Rank(A) < p or Rank([P; A; G]) < n") for g in ok],
                   obs_llh=np.where(ok, -tags.astype(float), 0.0), obs_rss=np.where(ok, 0.5 * tags, 0.0),
                   obs_tau_indices=[(0, nt)] * num, obs_drt_var=np.where(ok[:, None], 2.0 * obs_x, 0.0))
        assert llh_kw == {'normalize': True, 'weights': 'uniform'} and fit_kw == {'nonneg': True}
        return obs_x, {'R_inf': np.where(ok, 10.0 + tags, 0.0)}, res

    monkeypatch.setattr(store._driver, "fit_observations", fit_observations)
    return calls


def _obs(tag):
    f = np.logspace(3, 0, 5)
    return None, (f, np.full(5, tag + 0j))


def test_fit_all_only_fits_what_is_neither_fitted_nor_ignored(fake_driver):
    md = store.DRTMD(np.logspace(-4, 1, 7), drt=_NoDevice())
    for t in (1, 2, 3):
        md.add_observation([t], *_obs(t))
    assert md.num_obs == 3 and not md.obs_fit_status.any()
    assert md.fit_all().tolist() == [0, 1, 2] and fake_driver[-1] == [1, 2, 3]
    assert md.obs_fit_status.all() and md.obs_x[:, 0].tolist() == [1, 2, 3] and md.obs_special['R_inf'].tolist() == [11, 12, 13]
    md.add_observation([4], *_obs(4))
    md.add_observation([5], *_obs(5))
    assert md.obs_special['R_inf'].shape == (5,) and md.obs_x.shape == (5, 7)
    assert md.fit_all(refit=False).tolist() == [3, 4] and fake_driver[-1] == [4, 5]          # drtmd.py:326-327
    assert md.fit_all(refit=False).tolist() == [] and len(fake_driver) == 2                  # nothing left: no device job
    assert md.fit_all(refit=True).tolist() == [0, 1, 2, 3, 4] and fake_driver[-1] == [1, 2, 3, 4, 5]
    assert md.obs_llh.tolist() == [-1, -2, -3, -4, -5] and md.obs_tau_indices[4] == (0, 7)
    np.testing.assert_array_equal(md.obs_drt_var, 2.0 * md.obs_x)


def test_failed_observation_is_flagged_and_skipped_later(fake_driver):
    md = store.DRTMD(np.logspace(-4, 1, 7), drt=_NoDevice())
    for t in (1, -2, 3):
        md.add_observation([abs(t)], *_obs(t))
    md.fit_all(ignore_errors=True)                                                           # drtmd.py:292-299
    assert md.obs_fit_status.tolist() == [True, False, True] and md.obs_ignore_flag.tolist() == [False, True, False]
    assert isinstance(md.obs_fit_errors[1], ValueError) and md.obs_x[1].tolist() == [0.0] * 7
    md.add_observation([4], *_obs(4))
    assert md.fit_all().tolist() == [3]                                                      # the ignored one is not retried
    assert md.fit_all(refit=True, ignore_errors=True).tolist() == [0, 1, 2, 3]               # refit=True retries everything


def test_without_ignore_errors_the_first_failure_raises_and_later_observations_stay_unfitted(fake_driver):
    md = store.DRTMD(np.logspace(-4, 1, 7), drt=_NoDevice())
    for t in (1, -2, 3):
        md.add_observation([abs(t)], *_obs(t))
    with pytest.raises(ValueError, match="Rank"):
        md.fit_all()                                                                          # drtmd.py:300-301
    assert md.obs_fit_status.tolist() == [True, False, False] and not md.obs_ignore_flag.any()
    assert md.fit_all(ignore_errors=True).tolist() == [1, 2]
    assert md.obs_fit_status.tolist() == [True, False, True] and md.obs_ignore_flag.tolist() == [False, True, False]


def test_add_observation_with_fit_and_psi_checks(fake_driver):
    md = store.DRTMD(np.logspace(-4, 1, 7), drt=_NoDevice(), psi_dim_names=['T', 'p'])
    md.add_observation([300.0, 1.0], *_obs(7), fit=True)
    assert fake_driver[-1] == [7] and md.obs_fit_status.tolist() == [True] and md.obs_psi.shape == (1, 2)
    with pytest.raises(ValueError):
        md.add_observation([300.0], *_obs(8))
    with pytest.raises(ValueError):
        store.DRTMD(np.logspace(-4, 1, 7), drt=_NoDevice(), fit_type='nope')
