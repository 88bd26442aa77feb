"""GPU: the anti-aliasing filter of the chrono down-sampling (device kernel) against its scipy-based oracle and against
the reference run, and the down-sampled fit end to end."""
import numpy as np
import pytest

from hybrid_util import load_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,smax", [(37, 0.2), (400, 3.0), (5000, 40.0), (12, 30.0)])
def test_nonuniform_gaussian_filter_matches_scipy_oracle(n, smax):
    """random widths incl. zero and sub-minimum ones, widths larger than the array (multiple reflections)"""
    from hipdrt import filters
    from oracle import filters_oracle
    rng = np.random.default_rng(n)
    a = np.cumsum(rng.standard_normal(n))
    sigma = smax * rng.random(n) ** 3
    sigma[rng.random(n) < 0.2] = 0.0
    got = filters.nonuniform_gaussian_filter1d(a, sigma.copy())
    ref = filters_oracle.nonuniform_gaussian_filter1d(a, sigma.copy())
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-13 * np.abs(a).max())
    np.testing.assert_array_equal(filters.nonuniform_gaussian_filter1d(a, np.zeros(n)), a)


def test_downsample_data_matches_reference_run():
    """preprocessing.downsample_data on the dense record of the fixture: same kept samples, same filtered voltages"""
    from hipdrt import preprocessing as pp
    g, _ = load_case("hybrid_downsample")
    st, sa, _ = pp.process_input_signal(g["times"], g["i_signal"], 'ideal', True)
    t_s, i_s, v_s, idx = pp.downsample_data(g["times"], g["i_signal"], g["v_signal"], step_times=st,
                                            target_times=g["downsample_target_times"], prestep_samples=10)
    np.testing.assert_array_equal(idx, g["sample_index"])
    np.testing.assert_array_equal(t_s, g["sample_times"])
    np.testing.assert_allclose(v_s, g["sample_v"], rtol=0, atol=1e-13 * np.abs(g["sample_v"]).max())
    assert np.abs(v_s - g["v_signal"][idx]).max() > 1e-6           # the filter really acted on the decimated stretch


def test_fit_hybrid_with_downsampling_matches_reference_run():
    from hipdrt.models import DRT
    from test_gpu_hybrid import _check_fit
    g, special = load_case("hybrid_downsample")
    drt = DRT(warn=False)
    fp = drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"], downsample=True,
                        downsample_kw=dict(prestep_samples=10, target_times=g["downsample_target_times"]))
    assert drt.qphb_params["num_chrono"] == len(g["sample_index"])
    _check_fit(drt, g, special, False, data_rtol=1e-11)
    np.testing.assert_allclose(fp["v_baseline"], g["v_baseline"], rtol=1e-7)
