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


def test_decimated_records_match_reference_run():
    """method='decimate' with the anti-alias filter on the device: same kept indices, filtered currents / voltages within
    1e-13 of the reference's scipy filter"""
    import os
    from conftest import GOLDEN
    from hipdrt import preprocessing as pp
    from oracle.make_golden import DECIMATE_CASES, decimate_records
    g = np.load(os.path.join(GOLDEN, "refrun_decimate.npz"))
    recs = decimate_records()
    acted = 0
    for k, (rec, kw) in enumerate(DECIMATE_CASES):
        times, i_sig, v_sig = recs[rec][:3]
        st = times[pp.identify_steps(i_sig, allow_consecutive=False)]
        kw = dict(kw)
        kw.setdefault("antialiased", True)
        t_s, i_s, v_s, idx = pp.downsample_data(times, i_sig, v_sig, step_times=st, **kw)
        np.testing.assert_array_equal(idx, g[f"case{k}_index_aa1"], err_msg=str(kw))
        np.testing.assert_allclose(v_s, g[f"case{k}_v_aa1"], rtol=0, atol=1e-13 * np.abs(v_sig).max(), err_msg=str(kw))
        np.testing.assert_allclose(i_s, g[f"case{k}_i_aa1"], rtol=0, atol=1e-13 * np.abs(i_sig).max(), err_msg=str(kw))
        acted += int(np.abs(v_s - g[f"case{k}_v_aa0"]).max() > 1e-6) if len(v_s) == len(g[f"case{k}_v_aa0"]) else 0
    assert acted >= 6            # the filter changed the kept samples in the decimated cases


def test_fit_hybrid_with_decimation_matches_reference_run():
    from hipdrt.models import DRT
    from test_gpu_hybrid import _check_fit
    g, special = load_case("hybrid_decimate")
    drt = DRT(warn=False)
    fp = drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"], downsample=True,
                        downsample_kw=dict(method='decimate', prestep_samples=10, decimation_interval=20,
                                           decimation_factor=1.5, decimation_max_period=0.05))
    np.testing.assert_array_equal(drt.sample_index, g["sample_index"])
    assert drt.qphb_params["num_chrono"] == len(g["sample_index"])
    _check_fit(drt, g, special, False, data_rtol=1e-11)
    np.testing.assert_allclose(fp["v_baseline"], g["v_baseline"], rtol=1e-7)
