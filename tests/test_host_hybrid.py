"""CPU: host logic of the chrono / hybrid fits (step detection, scaling, grids, vz_offset strength, DOP column scaling)
against the values the reference produced on the same measurement (tests/golden/refrun_hybrid_s0*.npz)."""
import numpy as np
import pytest

from hybrid_util import load_case


@pytest.fixture(scope="module")
def case():
    return load_case("hybrid_s0_dop")


def test_synthetic_measurement_is_reproducible(case):
    from hipdrt import synth
    g, _ = case
    times, i_sig, v_sig, freq, z = synth.hybrid_measurement(seed=0)
    np.testing.assert_array_equal(times, g["times"])
    np.testing.assert_array_equal(i_sig, g["i_signal"])
    np.testing.assert_array_equal(v_sig, g["v_signal"])
    np.testing.assert_array_equal(z, g["z"])


def test_step_detection_and_scaling(case):
    from hipdrt import preprocessing as pp
    g, _ = case
    st, sa, tr = pp.process_input_signal(g["times"], g["i_signal"], 'ideal', True)
    assert tr is None
    np.testing.assert_array_equal(st, g["step_times"])
    np.testing.assert_array_equal(sa, g["step_sizes"])
    rp = pp.estimate_rp(g["times"], st, sa, g["v_signal"], 'ideal', g["z"])
    np.testing.assert_allclose(rp / 14, g["coefficient_scale"], rtol=1e-14)
    np.testing.assert_allclose(np.max(np.abs(sa)) * rp / 14, g["response_signal_scale"], rtol=1e-14)
    np.testing.assert_allclose(pp.get_basis_tau(g["freq"], g["times"], st), g["basis_tau"], rtol=1e-13)
    # step sizes from given step times, consecutive-step condensation in estimate_rp
    np.testing.assert_allclose(pp.get_step_sizes(g["times"], g["i_signal"], st), sa)
    two = np.array([st[0], st[0] + 1e-5])
    assert np.isfinite(pp.estimate_rp(g["times"], two, np.array([sa[0] / 2, sa[0] / 2]), g["v_signal"], 'ideal', None))


def test_time_since_step_and_model_signal(case):
    from hipdrt import preprocessing as pp
    g, _ = case
    t = g["times"]
    d = pp.get_time_since_step(t, g["step_times"], prestep_value=-1)
    assert len(d) == len(t) and np.all(d[t < g["step_times"][0]] == -1)
    assert np.min(d[d > 0]) >= np.min(np.diff(t))
    sig = pp.generate_model_signal(t, g["step_times"], g["step_sizes"])
    np.testing.assert_allclose(sig * 1.0 / g["input_signal_scale"], g["inf_response"] / g["input_signal_scale"])


def test_vz_strength_and_dop_scale(case):
    from hipdrt.models import DRT
    from hipdrt.matrices import phasance
    g, _ = case
    drt = DRT.__new__(DRT)
    cs, es = drt._vz_strength(g["times"], g["freq"], g["nonconsec_step_times"], 1)
    np.testing.assert_allclose(np.concatenate([cs, np.tile(es, 2)]), g["vz_strength_vec"], rtol=1e-13)
    sv = phasance.phasor_scale_vector(g["basis_nu"], g["basis_tau"]) / (np.sqrt(np.pi) / g["nu_epsilon"])
    np.testing.assert_allclose(sv, g["dop_scale_vector"], rtol=1e-13)


def test_special_parameter_layout(case):
    from hipdrt.models import DRT
    g, special = case
    drt = DRT.__new__(DRT)
    drt.fit_ohmic = drt.fit_inductance = drt.fit_dop = True
    drt.fit_capacitance = False
    drt.basis_nu = None
    drt.nu_epsilon = None
    sp = drt._general_special_params(True, True, True)
    assert sp == special
    np.testing.assert_allclose(drt.nu_epsilon, g["nu_epsilon"])
    np.testing.assert_allclose(drt.basis_nu, g["basis_nu"])
    drt.fit_dop = False
    assert list(drt._general_special_params(False, True, True)) == ["R_inf", "inductance"]


def test_antialias_filter_oracle_matches_reference_run():
    """oracle/filters_oracle.py (scipy-based restatement of the blended Gaussian filter and its width rule) reproduces the
    down-sampled, filtered voltage record of the reference run; the host-side width rule of the product agrees with it"""
    from oracle import filters_oracle
    from hipdrt import preprocessing as pp
    g, _ = load_case("hybrid_downsample")
    idx = g["sample_index"]
    step_index = pp.identify_steps(g["i_signal"], allow_consecutive=False)
    vf = filters_oracle.filter_chrono_signal(g["times"], g["v_signal"], step_index, idx)
    np.testing.assert_allclose(vf[idx], g["sample_v"], rtol=0, atol=1e-14 * np.abs(g["sample_v"]).max())
    sd = pp.sigma_from_decimate_index(g["v_signal"], idx)
    assert sd.max() > 1 and np.all(sd[np.setdiff1d(np.arange(len(sd)), idx)] == 0)


def test_decimation_indices_match_reference_run():
    """preprocessing.downsample_data, method='decimate' / discard_first_n_points / discard_only / one-step mode, without the
    (device) anti-alias filter: kept indices and samples bit-identical to the reference's on the same records"""
    import os
    from conftest import GOLDEN
    from hipdrt import preprocessing as pp
    from oracle.make_golden import DECIMATE_CASES, decimate_records
    g = np.load(os.path.join(GOLDEN, "refrun_decimate.npz"))
    recs = decimate_records()
    for k, (rec, kw) in enumerate(DECIMATE_CASES):
        if kw.get("antialiased"):
            continue
        times, i_sig, v_sig = recs[rec][:3]
        st = times[pp.identify_steps(i_sig, allow_consecutive=False)]
        t_s, i_s, v_s, idx = pp.downsample_data(times, i_sig, v_sig, step_times=st, antialiased=False, **kw)
        np.testing.assert_array_equal(idx, g[f"case{k}_index_aa0"], err_msg=str(kw))
        np.testing.assert_array_equal(v_s, g[f"case{k}_v_aa0"])
        np.testing.assert_array_equal(i_s, g[f"case{k}_i_aa0"])
        np.testing.assert_array_equal(t_s, times[idx])
    with pytest.raises(ValueError):
        pp.downsample_data(*recs["one_step"][:3], method='nearest')
