"""GPU: the config-5 family (distribution of phasances inside the loop, joint chrono + EIS fits) through the prepared-plan
entry points of the C-ABI, against the reference-run fixtures and the oracle's general loop."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, parity, parity_close

from oracle import drt_oracle as orc
from hybrid_util import load_case, initial_rzm_and_vz

pytestmark = pytest.mark.gpu


def make_desc(ffi, g, special, rzm0, vz, hyp):
    d = ffi.PreparedDesc()
    d.m, d.n = rzm0.shape
    d.ns = int(sum(v["size"] for v in special.values()))
    d.dop_start, d.dop_size = (special["x_dop"]["index"], special["x_dop"]["size"]) if "x_dop" in special else (0, 0)
    d.vz_index = vz["index"] if vz else -1
    d.vb_start, d.vb_size = (vz["vb"][0], vz["vb"][1] - vz["vb"][0]) if vz else (0, 0)
    d.num_chrono = vz["num_chrono"] if vz else 0
    d.toeplitz_m = 0
    if "x_dop" in special:
        d.dop_l2_lambda_0 = hyp["dop_l2_lambda_0"]
        for k in range(3):
            d.dop_derivative_weights[k] = hyp["dop_derivative_weights"][k]
            d.dop_s_alpha[k] = hyp["dop_s_alpha"][k]
            d.dop_rho_alpha[k] = hyp["dop_rho_alpha"][k]
            d.dop_s_0[k] = hyp["dop_s_0"][k]
            d.dop_rho_0[k] = hyp["dop_rho_0"][k]
    return d


@pytest.mark.parametrize("name", ["golden71x91_dop", "hybrid_s0", "hybrid_s0_dop", "chrono_s1", "hybrid_3step"])
@pytest.mark.parametrize("batched", [False, True])
def test_prepared_plan_reproduces_reference_trajectory(name, batched):
    from hipdrt import _ffi as ffi
    g, special = load_case(name)
    hyp = orc.get_default_hypers()
    if "x_dop" in special:
        hyp.update(orc.get_default_dop_hypers())
    rzm0, vz = initial_rzm_and_vz(g, special)
    if vz is not None and not batched:
        pytest.skip("a vz_offset column needs per-measurement matrices")
    ref = orc.qphb_fit_prepared(rzm0, g["rv"], [g["m0"], g["m1"], g["m2"]], g["vmm"], special, hyp, vz=vz)
    n = rzm0.shape[1]
    h = orc.make_h_constraint(n, special, True)
    ctx = ffi.get_context()
    desc = make_desc(ffi, g, special, rzm0, vz, hyp)
    B = 3 if batched else 1
    plan = ffi.PreparedPlan(ctx, desc, [g["m0"], g["m1"], g["m2"]], g["vmm"], h, ref["l1_lambda_vector"],
                            vz_strength=vz["strength"] if vz else None, capacity=B)
    rzv = np.tile(g["rv"], (B, 1))
    if batched:     # members 0 and 2 are the fixture's measurement, member 1 a scaled copy (different trajectory)
        rzv[1] *= 0.5
    plan.upload(np.tile(rzm0, (B, 1, 1)) if batched else rzm0, rzv)
    b = B - 1
    plan.record_history(b)
    plan.fit()
    out = plan.download(s_vectors=True)
    hist = plan.history()
    assert out["status"][b] == 0
    assert hist["qp_iterations"].tolist() == g["qp_iterations"].tolist()
    assert out["outer_iters"][b] == int(g["outer_iterations"])
    parity("x", hist["x"], g["hist_x"], default=1e-7)
    parity("rho", hist["rho"], g["hist_rho"], default=1e-6, rel=True)
    parity("weights", out["weights"][b], g["weights"], default=1e-6, rel=True)
    parity("s_vectors", out["s_vectors"][b], g["s_vectors"], default=1e-5, rel=True)
    parity("xmx", plan.get("xmx")[b], g["xmx_norms"], default=1e-6, rel=True)
    if "x_dop" in special:
        parity("dop_rho", hist["dop_rho"], g["hist_dop_rho"], default=1e-6, rel=True)
        parity("dop_xmx", plan.get("dop_xmx")[b], g["dop_xmx_norms"], default=1e-6, rel=True)
    pm = plan.p_matrix(b)
    parity("pm", pm, g["p_matrix"], default=1e-8)
    parity("q_vector", out["q_vector"][b], g["q_vector"], default=1e-8)
    if vz is not None:
        np.testing.assert_allclose(plan.get("rzm")[b], g["rm"], rtol=0, atol=1e-7)
    if batched:
        np.testing.assert_array_equal(out["x"][0], out["x"][2])           # identical inputs -> identical bits
        assert not np.allclose(out["x"][1], out["x"][0])
        # the scaled member against its own oracle run
        r1 = orc.qphb_fit_prepared(rzm0, rzv[1], [g["m0"], g["m1"], g["m2"]], g["vmm"], special, hyp, vz=vz)
        assert out["outer_iters"][1] == len(r1["history"])
        parity("x_2", out["x"][1], r1["x"], default=1e-7)


def _check_fit(drt, g, special, dop, data_rtol=1e-12, mat_rtol=1e-9):
    """data_rtol / mat_rtol are loosened for solve_rp runs, whose data and DOP columns carry a factor taken from a QP
    solution (agreement 1e-10 rather than rounding level)"""
    fp, qp = drt.fit_parameters, drt.qphb_params
    assert drt.special_qp_params == special
    assert qp["qp_iterations"].tolist() == g["qp_iterations"].tolist()
    assert qp["outer_iterations"] == int(g["outer_iterations"])
    # matrices the host layer composed from the device builders
    rzm0 = g["rm"].copy()
    mine = qp["rm"].copy()
    if "vz_offset" in special:
        np.testing.assert_allclose(mine[:, special["vz_offset"]["index"]], g["rm"][:, special["vz_offset"]["index"]],
                                   rtol=0, atol=1e-7)
        mine[:, special["vz_offset"]["index"]] = 0
        rzm0[:, special["vz_offset"]["index"]] = 0
    np.testing.assert_allclose(mine, rzm0, rtol=mat_rtol, atol=mat_rtol * 1e-2 * np.abs(rzm0).max())
    np.testing.assert_allclose(qp["rv"], g["rv"], rtol=data_rtol, atol=data_rtol * np.abs(g["rv"]).max())
    np.testing.assert_allclose(qp["vmm"], g["vmm"], rtol=1e-12, atol=1e-15)
    for k in range(3):
        ref = g[f"m{k}"]
        np.testing.assert_allclose(qp["penalty_matrices"][f"m{k}"], ref, rtol=1e-11, atol=1e-13 * np.abs(ref).max())
    np.testing.assert_allclose(qp["l1_lambda_vector"], g["l1_lambda_vector"])
    # results, reference's own criterion (tests/test_drt_fit.py: np.allclose) and tighter on the coefficients
    parity("x", fp["x"], g["x"], default=1e-7)
    parity("x_2", drt.cvx_result["x"], g["x_scaled"], default=1e-7)
    parity("R_inf", fp["R_inf"], g["R_inf"], default=1e-6, rel=True)
    parity("inductance", fp["inductance"], g["inductance"], default=1e-5, rel=True)
    if "C_inv" in g:
        parity("C_inv", fp["C_inv"], g["C_inv"], default=1e-5, rel=True)
    if "z_sigma_tot" in g:
        parity("z_sigma_tot", fp["z_sigma_tot"], g["z_sigma_tot"], default=1e-6, rel=True)
    parity("q_vector", fp["q_vector"], g["q_vector"], default=1e-8)
    parity("p_matrix", fp["p_matrix"], g["p_matrix"], default=1e-8)
    parity("rho_vector", qp["rho_vector"], g["rho_vector"], default=1e-6, rel=True)
    if dop:
        parity("x_dop", fp["x_dop"], g["x_dop"], default=1e-8)
        parity("dop_rho_vector", qp["dop_rho_vector"], g["dop_rho_vector"], default=1e-6, rel=True)
        np.testing.assert_allclose(drt.dop_scale_vector, g["dop_scale_vector"], rtol=max(1e-13, data_rtol))


def test_fit_eis_with_dop_matches_reference_run():
    """DRT(fit_dop=True).fit_eis on the reference test's own spectrum (drt1d.py:1215-1241 with the x_dop block)."""
    from hipdrt.models import DRT
    g, special = load_case("golden71x91_dop")
    drt = DRT(fit_dop=True)
    drt.fit_eis(g["freq"], g["z"])
    np.testing.assert_allclose(drt.basis_tau, g["basis_tau"], rtol=1e-13)
    _check_fit(drt, g, special, True)
    assert drt.fit_type == "qphb_eis"


@pytest.mark.parametrize("dop", [False, True])
def test_fit_hybrid_matches_reference_run(dop):
    """DRT.fit_hybrid (drt1d.py:1244-1268) end to end: step detection, scaling, device-built response / impedance /
    phasance / penalty / variance matrices, the device loop with the vz_offset column rewrite, parameter extraction."""
    from hipdrt.models import DRT
    g, special = load_case("hybrid_s0_dop" if dop else "hybrid_s0")
    drt = DRT(fit_dop=dop)
    fp = drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"])
    np.testing.assert_allclose(drt.basis_tau, g["basis_tau"], rtol=1e-13)
    np.testing.assert_array_equal(drt.step_times, g["step_times"])
    np.testing.assert_allclose(drt.response_signal_scale, g["response_signal_scale"], rtol=1e-14)
    _check_fit(drt, g, special, dop)
    parity("v_baseline", fp["v_baseline"], g["v_baseline"], default=1e-7, rel=True)
    parity("vz_offset", fp["vz_offset"], g["vz_offset"], default=1e-5, rel=True)
    parity("v_sigma_tot", fp["v_sigma_tot"], g["v_sigma_tot"], default=1e-6, rel=True)
    assert drt.fit_type == "qphb_hybrid"
    with pytest.raises(NotImplementedError):
        drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"], subtract_background=True)
    with pytest.raises(ValueError):
        drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"], not_a_keyword=1)


def test_fit_hybrid_batch_members_match_single_fits():
    """four jittered cells measured with one protocol, fitted concurrently: each member equals its own single fit"""
    from hipdrt.models import DRT
    from hipdrt import synth
    meas = [synth.hybrid_measurement(seed=s) for s in range(4)]
    times, freq = meas[0][0], meas[0][3]
    drt = DRT()
    res = drt.fit_hybrid_batch(times, [m_[1] for m_ in meas], [m_[2] for m_ in meas], freq, [m_[4] for m_ in meas])
    assert res["x"].shape[0] == 4 and np.all(res["status"] == 0)
    for b in (0, 3):
        single = DRT()
        fp = single.fit_hybrid(*meas[b])
        np.testing.assert_array_equal(res["x"][b], fp["x"])
        np.testing.assert_array_equal(res["vz_offset"][b], fp["vz_offset"])
        assert res["outer_iters"][b] == single.qphb_params["outer_iterations"]


def test_hyper_step_on_many_workgroups_is_bit_identical():
    """One large joint fit alone has the matrix-vector products of its hyper-parameter step (rm @ x, vmm @ resid^2, the
    vz_offset column) computed by a many-workgroup kernel before hyper_kernel (hyper.hip, premv_kernel: m n >= 2^20 and at
    most 32 fits); the same measurement as member of a batch of 40 runs them inside hyper_kernel.  Same rows_matvec, same
    bits (coneqp pinned to the one-workgroup kernel for both, which is what a batch of 40 gets anyway)."""
    from hipdrt.models import DRT
    from hipdrt import synth, _ffi
    meas = synth.hybrid_measurement(seed=1, n_pre=96, n_post=1500, nf=128)
    tau = np.logspace(-7, 3, 640)
    ctx = _ffi.get_context()
    try:
        ctx.debug_qp_group(0)
        one = DRT(fixed_basis_tau=tau, warn=False)
        r1 = one.fit_hybrid_batch(meas[0], [meas[1]], [meas[2]], meas[3], [meas[4]])
        assert (len(meas[0]) + 2 * len(meas[3])) * (len(tau) + 4) >= 1 << 20       # rows x columns of the response matrix
        many = DRT(fixed_basis_tau=tau, warn=False)
        r40 = many.fit_hybrid_batch(meas[0], [meas[1]] * 40, [meas[2]] * 40, meas[3], [meas[4]] * 40)
        three = DRT(fixed_basis_tau=tau, warn=False)            # several fits per launch of the many-workgroup kernel
        r3 = three.fit_hybrid_batch(meas[0], [meas[1]] * 3, [meas[2]] * 3, meas[3], [meas[4]] * 3)
    finally:
        ctx.debug_qp_group(-1)
    assert r1["outer_iters"][0] == r40["outer_iters"][0] == r40["outer_iters"][39] == r3["outer_iters"][2]
    for key in ("x", "vz_offset", "R_inf"):
        np.testing.assert_array_equal(r40[key][0], r1[key][0])
        np.testing.assert_array_equal(r40[key][39], r1[key][0])
        np.testing.assert_array_equal(r3[key][1], r1[key][0])
        np.testing.assert_array_equal(r3[key][2], r1[key][0])


def test_joint_fits_are_scale_and_order_equivariant():
    """size-independent properties of the joint chrono + EIS path: multiplying a cell's voltages and impedances by a power
    of two multiplies its resistances by exactly that factor and leaves the scaled trajectory untouched; the order of the
    measurements in a batch is irrelevant (bit for bit)"""
    from hipdrt.models import DRT
    from hipdrt import synth
    meas = [synth.hybrid_measurement(seed=s, jitter=True) for s in range(6)]
    times, freq = meas[0][0], meas[0][3]
    drt = DRT(fit_dop=True, warn=False)
    res = drt.fit_hybrid_batch(times, [m_[1] for m_ in meas], [m_[2] for m_ in meas], freq, [m_[4] for m_ in meas])
    factor = np.array([4.0, 0.25, 2.0, 1.0, 0.5, 8.0])
    perm = np.array([3, 0, 5, 1, 4, 2])
    res2 = drt.fit_hybrid_batch(times, [meas[b][1] for b in perm], [meas[b][2] * factor[b] for b in perm], freq,
                                [meas[b][4] * factor[b] for b in perm])
    np.testing.assert_array_equal(res2["outer_iters"], res["outer_iters"][perm])
    np.testing.assert_array_equal(res2["x"], (res["x"] * factor[:, None])[perm])
    np.testing.assert_array_equal(res2["R_inf"], (res["R_inf"] * factor)[perm])
    np.testing.assert_array_equal(res2["x_dop"], (res["x_dop"] * factor[:, None])[perm])


def test_config5_full_size_joint_fit_with_dop():
    """BASELINE config 5 at full size against the REFERENCE ITSELF: 512 frequencies + 4096 time samples x 1024 tau with the
    distribution of phasances (m = 5120 rows, n = 1078 unknowns), one measurement, every matrix built by the device
    kernels.  tests/golden/refrun_config5_full.npz holds the first twelve outer iterations of hybrid-drt's own fit_hybrid
    on this workload (oracle/make_golden.py --only-config5; 20 uV of voltage noise, for which the reference's outer
    iteration is contractive: step sizes 0.11, 0.03, 0.02, ... 0.01).  Pinned: the interior-point iteration count of all
    thirteen QPs, every iterate within 1e-6 of its largest coefficient, the hyper-parameters and the extracted parameters."""
    import time
    from hipdrt.models import DRT
    from hipdrt import synth
    g = np.load(os.path.join(GOLDEN, "refrun_config5_full.npz"))
    K = int(g["K"])
    meas = synth.hybrid_measurement(seed=0, n_pre=96, n_post=4000, nf=512, v_noise=float(g["v_noise"]))
    tau = np.logspace(-7, 3, 1024)
    drt = DRT(fixed_basis_tau=tau, fit_dop=True, warn=False)
    fp = drt.fit_hybrid(*meas, max_iter=K)
    qp = drt.qphb_params
    assert qp["rm"].shape == tuple(g["rm_shape"]) == (5120, 1078) and qp["num_chrono"] == 4096 and qp["num_eis"] == 512
    assert qp["qp_iterations"].tolist() == g["qp_iterations"].tolist()
    dx = np.array([h["x"] for h in drt.qphb_history])
    assert dx.shape == g["hist_x"].shape == (K, 1078)
    scale = np.abs(g["hist_x"]).max(axis=1)
    dev_err = np.abs(dx - g["hist_x"]).max(axis=1) / scale
    print("device vs reference per outer iteration:", np.array2string(dev_err, precision=2))
    parity_close("config5_full.hist_x", dev_err, np.zeros_like(dev_err), 1e-8, scale=1.0)       # measured 4.8e-10
    parity("rho_vector", np.array([h["rho_vector"] for h in drt.qphb_history]), g["hist_rho"], default=1e-6, rel=True)
    parity("dop_rho_vector", np.array([h["dop_rho_vector"] for h in drt.qphb_history]), g["hist_dop_rho"], default=1e-6, rel=True)
    np.testing.assert_allclose(qp["rv"], g["rv"], rtol=1e-12, atol=1e-13 * np.abs(g["rv"]).max())
    np.testing.assert_allclose(drt.coefficient_scale, float(g["coefficient_scale"]), rtol=1e-13)
    # (bounds = ten to twenty times the deviations measured in round 4: 5.5e-11, 1.4e-9, 8.8e-11 / 8.2e-9 / 2.4e-11, 8.3e-12)
    parity_close("config5_full.x", fp["x"], g["x"], 1e-9)
    parity_close("config5_full.x_dop", fp["x_dop"], g["x_dop"], 2e-8)
    for key, bound in (("R_inf", 1e-9), ("inductance", 1e-7), ("vz_offset", 1e-9)):
        parity_close("config5_full." + key, np.atleast_1d(fp[key]), np.atleast_1d(float(g[key])), bound)
    parity_close("config5_full.est_weights", qp["est_weights"] / g["est_weights"], np.ones_like(g["est_weights"]), 1e-10, scale=1.0)

    # the full run (defaults, 50 outer iterations at most): properties of the result
    t0 = time.time()
    fp = drt.fit_hybrid(*meas)
    print(f"config 5, {qp['outer_iterations']} outer iterations: {time.time() - t0:.2f} s wall incl. matrix builds and "
          f"transfers; device timings {drt._plan.timings()[0]}")
    qp = drt.qphb_params
    resid = (qp["rm"] @ drt.cvx_result["x"] - qp["rv"]) * qp["weights"]
    assert 0.3 < np.sqrt(np.mean(resid ** 2)) < 3.0            # both data sets reproduced at their noise level
    assert np.all(fp["x"] >= -1e-12) and np.all(fp["x_dop"] >= -1e-12) and fp["R_inf"] >= 0
    assert abs(fp["R_inf"] - 1.0) < 0.3                        # the synthetic cell's series resistance (DOP terms share it)


def test_fit_eis_batch_with_dop_shares_one_matrix_set():
    """fit_eis_batch with fit_dop=True: one shared response matrix for the batch, members equal their single fits"""
    from hipdrt.models import DRT
    from hipdrt import synth
    freq = np.logspace(5, 0, 61)
    zb = synth.zarc2_batch(freq, 5)
    drt = DRT(fit_dop=True, warn=False)
    res = drt.fit_eis_batch(freq, zb)
    assert not drt._plan.rm_batched and res["x_dop"].shape == (5, 50)
    single = DRT(fit_dop=True, warn=False)
    fp = single.fit_eis(freq, zb[3])
    np.testing.assert_array_equal(res["fit_x"][3], fp["x"])
    np.testing.assert_array_equal(res["x_dop"][3], fp["x_dop"])
    assert res["outer_iters"][3] == single.qphb_params["outer_iterations"]


def test_fit_chrono_matches_reference_run():
    """DRT.fit_chrono (drt1d.py:1195-1213): chrono-only fit, no vz_offset column, uniform error structure"""
    from hipdrt.models import DRT
    g, special = load_case("chrono_s1")
    drt = DRT()
    fp = drt.fit_chrono(g["times"], g["i_signal"], g["v_signal"])
    np.testing.assert_allclose(drt.basis_tau, g["basis_tau"], rtol=1e-13)
    _check_fit(drt, g, special, False)
    parity("v_baseline", fp["v_baseline"], g["v_baseline"], default=1e-7, rel=True)
    parity("v_sigma_tot", fp["v_sigma_tot"], g["v_sigma_tot"], default=1e-6, rel=True)
    assert fp["z_sigma_tot"] is None and drt.fit_type == "qphb_chrono"


@pytest.mark.parametrize("name,kw", [
    ("hybrid_3step", dict(vz_offset_scale=0.5, vz_offset_eps=2)),
    ("hybrid_3step_opts", dict(vz_offset=False, chrono_error_structure=None, smooth_inf_response=False,
                               offset_baseline=False, chrono_vmm_epsilon=2, vz_offset_eps=2)),
])
def test_fit_hybrid_three_step_protocol_and_options(name, kw):
    """three current steps (step detection, per-step response layers, segment-wise chrono variance matrix) and the
    non-default chrono keywords"""
    from hipdrt.models import DRT
    g, special = load_case(name)
    drt = DRT(warn=False)
    fp = drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"], **kw)
    np.testing.assert_array_equal(drt.step_times, g["step_times"])
    np.testing.assert_allclose(drt.step_sizes, g["step_sizes"], rtol=1e-14)
    _check_fit(drt, g, special, False)
    parity("v_baseline", fp["v_baseline"], g["v_baseline"], default=1e-6, rel=True)
    if "vz_offset" in special:
        parity("vz_offset", fp["vz_offset"], g["vz_offset"], default=1e-5, rel=True)


@pytest.mark.parametrize("name", ["golden71x91_solverp", "golden71x91_dop_solverp", "hybrid_s0_dop_solverp"])
def test_solve_rp_matches_reference_run(name):
    """solve_rp=True (drt1d.py:568-606): the extra Rp-estimation QP, the data rescale and the DOP column rescale"""
    from hipdrt.models import DRT
    g, special = load_case(name)
    dop = "x_dop" in special
    drt = DRT(fit_dop=dop, warn=False)
    if "times" in g:
        fp = drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"], solve_rp=True)
    else:
        fp = drt.fit_eis(g["freq"], g["z"], solve_rp=True)
    assert drt._prep["rp_qp_iterations"] == int(g["qp_iterations"][0])
    np.testing.assert_allclose(drt.coefficient_scale, g["coefficient_scale"], rtol=1e-8)
    # the recorded QP list of the reference starts with the Rp QP
    g2 = {k: g[k] for k in g.files}
    g2["qp_iterations"] = g["qp_iterations"][1:]
    _check_fit(drt, g2, special, dop, data_rtol=1e-8, mat_rtol=1e-8)
    if "times" in g:
        parity("v_baseline", fp["v_baseline"], g["v_baseline"], default=1e-7, rel=True)
        np.testing.assert_allclose(drt.response_signal_scale, g["response_signal_scale"], rtol=1e-8)


def test_prepared_plan_argument_validation():
    """error behaviour of the prepared-plan entry points: bad layouts are refused with HIPDRT_E_INVALID and a message"""
    from hipdrt import _ffi as ffi
    ctx = ffi.get_context()
    n, m = 12, 20
    eye = np.eye(n)
    vmm = np.full((m, m), 1.0 / m)

    def desc(**kw):
        d = ffi.PreparedDesc()
        d.m, d.n, d.ns = m, n, 4
        d.vz_index = -1
        for k, v in kw.items():
            setattr(d, k, v)
        return d
    with pytest.raises(ffi.HipDrtError, match="x_dop block"):
        ffi.PreparedPlan(ctx, desc(dop_start=3, dop_size=3), [eye] * 3, vmm, np.zeros(n), np.zeros(n))
    with pytest.raises(ffi.HipDrtError, match="vz_offset"):
        ffi.PreparedPlan(ctx, desc(vz_index=1), [eye] * 3, vmm, np.zeros(n), np.zeros(n))      # no strength vector
    with pytest.raises(ffi.HipDrtError, match="larger than the DRT block"):
        ffi.PreparedPlan(ctx, desc(ns=10, dop_start=0, dop_size=10), [eye] * 3, vmm, np.zeros(n), np.zeros(n))
    plan = ffi.PreparedPlan(ctx, desc(vz_index=1, vb_start=0, vb_size=1, num_chrono=8), [eye] * 3, vmm, np.zeros(n),
                            np.zeros(n), vz_strength=np.ones(m), capacity=2)
    rng = np.random.default_rng(0)
    with pytest.raises(ffi.HipDrtError, match="per measurement"):
        plan.upload(rng.standard_normal((m, n)), rng.standard_normal((2, m)))                    # shared matrix + vz column
    with pytest.raises(ffi.HipDrtError, match="capacity"):
        plan.upload(rng.standard_normal((3, m, n)), rng.standard_normal((3, m)))
    with pytest.raises(ffi.HipDrtError, match="prepared plans take"):
        ffi.Plan.upload(plan, np.zeros((1, 5), dtype=complex))
    plan.upload(rng.standard_normal((2, m, n)), rng.standard_normal((2, m)))
    plan.fit()                                                                                   # tiny random problem runs
    out = plan.download()
    assert np.all(np.isfinite(out["x"])) and np.all(out["outer_iters"] >= 1)


def test_fit_capacitance_matches_reference_run():
    """fit_capacitance=True: the C_inv column (1/(j omega) in the impedance block, the integrated current in the chrono
    block; drt1d.py:5803, 5838, 5888, mat1d.py:423-451) in an EIS fit and in a joint fit of a cell with a series capacitor"""
    from hipdrt.models import DRT
    g, special = load_case("eis_cap")
    drt = DRT(fit_capacitance=True, warn=False)
    drt.fit_eis(g["freq"], g["z"])
    _check_fit(drt, g, special, False)
    g, special = load_case("hybrid_cap")
    drt = DRT(fit_capacitance=True, warn=False)
    fp = drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"])
    _check_fit(drt, g, special, False)
    assert abs(fp["C_inv"] - 1 / 20.0) < 0.01          # the synthetic cell's 20 F series capacitor


def test_polynomial_baseline_matches_reference_run():
    """v_baseline_deg=1, v_baseline_sqrt=True with one penalty per coefficient: three baseline columns, all excluded from
    the vz_offset prediction; extract_qphb_parameters' per-column scaling"""
    from hipdrt.models import DRT
    g, special = load_case("hybrid_vb")
    assert special["v_baseline"]["size"] == 3
    drt = DRT(warn=False)
    fp = drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"], v_baseline_deg=1,
                        v_baseline_sqrt=True, v_baseline_penalty=[1e-6, 1e-4, 1e-5])
    _check_fit(drt, g, special, False)
    np.testing.assert_allclose(drt.v_baseline_scale, g["v_baseline_scale"], rtol=1e-14)
    parity("v_baseline", fp["v_baseline"], g["v_baseline"], default=1e-9)
    with pytest.raises(ValueError):
        drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"], v_baseline_deg=1,
                       v_baseline_penalty=[1e-6, 1e-4, 1e-5])


def test_weight_factors_match_reference_runs():
    """weight_factor on an EIS fit (device plan path), explicit chrono / EIS factors and the 'rp' rule on hybrid fits"""
    from hipdrt.models import DRT
    g, special = load_case("golden71x91_wf")
    drt = DRT(warn=False)
    fp = drt.fit_eis(g["freq"], g["z"], weight_factor=0.7)
    assert drt.qphb_params["qp_iterations"].tolist() == g["qp_iterations"].tolist()
    parity("x", fp["x"], g["x"], default=1e-7)
    parity("true_weights", drt.qphb_params["true_weights"], g["weights"], default=1e-6, rel=True)
    parity("q_vector", fp["q_vector"], g["q_vector"], default=1e-8)
    parity("p_matrix", fp["p_matrix"], g["p_matrix"], default=1e-8)
    parity("z_sigma_tot", fp["z_sigma_tot"], g["z_sigma_tot"], default=1e-6, rel=True)
    drt.fit_eis(g["freq"], g["z"])                       # the factor does not stick to the plan
    assert drt.qphb_params["outer_iterations"] == 6 and drt.qphb_params["qp_iterations"].tolist() == [6, 2, 3, 2, 2, 2, 2]
    for name, kw in (("hybrid_s0_wf", dict(weight_factor=1.5, eis_weight_factor=2.0, chrono_weight_factor=0.5)),
                     ("hybrid_s0_wfrp", dict(hybrid_weight_factor_method='rp'))):
        g, special = load_case(name)
        drt = DRT(warn=False)
        fp = drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"], **kw)
        np.testing.assert_allclose(drt.qphb_params["eis_weight_factor"], g["eis_weight_factor"], rtol=1e-12)
        np.testing.assert_allclose(drt.qphb_params["chrono_weight_factor"], g["chrono_weight_factor"], rtol=1e-12)
        _check_fit(drt, g, special, False)
        parity("true_weights_2", drt.qphb_params["true_weights"], g["weights"], default=1e-6, rel=True)
        parity("weights", drt.qphb_params["weights"], g["scaled_weights"], default=1e-6, rel=True)
    with pytest.raises(ValueError):
        drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"], hybrid_weight_factor_method='nope')


def test_outlier_p_in_a_joint_fit_matches_reference_run():
    """outlier_p on a prepared plan: self-excluded variance matrix, two initial QPs, outlier-aware weights each iteration"""
    from hipdrt.models import DRT
    g, special = load_case("hybrid_s0_outlier")
    drt = DRT(warn=False)
    fp = drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"], outlier_p=0.05)
    qp = drt.qphb_params
    # history of the plan starts at the last initial QP
    assert qp["qp_iterations"].tolist() == g["qp_iterations"].tolist()[1:]
    assert qp["outer_iterations"] == int(g["outer_iterations"])
    parity_close("hybrid_outlier.x_scaled", drt.cvx_result["x"], g["x_scaled"], 1e-9)        # measured 2.2e-11
    parity("true_weights", qp["true_weights"], g["weights"], default=1e-5, rel=True)
    parity_close("hybrid_outlier.x", fp["x"], g["x"], 1e-9)                                   # measured 1.4e-11


@pytest.mark.parametrize("name", ["eis_rmout", "hybrid_rmout"])
def test_remove_outliers_matches_reference_run(name):
    """remove_outliers=True: the initialize_weights-only detection pass flags the same points as the reference, the fit on
    the cleaned data follows the same trajectory"""
    from hipdrt.models import DRT
    g, special = load_case(name)
    drt = DRT(warn=False)
    if "times" in g:
        drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"], remove_outliers=True, outlier_p=0.05)
        np.testing.assert_array_equal(drt.chrono_outlier_index, g["chrono_outlier_index"])
        n_pass1 = 2
    else:
        drt.fit_eis(g["freq"], g["z"], remove_outliers=True, outlier_p=0.05)
        n_pass1 = 2
    np.testing.assert_array_equal(drt.eis_outlier_index, g["eis_outlier_index"])
    assert np.array_equal(np.where(g["eis_outlier_index"])[0], [10, 25])            # the two corrupted impedance points
    g2 = {k: g[k] for k in g.files}
    g2["qp_iterations"] = g["qp_iterations"][n_pass1:]                              # the detection pass' two QPs come first
    _check_fit(drt, g2, special, False)
    with pytest.raises(ValueError):
        drt.fit_eis(g["freq"], g["z"], remove_outliers=True)


def test_remove_extremes_matches_reference_run():
    """remove_extremes=True: quantile-range pre-filter of the raw data (preprocessing.py:844-857), then the normal fit"""
    from hipdrt.models import DRT
    from hipdrt import preprocessing as pp
    g, special = load_case("eis_rmext")
    flag = pp.identify_extreme_values(g["z"].real, 0.8, 1.5) | pp.identify_extreme_values(g["z"].imag, 0.8, 1.5)
    assert np.sum(flag) == len(g["z"]) - g["rm"].shape[0] // 2 and flag[10] and flag[40]
    drt = DRT(warn=False)
    drt.fit_eis(g["freq"], g["z"], remove_extremes=True)
    _check_fit(drt, g, special, False)


def test_update_scale_matches_reference_runs():
    """update_scale=True on the EIS plan path and on a prepared (hybrid + DOP) plan: the rescale runs inside the hyper kernel"""
    from hipdrt.models import DRT
    g, special = load_case("golden71x91_upscale")
    drt = DRT(warn=False)
    fp = drt.fit_eis(g["freq"], g["z"], update_scale=True)
    assert drt.qphb_params["qp_iterations"].tolist() == g["qp_iterations"].tolist()
    np.testing.assert_allclose(drt.coefficient_scale, g["coefficient_scale"], rtol=1e-8)
    parity("x", fp["x"], g["x"], default=1e-7)
    parity("R_inf", fp["R_inf"], g["R_inf"], default=1e-6, rel=True)
    parity("z_sigma_tot", fp["z_sigma_tot"], g["z_sigma_tot"], default=1e-6, rel=True)
    parity("rv", drt.qphb_params["rv"], g["rv"], default=1e-7, rel=True)
    parity("est_weights", drt.qphb_params["est_weights"], g["est_weights"], default=1e-6, rel=True)
    parity("q_vector", fp["q_vector"], g["q_vector"], default=1e-8)
    parity("p_matrix", fp["p_matrix"], g["p_matrix"], default=1e-8)
    g, special = load_case("hybrid_s0_dop_upscale")
    drt = DRT(fit_dop=True, warn=False)
    fp = drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"], update_scale=True)
    np.testing.assert_allclose(drt.coefficient_scale, g["coefficient_scale"], rtol=1e-8)
    np.testing.assert_allclose(drt.response_signal_scale, g["response_signal_scale"], rtol=1e-8)
    _check_fit(drt, g, special, True, data_rtol=1e-7, mat_rtol=1e-8)
    parity("v_baseline", fp["v_baseline"], g["v_baseline"], default=1e-6, rel=True)
    parity("xmx_norms", drt.qphb_params["xmx_norms"], g["xmx_norms"], default=1e-6, rel=True)


def test_eff_hp_false_and_negative_window_match_reference_runs():
    from hipdrt.models import DRT
    g, special = load_case("golden71x91_noeff")
    drt = DRT(warn=False)
    fp = drt.fit_eis(g["freq"], g["z"], eff_hp=False)           # EIS plan path, Toeplitz branch of the hyper kernel
    qp = drt.qphb_params
    assert qp["qp_iterations"].tolist() == g["qp_iterations"].tolist() and qp["outer_iterations"] == 24
    parity("x", fp["x"], g["x"], default=1e-7)
    parity("rho_vector", qp["rho_vector"], g["rho_vector"], default=1e-6, rel=True)
    parity("s_vectors", np.array(qp["s_vectors"]), g["s_vectors"], default=1e-5, rel=True)
    parity("p_matrix", fp["p_matrix"], g["p_matrix"], default=1e-8)
    assert drt.fit_kwargs["s_alpha"].tolist() == [1.05, 1.15, 2.5]
    g, special = load_case("golden71x91_dop_noeff")             # dop_rho_k in the DOP block's solve_s
    drt = DRT(fit_dop=True, warn=False)
    drt.fit_eis(g["freq"], g["z"], eff_hp=False)
    _check_fit(drt, g, special, True)
    g, special = load_case("golden71x91_negwin")
    drt = DRT(warn=False)
    fp = drt.fit_eis(g["freq"], g["z"], nonneg=False, neg_allowed_tau_range=(1e-5, 1e-3))
    _check_fit(drt, g, special, False)
    inside = (g["basis_tau"] >= 1e-5) & (g["basis_tau"] <= 1e-3)
    assert fp["x"][~inside].min() >= -1e-12
    with pytest.raises(ValueError):
        drt.fit_eis(g["freq"], g["z"], neg_allowed_tau_range=(1e-5, 1e-3))


@pytest.mark.parametrize("name,kw", [
    ("hybrid_s0_iwsep", dict(init_weights_separately=True)),
    ("hybrid_s0_wfw", dict(init_weights_separately=True, hybrid_weight_factor_method='weight')),
])
def test_separate_initial_weights_and_weight_rule_match_reference_runs(name, kw):
    """both run inside hipdrt_plan_fit: two masked initial QPs with per-block variance floors, row factors from the
    initial weights' block scales"""
    from hipdrt.models import DRT
    g, special = load_case(name)
    drt = DRT(warn=False)
    drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"], **kw)
    g2 = {k: g[k] for k in g.files}
    g2["qp_iterations"] = g["qp_iterations"][1:]            # the plan's history starts at the last initial QP
    _check_fit(drt, g2, special, False)
    parity("est_weights", drt.qphb_params["est_weights"], g["est_weights"], default=1e-6, rel=True)
    if "weight" in str(kw.get("hybrid_weight_factor_method")):
        parity("eis_weight_factor", drt.qphb_params["eis_weight_factor"], g["eis_weight_factor"], default=1e-7, rel=True)
        parity("chrono_weight_factor", drt.qphb_params["chrono_weight_factor"], g["chrono_weight_factor"], default=1e-7, rel=True)
        parity("weights", drt.qphb_params["weights"], g["scaled_weights"], default=1e-6, rel=True)


def test_series_neg_matches_reference_run():
    """series_neg=True: 2 x ntau non-negative coefficients over [A, -A], block-diagonal penalties (non-Toeplitz branch of the
    hyper kernel)"""
    from hipdrt.models import DRT
    g, special = load_case("golden71x91_sneg")
    drt = DRT(warn=False)
    fp = drt.fit_eis(g["freq"], g["z"], series_neg=True)
    assert len(fp["x"]) == 2 * len(g["basis_tau"])
    _check_fit(drt, g, special, False)
    with pytest.raises(ValueError):
        drt.fit_eis(g["freq"], g["z"], series_neg=True, nonneg=False)


def test_discard_first_n_matches_reference_run():
    """discard_first_n=2: first two samples of every segment dropped, step time inferred from the shortened record"""
    from hipdrt.models import DRT
    from hipdrt import preprocessing as pp
    g, special = load_case("hybrid_s0_discard")
    keep, (t, i_, v_) = pp.discard_first_n_chrono(g["times"], g["i_signal"], g["v_signal"], 2)
    assert len(t) == int(g["num_chrono"]) == len(g["times"]) - 4 and keep[0] == 2
    drt = DRT(warn=False)
    fp = drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"], discard_first_n=2)
    np.testing.assert_allclose(drt.step_times, g["step_times"], rtol=1e-14)
    _check_fit(drt, g, special, False)
    parity("v_baseline", fp["v_baseline"], g["v_baseline"], default=1e-7, rel=True)


@pytest.mark.parametrize("seed", range(32))
def test_randomised_joint_fits_follow_the_oracle(seed):
    """randomised differential test of the prepared path: protocol (samples, steps, frequencies), cell, noise, DOP, series
    capacitance, baseline degree and vz options vary; the device loop must reproduce the oracle's loop on the same
    (device-built) matrices -- identical outer / IPM iteration counts over the first twelve outer iterations (the
    reference's iteration is not always contractive beyond that, see the config-5 test) and the iterates."""
    from hipdrt.models import DRT
    from hipdrt import synth
    rng = np.random.default_rng(1000 + seed)
    dop = bool(rng.integers(2))
    cap = bool(rng.integers(2)) and not dop
    steps = ((float(rng.uniform(0.5, 3.0)), float(-rng.uniform(0.5, 2.0) * 1e-3)),) if rng.integers(2) else ()
    meas = synth.hybrid_measurement(seed=seed, n_pre=int(rng.integers(8, 40)), n_post=int(rng.integers(50, 160)),
                                    nf=int(rng.integers(21, 52)), f_hi=10 ** rng.uniform(4, 5.5), f_lo=10 ** rng.uniform(0, 1.5),
                                    v_noise=10 ** rng.uniform(-6.5, -5), jitter=True, extra_steps=steps,
                                    c_series=float(rng.uniform(5, 50)) if cap else None, t_hi=float(rng.uniform(5, 50)))
    kw = dict(max_iter=12)
    if rng.integers(2):
        kw.update(vz_offset_eps=float(rng.uniform(0.5, 3)), vz_offset_scale=float(rng.uniform(0.3, 3)))
    if rng.integers(3) == 0:
        kw.update(vz_offset=False)
    if rng.integers(3) == 0:
        kw.update(v_baseline_deg=1)
    if rng.integers(3) == 0:
        kw.update(chrono_error_structure=None, chrono_vmm_epsilon=float(rng.uniform(1, 6)))
    drt = DRT(fit_dop=dop, fit_capacitance=cap, warn=False)
    drt.fit_hybrid(*meas, **kw)
    qp, special = drt.qphb_params, drt.special_qp_params
    rzm0 = qp["rm"].copy()
    vz = None
    if "vz_offset" in special:
        vi = special["vz_offset"]["index"]
        rzm0[:, vi] = 0
        vb = special["v_baseline"]
        vz = dict(index=vi, strength=qp["vz_strength_vec"], num_chrono=qp["num_chrono"],
                  vb=(vb["index"], vb["index"] + vb["size"]))
    hyp = orc.get_default_hypers()
    if dop:
        hyp.update(orc.get_default_dop_hypers())
    ref = orc.qphb_fit_prepared(rzm0, qp["rv"], [qp["penalty_matrices"][f"m{k}"] for k in range(3)], qp["vmm"], special, hyp,
                                vz=vz, max_iter=12)
    assert [l["iterations"] for l in ref["qp_log"]] == qp["qp_iterations"].tolist(), (seed, kw, dop, cap)
    hx = np.array([h["x"] for h in ref["history"]])
    dx = np.array([h["x"] for h in drt.qphb_history])
    assert hx.shape == dx.shape
    scale = np.abs(hx).max(axis=1, keepdims=True)
    # (start-point QPs are direct solves: cond * eps, test_randomized_fits_vs_oracle's docstring)
    parity_close("random_joint_fits.hist_x", dx, hx, 1e-7, scale=scale.max())         # measured 3.3e-9 over the 24 draws
    parity("true_weights", qp["true_weights"], ref["weights"], default=1e-5, rel=True)


@pytest.mark.parametrize("seed", range(16))
def test_randomised_option_combinations_follow_the_oracle(seed):
    """random combinations of the _qphb_fit_core options on an EIS fit (prepared path): solve_rp, update_scale,
    weight_factor, eff_hp, series_neg, outlier_p, DOP -- each pinned alone by a reference-run fixture, here together
    against the oracle's general loop, which applies them in the reference's order"""
    from hipdrt.models import DRT
    from hipdrt import synth
    rng = np.random.default_rng(2000 + seed)
    freq = np.logspace(rng.uniform(4.5, 6), rng.uniform(-1, 0.5), int(rng.integers(31, 72)))
    z = synth.zarc2_spectrum(freq, seed=seed, jitter=True)
    dop = bool(rng.integers(2))
    kw = dict(max_iter=10, solve_rp=True)            # solve_rp routes every draw through the prepared path
    if rng.integers(2):
        kw["update_scale"] = True
    if rng.integers(2):
        kw["weight_factor"] = float(rng.uniform(0.6, 1.6))
    if rng.integers(3) == 0:
        kw["eff_hp"] = False
    if rng.integers(3) == 0:
        kw["series_neg"] = True
    if rng.integers(3) == 0:
        kw["outlier_p"] = float(rng.uniform(0.01, 0.1))
    drt = DRT(fit_dop=dop, warn=False)
    drt.fit_eis(freq, z, **kw)
    qp, special, prep = drt.qphb_params, drt.special_qp_params, drt._prep
    hyp = dict(qp["hypers"])
    hyp["eff_hp"] = kw.get("eff_hp", True)
    # un-do the host-side rescales to get the loop's inputs: data vector before solve_rp / update_scale, DOP columns before
    # the solve_rp rescale
    cs0 = (z.real.max() - z.real.min()) / 14
    rzv0 = np.concatenate([z.real, z.imag]) / cs0
    rzm0 = qp["rm"].copy()
    if dop:
        a, b = prep["dop"]
        from hipdrt.matrices import phasance
        scale0 = phasance.phasor_scale_vector(drt.basis_nu, drt.basis_tau) / (np.sqrt(np.pi) / drt.nu_epsilon)
        rzm0[:, a:b] *= scale0 / prep["dop_scale_vector"]
    area = np.sqrt(np.pi) / drt.tau_epsilon
    ref = orc.qphb_fit_prepared(rzm0, rzv0, [qp["penalty_matrices"][f"m{k}"] for k in range(3)], qp["vmm"], special, hyp,
                                max_iter=10, solve_rp=dict(basis_area=area),
                                update_scale=dict(basis_area=area) if kw.get("update_scale") else None,
                                weight_factor=kw.get("weight_factor", 1))
    n_extra = 1 + (1 if kw.get("outlier_p") else 0)          # Rp QP (+ first outlier QP) precede the plan's history
    assert [l["iterations"] for l in ref["qp_log"]][n_extra:] == qp["qp_iterations"].tolist(), \
        (seed, kw, dop, [l["iterations"] for l in ref["qp_log"]][n_extra:], qp["qp_iterations"].tolist())
    assert prep["rp_qp_iterations"] == ref["qp_log"][0]["iterations"]
    hx = np.array([h["x"] for h in ref["history"]])
    dx = np.array([h["x"] for h in drt.qphb_history])
    assert hx.shape == dx.shape
    parity_close("random_option_eis_fits.hist_x", dx, hx, 1e-7, scale=np.abs(hx).max())       # measured 1.2e-9 over the 16 draws
    parity("true_weights", qp["true_weights"], ref["weights"], default=1e-5, rel=True)
    parity("drt_coefficient_scale", drt.coefficient_scale, cs0 / (ref["scale_factor"] * ref["data_scale"]), default=1e-7, rel=True)


@pytest.mark.parametrize("seed", range(24))
def test_randomised_joint_fits_with_option_combinations(seed):
    """random option combinations on joint fits: solve_rp, update_scale, weight factors (given / 'rp' / 'weight'),
    init_weights_separately, eff_hp, series_neg, outlier_p, DOP, series capacitance -- device loop vs the oracle's general
    loop from the same un-rescaled inputs"""
    from hipdrt.models import DRT
    from hipdrt import synth
    rng = np.random.default_rng(3000 + seed)
    dop = bool(rng.integers(2))
    cap = bool(rng.integers(3) == 0) and not dop
    meas = synth.hybrid_measurement(seed=seed, n_pre=int(rng.integers(8, 30)), n_post=int(rng.integers(50, 120)),
                                    nf=int(rng.integers(21, 45)), v_noise=10 ** rng.uniform(-6.5, -5), jitter=True,
                                    c_series=float(rng.uniform(5, 50)) if cap else None)
    kw = dict(max_iter=10)
    pick = lambda p: rng.random() < p
    if pick(0.5): kw["solve_rp"] = True
    if pick(0.5): kw["update_scale"] = True
    if pick(0.4): kw["weight_factor"] = float(rng.uniform(0.6, 1.6))
    mode = rng.integers(4)
    if mode == 1: kw.update(eis_weight_factor=float(rng.uniform(0.5, 2)), chrono_weight_factor=float(rng.uniform(0.5, 2)))
    if mode == 2: kw["hybrid_weight_factor_method"] = "rp"
    if mode == 3: kw["hybrid_weight_factor_method"] = "weight"
    if pick(0.4): kw["init_weights_separately"] = True
    if pick(0.3): kw["eff_hp"] = False
    if pick(0.25): kw["series_neg"] = True
    if pick(0.3) and not kw.get("init_weights_separately"): kw["outlier_p"] = float(rng.uniform(0.01, 0.1))
    drt = DRT(fit_dop=dop, fit_capacitance=cap, warn=False)
    drt.fit_hybrid(*meas, **kw)
    qp, special, prep = drt.qphb_params, drt.special_qp_params, drt._prep
    hyp = dict(qp["hypers"])
    hyp["eff_hp"] = kw.get("eff_hp", True)
    rzm0 = prep["rzm_initial"].copy()
    vi = special["vz_offset"]["index"]
    vb = special["v_baseline"]
    vz = dict(index=vi, strength=qp["vz_strength_vec"], num_chrono=qp["num_chrono"], vb=(vb["index"], vb["index"] + vb["size"]))
    nc, m = qp["num_chrono"], len(qp["rv"])
    area = np.sqrt(np.pi) / drt.tau_epsilon
    rows = None
    if mode in (1, 2):
        rows = np.concatenate([np.full(nc, qp["chrono_weight_factor"]), np.full(m - nc, qp["eis_weight_factor"])])
    ref = orc.qphb_fit_prepared(rzm0, prep["rzv_initial"], [qp["penalty_matrices"][f"m{k}"] for k in range(3)], qp["vmm"],
                                special, hyp, vz=vz, max_iter=10,
                                solve_rp=dict(basis_area=area) if kw.get("solve_rp") else None,
                                update_scale=dict(basis_area=area) if kw.get("update_scale") else None,
                                weight_factor=kw.get("weight_factor", 1), row_factors=rows,
                                init_separately=dict(num_chrono=nc) if kw.get("init_weights_separately") else None,
                                weight_method=dict(num_chrono=nc) if mode == 3 else None)
    n_extra = (1 if kw.get("solve_rp") else 0) + (1 if kw.get("outlier_p") or kw.get("init_weights_separately") else 0)
    assert [l["iterations"] for l in ref["qp_log"]][n_extra:] == qp["qp_iterations"].tolist(), \
        (seed, kw, dop, cap, [l["iterations"] for l in ref["qp_log"]][n_extra:], qp["qp_iterations"].tolist())
    hx = np.array([h["x"] for h in ref["history"]])
    dx = np.array([h["x"] for h in drt.qphb_history])
    assert hx.shape == dx.shape
    parity_close("random_option_fits.hist_x", dx, hx, 5e-7, scale=np.abs(hx).max())           # measured 2.4e-8 over the 24 draws
    parity("true_weights", qp["true_weights"], ref["weights"], default=1e-5, rel=True)
    parity_close("random_option_fits.rzm", qp["rm"], ref["rzm"], 1e-7)       # measured 3.9e-10 here, 1.6e-8 over 60 further seeds (tools/fuzz_parity.py)


def test_uniform_chrono_variance_shortcut_is_bit_identical():
    """chrono_vmm_uniform=1 reads one chrono row of the variance matrix instead of all of them: same bits everywhere"""
    from hipdrt import _ffi as ffi
    g, special = load_case("hybrid_s0_dop")
    hyp = dict(orc.get_default_hypers(), **orc.get_default_dop_hypers())
    rzm0, vz = initial_rzm_and_vz(g, special)
    nc = int(g["num_chrono"])
    assert np.all(g["vmm"][:nc, :nc] == g["vmm"][0, 0]) and np.all(g["vmm"][:nc, nc:] == 0)
    h = orc.make_h_constraint(rzm0.shape[1], special, True)
    l1 = g["l1_lambda_vector"]
    outs = []
    for flag in (0, 1):
        desc = make_desc(ffi, g, special, rzm0, vz, hyp)
        desc.chrono_vmm_uniform = flag
        plan = ffi.PreparedPlan(ffi.get_context(), desc, [g["m0"], g["m1"], g["m2"]], g["vmm"], h, l1,
                                vz_strength=vz["strength"], capacity=1)
        plan.upload(rzm0[None], g["rv"][None])
        plan.fit()
        outs.append(plan.download(s_vectors=True))
    for key in ("x", "weights", "rho", "s_vectors", "q_vector"):
        np.testing.assert_array_equal(outs[0][key], outs[1][key])
    assert outs[0]["outer_iters"][0] == outs[1]["outer_iters"][0] == int(g["outer_iterations"])


def test_parameter_variances_of_a_joint_fit():
    """diag(inv(P)) cs^2 for a prepared-plan fit against numpy on the downloaded P (estimate_param_cov, drt1d.py:4116-4138)"""
    from hipdrt.models import DRT
    g, _ = load_case("hybrid_s0_dop")
    drt = DRT(fit_dop=True, warn=False)
    fp = drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"])
    var, ok = drt.estimate_param_var_batch()
    assert ok[0]
    ref = np.diag(np.linalg.inv(fp["p_matrix"])) * drt.coefficient_scale ** 2
    parity("var_0", var[0], ref, default=1e-6, rel=True)


def test_config5_bench_workload_first_outer_iterations():
    """The bench's own configs[4] workload (2 uV of voltage noise) against the reference itself, for the
    outer iterations in which the reference is still reproducible: tests/golden/refrun_config5_2uV.npz (oracle/make_golden.py
    --only-config5-2uV) holds the first six (outer steps 0.10, 0.02, 0.03, 0.03, 0.03 of the peak; after that the loop
    wanders and any two implementations drift apart -- oracle/probe_c5.py).  Pinned: the interior-point iteration count of
    all seven QPs (coneqp on the group kernel: n = 1078, one problem), every iterate, the hyper-parameters."""
    from hipdrt.models import DRT
    from hipdrt import synth
    g = np.load(os.path.join(GOLDEN, "refrun_config5_2uV.npz"))
    K = int(g["K"])
    assert float(g["v_noise"]) == 2e-6
    meas = synth.hybrid_measurement(seed=0, n_pre=96, n_post=4000, nf=512)        # (v_noise defaults to 2e-6: bench.py's call)
    drt = DRT(fixed_basis_tau=np.logspace(-7, 3, 1024), fit_dop=True, warn=False)
    fp = drt.fit_hybrid(*meas, max_iter=K)
    qp = drt.qphb_params
    assert qp["rm"].shape == tuple(g["rm_shape"]) == (5120, 1078)
    assert qp["qp_iterations"].tolist() == g["qp_iterations"].tolist()
    dx = np.array([h["x"] for h in drt.qphb_history])
    assert dx.shape == g["hist_x"].shape == (K, 1078)
    dev_err = np.abs(dx - g["hist_x"]).max(axis=1) / np.abs(g["hist_x"]).max(axis=1)
    print("device vs reference per outer iteration (2 uV):", np.array2string(dev_err, precision=2))
    parity_close("config5_2uV.hist_x", dev_err, np.zeros_like(dev_err), 1e-7, scale=1.0)        # measured 7.4e-9
    parity("rho_vector", np.array([h["rho_vector"] for h in drt.qphb_history]), g["hist_rho"], default=1e-5, rel=True)
    parity("dop_rho_vector", np.array([h["dop_rho_vector"] for h in drt.qphb_history]), g["hist_dop_rho"], default=1e-5, rel=True)
    parity_close("config5_2uV.x", fp["x"], g["x"], 1e-7)                    # measured 5.1e-9
    parity_close("config5_2uV.x_dop", fp["x_dop"], g["x_dop"], 5e-7)        # measured 5.0e-8 (this loop is not contractive)


# ---- survey 8f rank 3 beyond EIS: warm restarts on prepared plans (drt1d.py:1270-1365), PFRT (2558-2715), candidates (1497-1632) --
def _warm(name):
    return np.load(os.path.join(GOLDEN, f"refrun_warm_{name}.npz"), allow_pickle=False)


@pytest.mark.parametrize("name", ["hybrid_s0", "hybrid_s0_dop", "chrono_s1", "hybrid_s0_outlier"])
def test_pfrt_on_chrono_and_joint_fits_vs_reference_run(name):
    """DRT.pfrt_fit_hybrid / pfrt_fit_chrono with DRTMD's eleven factors against the reference's own run: per-step iteration
    counts, every iterate of every step (the device loop re-entered ten times on the prepared plan: weight factors on every
    iteration's weights, the vz_offset column rewritten from the copy frozen at entry), the step log-likelihoods and the
    matrix after the last rewrite."""
    from hipdrt.models import DRT
    g, special = load_case(name)
    w = _warm(name)
    drt = DRT(fit_dop="x_dop" in special, warn=False)
    if name == "chrono_s1":
        pr = drt.pfrt_fit_chrono(g["times"], g["i_signal"], g["v_signal"], factors=w["pfrt_factors"])
    elif name.endswith("_outlier"):        # outlier_p in the first fit and in every warm restart of a prepared plan
        pr = drt.pfrt_fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"], factors=w["pfrt_factors"],
                                 outlier_p=0.05)
    else:
        pr = drt.pfrt_fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"], factors=w["pfrt_factors"])
    assert pr["step_iters"][:, 0].tolist() == w["pfrt_step_iters"].tolist()
    peak = np.abs(w["pfrt_hist_x"]).max()
    hx = np.array([h["x"] for h in drt.pfrt_history])
    # several steps stop at max_iter_per_step = 10 without converging: rounding is amplified from step to step in both
    parity("hist_x", hx, w["pfrt_hist_x"], default=1e-6, scale=peak)
    parity("step_x", pr["step_x"][:, 0], w["pfrt_step_x"], default=1e-6, scale=peak)
    parity("step_llh", pr["step_llh"][:, 0], w["pfrt_step_llh"], default=1e-7, rel=True)
    parity("hist_weights", np.array([h["weights"] for h in drt.pfrt_history]), w["pfrt_hist_weights"], default=1e-5, rel=True)
    rm = drt._plan.get("rzm")
    parity("final_rm", rm[0] if rm.ndim == 3 else rm, w["pfrt_final_rm"], default=1e-6)
    assert len(drt.qphb_history) == int(w["pfrt_init_len"])          # fit_parameters / qphb_history describe the first step


@pytest.mark.parametrize("name", ["hybrid_s0", "hybrid_s0_dop", "chrono_s1"])
def test_candidates_on_chrono_and_joint_fits_vs_reference_run(name):
    """generate_candidates' two passes on one fitted object, weights first (drt1d.py:1660-1664): three weight steps x 0.5 from the
    fit's x / rho / (scaled) weights, then two s_0 steps x 4 from the same baseline but with the s vectors and the rewritten
    matrix the weight steps left behind -- the reference's shallow copies, reproduced."""
    from hipdrt.models import DRT
    g, special = load_case(name)
    w = _warm(name)
    drt = DRT(fit_dop="x_dop" in special, warn=False)
    if name == "chrono_s1":
        drt.fit_chrono(g["times"], g["i_signal"], g["v_signal"])
    else:
        drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"])
    assert len(drt.qphb_history) == int(w["cand_base_len"])
    steps_w = drt.generate_candidates_weights(0.5, 3, history_of=0)
    steps_s = drt.generate_candidates_s0(4, 2, history_of=0)
    assert [int(r["outer_iters"][0]) for r in steps_w + steps_s] == w["cand_call_iters"].tolist()
    peak = np.abs(w["cand_w_x"]).max()
    parity("w_x", np.concatenate([r["history"]["x"] for r in steps_w]), w["cand_w_x"], default=1e-6, scale=peak)
    parity("w_weights", np.concatenate([r["history"]["weights"] for r in steps_w]), w["cand_w_weights"], default=1e-5, rel=True)
    parity("s0_x", np.concatenate([r["history"]["x"] for r in steps_s]), w["cand_s0_x"], default=1e-6, scale=peak)
    parity("s0_rho", np.concatenate([r["history"]["rho"] for r in steps_s]), w["cand_s0_rho"], default=1e-5, rel=True)
    parity("s0_s", steps_s[-1]["s_vectors"][0], w["cand_s0_s"][-1], default=1e-4, rel=True)


def test_pfrt_batch_of_joint_fits_matches_single_runs():
    """pfrt_fit_hybrid_batch: three joint measurements of one protocol through the eleven steps at once; every member's steps
    are those of its own single run (iteration counts equal, iterates to rounding: the batch and the single plan run the same
    kernels per measurement); a restart with outlier_p switched on runs."""
    from hipdrt import synth
    from hipdrt.models import DRT
    meas = [synth.hybrid_measurement(seed=s_, jitter=True) for s_ in (0, 1, 2)]
    drt = DRT(warn=False)
    pr = drt.pfrt_fit_hybrid_batch(meas[0][0], [m_[1] for m_ in meas], [m_[2] for m_ in meas], meas[0][3], [m_[4] for m_ in meas])
    assert pr["step_x"].shape[:2] == (11, 3) and np.isfinite(pr["step_llh"]).all()
    one = DRT(warn=False)
    p1 = one.pfrt_fit_hybrid(*meas[1])
    assert pr["step_iters"][:, 1].tolist() == p1["step_iters"][:, 0].tolist()
    parity("step_x", pr["step_x"][:, 1], p1["step_x"][:, 0], default=1e-9)
    res = one.continue_from_init(outlier_p=0.05, max_iter=3)          # (built in round 6: any plan, any outlier_p)
    assert np.isfinite(res["x"]).all() and (res["status"] >= 0).all()


def test_full_covariance_of_a_joint_fit_with_dop():
    """estimate_param_cov on a joint chrono + EIS fit with a distribution of phasances: the device loop of a prepared plan runs
    at unit scale, so coefficient_scale^2 and the DOP rescaling (drt1d.py:4126-4131) are applied by the host layer -- against
    the reference's formula on the reference run's own P (refrun_hybrid_s0_dop) and on the device's P."""
    from hipdrt.models import DRT
    g, special = load_case("hybrid_s0_dop")
    drt = DRT(fit_dop=True)
    drt.fit_hybrid(g["times"], g["i_signal"], g["v_signal"], g["freq"], g["z"])
    a = special["x_dop"]["index"]
    e = a + special["x_dop"]["size"]

    def reference_formula(P):
        inv = np.linalg.inv(P)
        inv[:, a:e] *= g["dop_scale_vector"][None, :]
        inv[a:e, :] *= g["dop_scale_vector"][:, None]
        return inv * float(g["coefficient_scale"]) ** 2

    cov = drt.estimate_param_cov()
    parity("param_cov_vs_own_P", cov, reference_formula(drt.fit_parameters["p_matrix"]), default=1e-8)
    parity("param_cov_vs_reference_P", cov, reference_formula(g["p_matrix"]), default=1e-5)
    dcov = drt.estimate_distribution_cov(ppd=10)
    var, _ = drt.estimate_distribution_var_batch(ppd=10)
    parity("dist_cov_diag", np.diag(dcov), var[0], default=1e-9, rel=True, floor=1e-9)


def test_posterior_entry_points_after_a_warm_restart_with_row_factors():
    """hipdrt_plan_continue on a prepared plan with chrono / eis row factors used to scale the weights in
    place and leave w_eff -- what hipdrt_plan_get_p_matrix / _param_cov / _param_var build P from -- as the FIRST fit's scaled
    weights (or uninitialised when the factors came with the restart).  Now the restart ends by refreshing it: P after a
    restart is calculate_pq's construction on the restart's final state (last re-estimated weights x the factors its QPs saw,
    last s / rho, the rewritten matrix), q_vector to match.  Checked against numpy on the downloaded state, with factors that
    were set by the fit AND with factors first introduced by the restart."""
    from hipdrt.models import DRT
    from hipdrt import synth
    meas = synth.hybrid_measurement(seed=0)
    for fit_kw, cont_kw in ((dict(eis_weight_factor=2.0, chrono_weight_factor=0.5), dict()),
                            (dict(), dict(eis_weight_factor=1.7, chrono_weight_factor=0.6))):
        drt = DRT(warn=False)
        drt.fit_hybrid(*meas, **fit_kw)
        p_fit = drt.fit_parameters["p_matrix"].copy()
        res = drt._continue_prepared(max_iter=3, weight_factor=1.3, **cont_kw)
        plan, prep = drt._plan, drt._prep
        nc, m, n = prep["num_chrono"], prep["m"], plan.n
        cf = cont_kw.get("chrono_weight_factor", prep["chrono_weight_factor"])
        ef = cont_kw.get("eis_weight_factor", prep["chrono_weight_factor"])      # (upstream's fallback: the CHRONO factor for both)
        rows = np.concatenate([np.full(nc, cf), np.full(m - nc, ef)])
        w_eff = res["weights"][0] * rows * 1.3
        rm = plan.get("rzm")
        rm = rm[0] if rm.ndim == 3 else rm
        hyp = orc.get_default_hypers()
        pen = [drt.qphb_params["penalty_matrices"][f"m{k}"] for k in range(3)]
        ns = n - len(drt.basis_tau)
        l2 = orc.calculate_qp_l2_matrix(hyp, res["rho"][0], pen, list(res["s_vectors"][0]), ns)
        wrm = w_eff[:, None] * rm
        p_ref = wrm.T @ wrm + l2
        pm = plan.p_matrix(0)
        assert np.abs(pm - p_ref).max() <= 1e-10 * np.abs(p_ref).max()
        assert np.abs(pm - p_fit).max() > 1e-3 * np.abs(p_fit).max()            # (not the first fit's matrix any more)
        q_ref = -wrm.T @ (w_eff * prep["rzv"] if "rzv" in prep else w_eff * drt.qphb_params["rv"])
        assert np.abs(res["q_vector"][0] - q_ref).max() <= 1e-9 * np.abs(q_ref).max()
        cov, ok = plan.param_cov(0)
        assert ok and np.abs(cov - np.linalg.inv(p_ref)).max() <= 1e-7 * np.abs(np.linalg.inv(p_ref)).max()


def test_protocol_matrices_are_kept_between_calls_and_rebuilt_when_the_sampling_changes():
    """the penalty / variance / impedance blocks of a protocol are built once per DRT object (upstream: the _recalc flags): a second
    fit of the same sampling reuses the SAME arrays and gives the same bits as a fresh object; another frequency grid rebuilds them"""
    from hipdrt import synth
    from hipdrt.models import DRT
    m = synth.hybrid_measurement(seed=11, n_post=80, nf=25)
    m2 = synth.hybrid_measurement(seed=12, n_post=80, nf=25)
    d = DRT(warn=False)
    d.fit_hybrid(*m)
    vmm1, pen1 = d.qphb_params["vmm"], d.qphb_params["penalty_matrices"]["m1"]
    d.fit_hybrid(*m2)
    assert d.qphb_params["vmm"] is vmm1                                  # kept
    fresh = DRT(warn=False)
    fresh.fit_hybrid(*m2)
    np.testing.assert_array_equal(d.fit_parameters["x"], fresh.fit_parameters["x"])
    np.testing.assert_array_equal(d.qphb_params["vmm"], fresh.qphb_params["vmm"])
    np.testing.assert_array_equal(d.qphb_params["penalty_matrices"]["m1"], pen1)
    m3 = synth.hybrid_measurement(seed=12, n_post=80, nf=31)             # another frequency grid: everything is rebuilt
    d.fit_hybrid(*m3)
    assert d.qphb_params["vmm"] is not vmm1 and d.qphb_params["vmm"].shape != vmm1.shape
    fresh3 = DRT(warn=False)
    fresh3.fit_hybrid(*m3)
    np.testing.assert_array_equal(d.fit_parameters["x"], fresh3.fit_parameters["x"])
    assert len(d._build_memo_kept) <= 6                                  # one protocol's builds, not a growing cache
