"""Pin the oracle (CPU restatement, oracle/) against (1) the reference's own known-answer test vectors and
(2) outputs of the reference itself captured in the build container (tests/golden/refrun_*.npz).
CPU only."""
import os

import numpy as np
import pytest

from oracle import drt_oracle as orc
from oracle.coneqp import coneqp_boxlow

from conftest import GOLDEN


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def test_reference_known_answer_test():
    """Mirror of /root/reference/tests/test_drt_fit.py: same inputs, same np.allclose criterion."""
    g = load("ref_test_drt_fit_eis.npz")
    drt = orc.OracleDRT(fit_inductance=True, fit_ohmic=True)
    hypers = dict(rp_scale=14, derivative_weights=np.array([1.5, 1.0, 0.5]), sigma_ds=np.array([1, 1000, 1000]),
                  l1_lambda_0=0, l2_lambda_0=142, s_alpha=np.array([5, 10, 25]),
                  rho_alpha=np.array([0.15, 0.2, 0.25]), iw_alpha=None, iw_beta=None, s_0=np.ones(3),
                  rho_0=np.ones(3), outlier_p=None)
    fp = drt.fit_eis(g["freq"], g["z"], **hypers)
    for key in ("x", "R_inf", "inductance", "C_inv", "z_sigma_tot", "vz_offset_eps", "q_vector"):
        assert np.allclose(g[key], fp[key]), key
    assert fp["v_sigma_tot"] is None and fp["v_sigma_res"] is None


def test_lookup_and_matrices_vs_reference_run():
    g = load("refrun_golden71x91.npz")
    eps = float(g["tau_epsilon"])
    assert eps == orc.get_epsilon_from_ppd(10)
    (lre, zre), (lim, zim) = orc.generate_impedance_lookup(eps)
    for a, b in ((lre, g["lut_log_wt_re"]), (zre, g["lut_z_re"]), (lim, g["lut_log_wt_im"]), (zim, g["lut_z_im"])):
        np.testing.assert_array_equal(a, b)
    tau = orc.get_basis_tau(g["freq"])
    np.testing.assert_array_equal(tau, g["basis_tau"])
    assert orc.impedance_matrix_is_toeplitz(g["freq"], tau)
    zr = orc.construct_impedance_matrix(g["freq"], "real", tau, eps, "interp", interpolate_grids=(lre, zre))
    zi = orc.construct_impedance_matrix(g["freq"], "imag", tau, eps, "interp", interpolate_grids=(lim, zim))
    np.testing.assert_array_equal(zr, g["zm_re"])
    np.testing.assert_array_equal(zi, g["zm_im"])
    for k in range(3):
        m = orc.construct_integrated_derivative_matrix(np.log(tau), k, eps)
        np.testing.assert_array_equal(m, g[f"m{k}"])
    np.testing.assert_array_equal(orc.construct_eis_var_matrix(g["freq"]), g["vmm"])


def test_trapz_matrices_vs_reference_run():
    for name in ("refrun_trapz_32x64.npz", "refrun_trapz_71x91_toeplitz.npz"):
        g = load(name)
        for part in ("real", "imag"):
            a = orc.construct_impedance_matrix(g["freq"], part, g["tau"], float(g["eps"]), "trapz")
            np.testing.assert_array_equal(a, g[f"A_{part}"])


@pytest.mark.parametrize("name", ["refrun_golden71x91.npz", "refrun_golden71x91_neg.npz", "refrun_c1_71x121.npz"])
def test_every_qp_vs_reference_run(name):
    g = load(name)
    for i in range(len(g["qp_iterations"])):
        res = coneqp_boxlow(g[f"qp{i}_P"], g[f"qp{i}_q"], g[f"qp{i}_h"])
        assert res["iterations"] == g["qp_iterations"][i]
        np.testing.assert_array_equal(res["x"], g[f"qp{i}_x"])


def _fit(g, structure="fast"):
    fixed = None
    if len(g["basis_tau"]) != len(orc.get_basis_tau(g["freq"])):
        fixed = g["basis_tau"]
    drt = orc.OracleDRT(fixed_basis_tau=fixed)
    drt.fit_eis(g["freq"], g["z"], nonneg=bool(g["nonneg"]), keep_history=True, structure=structure)
    return drt


@pytest.mark.parametrize("name", ["refrun_golden71x91.npz", "refrun_golden71x91_neg.npz", "refrun_c1_71x121.npz",
                                  "refrun_c2_256x512_s2.npz", "refrun_c3_member2.npz"])
def test_full_fit_trajectory_vs_reference_run(name):
    g = load(name)
    drt = _fit(g)
    assert [q["iterations"] for q in drt.qp_log] == g["qp_iterations"].tolist()
    assert len(drt.qphb_history) == int(g["outer_iterations"])
    tol = dict(rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(np.array([h["x"] for h in drt.qphb_history]), g["hist_x"], **tol)
    np.testing.assert_allclose(np.array([h["rho_vector"] for h in drt.qphb_history]), g["hist_rho"], **tol)
    np.testing.assert_allclose(np.array([h["weights"] for h in drt.qphb_history]), g["hist_weights"], **tol)
    fp, qp = drt.fit_parameters, drt.qphb_params
    for key in ("x", "R_inf", "inductance", "z_sigma_tot", "q_vector"):
        np.testing.assert_allclose(fp[key], g[key], **tol)
    np.testing.assert_allclose(qp["est_weights"], g["est_weights"], **tol)
    np.testing.assert_allclose(np.array(qp["s_vectors"]), g["s_vectors"], **tol)
    np.testing.assert_allclose(qp["xmx_norms"], g["xmx_norms"], **tol)
    if "p_matrix" in g:
        np.testing.assert_allclose(fp["p_matrix"], g["p_matrix"], rtol=1e-7, atol=1e-9)


def test_reference_structure_equals_fast():
    g = load("refrun_golden71x91.npz")
    a = _fit(g, "fast").fit_parameters
    b = _fit(g, "reference").fit_parameters
    np.testing.assert_allclose(a["x"], b["x"], rtol=1e-10, atol=1e-14)


def test_oracle_response_path_matches_reference_run():
    """survey row a3: response lookup and response matrices (interp with 1 / 3 steps / a step after the last sample,
    trapz) against the reference's own outputs -- bit-exact, same numpy calls."""
    from oracle import drt_oracle as orc
    g = np.load(os.path.join(GOLDEN, "refrun_response.npz"))
    for tag in ("eps_grid", "eps_4p34"):
        lg, rg = orc.generate_response_lookup(float(g[f"lookup_{tag}_eps"]))
        np.testing.assert_array_equal(lg, g[f"lookup_{tag}_log_td"])
        np.testing.assert_array_equal(rg, g[f"lookup_{tag}_v"])
    grids = (g["lookup_eps_grid_log_td"], g["lookup_eps_grid_v"])
    for case in ("one_step", "three_steps", "step_after_end"):
        a, lay = orc.construct_response_matrix(g["tau"], g["times"], g[f"{case}_step_times"], g[f"{case}_step_sizes"],
                                               float(g["epsilon"]), 'interp', interpolate_grids=grids)
        np.testing.assert_array_equal(a, g[f"{case}_A"])
        np.testing.assert_array_equal(lay, g[f"{case}_layered"])
    a, lay = orc.construct_response_matrix(g["trapz_tau"], g["trapz_times"], g["trapz_step_times"],
                                           g["trapz_step_sizes"], float(g["trapz_epsilon"]), 'trapz')
    np.testing.assert_array_equal(a, g["trapz_A"])
    np.testing.assert_array_equal(lay, g["trapz_layered"])
    # distribution-of-phasances matrices (row a18)
    eps = float(g["dop_epsilon"])
    np.testing.assert_array_equal(orc.construct_phasor_z_matrix(g["dop_freq"], g["dop_nu"], eps), g["dop_zm"])
    vm, vl = orc.construct_phasor_v_matrix(g["times"], g["dop_nu"], eps, g["three_steps_step_times"], g["three_steps_step_sizes"])
    np.testing.assert_array_equal(vm, g["dop_vm"])
    np.testing.assert_array_equal(vl, g["dop_vm_layered"])
    np.testing.assert_array_equal(orc.phasor_scale_vector(g["dop_nu"], g["tau"]), g["dop_scale"])
    # chrono variance-estimation matrices (row a5)
    for case in ("one_step", "three_steps"):
        np.testing.assert_array_equal(orc.construct_chrono_var_matrix(g["times"], g[f"{case}_step_times"], 0.25),
                                      g[f"{case}_vmm"])
    np.testing.assert_array_equal(orc.construct_chrono_var_matrix(g["times"], g["one_step_step_times"], 0.25, 'uniform'),
                                  g["uniform_vmm"])


def test_oracle_posterior_quantities_match_reference_run():
    """survey 8f rank 1 (what DRTMD stores per observation, drtmd.py:258-279): distribution variance on a supergrid
    (with / without extend_var), parameter variance, llh and rss -- bit-exact against the reference's own outputs."""
    g = np.load(os.path.join(GOLDEN, "refrun_posterior_golden71x91.npz"))
    P, ns, cs = g["p_matrix"], int(g["num_special"]), float(g["coefficient_scale"])
    eps = float(g["tau_epsilon"])
    v = orc.estimate_distribution_var(P, g["basis_tau"], g["tau_eval"], eps, ns, cs)
    np.testing.assert_array_equal(v, g["dist_var"])
    ve = orc.estimate_distribution_var(P, g["basis_tau"], g["tau_eval"], eps, ns, cs, True, g["freq"])
    np.testing.assert_array_equal(ve, g["dist_var_ext"])
    np.testing.assert_array_equal(np.diag(orc.estimate_param_cov(P, cs)), g["param_var"])
    assert orc.evaluate_rss(g["x_scaled"], g["rm"], g["rv"], g["est_weights"]) == pytest.approx(float(g["rss"]), rel=1e-13)
    assert orc.evaluate_llh(g["x_scaled"], g["rm"], g["rv"], g["est_weights"]) == pytest.approx(float(g["llh"]), rel=1e-13)


def test_oracle_warm_restarts_match_reference_run():
    """survey 8f rank 3: _continue_from_init through the reference's two candidate generators after its known-answer
    fit -- same number of iterations in every step, every intermediate x / rho / weights (the oracle's own fit differs
    from the reference run by ~1e-12, hence not bit-exact)."""
    g = np.load(os.path.join(GOLDEN, "refrun_candidates_golden71x91.npz"))
    d = orc.OracleDRT()
    d.fit_eis(g["freq"], g["z"])
    hs, cs = d.generate_candidates_s0(4, 2)
    hw, cw = d.generate_candidates_weights(0.5, 3)
    assert cs == [9, 10] and cw == [4, 4, 4]
    for tag, h in (("s0", hs), ("w", hw)):
        np.testing.assert_allclose(np.array([e["x"] for e in h]), g[f"{tag}_x"], rtol=0, atol=1e-10 * np.abs(g[f"{tag}_x"]).max())
        np.testing.assert_allclose(np.array([e["rho_vector"] for e in h]), g[f"{tag}_rho"], rtol=1e-9)
        np.testing.assert_allclose(np.array([e["weights"] for e in h]), g[f"{tag}_weights"], rtol=1e-10)


def test_oracle_pfrt_matches_reference_run():
    g = np.load(os.path.join(GOLDEN, "refrun_candidates_golden71x91.npz"))
    d = orc.OracleDRT()
    r = d.pfrt_fit_eis(g["freq"], g["z"])
    assert sum(r["counts"]) == int(g["pfrt_history_len"]) and r["counts"][0] == int(g["pfrt_init_len"])
    np.testing.assert_allclose(r["step_x"], g["pfrt_step_x"], rtol=0, atol=1e-10 * np.abs(g["pfrt_step_x"]).max())
    np.testing.assert_allclose(r["step_llh"], g["pfrt_step_llh"], rtol=1e-9)


@pytest.mark.parametrize("name,kw", [("outlier", dict(outlier_p=0.05)), ("iw", dict(iw_alpha=1.5, iw_beta=0.5))])
def test_oracle_optional_weight_branches_match_reference_run(name, kw):
    g = np.load(os.path.join(GOLDEN, f"refrun_golden71x91_{name}.npz"))
    d = orc.OracleDRT()
    d.fit_eis(g["freq"], g["z"], keep_history=True, **kw)
    assert [l["iterations"] for l in d.qp_log] == list(g["qp_iterations"])
    np.testing.assert_array_equal(d.qphb_params["est_weights"], g["est_weights"])
    np.testing.assert_allclose(d.qphb_params["x_scaled"], g["x_scaled"], rtol=0, atol=1e-11 * np.abs(g["x_scaled"]).max())


# ---- config-5 family: distribution of phasances inside the loop, joint chrono + EIS fits ----------------------------
@pytest.mark.parametrize("name", ["golden71x91_dop", "hybrid_s0", "hybrid_s0_dop", "chrono_s1", "hybrid_3step",
                                  "hybrid_3step_opts", "eis_cap", "hybrid_cap", "hybrid_vb", "golden71x91_sneg"])
def test_general_loop_reproduces_reference_trajectory(name):
    """oracle.qphb_fit_prepared (the loop of drt1d.py:551-1006 with the DOP pass of qphb.py:822-933 and the vz_offset
    column rewrite of drt1d.py:973-979) on the reference run's own matrices: identical outer / IPM iteration counts,
    every iterate, hyper-parameters, weights, the rewritten matrix and calculate_pq's outputs."""
    from hybrid_util import load_case, initial_rzm_and_vz
    g, special = load_case(name)
    hyp = orc.get_default_hypers()
    if "x_dop" in special:
        hyp.update(orc.get_default_dop_hypers())
    rzm0, vz = initial_rzm_and_vz(g, special)
    r = orc.qphb_fit_prepared(rzm0, g["rv"], [g["m0"], g["m1"], g["m2"]], g["vmm"], special, hyp, vz=vz)
    assert [l["iterations"] for l in r["qp_log"]] == g["qp_iterations"].tolist()
    assert len(r["history"]) == int(g["outer_iterations"])
    np.testing.assert_allclose(np.array([h["x"] for h in r["history"]]), g["hist_x"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(np.array([h["rho_vector"] for h in r["history"]]), g["hist_rho"], rtol=1e-8)
    np.testing.assert_allclose(r["weights"], g["weights"], rtol=1e-8)
    np.testing.assert_allclose(np.array(r["s_vectors"]), g["s_vectors"], rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(r["rzm"], g["rm"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(r["p_matrix"], g["p_matrix"], rtol=1e-8, atol=1e-8 * np.abs(g["p_matrix"]).max())
    np.testing.assert_allclose(r["q_vector"], g["q_vector"], rtol=1e-8, atol=1e-8 * np.abs(g["q_vector"]).max())
    np.testing.assert_allclose(r["l1_lambda_vector"], g["l1_lambda_vector"])
    if "x_dop" in special:
        np.testing.assert_allclose(np.array([h["dop_rho_vector"] for h in r["history"]]), g["hist_dop_rho"], rtol=1e-8)
        np.testing.assert_allclose(r["dop_xmx_norms"], g["dop_xmx_norms"], rtol=1e-7)
    np.testing.assert_allclose(r["xmx_norms"], g["xmx_norms"], rtol=1e-7)


@pytest.mark.parametrize("name,base", [("golden71x91_solverp", None), ("golden71x91_dop_solverp", "golden71x91_dop"),
                                       ("hybrid_s0_dop_solverp", "hybrid_s0_dop")])
def test_general_loop_solve_rp_branch(name, base):
    """solve_rp=True (drt1d.py:568-606): the oracle re-derives the Rp QP, the data rescale and the DOP column rescale from
    the un-rescaled inputs (recovered from the companion fixture of the same measurement without solve_rp) and then
    follows the reference's trajectory."""
    from hybrid_util import load_case, initial_rzm_and_vz
    g, special = load_case(name)
    hyp = orc.get_default_hypers()
    if "x_dop" in special:
        hyp.update(orc.get_default_dop_hypers())
    rzm0, vz = initial_rzm_and_vz(g, special)
    if base is not None:
        gb, _ = load_case(base)
        sf = float(gb["coefficient_scale"] / g["coefficient_scale"])
        rzv0 = gb["rv"]
        a = special["x_dop"]["index"]
        rzm0[:, a:a + special["x_dop"]["size"]] *= gb["dop_scale_vector"] / g["dop_scale_vector"]
    else:
        cs0 = (g["z"].real.max() - g["z"].real.min()) / 14
        sf = cs0 / float(g["coefficient_scale"])
        rzv0 = g["rv"] / sf
    r = orc.qphb_fit_prepared(rzm0, rzv0, [g["m0"], g["m1"], g["m2"]], g["vmm"], special, hyp, vz=vz,
                              solve_rp=dict(basis_area=np.sqrt(np.pi) / float(g["tau_epsilon"])))
    assert [l["iterations"] for l in r["qp_log"]] == g["qp_iterations"].tolist()
    np.testing.assert_allclose(r["scale_factor"], sf, rtol=1e-8)
    np.testing.assert_allclose(r["rzv"], g["rv"], rtol=1e-7, atol=1e-8 * np.abs(g["rv"]).max())
    np.testing.assert_allclose(np.array([h["x"] for h in r["history"]]), g["hist_x"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(r["rzm"], g["rm"], rtol=0, atol=1e-7 * np.abs(g["rm"]).max())


@pytest.mark.parametrize("name,wf", [("golden71x91_wf", 0.7), ("hybrid_s0_wf", 1.5), ("hybrid_s0_wfrp", 1.0)])
def test_general_loop_weight_factors(name, wf):
    """weight_factor / chrono- and EIS-row factors (drt1d.py:887-901, 990-1006) in the oracle's general loop"""
    from hybrid_util import load_case, initial_rzm_and_vz
    g, special = load_case(name)
    rzm0, vz = initial_rzm_and_vz(g, special)
    m = len(g["rv"])
    rows = None
    if "num_chrono" in g:
        nc = int(g["num_chrono"])
        rows = np.concatenate([np.full(nc, float(g["chrono_weight_factor"])), np.full(m - nc, float(g["eis_weight_factor"]))])
    r = orc.qphb_fit_prepared(rzm0, g["rv"], [g["m0"], g["m1"], g["m2"]], g["vmm"], special, orc.get_default_hypers(),
                              vz=vz, weight_factor=wf, row_factors=rows)
    assert [l["iterations"] for l in r["qp_log"]] == g["qp_iterations"].tolist()
    np.testing.assert_allclose(np.array([h["x"] for h in r["history"]]), g["hist_x"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(r["weights"], g["weights"], rtol=1e-8)                  # true_weights
    np.testing.assert_allclose(r["p_matrix"], g["p_matrix"], rtol=1e-8, atol=1e-8 * np.abs(g["p_matrix"]).max())
    np.testing.assert_allclose(r["q_vector"], g["q_vector"], rtol=1e-8, atol=1e-8 * np.abs(g["q_vector"]).max())


def test_general_loop_outlier_branch_hybrid():
    """outlier_p in a joint fit: both initial QPs and the outlier-aware weights of every iteration (48 outer iterations)"""
    from hybrid_util import load_case, initial_rzm_and_vz
    g, special = load_case("hybrid_s0_outlier")
    rzm0, vz = initial_rzm_and_vz(g, special)
    hyp = dict(orc.get_default_hypers(), outlier_p=0.05)
    r = orc.qphb_fit_prepared(rzm0, g["rv"], [g["m0"], g["m1"], g["m2"]], g["vmm"], special, hyp, vz=vz)
    assert [l["iterations"] for l in r["qp_log"]] == g["qp_iterations"].tolist()
    np.testing.assert_allclose(np.array([h["x"] for h in r["history"]]), g["hist_x"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(r["weights"], g["weights"], rtol=1e-6)


@pytest.mark.parametrize("name,base", [("golden71x91_upscale", "golden71x91_dop"), ("hybrid_s0_dop_upscale", "hybrid_s0_dop")])
def test_general_loop_update_scale(name, base):
    """update_scale=True (drt1d.py:903-927): per-iteration rescale of the data, x_in, xmx norms and weights; the oracle
    starts from the un-rescaled data vector (final one divided by the accumulated factor) and must end at the fixture's"""
    from hybrid_util import load_case, initial_rzm_and_vz
    g, special = load_case(name)
    hyp = orc.get_default_hypers()
    if "x_dop" in special:
        hyp.update(orc.get_default_dop_hypers())
    rzm0, vz = initial_rzm_and_vz(g, special)
    if "times" in g:
        gb, _ = load_case(base)
        total = float(gb["coefficient_scale"] / g["coefficient_scale"])
        rzv0 = gb["rv"]
    else:
        cs0 = (g["z"].real.max() - g["z"].real.min()) / 14
        total = cs0 / float(g["coefficient_scale"])
        rzv0 = g["rv"] / total
    r = orc.qphb_fit_prepared(rzm0, rzv0, [g["m0"], g["m1"], g["m2"]], g["vmm"], special, hyp, vz=vz,
                              update_scale=dict(basis_area=np.sqrt(np.pi) / float(g["tau_epsilon"])))
    assert [l["iterations"] for l in r["qp_log"]] == g["qp_iterations"].tolist()
    assert abs(total - 1) > 1e-4                      # the scale really moved
    np.testing.assert_allclose(r["data_scale"], total, rtol=1e-8)
    np.testing.assert_allclose(r["rzv"], g["rv"], rtol=1e-7, atol=1e-9 * np.abs(g["rv"]).max())
    np.testing.assert_allclose(np.array([h["x"] for h in r["history"]]), g["hist_x"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(r["weights"], g["weights"], rtol=1e-7)
    np.testing.assert_allclose(r["est_weights"], g["est_weights"], rtol=1e-7)
    np.testing.assert_allclose(r["xmx_norms"], g["xmx_norms"], rtol=1e-6)


def test_general_loop_eff_hp_false_and_negative_window():
    """eff_hp=False (rho_k enters solve_s, qphb.py:747-750, with the other default alphas) and neg_allowed_tau_range
    (loop QPs allow negative coefficients only inside the window, the initial QP everywhere)"""
    from hybrid_util import load_case, initial_rzm_and_vz
    g, special = load_case("golden71x91_noeff")
    hyp = dict(orc.get_default_hypers(), s_alpha=np.array([1.05, 1.15, 2.5]), rho_alpha=np.array([0.05, 0.1, 0.05]),
               eff_hp=False)
    rzm0, _ = initial_rzm_and_vz(g, special)
    r = orc.qphb_fit_prepared(rzm0, g["rv"], [g["m0"], g["m1"], g["m2"]], g["vmm"], special, hyp)
    assert [l["iterations"] for l in r["qp_log"]] == g["qp_iterations"].tolist()
    np.testing.assert_allclose(np.array([h["x"] for h in r["history"]]), g["hist_x"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(np.array([h["rho_vector"] for h in r["history"]]), g["hist_rho"], rtol=1e-8)
    g, special = load_case("golden71x91_negwin")
    rzm0, _ = initial_rzm_and_vz(g, special)
    idx = np.where((g["basis_tau"] >= 1e-5) & (g["basis_tau"] <= 1e-3))[0] + 2
    r = orc.qphb_fit_prepared(rzm0, g["rv"], [g["m0"], g["m1"], g["m2"]], g["vmm"], special, orc.get_default_hypers(),
                              nonneg=False, neg_allowed_indices=idx)
    assert [l["iterations"] for l in r["qp_log"]] == g["qp_iterations"].tolist()
    np.testing.assert_allclose(np.array([h["x"] for h in r["history"]]), g["hist_x"], rtol=1e-6, atol=1e-8)
    assert np.min(np.delete(r["x"], idx)) >= -1e-9          # outside the window the coefficients stay non-negative


@pytest.mark.parametrize("name", ["hybrid_s0_iwsep", "hybrid_s0_wfw"])
def test_general_loop_separate_initial_weights_and_weight_rule(name):
    """init_weights_separately (one initialize_weights per data block, drt1d.py:648-672) and
    hybrid_weight_factor_method='weight' (factors from the blocks' weight scales, 748-760)"""
    from hybrid_util import load_case, initial_rzm_and_vz
    g, special = load_case(name)
    rzm0, vz = initial_rzm_and_vz(g, special)
    nc = int(g["num_chrono"])
    r = orc.qphb_fit_prepared(rzm0, g["rv"], [g["m0"], g["m1"], g["m2"]], g["vmm"], special, orc.get_default_hypers(), vz=vz,
                              init_separately=dict(num_chrono=nc),
                              weight_method=dict(num_chrono=nc) if name.endswith("wfw") else None)
    assert [l["iterations"] for l in r["qp_log"]] == g["qp_iterations"].tolist()
    np.testing.assert_allclose(r["est_weights"], g["est_weights"], rtol=1e-8)
    np.testing.assert_allclose(np.array([h["x"] for h in r["history"]]), g["hist_x"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(r["p_matrix"], g["p_matrix"], rtol=1e-8, atol=1e-8 * np.abs(g["p_matrix"]).max())


def test_general_loop_dop_pass_without_eff_hp():
    """eff_hp=False with the distribution of phasances: dop_rho_k enters the DOP block's solve_s (qphb.py:858-861)"""
    from hybrid_util import load_case, initial_rzm_and_vz
    g, special = load_case("golden71x91_dop_noeff")
    hyp = dict(orc.get_default_hypers(), s_alpha=np.array([1.05, 1.15, 2.5]), rho_alpha=np.array([0.05, 0.1, 0.05]),
               eff_hp=False)
    hyp.update(orc.get_default_dop_hypers())
    rzm0, _ = initial_rzm_and_vz(g, special)
    r = orc.qphb_fit_prepared(rzm0, g["rv"], [g["m0"], g["m1"], g["m2"]], g["vmm"], special, hyp)
    assert [l["iterations"] for l in r["qp_log"]] == g["qp_iterations"].tolist()
    np.testing.assert_allclose(np.array([h["x"] for h in r["history"]]), g["hist_x"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(np.array([h["dop_rho_vector"] for h in r["history"]]), g["hist_dop_rho"], rtol=1e-8)


@pytest.mark.parametrize("name", ["hybrid_s0", "hybrid_s0_dop", "chrono_s1", "hybrid_s0_outlier"])
def test_warm_restarts_on_prepared_fits_reproduce_the_reference(name):
    """survey 8f rank 3 beyond EIS: oracle.pfrt_fit_prepared / continue_prepared (drt1d.py:1270-1365, 2558-2715: the
    vz_offset column rewritten from a copy frozen at entry, chrono / eis factors and weight_factor on every iteration's
    weights, min_iter = 2) against the reference's own pfrt_fit_hybrid / pfrt_fit_chrono with DRTMD's factors
    (refrun_warm_*.npz: every iterate of the eleven steps, per-step iteration counts, step log-likelihoods, the matrix after
    the last rewrite)."""
    from hybrid_util import load_case, initial_rzm_and_vz
    g, special = load_case(name)
    w = np.load(os.path.join(GOLDEN, f"refrun_warm_{name}.npz"), allow_pickle=False)
    hyp = orc.get_default_hypers()
    if "x_dop" in special:
        hyp.update(orc.get_default_dop_hypers())
    if name.endswith("_outlier"):          # outlier_p in the first fit and in every warm restart (five factors)
        hyp["outlier_p"] = 0.05
    rzm0, vz = initial_rzm_and_vz(g, special)
    r = orc.pfrt_fit_prepared(rzm0, g["rv"], [g["m0"], g["m1"], g["m2"]], g["vmm"], special, hyp, w["pfrt_factors"], vz=vz)
    assert r["step_iters"].tolist() == w["pfrt_step_iters"].tolist()
    peak = np.abs(w["pfrt_hist_x"]).max()
    hx = np.array([h["x"] for h in r["history"]])
    np.testing.assert_allclose(hx, w["pfrt_hist_x"], rtol=0, atol=1e-8 * peak)
    np.testing.assert_allclose(r["step_x"], w["pfrt_step_x"], rtol=0, atol=1e-8 * peak)
    np.testing.assert_allclose(r["step_llh"], w["pfrt_step_llh"], rtol=1e-9)
    np.testing.assert_allclose(np.array([h["weights"] for h in r["history"]]), w["pfrt_hist_weights"], rtol=1e-7)
    np.testing.assert_allclose(r["rm"], w["pfrt_final_rm"], rtol=0, atol=1e-8 * np.abs(w["pfrt_final_rm"]).max())
