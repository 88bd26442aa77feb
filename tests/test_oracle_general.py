"""The oracle's two restatements of cvxopt's coneqp check each other (CPU).

oracle/coneqp.py is specialised to G = -I; oracle/coneqp_general.py restates coneqp + kkt_chol2 for a general dense G and shares
no code with it.  Real cvxopt pins the trajectory through one case only (the reference's known-answer vectors, 71 x 91); at the
sizes the bench runs, with nonneg=False and with the distribution of phasances, the fixtures' QP legs were produced through the
specialisation -- here every such QP goes through the general form as well: same interior-point iteration count, same x to
rounding.  oracle/check_general_shim.py does the same with the REFERENCE ITSELF in the loop (build container only) and leaves
its verdict in tests/golden/general_shim_check.json, which is checked at the end."""
import json
import os

import numpy as np
import pytest

from oracle import drt_oracle as orc
from oracle import resolve_oracle as ro
from oracle.coneqp import coneqp_boxlow
from oracle.coneqp_general import coneqp_dense

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
X_TOL = 1e-10          # of the largest |x|; measured <= 3e-13 (see the asserts' messages when they fire)


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def general(P, q, h):
    """the specialisation's call signature on top of the general solver: G = -I as a dense matrix, as the reference builds it"""
    return coneqp_dense(P, q, -np.eye(len(q)), h)


def both(P, q, h):
    a, b = coneqp_boxlow(P, q, h), general(P, q, h)
    assert a["iterations"] == b["iterations"], (a["iterations"], b["iterations"])
    assert a["status"] == b["status"]
    err = np.max(np.abs(a["x"] - b["x"])) / np.max(np.abs(a["x"]))
    assert err < X_TOL, err
    return a["iterations"], err


def test_known_answer_vectors_of_the_reference_through_the_general_solver(monkeypatch):
    """/root/reference/tests/test_drt_fit.py:6-134 (real cvxopt) with the general solver inside the oracle's fit: the same
    np.allclose the reference's own test applies, and the iteration counts of the seven solves"""
    g = load("ref_test_drt_fit_eis.npz")
    monkeypatch.setattr(orc, "coneqp_boxlow", general)
    drt = orc.OracleDRT()
    drt.fit_eis(g["freq"], g["z"], keep_history=True)
    assert [q["iterations"] for q in drt.qp_log] == [6, 2, 3, 2, 2, 2, 2]
    fp = drt.fit_parameters
    for key in ("x", "R_inf", "inductance", "z_sigma_tot", "q_vector"):
        assert np.allclose(fp[key], g[key], rtol=1e-5, atol=1e-8), key


@pytest.mark.parametrize("name", ["refrun_golden71x91.npz", "refrun_golden71x91_neg.npz", "refrun_c1_71x121.npz"])
def test_every_stored_qp_of_the_reference_runs(name):
    """the QPs the reference built (P, q, h as it handed them to cvxopt.solvers.qp) incl. nonneg=False (h = 1e5: 19 iterations)"""
    g = load(name)
    for i in range(len(g["qp_iterations"])):
        res = general(g[f"qp{i}_P"], g[f"qp{i}_q"], g[f"qp{i}_h"])
        assert res["iterations"] == g["qp_iterations"][i]
        err = np.max(np.abs(res["x"] - g[f"qp{i}_x"])) / np.max(np.abs(g[f"qp{i}_x"]))
        assert err < X_TOL, (i, err)


@pytest.mark.parametrize("name", ["refrun_c2_256x512_s0.npz", "refrun_c2_256x512_s1.npz", "refrun_c2_256x512_s2.npz"])
def test_every_qp_of_a_config2_fit(name):
    """n = 514 (BASELINE configs[1..3]): the oracle's fit regenerates the 30-odd QPs of the reference run (iteration counts
    equal to the fixture's), each is then solved by the general form"""
    g = load(name)
    drt = orc.OracleDRT(fixed_basis_tau=g["basis_tau"])
    drt.fit_eis(g["freq"], g["z"], nonneg=bool(g["nonneg"]), keep_history=True)
    assert [q["iterations"] for q in drt.qp_log] == g["qp_iterations"].tolist()
    worst = 0.0
    for q in drt.qp_log:
        res = general(q["P"], q["q"], q["h"])
        assert res["iterations"] == q["iterations"]
        worst = max(worst, np.max(np.abs(res["x"] - q["x"])) / np.max(np.abs(q["x"])))
    assert worst < X_TOL, worst


@pytest.mark.parametrize("name", ["golden71x91_dop", "hybrid_s0_dop", "hybrid_3step"])
def test_dop_and_hybrid_qps(name):
    """fits with the distribution of phasances and joint chrono + EIS fits (n = 141 ... 150, 11-14 iterations per QP): the
    oracle's prepared-matrix loop regenerates the reference run's QPs (same iteration counts as the fixture), each is then
    solved by the general form"""
    from hybrid_util import load_case, initial_rzm_and_vz
    g, special = load_case(name)
    hyp = orc.get_default_hypers()
    if "x_dop" in special:
        hyp.update(orc.get_default_dop_hypers())
    rzm0, vz = initial_rzm_and_vz(g, special)
    ref = orc.qphb_fit_prepared(rzm0, g["rv"], [g["m0"], g["m1"], g["m2"]], g["vmm"], special, hyp, vz=vz)
    assert [q["iterations"] for q in ref["qp_log"]] == g["qp_iterations"].tolist()
    for q in ref["qp_log"]:
        res = general(q["P"], q["q"], q["h"])
        assert res["iterations"] == q["iterations"]
        assert np.max(np.abs(res["x"] - q["x"])) / np.max(np.abs(q["x"])) < X_TOL


def test_resolve_qp_686_unknowns():
    """mapping/resolve.py:301-341: the coupled QP of seven observations (block-diagonal P + smoothness across observations)"""
    g = load("refrun_resolve_hybrid7.npz")
    special = {str(nm): dict(index=int(i), size=int(s), nonneg=bool(nn)) for nm, i, s, nn in
               zip(g["special_names"], g["special_index"], g["special_size"], g["special_nonneg"])}
    obs = [dict(p_matrix=g["p_matrix"][k], q_vector=g["q_vector"][k], v_baseline=g["v_baseline"][k], vz_offset=g["vz_offset"][k],
                R_inf=g["R_inf"][k], coefficient_scale=g["coefficient_scale"][k], response_signal_scale=g["response_signal_scale"][k],
                scaled_response_offset=g["scaled_response_offset"][k], v_baseline_scale=g["v_baseline_scale"][k])
           for k in range(int(g["n_obs"]))]
    _, res, (P, q, h) = ro.resolve_observations(obs, special)
    assert res["iterations"] == int(g["qp_iterations"][0])
    its, err = both(P, q, h)
    assert its == int(g["qp_iterations"][0])


def test_general_form_handles_a_g_that_is_not_minus_identity():
    """what makes it 'general': box constraints l <= x <= u as G = [-I; I] (2n inequalities) against the optimum of the same
    problem found by projected coordinate descent -- the specialisation cannot even state this problem"""
    rng = np.random.default_rng(3)
    n = 12
    A = rng.standard_normal((30, n))
    P = A.T @ A + 0.1 * np.eye(n)
    q = rng.standard_normal(n) * 3
    lo, up = -0.2 * np.ones(n), 0.3 * np.ones(n)
    G = np.vstack([-np.eye(n), np.eye(n)])
    h = np.concatenate([-lo, up])
    res = coneqp_dense(P, q, G, h)
    assert res["status"] == "optimal"
    x = np.clip(np.zeros(n), lo, up)
    for _ in range(4000):
        for i in range(n):
            r = q[i] + P[i] @ x - P[i, i] * x[i]
            x[i] = np.clip(-r / P[i, i], lo[i], up[i])
    assert np.max(np.abs(res["x"] - x)) < 1e-6


def test_reference_in_the_loop_verdict():
    """oracle/check_general_shim.py (build container, reads /root/reference): the reference's own fits and its resolve with
    cvxopt.solvers.qp routed to the GENERAL form, against the committed fixtures (generated through the specialisation)"""
    path = os.path.join(GOLDEN, "general_shim_check.json")
    assert os.path.exists(path), "run python -m oracle.check_general_shim in the build container"
    res = json.load(open(path))
    for name in ("refrun_golden71x91", "refrun_golden71x91_neg", "refrun_c2_256x512_s0", "refrun_c2_256x512_s1",
                 "refrun_c2_256x512_s2", "refrun_config5_full", "refrun_resolve_c2grid"):
        r = res[name]
        assert r["iterations"] == r["fixture_iterations"], name
        # (resolve: its inputs are seven fits of which two stop at max_iter = 50 -- such a fit repeats to 5e-8 only, whatever
        # solves its QPs, and the coupled QP inherits that; the json carries the inputs' own deviation beside the result's)
        bound = max(1e-9, 2.0 * r.get("x_fit_rel", 0.0))
        assert r["x_rel"] < bound, (name, r["x_rel"], bound)
    assert res["refrun_config5_full"]["n"] == 1078 and res["refrun_resolve_c2grid"]["n"] == 3598
