"""Generate the committed fixtures under tests/golden/ (BUILD-CONTAINER ONLY: reads /root/reference).

Two kinds of fixture, both pure data (inputs + expected outputs):

1. ``ref_test_drt_fit_eis.npz`` -- the known-answer vectors the reference's own test holds
   (/root/reference/tests/test_drt_fit.py:6-134: 71 frequencies, 71 noisy impedances, expected x / R_inf /
   inductance / z_sigma_tot / q_vector).  Extracted by parsing the literals out of that file's AST; these
   numbers were produced by the reference author with the REAL cvxopt, so they pin the oracle's coneqp
   restatement independently of anything in this repository.

2. ``refrun_*.npz`` -- outputs of the reference itself (hybdrt imported read-only from /root/reference
   through oracle/refshim, whose only arithmetic substitution is cvxopt.solvers.qp -> oracle/coneqp.py)
   on seeded synthetic inputs: lookup tables, Z'/Z'' matrices (interp and trapz), penalty matrices, vmm,
   every QP (P, q, h, x, iterations), per-iteration hyper-parameter history, final fit parameters.

Run:  python -m oracle.make_golden      (from the repo root)
"""
import ast
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
OUT = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"


def extract_reference_test_vectors():
    src = open(os.path.join(REF, "tests", "test_drt_fit.py")).read()
    tree = ast.parse(src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "test_drt_fit_eis")
    env = {"np": np}
    for node in fn.body:
        if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name) and \
                node.targets[0].id in ("freq", "z_noisy", "expected_result"):
            code = compile(ast.Module(body=[node], type_ignores=[]), "<golden>", "exec")
            exec(code, env)
    exp = env["expected_result"]
    np.savez(os.path.join(OUT, "ref_test_drt_fit_eis.npz"),
             freq=env["freq"], z=env["z_noisy"], x=exp["x"], R_inf=exp["R_inf"],
             inductance=exp["inductance"], C_inv=exp["C_inv"], z_sigma_tot=exp["z_sigma_tot"],
             vz_offset_eps=exp["vz_offset_eps"], q_vector=exp["q_vector"])
    return env["freq"], env["z_noisy"]


def _boot_reference():
    sys.path.insert(0, os.path.join(HERE, "refshim"))
    import oracle_boot  # noqa: F401
    import cvxopt
    from hybdrt.models import DRT
    return DRT, cvxopt


def _quiet():
    return contextlib.redirect_stdout(io.StringIO())


def run_case(DRT, cvxopt, name, freq, z, ctor_kw, fit_kw, save_mats=True, save_qps=True, row_stride=1):
    log = []
    cvxopt.solvers.options["_oracle_log"] = log
    with _quiet():
        drt = DRT(**ctor_kw)
        drt.fit_eis(freq, z, **fit_kw)
    cvxopt.solvers.options["_oracle_log"] = None
    fp, qp = drt.fit_parameters, drt.qphb_params
    out = dict(freq=freq, z=z, basis_tau=drt.basis_tau, tau_epsilon=drt.tau_epsilon,
               x=fp["x"], R_inf=fp["R_inf"], inductance=fp["inductance"], C_inv=fp["C_inv"], z_sigma_tot=fp["z_sigma_tot"],
               q_vector=fp["q_vector"], x_scaled=np.array(list(drt.cvx_result["x"])),
               coefficient_scale=drt.coefficient_scale,
               est_weights=qp["est_weights"], weights=qp["true_weights"], rho_vector=qp["rho_vector"],
               s_vectors=np.array(qp["s_vectors"]), xmx_norms=qp["xmx_norms"],
               x_overfit=qp["x_overfit_eis"], qp_iterations=np.array([l["iterations"] for l in log]),
               outer_iterations=len(drt.qphb_history),
               hist_x=np.array([h["x"] for h in drt.qphb_history]),
               hist_rho=np.array([h["rho_vector"] for h in drt.qphb_history]),
               hist_weights=np.array([h["weights"] for h in drt.qphb_history]),
               nonneg=fit_kw.get("nonneg", True))
    if save_mats:
        lk = drt.interpolate_lookups
        if lk["z_real"] is not None:
            out.update(lut_log_wt_re=lk["z_real"][0], lut_z_re=lk["z_real"][1],
                       lut_log_wt_im=lk["z_imag"][0], lut_z_im=lk["z_imag"][1])
        zm = drt.fit_matrices["impedance"]
        out.update(zm_re=zm.real[::row_stride], zm_im=zm.imag[::row_stride], row_stride=row_stride,
                   m0=drt.fit_matrices["m0"][::row_stride], m1=drt.fit_matrices["m1"][::row_stride],
                   m2=drt.fit_matrices["m2"][::row_stride], vmm=qp["vmm"][::row_stride],
                   rm=qp["rm"][::row_stride], rv=qp["rv"])
    if save_qps:
        for i, l in enumerate(log):
            out[f"qp{i}_P"] = l["P"]; out[f"qp{i}_q"] = l["q"]; out[f"qp{i}_h"] = l["h"]
            out[f"qp{i}_x"] = l["x"]; out[f"qp{i}_pcost"] = l["pcost"]
        out["p_matrix"] = fp["p_matrix"]
    np.savez_compressed(os.path.join(OUT, f"refrun_{name}.npz"), **out)
    print(f"{name}: outer={out['outer_iterations']} qp_iters={out['qp_iterations'].tolist()}")


def run_trapz_matrices(name, freq, tau, eps):
    sys.path.insert(0, os.path.join(HERE, "refshim"))
    import oracle_boot  # noqa: F401
    from hybdrt.matrices import mat1d
    out = dict(freq=freq, tau=tau, eps=eps)
    for part in ("real", "imag"):
        out[f"A_{part}"] = mat1d.construct_impedance_matrix(freq, part, tau=tau, epsilon=eps,
                                                            integrate_method="trapz")
    np.savez_compressed(os.path.join(OUT, f"refrun_{name}.npz"), **out)
    print(f"{name}: trapz matrices {out['A_real'].shape}")


def chrono_times(n_pre, n_post, t_step, dt_pre=5e-4, t_lo=1e-4, t_hi=50.0):
    """sample times of a step experiment: n_pre uniform samples ending at the step, n_post log-uniform after it"""
    pre = t_step - dt_pre * np.arange(n_pre, 0, -1) + dt_pre
    return np.concatenate([pre, t_step + np.logspace(np.log10(t_lo), np.log10(t_hi), n_post)])


def run_response_matrices():
    """survey row a3: basis.generate_response_lookup (basis.py:672-689) and mat1d.construct_response_matrix
    (mat1d.py:16-122), gaussian basis / galvanostatic / ideal step, through the reference itself"""
    sys.path.insert(0, os.path.join(HERE, "refshim"))
    import oracle_boot  # noqa: F401
    from hybdrt.matrices import basis, mat1d
    out = {}
    tau = np.logspace(-7, 3, 96)
    eps = 1 / np.mean(np.diff(np.log(tau)))
    for tag, e in (("eps_grid", eps), ("eps_4p34", 10 / np.log(10))):
        lg, rg = basis.generate_response_lookup('gaussian', 'galv', 'ideal', e, 2000)
        out[f"lookup_{tag}_eps"] = e
        out[f"lookup_{tag}_log_td"] = lg
        out[f"lookup_{tag}_v"] = rg
    grids = (out["lookup_eps_grid_log_td"], out["lookup_eps_grid_v"])
    times = chrono_times(24, 136, 0.05)
    cases = {
        "one_step": (np.array([0.05]), np.array([1e-3])),
        "three_steps": (np.array([0.05, 1.0, 5.0]), np.array([1e-3, -2e-3, 1e-3])),
        "step_after_end": (np.array([0.05, 1e3]), np.array([1e-3, 5e-4])),
    }
    out["tau"] = tau
    out["times"] = times
    out["epsilon"] = eps
    for name, (st, sa) in cases.items():
        a, lay = mat1d.construct_response_matrix(tau, times, 'ideal', st, sa, basis_type='gaussian', epsilon=eps,
                                                 op_mode='galv', integrate_method='interp', interpolate_grids=grids)
        out[f"{name}_step_times"] = st
        out[f"{name}_step_sizes"] = sa
        out[f"{name}_A"] = a
        out[f"{name}_layered"] = lay
    # trapz mode, small
    tau_s = np.logspace(-5, 1, 20)
    eps_s = 1 / np.mean(np.diff(np.log(tau_s)))
    times_s = chrono_times(4, 20, 0.01, t_lo=1e-4, t_hi=10.0)
    st, sa = np.array([0.01, 0.5]), np.array([2e-3, -1e-3])
    a, lay = mat1d.construct_response_matrix(tau_s, times_s, 'ideal', st, sa, basis_type='gaussian', epsilon=eps_s,
                                             op_mode='galv', integrate_method='trapz', integrate_points=1000)
    out.update(trapz_tau=tau_s, trapz_times=times_s, trapz_epsilon=eps_s, trapz_step_times=st, trapz_step_sizes=sa,
               trapz_A=a, trapz_layered=lay)
    # the non-default forms (mat1d.py:96-118): potentiostatic delta response; expdecay step model (trapz) + its inductance vector
    a, lay = mat1d.construct_response_matrix(tau_s, times_s, 'ideal', st, sa, basis_type='gaussian', epsilon=eps_s, op_mode='pot')
    out.update(pot_A=a, pot_layered=lay)
    tr = np.array([2e-4, 5e-3])
    a, lay = mat1d.construct_response_matrix(tau_s, times_s, 'expdecay', st, sa, basis_type='gaussian', epsilon=eps_s, tau_rise=tr,
                                             op_mode='galv', integrate_method='trapz', integrate_points=1000)
    out.update(expdecay_tau_rise=tr, expdecay_A=a, expdecay_layered=lay,
               expdecay_inductance_rv=mat1d.construct_inductance_response_vector(times_s, 'expdecay', st, sa, tr))
    # chrono variance-estimation matrices (survey row a5, mat1d.py:457-490): flexible and uniform error structure
    for name in ("one_step", "three_steps"):
        st = cases[name][0]
        out[f"{name}_vmm"] = mat1d.construct_chrono_var_matrix(times, st, 0.25, None)
    out["uniform_vmm"] = mat1d.construct_chrono_var_matrix(times, cases["one_step"][0], 0.25, 'uniform')
    # distribution-of-phasances matrices (survey row a18, phasance.py:108-184): the fit's nu grid and epsilon
    from hybdrt.matrices import phasance
    basis_nu = np.concatenate([np.linspace(-1, -0.4, 25), np.linspace(0.4, 1, 25)])
    nu_eps = 1 / np.median(np.diff(np.sort(basis_nu)))
    f_dop = np.logspace(5, 1, 64)
    out.update(dop_nu=basis_nu, dop_epsilon=nu_eps, dop_freq=f_dop,
               dop_zm=phasance.construct_phasor_z_matrix(f_dop, basis_nu, 'gaussian', nu_eps),
               dop_scale=phasance.phasor_scale_vector(basis_nu, tau))
    vm, vlay = phasance.construct_phasor_v_matrix(times, basis_nu, 'gaussian', nu_eps, 'ideal', cases["three_steps"][0],
                                                   cases["three_steps"][1])
    out.update(dop_vm=vm, dop_vm_layered=vlay)
    for eps_small in (2.0,):
        out[f"dop_zm_eps{eps_small:g}"] = phasance.construct_phasor_z_matrix(f_dop, basis_nu, 'gaussian', eps_small)
    np.savez_compressed(os.path.join(OUT, "refrun_response.npz"), **out)
    print("refrun_response.npz written:", {k: np.shape(v) for k, v in out.items()})


def run_posterior(DRT, name, freq, z, ctor_kw, tau_eval):
    """survey 8f rank 1: what DRTMD.fit_observation takes from a finished fit (drtmd.py:258-279): evaluate_llh,
    evaluate_rss and the diagonal of estimate_distribution_cov on the supergrid (with and without extend_var)"""
    with _quiet():
        drt = DRT(**ctor_kw)
        drt.fit_eis(freq, z)
        cov = drt.estimate_distribution_cov(tau=tau_eval)
        cov_ext = drt.estimate_distribution_cov(tau=tau_eval, extend_var=True)
        llh = drt.evaluate_llh()
        rss = drt.evaluate_rss()
        pcov = drt.estimate_param_cov()
    fp, qp = drt.fit_parameters, drt.qphb_params
    out = dict(freq=freq, z=z, basis_tau=drt.basis_tau, tau_epsilon=drt.tau_epsilon, tau_eval=tau_eval,
               coefficient_scale=drt.coefficient_scale, p_matrix=fp["p_matrix"], num_special=drt.get_qp_mat_offset(),
               dist_var=np.diag(cov), dist_var_ext=np.diag(cov_ext), param_var=np.diag(pcov), llh=llh, rss=rss,
               x_scaled=drt.qphb_history[-1]["x"], est_weights=qp["est_weights"], rm=qp["rm"], rv=qp["rv"],
               x=fp["x"])
    np.savez_compressed(os.path.join(OUT, f"refrun_posterior_{name}.npz"), **out)
    print(f"posterior_{name}: llh={llh:.6f} rss={rss:.6f} var range {out['dist_var'].min():.3e}..{out['dist_var'].max():.3e}")


def run_posterior_sneg(DRT, freq, z, ctor_kw, tau_eval):
    """series_neg=True (drt1d.py:3090-3103): the posterior of the positive copy (sign=1, what DRTMD stores), of the negative
    copy (sign=-1) and of their difference (sign=0: pos + neg - cross terms), plus the parameter variances of all 2 ntau + ns"""
    with _quiet():
        drt = DRT(**ctor_kw)
        drt.fit_eis(freq, z, series_neg=True)
        out = dict(freq=freq, z=z, tau_eval=tau_eval, basis_tau=drt.basis_tau, coefficient_scale=drt.coefficient_scale,
                   x=drt.fit_parameters["x"], param_var=np.diag(drt.estimate_param_cov()),
                   p_matrix=drt.fit_parameters["p_matrix"])
        for sign, tag in ((1, "pos"), (-1, "neg"), (0, "both")):
            cov = drt.estimate_distribution_cov(tau=tau_eval, sign=sign)
            out[f"dist_var_{tag}"] = np.diag(cov)
            out[f"dist_cov_{tag}_row40"] = cov[40]
        out["dist_var_pos_ext"] = np.diag(drt.estimate_distribution_cov(tau=tau_eval, extend_var=True))
    np.savez_compressed(os.path.join(OUT, "refrun_posterior_golden71x91_sneg.npz"), **out)
    print("posterior_sneg: var ranges", {t: (float(out[f'dist_var_{t}'].min()), float(out[f'dist_var_{t}'].max())) for t in ("pos", "neg", "both")})


def run_candidates(DRT, name, freq, z, ctor_kw):
    """survey 8f rank 3: warm restarts of the outer loop (_continue_from_init) as the candidate generators drive them
    (drt1d.py:1497-1632): 2 s_0 steps x4 and 3 weight steps x0.5, every per-iteration state."""
    with _quiet():
        drt = DRT(**ctor_kw)
        drt.fit_eis(freq, z)
        base_x = drt.qphb_history[-1]["x"].copy()
        cx_s, hist_s, _ = drt._generate_candidates_s0(4, 2, 1e-2, 10)
        cx_w, hist_w, _ = drt._generate_candidates_weights(0.5, 3, 1e-2, 10)
    out = dict(freq=freq, z=z, base_x=base_x)
    for tag, hist in (("s0", hist_s), ("w", hist_w)):
        out[f"{tag}_x"] = np.array([h["x"] for h in hist])
        out[f"{tag}_rho"] = np.array([h["rho_vector"] for h in hist])
        out[f"{tag}_weights"] = np.array([h["weights"] for h in hist])
        out[f"{tag}_s"] = np.array([np.array(h["s_vectors"]) for h in hist])
    # PFRT (drt1d.py:2558-2700): full fit at factor 0.1, ten warm restarts up to factor 10
    with _quiet():
        drt2 = DRT(**ctor_kw)
        drt2.pfrt_fit_eis(freq, z)
    pr = drt2.pfrt_result
    out.update(pfrt_factors=pr["factors"], pfrt_step_x=np.array(pr["step_x"]), pfrt_step_llh=np.array(pr["step_llh"]),
               pfrt_history_len=len(drt2.pfrt_history), pfrt_init_len=len(drt2.qphb_history))
    np.savez_compressed(os.path.join(OUT, f"refrun_candidates_{name}.npz"), **out)
    print(f"candidates_{name}: s0 history {len(hist_s)}, weights history {len(hist_w)}")


def run_warm_restarts_prepared(DRT, name, data, ctor_kw, factors=None, fit_kw=None, candidates=True):
    """survey 8f rank 3 on chrono / joint / DOP fits: _continue_from_init re-enters the loop with the vz_offset column rewrite
    and the chrono / eis weight factors (drt1d.py:1270-1365), driven by pfrt_fit_hybrid / pfrt_fit_chrono (2558-2715; DRTMD's
    factors logspace(-0.7, 0.7, 11), drtmd.py:98-100) and by the candidate generators (1497-1632).  Every warm restart's
    iterates are recorded through a wrapper around the reference's own _continue_from_init."""
    times, i_sig, v_sig, freq, z = data
    factors = np.logspace(-0.7, 0.7, 11) if factors is None else factors
    fit_kw = dict(fit_kw or {})        # e.g. outlier_p: the first fit AND every warm restart run with it (drt1d.py:2564-2566, 2673)
    calls = []

    def wrap(drt):
        inner = drt._continue_from_init

        def recording(*a, **k):
            hist = inner(*a, **k)
            calls.append(hist)
            return hist
        drt._continue_from_init = recording

    with _quiet():
        drt = DRT(**ctor_kw)
        wrap(drt)
        if freq is None:
            drt.pfrt_fit_chrono(times, i_sig, v_sig, factors=factors, **fit_kw)
        else:
            drt.pfrt_fit_hybrid(times, i_sig, v_sig, freq, z, factors=factors, **fit_kw)
    pr = drt.pfrt_result
    dop = bool(ctor_kw.get("fit_dop"))
    out = dict(pfrt_factors=np.asarray(pr["factors"]), pfrt_step_x=np.array(pr["step_x"]),
               pfrt_step_llh=np.array(pr["step_llh"]), pfrt_init_len=len(drt.qphb_history),
               pfrt_step_iters=np.array([len(drt.qphb_history)] + [len(h) for h in calls]),
               pfrt_hist_x=np.array([h["x"] for h in drt.pfrt_history]),
               pfrt_hist_rho=np.array([h["rho_vector"] for h in drt.pfrt_history]),
               pfrt_hist_weights=np.array([h["weights"] for h in drt.pfrt_history]),
               pfrt_final_rm=drt.qphb_params["rm"],
               pfrt_step_p_diag=np.array([np.diag(pm) for pm in pr["step_p_mat"]]))
    if dop:
        out["pfrt_hist_dop_rho"] = np.array([h["dop_rho_vector"] for h in drt.pfrt_history])
    if not candidates:
        np.savez_compressed(os.path.join(OUT, f"refrun_warm_{name}.npz"), **out)
        print(f"warm_{name}: pfrt step iterations {out['pfrt_step_iters'].tolist()}")
        return
    # candidates on a fresh fit (the weight candidates first, as generate_candidates runs them)
    calls.clear()
    with _quiet():
        drt2 = DRT(**ctor_kw)
        wrap(drt2)
        if freq is None:
            drt2.fit_chrono(times, i_sig, v_sig)
        else:
            drt2.fit_hybrid(times, i_sig, v_sig, freq, z)
        base_len = len(drt2.qphb_history)
        cx_w, hist_w, _ = drt2._generate_candidates_weights(0.5, 3, 1e-2, 10)
        cx_s, hist_s, _ = drt2._generate_candidates_s0(4, 2, 1e-2, 10)
    out.update(cand_base_len=base_len, cand_call_iters=np.array([len(h) for h in calls]),
               cand_w_x=np.array([h["x"] for h in hist_w]), cand_w_weights=np.array([h["weights"] for h in hist_w]),
               cand_w_rho=np.array([h["rho_vector"] for h in hist_w]),
               cand_s0_x=np.array([h["x"] for h in hist_s]), cand_s0_weights=np.array([h["weights"] for h in hist_s]),
               cand_s0_rho=np.array([h["rho_vector"] for h in hist_s]),
               cand_s0_s=np.array([np.array(h["s_vectors"]) for h in hist_s]))
    np.savez_compressed(os.path.join(OUT, f"refrun_warm_{name}.npz"), **out)
    print(f"warm_{name}: pfrt step iterations {out['pfrt_step_iters'].tolist()}, candidate calls {out['cand_call_iters'].tolist()}")


def run_hybrid_case(DRT, cvxopt, name, data, ctor_kw, fit_kw):
    """config-5 family (drt1d.py:1244-1268 -> 102-1104): joint chrono + EIS fit, optionally with the distribution of
    phasances; every prepared quantity the build's host layer must reproduce, the QP matrices, the per-iteration
    history and the extracted parameters.  ``data`` = (times, i, v, freq, z); times None -> fit_eis."""
    times, i_sig, v_sig, freq, z = data
    log = []
    cvxopt.solvers.options["_oracle_log"] = log
    with _quiet():
        drt = DRT(**ctor_kw)
        if times is None:
            drt.fit_eis(freq, z, **fit_kw)
        elif freq is None:
            drt.fit_chrono(times, i_sig, v_sig, **fit_kw)
        else:
            drt.fit_hybrid(times, i_sig, v_sig, freq, z, **fit_kw)
    cvxopt.solvers.options["_oracle_log"] = None
    fp, qp = drt.fit_parameters, drt.qphb_params
    sp = drt.special_qp_params
    out = dict(freq=freq, z=z, basis_tau=drt.basis_tau, tau_epsilon=drt.tau_epsilon,
               special_names=np.array(list(sp.keys())), special_index=np.array([v["index"] for v in sp.values()]),
               special_size=np.array([v.get("size", 1) for v in sp.values()]),
               special_nonneg=np.array([v["nonneg"] for v in sp.values()]),
               coefficient_scale=drt.coefficient_scale, impedance_scale=drt.impedance_scale,
               rm=qp["rm"], rv=qp["rv"], vmm=qp["vmm"], l1_lambda_vector=qp["l1_lambda_vector"],
               m0=qp["penalty_matrices"]["m0"], m1=qp["penalty_matrices"]["m1"], m2=qp["penalty_matrices"]["m2"],
               est_weights=qp["est_weights"], init_weights=qp["init_weights"], weights=qp["true_weights"],
               scaled_weights=qp["weights"], rho_vector=qp["rho_vector"], s_vectors=np.array(qp["s_vectors"]),
               xmx_norms=qp["xmx_norms"], qp_iterations=np.array([l["iterations"] for l in log]),
               eis_weight_factor=np.nan if qp["eis_weight_factor"] is None else qp["eis_weight_factor"],
               chrono_weight_factor=np.nan if qp["chrono_weight_factor"] is None else qp["chrono_weight_factor"],
               outer_iterations=len(drt.qphb_history),
               hist_x=np.array([h["x"] for h in drt.qphb_history]),
               hist_rho=np.array([h["rho_vector"] for h in drt.qphb_history]),
               hist_weights=np.array([h["weights"] for h in drt.qphb_history]),
               x=fp["x"], R_inf=fp["R_inf"], inductance=fp["inductance"], C_inv=fp["C_inv"],
               q_vector=fp["q_vector"], p_matrix=fp["p_matrix"], x_scaled=np.array(list(drt.cvx_result["x"])))
    if freq is not None:
        out["z_sigma_tot"] = fp["z_sigma_tot"]
    else:
        del out["freq"], out["z"], out["impedance_scale"]
    if fit_kw.get("remove_outliers"):
        out["eis_outlier_index"] = drt.eis_outlier_index
        if times is not None:
            out["chrono_outlier_index"] = drt.chrono_outlier_index
    if drt.fit_dop:
        out.update(basis_nu=drt.basis_nu, nu_epsilon=drt.nu_epsilon, dop_scale_vector=drt.dop_scale_vector,
                   dop_rho_vector=qp["dop_rho_vector"], dop_xmx_norms=qp["dop_xmx_norms"], x_dop=fp["x_dop"],
                   hist_dop_rho=np.array([h["dop_rho_vector"] for h in drt.qphb_history]))
    if fit_kw.get("downsample"):
        out["sample_index"] = drt.sample_index
        out["sample_v"] = drt.raw_response_signal
        if fit_kw["downsample_kw"].get("target_times") is not None:
            out["downsample_target_times"] = fit_kw["downsample_kw"]["target_times"]
    if times is not None:
        out.update(times=times, i_signal=i_sig, v_signal=v_sig, sample_times=drt.get_fit_times(),
                   step_times=drt.step_times, step_sizes=drt.step_sizes,
                   nonconsec_step_times=drt.nonconsec_step_times,
                   input_signal_scale=drt.input_signal_scale, response_signal_scale=drt.response_signal_scale,
                   scaled_response_offset=drt.scaled_response_offset, v_baseline_scale=drt.v_baseline_scale,
                   vz_strength_vec=qp["vz_strength_vec"], num_chrono=qp["num_chrono"],
                   v_baseline=fp["v_baseline"], vz_offset=fp.get("vz_offset", np.nan), v_sigma_tot=fp["v_sigma_tot"],
                   response_matrix=drt.fit_matrices["response"], inf_response=drt.fit_matrices["inf_response"])
    np.savez_compressed(os.path.join(OUT, f"refrun_{name}.npz"), **out)
    print(f"{name}: m x n = {qp['rm'].shape}, outer={out['outer_iterations']} qp_iters={out['qp_iterations'].tolist()}")


def run_hybrid_cases(DRT, cvxopt, freq_g, z_g):
    from hipdrt import synth
    base = dict(fit_inductance=True, fit_capacitance=False, fit_ohmic=True)
    run_hybrid_case(DRT, cvxopt, "golden71x91_dop", (None, None, None, freq_g, z_g), dict(base, fit_dop=True), {})
    meas = synth.hybrid_measurement(seed=0)
    run_hybrid_case(DRT, cvxopt, "hybrid_s0", meas, dict(base, fit_dop=False), {})
    run_hybrid_case(DRT, cvxopt, "hybrid_s0_dop", meas, dict(base, fit_dop=True), {})
    # chrono-only fit (drt1d.py:1195-1213) and a three-step protocol with the non-default chrono options
    run_hybrid_case(DRT, cvxopt, "chrono_s1", meas[:3] + (None, None), dict(base, fit_dop=False), {})
    meas3 = synth.hybrid_measurement(seed=2, n_post=80, extra_steps=((2.0, -2e-3), (3.0, 1e-3)))
    run_hybrid_case(DRT, cvxopt, "hybrid_3step_opts", meas3, dict(base, fit_dop=False),
                    dict(vz_offset=False, chrono_error_structure=None, smooth_inf_response=False, offset_baseline=False,
                         chrono_vmm_epsilon=2, vz_offset_eps=2))
    run_hybrid_case(DRT, cvxopt, "hybrid_3step", meas3, dict(base, fit_dop=False), dict(vz_offset_scale=0.5, vz_offset_eps=2))
    # solve_rp=True (drt1d.py:568-606, 5421-5437): one extra QP re-estimates Rp, data and DOP columns are rescaled
    run_hybrid_case(DRT, cvxopt, "golden71x91_dop_solverp", (None, None, None, freq_g, z_g), dict(base, fit_dop=True),
                    dict(solve_rp=True))
    run_hybrid_case(DRT, cvxopt, "hybrid_s0_dop_solverp", meas, dict(base, fit_dop=True), dict(solve_rp=True))
    run_hybrid_case(DRT, cvxopt, "golden71x91_solverp", (None, None, None, freq_g, z_g), dict(base, fit_dop=False),
                    dict(solve_rp=True))
    # outlier-aware weights in a joint fit (qphb.py:1497-1553, 1629-1656)
    run_hybrid_case(DRT, cvxopt, "hybrid_s0_outlier", meas, dict(base, fit_dop=False), dict(outlier_p=0.05))
    # remove_outliers (drt1d.py:214-302, 817-833): an initialize_weights-only pass flags points, the fit runs without them
    t_o, i_o, v_o, f_o, z_o = synth.hybrid_measurement(seed=4)
    z_o = z_o.copy(); v_o = v_o.copy()
    z_o[10] += 0.3; z_o[25] -= 0.25j; v_o[100] += 5e-5; v_o[180] -= 8e-5
    run_hybrid_case(DRT, cvxopt, "hybrid_rmout", (t_o, i_o, v_o, f_o, z_o), dict(base, fit_dop=False),
                    dict(remove_outliers=True, outlier_p=0.05))
    run_hybrid_case(DRT, cvxopt, "eis_rmout", (None, None, None, f_o, z_o), dict(base, fit_dop=False),
                    dict(remove_outliers=True, outlier_p=0.05))
    z_x = z_o.copy(); z_x[10] += 5.0; z_x[40] -= 4.0j        # gross errors: remove_extremes (drt1d.py:187-212)
    run_hybrid_case(DRT, cvxopt, "eis_rmext", (None, None, None, f_o, z_x), dict(base, fit_dop=False),
                    dict(remove_extremes=True))
    # update_scale=True (drt1d.py:903-927)
    run_hybrid_case(DRT, cvxopt, "golden71x91_upscale", (None, None, None, freq_g, z_g), dict(base, fit_dop=False),
                    dict(update_scale=True))
    run_hybrid_case(DRT, cvxopt, "hybrid_s0_dop_upscale", meas, dict(base, fit_dop=True), dict(update_scale=True))
    # eff_hp=False (qphb.py:208-222, 747-750) and a window of allowed negative coefficients (drt1d.py:82-91)
    run_hybrid_case(DRT, cvxopt, "golden71x91_noeff", (None, None, None, freq_g, z_g), dict(base, fit_dop=False),
                    dict(eff_hp=False))
    run_hybrid_case(DRT, cvxopt, "golden71x91_dop_noeff", (None, None, None, freq_g, z_g), dict(base, fit_dop=True),
                    dict(eff_hp=False))
    run_hybrid_case(DRT, cvxopt, "golden71x91_negwin", (None, None, None, freq_g, z_g), dict(base, fit_dop=False),
                    dict(nonneg=False, neg_allowed_tau_range=(1e-5, 1e-3)))
    # weight factors (drt1d.py:743-803, 887-901, 990-1000)
    run_hybrid_case(DRT, cvxopt, "golden71x91_wf", (None, None, None, freq_g, z_g), dict(base, fit_dop=False),
                    dict(weight_factor=0.7))
    run_hybrid_case(DRT, cvxopt, "hybrid_s0_wf", meas, dict(base, fit_dop=False),
                    dict(weight_factor=1.5, eis_weight_factor=2.0, chrono_weight_factor=0.5))
    run_hybrid_case(DRT, cvxopt, "hybrid_s0_wfrp", meas, dict(base, fit_dop=False), dict(hybrid_weight_factor_method='rp'))
    # separate initial weights per data block, and the 'weight' rule for the hybrid factors (drt1d.py:648-672, 748-760)
    run_hybrid_case(DRT, cvxopt, "hybrid_s0_iwsep", meas, dict(base, fit_dop=False), dict(init_weights_separately=True))
    run_hybrid_case(DRT, cvxopt, "hybrid_s0_wfw", meas, dict(base, fit_dop=False),
                    dict(init_weights_separately=True, hybrid_weight_factor_method='weight'))
    # series_neg: a sign-flipped second copy of the basis (drt1d.py:5497-5530)
    run_hybrid_case(DRT, cvxopt, "golden71x91_sneg", (None, None, None, freq_g, z_g), dict(base, fit_dop=False),
                    dict(series_neg=True))
    # discard_first_n (drt1d.py:167-178, preprocessing.py:471-504)
    run_hybrid_case(DRT, cvxopt, "hybrid_s0_discard", meas, dict(base, fit_dop=False), dict(discard_first_n=2))
    # anti-aliased down-sampling of a densely sampled record (drtbase.py:324-340, preprocessing.py:335-470, 507-589)
    dense = synth.hybrid_measurement(seed=5, n_pre=200, n_post=6000, uniform_dt=2.5e-4, v_noise=2e-5)
    tt = np.concatenate(([0], np.logspace(-3.5, np.log10(1.45), 81)))
    run_hybrid_case(DRT, cvxopt, "hybrid_downsample", dense, dict(base, fit_dop=False),
                    dict(downsample=True, downsample_kw=dict(prestep_samples=10, target_times=tt)))
    # polynomial + square-root voltage baseline (background.py:23-37; three v_baseline coefficients)
    run_hybrid_case(DRT, cvxopt, "hybrid_vb", meas, dict(base, fit_dop=False),
                    dict(v_baseline_deg=1, v_baseline_sqrt=True, v_baseline_penalty=[1e-6, 1e-4, 1e-5]))
    # series capacitance as a special parameter (C_inv column: drt1d.py:5803, 5838, 5888; mat1d.py:423-451)
    meas_c = synth.hybrid_measurement(seed=3, c_series=20.0, t_hi=5.0)
    capb = dict(base, fit_capacitance=True)
    run_hybrid_case(DRT, cvxopt, "eis_cap", (None, None, None) + meas_c[3:], dict(capb, fit_dop=False), {})
    run_hybrid_case(DRT, cvxopt, "hybrid_cap", meas_c, dict(capb, fit_dop=False), {})


def run_resolve(DRT, cvxopt, name, fit_dop, n_obs=7, basis_tau=None, store_p=True):
    """survey 8f rank 2: mapping/resolve.py:189-341 -- coherent re-optimisation of neighbouring observations (block
    diagonal P of the single fits + a smoothness penalty across observations).  The reference only supports hybrid
    fits here (get_offset_pq indexes v_baseline / vz_offset unconditionally).  Saves what resolve consumes from every
    fitted DRT object and what it returns."""
    from hipdrt import synth
    from hybdrt.mapping import resolve
    base = dict(fit_inductance=True, fit_capacitance=False, fit_ohmic=True, fit_dop=fit_dop)
    if basis_tau is not None:
        base["fixed_basis_tau"] = basis_tau
    drts = []
    with _quiet():
        for s_ in range(n_obs):
            drt = DRT(**base)
            drt.fit_hybrid(*synth.hybrid_measurement(seed=s_, jitter=True, n_post=120, nf=31))
            drts.append(drt)
    ntau = len(drts[0].basis_tau)
    assert all(len(d.basis_tau) == ntau for d in drts)
    tau_idx = [(0, ntau)] * n_obs
    log = []
    cvxopt.solvers.options["_oracle_log"] = log
    with _quiet():
        x_opt, match = resolve.resolve_observations(drts, tau_idx, True)
        x_opt_s3, _ = resolve.resolve_observations(drts, tau_idx, True, sigma=2, lambda_psi=10)
    cvxopt.solvers.options["_oracle_log"] = None
    sp = drts[0].special_qp_params
    out = dict(n_obs=n_obs, ntau=ntau, special_names=np.array(list(sp.keys())),
               special_index=np.array([v["index"] for v in sp.values()]),
               special_size=np.array([v.get("size", 1) for v in sp.values()]),
               special_nonneg=np.array([v["nonneg"] for v in sp.values()]),
               p_matrix=_pad_stack([d.fit_parameters["p_matrix"] for d in drts]),
               q_vector=_pad_stack([d.fit_parameters["q_vector"] for d in drts]),
               n_params=np.array([len(d.fit_parameters["q_vector"]) for d in drts]),
               v_baseline=np.array([d.fit_parameters["v_baseline"] for d in drts]),
               vz_offset=np.array([d.fit_parameters["vz_offset"] for d in drts]),
               R_inf=np.array([d.fit_parameters["R_inf"] for d in drts]),
               coefficient_scale=np.array([d.coefficient_scale for d in drts]),
               response_signal_scale=np.array([d.response_signal_scale for d in drts]),
               scaled_response_offset=np.array([d.scaled_response_offset for d in drts]),
               v_baseline_scale=np.array([d.v_baseline_scale for d in drts]),
               x_opt=x_opt, match_tau_indices=np.array(match), x_opt_sigma2_lambda10=x_opt_s3,
               qp_iterations=np.array([l["iterations"] for l in log]),
               qp0_P_diag=np.diag(log[0]["P"]), qp0_q=log[0]["q"], qp0_h=log[0]["h"])
    if fit_dop:
        out.update(x_dop=np.array([d.fit_parameters["x_dop"] for d in drts]),
                   dop_scale_vector=np.array([d.dop_scale_vector for d in drts]))
    if not store_p:
        # large grids: the single fits' P matrices (n_obs x n x n doubles) stay out of the repository; the fixture keeps the
        # fitted coefficients (the device test fits the same synthetic cells itself and compares them first), P's diagonal
        # and the resolved result
        out["p_diag"] = np.array([np.diag(p_) for p_ in out.pop("p_matrix")])
        out["x_fit"] = np.array([d.fit_parameters["x"] for d in drts])
        out["basis_tau"] = np.asarray(basis_tau)
    np.savez_compressed(os.path.join(OUT, f"refrun_resolve_{name}.npz"), **out)
    print(f"resolve_{name}: {n_obs} obs x {x_opt.shape[1]} params, qp iterations {out['qp_iterations'].tolist()}")


def _pad_stack(arrays):
    """arrays of one rank and different sizes -> one zero-padded stack (the sizes are stored beside it)"""
    arrays = [np.asarray(a) for a in arrays]
    shape = tuple(max(a.shape[d] for a in arrays) for d in range(arrays[0].ndim))
    out = np.zeros((len(arrays),) + shape)
    for i, a in enumerate(arrays):
        out[(i,) + tuple(slice(0, n) for n in a.shape)] = a
    return out


def run_resolve_group(cvxopt, name, n_obs=16, batch_size=7, overlap=2, t_hi_of=None):
    """the reference's own DRTMD (mapping/drtmd.py:186-329, 432-559) driven with array data: n_obs joint fits along one psi
    axis, then resolve_group -- overlapping batches of coupled QPs, margin-weighted average of the overlaps."""
    from hipdrt import synth
    from hybdrt.mapping.drtmd import DRTMD
    sup = np.logspace(-8, 4, 121)
    with _quiet():
        dmd = DRTMD(tau_supergrid=sup, psi_dim_names=['T'], print_progress=False, warn=False)
        for k in range(n_obs):
            # t_hi_of: observations with a shorter record are fitted on a shorter slice of the supergrid -- batches whose common
            # tau ranges differ in length
            extra = {} if t_hi_of is None else dict(t_hi=t_hi_of(k))
            m = synth.hybrid_measurement(seed=100 + k, jitter=True, n_post=100, nf=31, **extra)
            dmd.add_observation(np.array([float(k)]), (m[0], m[1], m[2]), (m[3], m[4]), group_id='g')
        dmd.fit_all()
    log = []
    cvxopt.solvers.options["_oracle_log"] = log
    with _quiet():
        dmd.resolve_group('g', batch_size=batch_size, overlap=overlap, psi_sort_dims=['T'])
    cvxopt.solvers.options["_oracle_log"] = None
    drts = [dmd.get_fit(i) for i in range(n_obs)]
    sp = drts[0].special_qp_params
    out = dict(n_obs=n_obs, batch_size=batch_size, overlap=overlap, n_super=len(sup),
               t_hi=np.array([50.0 if t_hi_of is None else t_hi_of(k) for k in range(n_obs)]),
               obs_tau_indices=np.array(dmd.obs_tau_indices), special_names=np.array(list(sp.keys())),
               special_index=np.array([v["index"] for v in sp.values()]),
               special_size=np.array([v.get("size", 1) for v in sp.values()]),
               special_nonneg=np.array([v["nonneg"] for v in sp.values()]),
               p_matrix=_pad_stack([d.fit_parameters["p_matrix"] for d in drts]),
               q_vector=_pad_stack([d.fit_parameters["q_vector"] for d in drts]),
               n_params=np.array([len(d.fit_parameters["q_vector"]) for d in drts]),
               v_baseline=np.array([d.fit_parameters["v_baseline"] for d in drts]),
               vz_offset=np.array([d.fit_parameters["vz_offset"] for d in drts]),
               R_inf=np.array([d.fit_parameters["R_inf"] for d in drts]),
               coefficient_scale=np.array([d.coefficient_scale for d in drts]),
               response_signal_scale=np.array([d.response_signal_scale for d in drts]),
               scaled_response_offset=np.array([d.scaled_response_offset for d in drts]),
               v_baseline_scale=np.array([d.v_baseline_scale for d in drts]),
               inductance_scale=np.array([d.inductance_scale for d in drts]),
               obs_x=dmd.obs_x, obs_x_resolved=dmd.obs_x_resolved,
               R_inf_resolved=dmd.obs_special_resolved["R_inf"], inductance_resolved=dmd.obs_special_resolved["inductance"],
               qp_iterations=np.array([l["iterations"] for l in log]))
    np.savez_compressed(os.path.join(OUT, f"refrun_resolve_group_{name}.npz"), **out)
    print(f"resolve_group_{name}: {n_obs} obs, {len(log)} batches, qp iterations {out['qp_iterations'].tolist()}")


def mixed_map_observations(n_obs=16):
    """The 16-observation map of the heterogeneous batch-driver test (tests/test_gpu_mapping.py), as DRTMD.add_observation
    takes it: (chrono_data | None, eis_data) per observation.  Three groups, interleaved: joint chrono + EIS measurements
    (k % 3 == 0), impedance spectra on a 41-point grid 1e5 .. 1 Hz (k % 3 == 1) and on a 36-point grid 1e4 .. 0.1 Hz
    (k % 3 == 2): different data types, different frequency ranges, hence different slices of the tau supergrid."""
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


def outlier_map_observations(n_obs=9):
    """the first observations of the mixed map with gross errors planted in some of them (tests/test_gpu_mapping.py holds the
    same recipe): what DRTMD(fit_kw=dict(remove_outliers=True, outlier_p=0.05)) has to find and drop per observation"""
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


def run_drtmd_outliers(n_obs=9):
    """the reference's own DRTMD with remove_outliers among its fit keywords (every observation through the detection pass and
    the refit of drt1d.py:214-302) on the map above: obs_x, specials, llh / rss, tau slices"""
    from hybdrt.mapping.drtmd import DRTMD
    sup = np.logspace(-8, 4, 121)
    obs = outlier_map_observations(n_obs)
    removed = []
    with _quiet():
        dmd = DRTMD(tau_supergrid=sup, psi_dim_names=['T'], print_progress=False, warn=False,
                    fit_kw=dict(nonneg=True, remove_outliers=True, outlier_p=0.05))
        for k, (chrono, eis) in enumerate(obs):
            dmd.add_observation(np.array([float(k)]), chrono, eis, group_id='g')
            dmd.fit_observation(k)
            d = dmd.drt1d
            removed.append((0 if d.chrono_outlier_index is None else int(np.sum(d.chrono_outlier_index)),
                            0 if d.eis_outlier_index is None else int(np.sum(d.eis_outlier_index))))
    assert dmd.obs_fit_status.all()
    out = dict(n_obs=n_obs, tau_supergrid=sup, obs_x=dmd.obs_x, obs_llh=dmd.obs_llh, obs_rss=dmd.obs_rss,
               obs_tau_indices=np.array(dmd.obs_tau_indices), removed=np.array(removed),
               special_names=np.array(list(dmd.obs_special.keys())))
    for key, val in dmd.obs_special.items():
        out["special_" + key] = np.asarray(val)
    np.savez_compressed(os.path.join(OUT, "refrun_drtmd_outliers9.npz"), **out)
    print(f"drtmd_outliers9: removed (chrono, eis) per observation {removed}")


def run_drtmd_mixed(n_obs=16):
    """the reference's own DRTMD (mapping/drtmd.py:186-329, 1136-1158) on a map that mixes data types and frequency ranges:
    what fit_all records per observation -- obs_x in supergrid slots, every special parameter, obs_llh / obs_rss (with
    DRTMD's default llh_kw / rss_kw: weights='uniform', normalize=True), obs_tau_indices, obs_drt_var."""
    from hybdrt.mapping.drtmd import DRTMD
    sup = np.logspace(-8, 4, 121)
    obs = mixed_map_observations(n_obs)
    with _quiet():
        dmd = DRTMD(tau_supergrid=sup, psi_dim_names=['T'], print_progress=False, warn=False)
        for k, (chrono, eis) in enumerate(obs):
            dmd.add_observation(np.array([float(k)]), chrono, eis, group_id='g')
        dmd.fit_all()
    assert dmd.obs_fit_status.all()
    out = dict(n_obs=n_obs, tau_supergrid=sup, obs_x=dmd.obs_x, obs_llh=dmd.obs_llh, obs_rss=dmd.obs_rss,
               obs_tau_indices=np.array(dmd.obs_tau_indices), obs_drt_var=dmd.obs_drt_var,
               special_names=np.array(list(dmd.obs_special.keys())))
    for key, val in dmd.obs_special.items():
        out["special_" + key] = np.asarray(val)
    np.savez_compressed(os.path.join(OUT, "refrun_drtmd_mixed16.npz"), **out)
    print(f"drtmd_mixed16: {n_obs} obs, specials {list(dmd.obs_special)}, tau slices {sorted(set(map(tuple, dmd.obs_tau_indices)))}")


def run_drtmd_pfrt(n_obs=6):
    """the reference's own DRTMD with fit_type='pfrt' (mapping/drtmd.py:98-100, 1136-1158, 1338-1342: every observation through
    _pfrt_fit_core with the factors logspace(-0.7, 0.7, 11)) on the first observations of the mixed map: obs_x
    (n_obs, 11, supergrid), every special parameter per factor, obs_llh / obs_rss, obs_tau_indices, obs_drt_var."""
    from hybdrt.mapping.drtmd import DRTMD
    sup = np.logspace(-8, 4, 121)
    # (impedance observations only: with a joint observation the reference's DRTMD stops in its own bookkeeping --
    # v_baseline comes back as (11, 1) per observation and obs_special['v_baseline'] was initialised (num, 11), drtmd.py:287)
    obs = [o for o in mixed_map_observations(3 * n_obs) if o[0] is None][:n_obs]
    with _quiet():
        dmd = DRTMD(tau_supergrid=sup, psi_dim_names=['T'], print_progress=False, warn=False, fit_type='pfrt')
        for k, (chrono, eis) in enumerate(obs):
            dmd.add_observation(np.array([float(k)]), chrono, eis, group_id='g')
        dmd.fit_all()
    assert dmd.obs_fit_status.all()
    out = dict(n_obs=n_obs, tau_supergrid=sup, pfrt_factors=np.asarray(dmd.pfrt_factors), obs_x=dmd.obs_x, obs_llh=dmd.obs_llh,
               obs_rss=dmd.obs_rss, obs_tau_indices=np.array(dmd.obs_tau_indices), obs_drt_var=dmd.obs_drt_var,
               special_names=np.array(list(dmd.obs_special.keys())))
    for key, val in dmd.obs_special.items():
        out["special_" + key] = np.asarray(val)
    np.savez_compressed(os.path.join(OUT, "refrun_drtmd_pfrt6.npz"), **out)
    print(f"drtmd_pfrt6: obs_x {dmd.obs_x.shape}, specials { {k: np.shape(v) for k, v in dmd.obs_special.items()} }")


DECIMATE_CASES = [     # (record, keywords of preprocessing.downsample_data)
    ("one_step", dict(method='decimate', prestep_samples=10)),
    ("one_step", dict(method='decimate', prestep_samples=7, decimation_interval=25, decimation_factor=1.5)),
    ("one_step", dict(method='decimate', prestep_samples=10, decimation_interval=5, decimation_factor=3,
                      decimation_max_period=0.02)),
    ("one_step", dict(method='decimate', prestep_samples=10, target_size=400, decimation_factor=1.3)),
    ("two_step", dict(method='decimate', prestep_samples=12, decimation_interval=8)),
    ("two_step", dict(method='decimate', prestep_samples=12, target_size=250, decimation_max_period=0.1)),
    ("two_step", dict(method='decimate', prestep_samples=5, decimation_interval=8, discard_first_n_points=2)),
    ("two_step", dict(method='match', prestep_samples=5, discard_first_n_points=3, discard_only=True)),
    ("one_step", dict(method='match', stepwise_sample_times=False, antialiased=True,
                      target_times=np.concatenate(([0], np.logspace(-3, 0, 40))))),
]


def decimate_records():
    from hipdrt import synth
    return dict(one_step=synth.hybrid_measurement(seed=5, n_pre=200, n_post=6000, uniform_dt=2.5e-4, v_noise=2e-5),
                two_step=synth.hybrid_measurement(seed=6, n_pre=150, n_post=3000, uniform_dt=2.5e-4, v_noise=2e-5,
                                                  extra_steps=((0.9, -1e-3),)))


def run_decimate(DRT, cvxopt):
    """preprocessing.downsample_data with method='decimate' (335-470, 603-689): kept indices for every keyword set of
    DECIMATE_CASES (with and without the anti-alias filter, whose filtered voltages are saved too), and one fit_hybrid run
    that down-samples this way."""
    from hybdrt import preprocessing as rpp
    recs = decimate_records()
    out = {}
    for k, (rec, kw) in enumerate(DECIMATE_CASES):
        times, i_sig, v_sig = recs[rec][:3]
        step_times = times[rpp.identify_steps(i_sig, allow_consecutive=False)]
        for aa in (False, True):
            kw2 = dict(kw)
            kw2.setdefault("antialiased", aa)
            t_s, i_s, v_s, idx = rpp.downsample_data(times, i_sig, v_sig, step_times=step_times, **kw2)
            out[f"case{k}_index_aa{int(aa)}"] = idx
            out[f"case{k}_v_aa{int(aa)}"] = v_s
            out[f"case{k}_i_aa{int(aa)}"] = i_s
        print(f"decimate case {k}: {len(times)} -> {len(idx)} samples")
    np.savez_compressed(os.path.join(OUT, "refrun_decimate.npz"), **out)
    base = dict(fit_inductance=True, fit_capacitance=False, fit_ohmic=True)
    run_hybrid_case(DRT, cvxopt, "hybrid_decimate", recs["one_step"], dict(base, fit_dop=False),
                    dict(downsample=True, downsample_kw=dict(method='decimate', prestep_samples=10, decimation_interval=20,
                                                             decimation_factor=1.5, decimation_max_period=0.05)))


def run_config5(DRT, cvxopt, K=12, v_noise=2e-5, name="refrun_config5_full"):
    """BASELINE configs[4] at FULL size through the reference itself: fit_hybrid with the distribution of phasances, 512
    frequencies + 4096 time samples x 1024 tau (m = 5120 rows, n = 1078 unknowns), the first K outer iterations.  The
    workload is synth.hybrid_measurement(seed=0, n_pre=96, n_post=4000, nf=512, v_noise=2e-5): with 20 uV of voltage noise
    the reference's outer iteration is contractive (step sizes 0.11, 0.03, 0.02, ... 0.01; oracle/probe_c5.py), so the
    iterates are reproducible to rounding and can be pinned tightly -- with the 2 uV of the other fixtures the loop
    wanders (steps grow again after the sixth iteration) and any two implementations drift apart.  The second fixture
    (refrun_config5_2uV: v_noise = 2e-6, the bench's own workload, K = 6) pins the iterations before that happens."""
    from hipdrt import synth
    meas = synth.hybrid_measurement(seed=0, n_pre=96, n_post=4000, nf=512, v_noise=v_noise)
    log = []
    cvxopt.solvers.options["_oracle_log"] = log
    with _quiet():
        drt = DRT(fixed_basis_tau=np.logspace(-7, 3, 1024), fit_dop=True, fit_inductance=True, fit_ohmic=True,
                  fit_capacitance=False)
        drt.fit_hybrid(*meas, max_iter=K)
    cvxopt.solvers.options["_oracle_log"] = None
    fp, qp = drt.fit_parameters, drt.qphb_params
    out = dict(K=K, v_noise=v_noise, qp_iterations=np.array([l["iterations"] for l in log]),
               hist_x=np.array([h["x"] for h in drt.qphb_history]),
               hist_rho=np.array([h["rho_vector"] for h in drt.qphb_history]),
               hist_dop_rho=np.array([h["dop_rho_vector"] for h in drt.qphb_history]),
               x=fp["x"], x_dop=fp["x_dop"], R_inf=fp["R_inf"], inductance=fp["inductance"], v_baseline=fp["v_baseline"],
               vz_offset=fp["vz_offset"], coefficient_scale=drt.coefficient_scale, rm_shape=np.array(qp["rm"].shape),
               rv=qp["rv"], est_weights=qp["est_weights"], xmx_norms=qp["xmx_norms"], dop_xmx_norms=qp["dop_xmx_norms"])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    hx = out["hist_x"]
    steps = np.abs(np.diff(hx, axis=0)).max(axis=1) / np.abs(hx[1:]).max(axis=1)
    print(f"{name}: m x n = {qp['rm'].shape}, qp_iters={out['qp_iterations'].tolist()}, steps {np.round(steps, 3).tolist()}")


def run_posteriors(DRT, freq_g, z_g, default):
    from oracle.drt_oracle import get_basis_tau
    bt = get_basis_tau(freq_g)
    sup = np.logspace(np.log10(bt[0]) - 0.5, np.log10(bt[-1]) + 0.5, 10 * 13 + 1)   # a wider "supergrid"
    run_posterior(DRT, "golden71x91", freq_g, z_g, default, sup)


def main():
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, REPO)
    if "--only-response" in sys.argv:
        run_response_matrices()
        return
    if "--only-config5" in sys.argv:
        DRT, cvxopt = _boot_reference()
        run_config5(DRT, cvxopt)
        return
    if "--only-config5-2uV" in sys.argv:
        DRT, cvxopt = _boot_reference()
        run_config5(DRT, cvxopt, K=int(os.environ.get("C5_K", "6")), v_noise=2e-6, name="refrun_config5_2uV")
        return
    if "--only-decimate" in sys.argv:
        DRT, cvxopt = _boot_reference()
        run_decimate(DRT, cvxopt)
        return
    if "--only-resolve" in sys.argv:
        DRT, cvxopt = _boot_reference()
        run_resolve(DRT, cvxopt, "hybrid7", False)
        run_resolve(DRT, cvxopt, "hybrid7_dop", True)
        run_resolve_group(cvxopt, "hybrid16")
        return
    if "--only-resolve-ranges" in sys.argv:
        # observations 9..15 have a ten times shorter record: three batches with common tau ranges of two lengths
        _, cvxopt = _boot_reference()
        run_resolve_group(cvxopt, "hybrid16_ranges", t_hi_of=lambda k: 50.0 if k < 9 else 5.0)
        return
    if "--only-resolve-c2grid" in sys.argv:
        # 7 joint fits on the 512-point tau grid of BASELINE configs[2]: a coupled QP of 7 x 514 = 3598 unknowns
        from hipdrt import synth
        DRT, cvxopt = _boot_reference()
        run_resolve(DRT, cvxopt, "c2grid", False, basis_tau=synth.config_c2()["tau"], store_p=False)
        return
    if "--only-drtmd-outliers" in sys.argv:
        _boot_reference()
        run_drtmd_outliers()
        return
    if "--only-drtmd-pfrt" in sys.argv:
        _boot_reference()
        run_drtmd_pfrt()
        return
    if "--only-drtmd" in sys.argv:
        _boot_reference()
        run_drtmd_mixed()
        return
    if "--only-hybrid" in sys.argv:
        freq_g, z_g = extract_reference_test_vectors()
        DRT, cvxopt = _boot_reference()
        run_hybrid_cases(DRT, cvxopt, freq_g, z_g)
        return
    if "--only-options" in sys.argv:
        freq_g, z_g = extract_reference_test_vectors()
        DRT, cvxopt = _boot_reference()
        default = dict(fit_inductance=True, fit_capacitance=False, fit_dop=False, fit_ohmic=True)
        run_case(DRT, cvxopt, "golden71x91_outlier", freq_g, z_g, default, dict(outlier_p=0.05), save_mats=False,
                 save_qps=False)
        run_case(DRT, cvxopt, "golden71x91_iw", freq_g, z_g, default, dict(iw_alpha=1.5, iw_beta=0.5), save_mats=False,
                 save_qps=False)
        return
    if "--only-warm-prepared" in sys.argv:
        from hipdrt import synth
        DRT, cvxopt = _boot_reference()
        base = dict(fit_inductance=True, fit_capacitance=False, fit_ohmic=True)
        meas = synth.hybrid_measurement(seed=0)
        run_warm_restarts_prepared(DRT, "hybrid_s0", meas, dict(base, fit_dop=False))
        run_warm_restarts_prepared(DRT, "hybrid_s0_dop", meas, dict(base, fit_dop=True))
        run_warm_restarts_prepared(DRT, "chrono_s1", meas[:3] + (None, None), dict(base, fit_dop=False))
        return
    if "--only-posterior-sneg" in sys.argv:
        from oracle.drt_oracle import get_basis_tau
        freq_g, z_g = extract_reference_test_vectors()
        DRT, cvxopt = _boot_reference()
        bt = get_basis_tau(freq_g)
        run_posterior_sneg(DRT, freq_g, z_g, dict(fit_inductance=True, fit_capacitance=False, fit_dop=False, fit_ohmic=True),
                           np.logspace(np.log10(bt[0]) - 0.5, np.log10(bt[-1]) + 0.5, 10 * 13 + 1))
        return
    if "--only-warm-outlier" in sys.argv:
        # outlier_p in the warm restarts of a joint fit: five factors, the first fit and every restart with it
        from hipdrt import synth
        DRT, cvxopt = _boot_reference()
        base = dict(fit_inductance=True, fit_capacitance=False, fit_ohmic=True)
        run_warm_restarts_prepared(DRT, "hybrid_s0_outlier", synth.hybrid_measurement(seed=0), dict(base, fit_dop=False),
                                   factors=np.logspace(-0.5, 0.5, 5), fit_kw=dict(outlier_p=0.05), candidates=False)
        return
    if "--only-candidates" in sys.argv:
        freq_g, z_g = extract_reference_test_vectors()
        DRT, cvxopt = _boot_reference()
        run_candidates(DRT, "golden71x91", freq_g, z_g, dict(fit_inductance=True, fit_capacitance=False, fit_dop=False,
                                                              fit_ohmic=True))
        return
    if "--only-posterior" in sys.argv:
        freq_g, z_g = extract_reference_test_vectors()
        DRT, cvxopt = _boot_reference()
        run_posteriors(DRT, freq_g, z_g, dict(fit_inductance=True, fit_capacitance=False, fit_dop=False, fit_ohmic=True))
        return
    from hipdrt import synth

    freq_g, z_g = extract_reference_test_vectors()
    DRT, cvxopt = _boot_reference()

    default = dict(fit_inductance=True, fit_capacitance=False, fit_dop=False, fit_ohmic=True)
    # (1) the reference's own test inputs, with all intermediates
    run_case(DRT, cvxopt, "golden71x91", freq_g, z_g, default, {})
    # (1b) optional branches of the weight estimation: outlier down-weighting and the initial-weight prior
    run_case(DRT, cvxopt, "golden71x91_outlier", freq_g, z_g, default, dict(outlier_p=0.05), save_mats=False, save_qps=False)
    run_case(DRT, cvxopt, "golden71x91_iw", freq_g, z_g, default, dict(iw_alpha=1.5, iw_beta=0.5), save_mats=False,
             save_qps=False)
    # (2) same inputs, DRT coefficients allowed negative (h = 1e5 branch of make_h_constraint)
    run_case(DRT, cvxopt, "golden71x91_neg", freq_g, z_g, default, dict(nonneg=False), save_mats=False)
    # (3) C1 variant: 71 freqs x fixed 121-point tau grid, 2-ZARC seed 0
    c1 = synth.config_c1()
    run_case(DRT, cvxopt, "c1_71x121", c1["freq"], synth.zarc2_spectrum(c1["freq"], 0),
             dict(default, fixed_basis_tau=c1["tau"]), {})
    # (4) C2: 256 x 512, seeds 0..2 (matrices sub-sampled, no per-QP P's: size)
    c2 = synth.config_c2()
    for seed in range(3):
        run_case(DRT, cvxopt, f"c2_256x512_s{seed}", c2["freq"], synth.zarc2_spectrum(c2["freq"], seed),
                 dict(default, fixed_basis_tau=c2["tau"]), {}, save_mats=(seed == 0), save_qps=False,
                 row_stride=16)
    # (5) batch members (jittered) of the C3/C4 workload, first 4
    for b in range(4):
        run_case(DRT, cvxopt, f"c3_member{b}", c2["freq"], synth.zarc2_spectrum(c2["freq"], b, jitter=True),
                 dict(default, fixed_basis_tau=c2["tau"]), {}, save_mats=False, save_qps=False)
    # (6) trapz-mode matrices: non-Toeplitz 32 x 64 and Toeplitz 71 x 91
    run_trapz_matrices("trapz_32x64", np.logspace(4, 0, 32), np.logspace(-6, 1, 64),
                       1 / np.mean(np.diff(np.log(np.logspace(-6, 1, 64)))))
    from oracle.drt_oracle import get_basis_tau, get_epsilon_from_ppd
    run_trapz_matrices("trapz_71x91_toeplitz", freq_g, get_basis_tau(freq_g), get_epsilon_from_ppd(10))
    # (7) chrono response lookup + matrices (survey row a3)
    run_response_matrices()
    # (8) post-fit quantities DRTMD stores per observation (survey 8f rank 1)
    run_posteriors(DRT, freq_g, z_g, default)
    # (9) warm restarts / candidate generation (survey 8f rank 3)
    run_candidates(DRT, "golden71x91", freq_g, z_g, default)
    # (10) distribution of phasances inside fit_eis, and joint chrono + EIS fits (config-5 family)
    run_hybrid_cases(DRT, cvxopt, freq_g, z_g)
    # (11) coherent multi-observation re-optimisation (survey 8f rank 2)
    run_resolve(DRT, cvxopt, "hybrid7", False)
    run_resolve(DRT, cvxopt, "hybrid7_dop", True)
    run_resolve_group(cvxopt, "hybrid16")
    run_resolve_group(cvxopt, "hybrid16_ranges", t_hi_of=lambda k: 50.0 if k < 9 else 5.0)
    run_drtmd_mixed()
    run_drtmd_outliers()
    # (12) evaluation of fitted models
    # (13) Kramers-Kronig test
    # (14) progressive decimation of raw chrono records
    run_decimate(DRT, cvxopt)


if __name__ == "__main__":
    main()
