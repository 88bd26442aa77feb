"""BUILD-CONTAINER ONLY (reads /root/reference): re-runs the reference itself with ``cvxopt.solvers.qp`` routed to the SECOND
restatement (oracle/coneqp_general.py: coneqp + kkt_chol2 for a general dense G, handed the reference's own G) and compares
with the committed fixtures, which oracle/make_golden.py generated through the FIRST one (oracle/coneqp.py, G = -I
specialisation).  Where real cvxopt pins the trajectory directly (the reference's known-answer test, 71 x 91) both agree with
it; at the sizes the bench runs (n = 514, 1078, 3598), with nonneg=False and with DOP, the two restatements check each other:
the same interior-point iteration count for every QP of every run and the same x to rounding.

    python -m oracle.check_general_shim            -> tests/golden/general_shim_check.json (committed; tests/test_oracle_general.py reads it)
"""
import json
import os
import sys
import time

import numpy as np

os.environ["ORACLE_CVXOPT_SHIM"] = "general"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")

from oracle import make_golden as mg      # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def eis_case(DRT, cvxopt, name, ctor_kw=None, fit_kw=None):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    log = []
    cvxopt.solvers.options["_oracle_log"] = log
    ctor = dict(ctor_kw or {})
    from hybdrt.models import DRT as _D      # noqa: F401
    if len(g["basis_tau"]) != 91 or name.startswith("refrun_c"):
        ctor["fixed_basis_tau"] = g["basis_tau"]
    t0 = time.time()
    with mg._quiet():
        drt = DRT(**ctor)
        drt.fit_eis(g["freq"], g["z"], nonneg=bool(g["nonneg"]), **(fit_kw or {}))
    cvxopt.solvers.options["_oracle_log"] = None
    out = dict(n=int(log[0]["q"].size), qps=len(log), iterations=[int(l["iterations"]) for l in log],
               fixture_iterations=[int(v) for v in g["qp_iterations"]],
               x_rel=rel(drt.fit_parameters["x"], g["x"]), seconds=round(time.time() - t0, 1))
    if "qp0_x" in g.files:
        out["qp_x_rel_max"] = max(rel(log[i]["x"], g[f"qp{i}_x"]) for i in range(len(log)) if f"qp{i}_x" in g.files)
    return out


def config5_case(DRT, cvxopt, name="refrun_config5_full", v_noise=2e-5):
    from hipdrt import synth
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    K = int(g["K"])
    meas = synth.hybrid_measurement(seed=0, n_pre=96, n_post=4000, nf=512, v_noise=v_noise)
    log = []
    cvxopt.solvers.options["_oracle_log"] = log
    t0 = time.time()
    with mg._quiet():
        drt = DRT(fixed_basis_tau=np.logspace(-7, 3, 1024), fit_dop=True, fit_inductance=True, fit_ohmic=True, fit_capacitance=False)
        drt.fit_hybrid(*meas, max_iter=K)
    cvxopt.solvers.options["_oracle_log"] = None
    hx = np.array([h["x"] for h in drt.qphb_history])
    return dict(n=int(log[0]["q"].size), qps=len(log), iterations=[int(l["iterations"]) for l in log],
                fixture_iterations=[int(v) for v in g["qp_iterations"]], x_rel=rel(drt.fit_parameters["x"], g["x"]),
                hist_x_rel=rel(hx, g["hist_x"]), seconds=round(time.time() - t0, 1))


def resolve_case(DRT, cvxopt, name="refrun_resolve_c2grid"):
    """the seven hybrid fits of the fixture on the 512-point grid, then hybdrt's own resolve_observations: one QP of 3598 unknowns"""
    from hipdrt import synth
    from hybdrt.mapping import resolve as ref_resolve
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n_obs, nt = int(g["n_obs"]), int(g["ntau"])
    t0 = time.time()
    fits = []
    with mg._quiet():
        for s_ in range(n_obs):
            d = DRT(fixed_basis_tau=g["basis_tau"], fit_dop=False)
            d.fit_hybrid(*synth.hybrid_measurement(seed=s_, jitter=True, n_post=120, nf=31))
            fits.append(d)
    log = []
    cvxopt.solvers.options["_oracle_log"] = log
    with mg._quiet():
        x_opt, _ = ref_resolve.resolve_observations(fits, [(0, nt)] * n_obs, True)
    cvxopt.solvers.options["_oracle_log"] = None
    return dict(n=int(log[0]["q"].size), qps=len(log), iterations=[int(l["iterations"]) for l in log],
                fixture_iterations=[int(g["qp_iterations"][0])], x_fit_rel=rel([d.fit_parameters["x"] for d in fits], g["x_fit"]),
                x_rel=rel(x_opt, g["x_opt"]), seconds=round(time.time() - t0, 1))


def main():
    DRT, cvxopt = mg._boot_reference()
    res = {}
    for name in ("refrun_golden71x91", "refrun_golden71x91_neg", "refrun_c2_256x512_s0", "refrun_c2_256x512_s1", "refrun_c2_256x512_s2"):
        res[name] = eis_case(DRT, cvxopt, name)
        print(name, res[name], flush=True)
    res["refrun_config5_full"] = config5_case(DRT, cvxopt)
    print("refrun_config5_full", res["refrun_config5_full"], flush=True)
    res["refrun_resolve_c2grid"] = resolve_case(DRT, cvxopt)
    print("refrun_resolve_c2grid", res["refrun_resolve_c2grid"], flush=True)
    with open(os.path.join(GOLDEN, "general_shim_check.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
