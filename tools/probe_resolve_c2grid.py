"""Diagnostic: the numbers behind tests/test_resolve.py::test_device_resolve_c2grid_3598_unknowns (device fits vs the
reference's fits on the 512-point grid, coupled QP vs the CPU checker and vs the reference's resolve), with timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt.models import DRT
from hipdrt.mapping import resolve
from hipdrt import synth
from oracle import resolve_oracle as ro
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "refrun_resolve_c2grid.npz"))
nt, nobs = int(g["ntau"]), int(g["n_obs"])
drts = []
t0 = time.time()
for s_ in range(nobs):
    d = DRT(fixed_basis_tau=g["basis_tau"], warn=False)
    d.fit_hybrid(*synth.hybrid_measurement(seed=s_, jitter=True, n_post=120, nf=31))
    drts.append(d)
    print("fit", s_, "outer iterations", d.qphb_params.get("outer_iterations") if hasattr(d, "qphb_params") else None, flush=True)
print("7 device fits: %.2f s" % (time.time() - t0))
x_fit = np.array([d.fit_parameters["x"] for d in drts])
print("fits vs reference: max |dx| / peak per observation", (np.abs(x_fit - g["x_fit"]).max(axis=1) / np.abs(g["x_fit"]).max(axis=1)).tolist())
print("coefficient_scale rel err", np.abs(np.array([d.coefficient_scale for d in drts]) / g["coefficient_scale"] - 1).max())
t0 = time.time(); x, match = resolve.resolve_observations(drts, [(0, nt)] * nobs, True); t1 = time.time() - t0
t0 = time.time(); x, match = resolve.resolve_observations(drts, [(0, nt)] * nobs, True); t2 = time.time() - t0
print("device resolve: %.2f s (first), %.2f s; qp" % (t1, t2), resolve.resolve_observations.last_qp)
special = drts[0].special_qp_params
obs = [dict(p_matrix=d.fit_parameters["p_matrix"], q_vector=d.fit_parameters["q_vector"], v_baseline=d.fit_parameters["v_baseline"],
            vz_offset=d.fit_parameters["vz_offset"], R_inf=d.fit_parameters["R_inf"], coefficient_scale=d.coefficient_scale,
            response_signal_scale=d.response_signal_scale, scaled_response_offset=d.scaled_response_offset,
            v_baseline_scale=d.v_baseline_scale) for d in drts]
t0 = time.time(); xo, res, (P, q, h) = ro.resolve_observations(obs, special); t3 = time.time() - t0
print("CPU checker: %.2f s, iterations %d; max |x - x_cpu| / peak %.2e" % (t3, res["iterations"], np.abs(x - xo).max() / np.abs(xo).max()))
print("reference: iterations", g["qp_iterations"].tolist(), "max |x - x_ref| / peak %.2e" % (np.abs(x - g["x_opt"]).max() / np.abs(g["x_opt"]).max()),
      "diag P rel %.2e" % np.abs(np.diag(P) / g["qp0_P_diag"] - 1).max(), "q %.2e" % (np.abs(q - g["qp0_q"]).max() / np.abs(g["qp0_q"]).max()),
      "h equal", bool(np.array_equal(h, g["qp0_h"])))
x2, _ = resolve.resolve_observations(drts, [(0, nt)] * nobs, True, sigma=2, lambda_psi=10)
print("sigma=2 lambda=10: qp", resolve.resolve_observations.last_qp, "max |x - x_ref| / peak %.2e" % (np.abs(x2 - g["x_opt_sigma2_lambda10"]).max() / np.abs(g["x_opt"]).max()))
