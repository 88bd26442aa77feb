"""Diagnostic for tests/test_gpu_hybrid.py::test_randomised_option_combinations_follow_the_oracle: per-iteration distance
between the device loop and the oracle loop for given seeds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt.models import DRT
from hipdrt import synth
from hipdrt.matrices import phasance
from oracle import drt_oracle as orc

for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(2000 + seed)
    freq = np.logspace(rng.uniform(4.5, 6), rng.uniform(-1, 0.5), int(rng.integers(31, 72)))
    z = synth.zarc2_spectrum(freq, seed=seed, jitter=True)
    dop = bool(rng.integers(2))
    kw = dict(max_iter=10, solve_rp=True)
    if rng.integers(2): kw["update_scale"] = True
    if rng.integers(2): kw["weight_factor"] = float(rng.uniform(0.6, 1.6))
    if rng.integers(3) == 0: kw["eff_hp"] = False
    if rng.integers(3) == 0: kw["series_neg"] = True
    if rng.integers(3) == 0: kw["outlier_p"] = float(rng.uniform(0.01, 0.1))
    drt = DRT(fit_dop=dop, warn=False)
    drt.fit_eis(freq, z, **kw)
    qp, special, prep = drt.qphb_params, drt.special_qp_params, drt._prep
    hyp = dict(qp["hypers"]); hyp["eff_hp"] = kw.get("eff_hp", True)
    cs0 = (z.real.max() - z.real.min()) / 14
    rzv0 = np.concatenate([z.real, z.imag]) / cs0
    rzm0 = qp["rm"].copy()
    if dop:
        a, b = prep["dop"]
        scale0 = phasance.phasor_scale_vector(drt.basis_nu, drt.basis_tau) / (np.sqrt(np.pi) / drt.nu_epsilon)
        rzm0[:, a:b] *= scale0 / prep["dop_scale_vector"]
    area = np.sqrt(np.pi) / drt.tau_epsilon
    ref = orc.qphb_fit_prepared(rzm0, rzv0, [qp["penalty_matrices"][f"m{k}"] for k in range(3)], qp["vmm"], special, hyp,
                                max_iter=10, solve_rp=dict(basis_area=area),
                                update_scale=dict(basis_area=area) if kw.get("update_scale") else None,
                                weight_factor=kw.get("weight_factor", 1))
    print("seed", seed, "dop", dop, {k: v for k, v in kw.items() if k not in ("max_iter", "solve_rp")})
    print("  ref qp", [l["iterations"] for l in ref["qp_log"]]); print("  dev qp", qp["qp_iterations"].tolist(), "rp", prep["rp_qp_iterations"])
    hx = np.array([h["x"] for h in ref["history"]]); dx = np.array([h["x"] for h in drt.qphb_history])
    k = min(len(hx), len(dx))
    print("  per-iteration max|dx|/max|x|:", np.array2string(np.abs(dx[:k] - hx[:k]).max(axis=1) / np.abs(hx[:k]).max(axis=1), precision=1))
    print("  scale factor", ref["scale_factor"], "dev cs", drt.coefficient_scale, "expected", cs0 / (ref["scale_factor"] * ref["data_scale"]))
