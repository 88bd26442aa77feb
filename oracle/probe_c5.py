"""Test infrastructure (runs only in the build container, imports /root/reference through oracle/refshim): looks for a
BASELINE configs[4]-sized joint fit (512 f + 4096 t x 1024 tau, DOP) whose outer iteration is contractive, so that a
full-size reference-run fixture can pin the device loop tightly.  For every candidate workload the REFERENCE's
fit_hybrid runs K outer iterations; printed: seconds, interior-point counts, and the size of every outer step
max|x_k - x_(k-1)| / max|x_k| (a contractive loop shrinks it steadily).
    python oracle/probe_c5.py [K] [candidate ...]"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.make_golden import _boot_reference, _quiet  # noqa: E402

CANDIDATES = {
    "base": (dict(seed=0), {}),
    "noisy": (dict(seed=0, v_noise=2e-5), {}),
    "solverp": (dict(seed=0), dict(solve_rp=True)),
    "quiet": (dict(seed=0, v_noise=2e-7), {}),
    "seed3": (dict(seed=3), {}),
    "novz": (dict(seed=0), dict(vz_offset=False)),
}


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    names = sys.argv[2:] or list(CANDIDATES)
    DRT, cvxopt = _boot_reference()
    from hipdrt import synth
    for name in names:
        mkw, fkw = CANDIDATES[name]
        meas = synth.hybrid_measurement(n_pre=96, n_post=4000, nf=512, **mkw)
        log = []
        cvxopt.solvers.options["_oracle_log"] = log
        t0 = time.time()
        with _quiet():
            drt = DRT(fixed_basis_tau=np.logspace(-7, 3, 1024), fit_dop=True, fit_inductance=True, fit_ohmic=True)
            drt.fit_hybrid(*meas, max_iter=K, **fkw)
        cvxopt.solvers.options["_oracle_log"] = None
        hx = np.array([h["x"] for h in drt.qphb_history])
        steps = np.abs(np.diff(hx, axis=0)).max(axis=1) / np.abs(hx[1:]).max(axis=1)
        print(f"{name}: {time.time() - t0:.0f} s, outer {len(hx)}, ipm {[l['iterations'] for l in log]}")
        print("   step sizes", np.array2string(steps, precision=2), "R_inf", float(drt.fit_parameters["R_inf"]), flush=True)


if __name__ == "__main__":
    main()
