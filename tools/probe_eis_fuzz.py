"""Diagnostic for EIS fuzz seeds that fail tools/fuzz_parity.py --eis: prints, per spectrum, outer iterations, total QP
iterations, the coefficient difference relative to the peak, and how far the oracle itself moves when its input is
perturbed by 1e-13 (needs a GPU).  python tools/probe_eis_fuzz.py seed [seed ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import warnings
warnings.filterwarnings("ignore")
import numpy as np
from hipdrt.models import DRT
from oracle import drt_oracle as orc

for seed in map(int, sys.argv[1:]):
    rng = np.random.default_rng(1000 + seed)
    nf = int(rng.integers(30, 90))
    f_hi, f_lo = 10 ** rng.uniform(4, 6.5), 10 ** rng.uniform(-2, 0.5)
    freq = np.logspace(np.log10(f_hi), np.log10(f_lo), nf)
    ppd = int(rng.choice([6, 8, 10, 12]))
    nonneg = bool(rng.random() < 0.75)
    err = None if rng.random() < 0.7 else 'uniform'
    z = []
    for b in range(4):
        r_inf, r1, r2 = rng.uniform(0.1, 5), rng.uniform(0.2, 3), rng.uniform(0.1, 2)
        t1, t2 = 10 ** rng.uniform(-5, -2), 10 ** rng.uniform(-2, 0.5)
        b1, b2 = rng.uniform(0.6, 1.0), rng.uniform(0.6, 1.0)
        w = 2j * np.pi * freq
        zz = r_inf + r1 / (1 + (w * t1) ** b1) + r2 / (1 + (w * t2) ** b2) + w * 10 ** rng.uniform(-8, -6)
        sig = 10 ** rng.uniform(-4, -2)
        z.append(zz + sig * np.abs(zz) * (rng.standard_normal(nf) + 1j * rng.standard_normal(nf)))
    z = np.array(z)
    drt = DRT(basis_tau_ppd=ppd)
    res = drt.fit_eis_batch(freq, z, eis_error_structure=err, nonneg=nonneg)
    print(f"seed {seed}: nf={nf} ppd={ppd} nonneg={nonneg} err={err}")
    for b in range(4):
        od = orc.OracleDRT(basis_tau_ppd=ppd)
        od.fit_eis(freq, z[b], error_structure=err, keep_history=True, nonneg=nonneg)
        xo = od.qphb_params["x_scaled"]
        od2 = orc.OracleDRT(basis_tau_ppd=ppd)
        od2.fit_eis(freq, z[b] * (1 + 1e-13), error_structure=err, keep_history=True, nonneg=nonneg)
        x2 = od2.qphb_params["x_scaled"]
        sens = np.abs(x2 - xo).max() / np.abs(xo).max() if len(od2.qphb_history) == len(od.qphb_history) else float('nan')
        d = np.abs(res["x"][b] - xo).max() / np.abs(xo).max()
        its = [l["iterations"] for l in od.qp_log]
        print(f"  b={b} outer dev/orc/orc' {res['outer_iters'][b]}/{len(od.qphb_history)}/{len(od2.qphb_history)}  qp {res['qp_iters_total'][b]}/{sum(its)}"
              f"  dx {d:.2e}  oracle-sens {sens:.2e}  zero-iter QPs {sum(1 for i in its if i == 0)}/{len(its)}")
