"""Deviation of the device fit from the CPU checker over the draws of the randomised EIS differential test, as numbers
instead of pass / fail: per draw the largest |x - x_oracle| / peak over its 4 spectra and whether the outer and
interior-point iteration counts agree.  Used to compare two builds of the library on one box (HIPDRT_LIB selects the
build), e.g. the P x recurrence of the coneqp kernel against the direct product (-DHIPDRT_QP_MATVEC).
    python tools/probe_px_recur.py [first_seed] [count] [--perturb]
--perturb also reports how far the CHECKER itself moves when its input is scaled by 1 + 1e-13 (the noise floor of the
draw: fits whose outer iteration is not contractive have no reproducible answer below it)."""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")
import numpy as np  # noqa: E402

from hybrid_util import random_eis_problem  # noqa: E402
from hipdrt.models import DRT  # noqa: E402
from oracle import drt_oracle as orc  # noqa: E402  (checker)

args = [a for a in sys.argv[1:] if not a.startswith("--")]
first = int(args[0]) if args else 0
count = int(args[1]) if len(args) > 1 else 12
perturb = "--perturb" in sys.argv

worst = 0.0
for seed in range(first, first + count):
    freq, z, ppd, err, kw = random_eis_problem(seed)
    drt = DRT(basis_tau_ppd=ppd)
    res = drt.fit_eis_batch(freq, z, eis_error_structure=err, **kw)
    devs, same, floor = [], [], []
    for b in range(4):
        od = orc.OracleDRT(basis_tau_ppd=ppd)
        od.fit_eis(freq, z[b], error_structure=err, keep_history=True, **kw)
        xo = od.qphb_params["x_scaled"]
        devs.append(float(np.max(np.abs(res["x"][b] - xo)) / np.abs(xo).max()))
        same.append(bool(res["outer_iters"][b] == len(od.qphb_history) and
                         res["qp_iters_total"][b] == sum(l["iterations"] for l in od.qp_log)))
        if perturb:
            o2 = orc.OracleDRT(basis_tau_ppd=ppd)
            o2.fit_eis(freq, z[b] * (1 + 1e-13), error_structure=err, keep_history=True, **kw)
            x2 = o2.qphb_params["x_scaled"]
            floor.append(float(np.max(np.abs(x2 - xo)) / np.abs(xo).max()) if len(x2) == len(xo) else float("nan"))
    worst = max(worst, max(devs))
    print(f"seed {seed:4d} nonneg={kw['nonneg']!s:5} err={err!s:7} dev " + " ".join(f"{d:.1e}" for d in devs) +
          " iters_equal " + "".join("y" if s_ else "N" for s_ in same) +
          (" oracle_floor " + " ".join(f"{d:.1e}" for d in floor) if perturb else ""), flush=True)
print(f"worst deviation {worst:.2e}")
