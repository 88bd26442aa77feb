"""Where a device fit and the CPU checker part ways (diagnostic for a fuzz finding; needs a GPU): per outer iteration the
coefficient difference relative to the peak, for one spectrum of the bench's workload (`c2 <seed>`) or one member of a random EIS
draw (`eis <seed> <b>`).  Also how far the CHECKER itself moves when its input is perturbed by 1e-13.
python tools/probe_divergence.py c2 3694 | eis 5111 2"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import warnings
warnings.filterwarnings("ignore")
import numpy as np
from hipdrt import synth
from hipdrt.models import DRT
from oracle import drt_oracle as orc

kind, seed = sys.argv[1], int(sys.argv[2])
if kind == "c2":
    c2 = synth.config_c2()
    freq, z = c2["freq"], synth.zarc2_batch(c2["freq"], 1, first_seed=seed)[0]
    mk_d = lambda: DRT(fixed_basis_tau=c2["tau"])
    mk_o = lambda: orc.OracleDRT(fixed_basis_tau=c2["tau"])
    kw_d, kw_o = {}, {}
else:
    from hybrid_util import random_eis_problem
    freq, zb, ppd, err, kw = random_eis_problem(seed)
    z = zb[int(sys.argv[3])]
    mk_d = lambda: DRT(basis_tau_ppd=ppd)
    mk_o = lambda: orc.OracleDRT(basis_tau_ppd=ppd)
    kw_d, kw_o = dict(eis_error_structure=err, **kw), dict(error_structure=err, **kw)
    print(f"nf={len(freq)} ppd={ppd} err={err} {kw}")
drt = mk_d()
drt.fit_eis(freq, z, **kw_d)
od = mk_o()
od.fit_eis(freq, z, keep_history=True, **kw_o)
od2 = mk_o()
od2.fit_eis(freq, z * (1 + 1e-13), keep_history=True, **kw_o)
hd, ho, h2 = drt.qphb_history, od.qphb_history, od2.qphb_history
its, its2 = [l["iterations"] for l in od.qp_log], [l["iterations"] for l in od2.qp_log]
print(f"outer iterations device / checker / perturbed checker: {len(hd)} / {len(ho)} / {len(h2)};  ipm totals {drt.qphb_params.get('qp_iters_total', '?')} / {sum(its)} / {sum(its2)}")
print("it   |x_dev - x_chk|/peak   |x_chk' - x_chk|/peak   ipm chk / chk'")
for i in range(min(len(hd), len(ho))):
    pk = np.abs(ho[i]["x"]).max()
    d = np.abs(np.asarray(hd[i]["x"]) - ho[i]["x"]).max() / pk
    s = np.abs(h2[i]["x"] - ho[i]["x"]).max() / pk if i < len(h2) else float("nan")
    a = its[i + 1] if i + 1 < len(its) else -1
    b = its2[i + 1] if i + 1 < len(its2) else -1
    print(f"{i:3d}   {d:9.2e}   {s:9.2e}   {a} / {b}")
