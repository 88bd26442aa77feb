"""Diagnostic: BASELINE configs[3]-sized batch (10 000 spectra, 256 x 512) through one plan on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt import synth
from hipdrt.models import DRT

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
c2 = synth.config_c2()
z = synth.zarc2_batch(c2["freq"], B, first_seed=0)
drt = DRT(fixed_basis_tau=c2["tau"])
drt.fit_eis_batch(c2["freq"], z[:64])          # builds the plan matrices (not timed)
t = time.time(); res = drt.fit_eis_batch(c2["freq"], z); dt = time.time() - t
st = res["status"]
print(f"B={B}: {dt:.2f} s -> {B/dt:.0f} fits/s incl. upload/download; converged {np.mean(st == 0):.4f}, "
      f"mean outer {res['outer_iters'].mean():.2f}, max_iter exits {int(np.sum(st == 1))}, failed {int(np.sum(st < 0))}")
t = time.time(); var, ok = drt.estimate_distribution_var_batch(tau=np.logspace(-8.5, 2.5, 544), extend_var=True); dv = time.time() - t
print(f"posterior variance for all {B}: {dv*1e3:.0f} ms, ok {ok.all()}")
