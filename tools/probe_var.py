"""Diagnostic: time the posterior-variance pass (hipdrt_plan_distribution_var) after a 1024-spectrum C3 fit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt import synth
from hipdrt.models import DRT

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
c2 = synth.config_c2()
z = synth.zarc2_batch(c2["freq"], B, first_seed=0)
drt = DRT(fixed_basis_tau=c2["tau"])
t = time.time(); res = drt.fit_eis_batch(c2["freq"], z); t_fit = time.time() - t
sup = np.logspace(-8.5, 2.5, 544)
for rep in range(2):
    t = time.time(); var, ok = drt.estimate_distribution_var_batch(tau=sup, extend_var=True); dt = time.time() - t
    print(f"B={B}: fit {t_fit*1e3:.0f} ms; distribution variance on {len(sup)} points: {dt*1e3:.1f} ms "
          f"({dt/B*1e6:.0f} us per spectrum), ok={ok.all()}")
