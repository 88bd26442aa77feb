"""Diagnostic: phase split of hyper_kernel's workgroup 0 on the C3 batch (HIPDRT_LIB=.../libhipdrt_prof.so, PROFILE=1 build):
python tools/probe_hyper.py [B] [max_iter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt import synth
from hipdrt.models import DRT

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c2 = synth.config_c2()
freq, tau = c2["freq"], c2["tau"]
drt = DRT(fixed_basis_tau=tau)
z = synth.zarc2_batch(freq, B, first_seed=0)
drt.fit_eis_batch(freq, z, max_iter=iters)
ctx = drt._plan.ctx
ctx.qp_profile(reset=True)
drt.fit_eis_batch(freq, z, max_iter=iters)
t = drt._plan.timings()
prof = ctx.qp_profile(reset=True)[48:]
names = ["setup", "solve_s/rho k=0", "k=1", "k=2", "xmx (first pass)", "estimate_weights", "convergence + rescale", "-",
         "  rm @ x", "  vmm @ r^2"]
launches = max(prof[15], 1)
print(f"B={B} outer iterations {iters}: plan timings {t}")
print(f"hyper_kernel launches seen by workgroup 0: {launches}")
tot = sum(prof[:7])
for i, nm in enumerate(names):
    if nm != "-":
        print(f"  {nm:24s} {prof[i] / launches:12.0f} ticks/launch  {prof[i] / max(tot, 1):6.3f}")
print(f"  total {tot / launches:.0f} ticks/launch")
