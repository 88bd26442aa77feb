import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from hipdrt import synth
from hipdrt.models import DRT
c2 = synth.config_c2()
z = synth.zarc2_batch(c2["freq"], 8, first_seed=0)
drt = DRT(fixed_basis_tau=c2["tau"])
drt.fit_eis_batch(c2["freq"], z[:1])
for B in (1, 8):
    t = time.time(); r = drt.fit_eis_batch(c2["freq"], z[:B]); dt = time.time() - t
    print(f"B={B}: {dt*1e3:.1f} ms, outer {r['outer_iters'].tolist()}, timings {r['timings_ms']}")
