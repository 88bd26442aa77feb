"""Throughput of batched fits that take the prepared-plan route: EIS + DOP at the config-2 grid (n = 564, tile-packed QP kernel with its inverse diagonal blocks in global memory
kernel) and joint chrono + EIS fits.  python tools/probe_dop_batch.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt.models import DRT
from hipdrt import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
c2 = synth.config_c2()
zb = synth.zarc2_batch(c2["freq"], B)
for dop in (False, True):
    drt = DRT(fixed_basis_tau=c2["tau"], fit_dop=dop, warn=False)
    drt.fit_eis_batch(c2["freq"], zb[:8])
    t = time.time(); res = drt.fit_eis_batch(c2["freq"], zb); dt = time.time() - t
    tm = drt._plan.timings()[0]
    print(f"EIS 256x512 fit_dop={dop}: B={B} wall {dt:.2f} s -> {B/dt:.1f} fits/s; device {tm}; mean outer {res['outer_iters'].mean():.1f} "
          f"converged {np.mean(res['status'] == 0):.2f}")
meas = [synth.hybrid_measurement(seed=s, jitter=True) for s in range(min(B, 64))]
for dop in (False, True):
    drt = DRT(fit_dop=dop, warn=False)
    args = (meas[0][0], [m[1] for m in meas], [m[2] for m in meas], meas[0][3], [m[4] for m in meas])
    drt.fit_hybrid_batch(*args)
    t = time.time(); res = drt.fit_hybrid_batch(*args); dt = time.time() - t
    print(f"hybrid 224 t + 41 f fit_dop={dop}: B={len(meas)} wall {dt:.2f} s -> {len(meas)/dt:.1f} fits/s; device {drt._plan.timings()[0]}; "
          f"mean outer {res['outer_iters'].mean():.1f}")
