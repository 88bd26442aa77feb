"""fits/s of ONE plan, ONE caller thread for k = 1 .. 4 sub-batches inside the fit (hipdrt_plan_set_subbatches) at several batch
sizes -- the single-caller figure of bench.py and the per-rank share of an 8-GPU map (1250 spectra):
python tools/probe_subbatch.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt import synth
from hipdrt.models import DRT

c2 = synth.config_c2()
sizes = [int(a) for a in sys.argv[1:]] or [625, 1024, 1250, 2500, 5000]
for B in sizes:
    z = synth.zarc2_batch(c2["freq"], B)
    drt = DRT(fixed_basis_tau=c2["tau"])
    plan = drt.stage_batch(c2["freq"], z)
    drt.fit_staged()
    out = []
    for k in (1, 2, 3, 4, 6, 0):
        plan.set_subbatches(k)
        drt.fit_staged()
        reps = 3 if B <= 2500 else 2
        t0 = time.perf_counter()
        for _ in range(reps):
            drt.fit_staged()
        dt = (time.perf_counter() - t0) / reps
        out.append(f"k={k if k else 'auto'}: {B / dt:7.1f}")
    print(f"B = {B:5d} spectra, fits/s: " + "  ".join(out), flush=True)
