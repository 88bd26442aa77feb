"""fits/s of mapping.fit_observations (host pre/post-processing inside) over plans in flight x ranges per plan, at the per-rank
share of an 8-GPU map (1250), of a 4-GPU map (2500) and at the whole map (10 000): which split of ONE GPU's spectra over plans
(host threads, own streams) and ranges (one thread, hipdrt_plan_set_subbatches) is fastest under the hardware-queue count in
force.  python tools/probe_inflight_ranges.py [total ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt import synth, mapping
from hipdrt.mapping import drtmd
from hipdrt.models import DRT

c2 = synth.config_c2()
GRID = ((1, 1), (1, 2), (1, 4), (1, 6), (2, 1), (2, 2), (2, 3), (3, 1), (3, 2), (4, 1), (4, 2))
args = [a for a in sys.argv[1:] if not a.startswith("--grid=")]
for a in sys.argv[1:]:
    if a.startswith("--grid="):                      # e.g. --grid=1x1,1x4,2x1
        GRID = tuple(tuple(int(v) for v in g.split("x")) for g in a[7:].split(","))
totals = [int(a) for a in args] or [1250, 2500, 10000]
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
for total in totals:
    z = synth.zarc2_batch(c2["freq"], total)
    drt = DRT(fixed_basis_tau=c2["tau"])
    for plans, ranges in GRID:
        if total // plans < 128:
            continue
        drtmd._RANGES_PER_INFLIGHT_PLAN = ranges
        if plans == 1:
            drt.plan_subbatches = ranges
            if getattr(drt, "_plan", None) is not None:
                drt._plan.set_subbatches(ranges)
        run = lambda: mapping.fit_observations(drt, c2["freq"], z, inflight=plans, tau_supergrid=c2["tau"], drt_var=False)
        run()
        reps = 3 if total <= 2500 else 2
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        dt = (time.perf_counter() - t0) / reps
        print(f"total {total:6d}  plans {plans}  ranges/plan {ranges}: {total / dt:7.1f} fits/s", flush=True)
