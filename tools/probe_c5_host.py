"""Where the wall time of ONE configs[4] fit (joint chrono + EIS with DOP, 5120 x 1078) goes outside the device loop: cProfile of the
second fit_hybrid call, host functions by cumulative time.   python tools/probe_c5_host.py"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt import synth
from hipdrt.models import DRT

meas = synth.hybrid_measurement(seed=0, n_pre=96, n_post=4000, nf=512)
d5 = DRT(fixed_basis_tau=np.logspace(-7, 3, 1024), fit_dop=True, warn=False)
d5.fit_hybrid(*meas, max_iter=2)
d5.fit_hybrid(*meas)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
d5.fit_hybrid(*meas)
pr.disable()
wall = time.perf_counter() - t0
tm = d5._plan.timings()[0]
print(f"wall {wall:.3f} s, device loop {tm['total'] / 1e3:.3f} s")
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(35)
