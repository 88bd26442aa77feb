"""Where the HOST spends its time in one configs[3] map (10 000 spectra through mapping.fit_observations_sharded): cProfile of the third map.
python tools/profile_map_host.py [total]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt import synth
from hipdrt.mapping import fit_observations_sharded
from hipdrt.models import DRT

total = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
c2 = synth.config_c2()
z = synth.zarc2_batch(c2["freq"], total)
drt = DRT(fixed_basis_tau=c2["tau"])
run = lambda: fit_observations_sharded(drt, c2["freq"], z, rank=0, world=1)
run(); run()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable(); run(); pr.disable()
dt = time.perf_counter() - t0
print(f"one map of {total}: {dt:.3f} s = {total / dt:.1f} fits/s; device loop {drt._plan.timings()[0]['total'] / 1e3:.3f} s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
