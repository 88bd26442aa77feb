"""Where the wall time of one rank's share of a map goes (mapping.fit_observations_sharded on one rank, `total` spectra of the
configs[2] grids): cProfile of the second map (plans and lookup tables warm), host functions by cumulative time, next to the
device loop's own time.   python tools/probe_share.py [total] [inflight]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from hipdrt import synth
from hipdrt.mapping import fit_observations_sharded
from hipdrt.models import DRT

total = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
inflight = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c2 = synth.config_c2()
z = synth.zarc2_batch(c2["freq"], total)
drt = DRT(fixed_basis_tau=c2["tau"])
fit_observations_sharded(drt, c2["freq"], z, rank=0, world=1, inflight=inflight)
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
out = fit_observations_sharded(drt, c2["freq"], z, rank=0, world=1, inflight=inflight)
pr.disable()
wall = time.perf_counter() - t0
tm, _ = drt._plan.timings()
print(f"{total} spectra, inflight {inflight}: map {wall * 1e3:.1f} ms = {total / wall:.0f} fits/s; device loop of the first plan {tm['total']:.1f} ms")
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
