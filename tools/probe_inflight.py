"""Diagnostic: mapping.fit_observations throughput (upload + fit + llh/rss + download) against the number of batches in flight:
python tools/probe_inflight.py [num ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt import synth
from hipdrt.mapping import fit_observations
from hipdrt.models import DRT

c2 = synth.config_c2()
freq, tau = c2["freq"], c2["tau"]
for num in [int(a) for a in sys.argv[1:]] or [1024, 4096]:
    z = synth.zarc2_batch(freq, num, first_seed=0)
    for k in (1, 2, 3, 4):
        drt = DRT(fixed_basis_tau=tau)
        fit_observations(drt, freq, z, inflight=k)           # plans, tables
        t0 = time.perf_counter()
        fit_observations(drt, freq, z, inflight=k)
        dt = time.perf_counter() - t0
        print(f"num {num} inflight {k}: {num / dt:8.1f} fits/s ({dt:.3f} s)", flush=True)
