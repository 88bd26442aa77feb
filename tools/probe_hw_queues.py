"""How the placement of the plans' HIP streams on the runtime's hardware queues moves the configs[3] map (three plans in flight):
K idle contexts (= K streams) are created first, so the map's three streams land on queues K, K+1, K+2 (mod GPU_MAX_HW_QUEUES).
python tools/probe_hw_queues.py K [total]   (set GPU_MAX_HW_QUEUES in the environment to compare 4 and 8)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt import _ffi, synth
from hipdrt.models import DRT
from hipdrt.mapping import fit_observations_sharded
K = int(sys.argv[1]); total = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
cfg = synth.config_c2()
z = synth.zarc2_batch(cfg["freq"], total)
dummies = [_ffi.Context(0) for _ in range(K)]
drt = DRT(fixed_basis_tau=cfg["tau"], context=_ffi.Context(0))
run = lambda: fit_observations_sharded(drt, cfg["freq"], z, rank=0, world=1, inflight='auto')
run()
t0 = time.perf_counter(); run(); run(); el = time.perf_counter() - t0
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} idle streams first {K}: {2 * total / el:.1f} fits/s")
