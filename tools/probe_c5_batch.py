"""Config-5-sized joint fits as a batch: B measurements (jittered cells, one protocol) through fit_hybrid_batch.
python tools/probe_c5_batch.py [B] [max_iter]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt.models import DRT
from hipdrt import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
max_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 50
meas = [synth.hybrid_measurement(seed=s, jitter=True, n_pre=96, n_post=4000, nf=512) for s in range(B)]
drt = DRT(fixed_basis_tau=np.logspace(-7, 3, 1024), fit_dop=True, warn=False)
args = (meas[0][0], [m[1] for m in meas], [m[2] for m in meas], meas[0][3], [m[4] for m in meas])
t = time.time(); res = drt.fit_hybrid_batch(*args, max_iter=max_iter); dt = time.time() - t
tm = drt._plan.timings()[0]
print(f"config 5 x {B}: wall {dt:.1f} s (host prep + transfers + fit), device loop {tm['total']/1e3:.2f} s "
      f"(gram {tm['gram']/1e3:.2f}, qp {tm['qp']/1e3:.2f}, hyper {tm['hyper']/1e3:.2f}) -> {B/(tm['total']/1e3):.2f} fits/s on the device; "
      f"outer iterations {res['outer_iters'].tolist()}")
