"""The mapping pipeline on the device: B joint fits of one protocol as a batch, then DRTMD-style resolve_group over all of
them (overlapping batches of 7 as one batched QP launch).  python tools/probe_mapping.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt.models import DRT
from hipdrt.mapping import resolve
from hipdrt import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
meas = [synth.hybrid_measurement(seed=s, jitter=True, n_post=100, nf=31) for s in range(B)]
drt = DRT(warn=False)
args = (meas[0][0], [m[1] for m in meas], [m[2] for m in meas], meas[0][3], [m[4] for m in meas])
drt.fit_hybrid_batch(*args)
t = time.time(); res = drt.fit_hybrid_batch(*args); t_fit = time.time() - t
t = time.time(); fits = drt.batch_fits(); t_views = time.time() - t
nt = len(drt.basis_tau)
t = time.time(); x_res, sp = resolve.resolve_group(fits, [(0, nt)] * B, True, nt); t_res = time.time() - t
print(f"{B} joint fits: {t_fit:.2f} s ({B / t_fit:.0f} fits/s wall, device {drt._plan.timings()[0]['total']/1e3:.2f} s); "
      f"p_matrix views {t_views:.2f} s; resolve_group ({len(resolve.resolve_group.last_qp['iterations'])} coupled QPs of "
      f"{7 * (nt + 2)} unknowns in one launch): {t_res:.2f} s; |x_resolved - x| max rel "
      f"{np.abs(x_res - res['x']).max() / np.abs(res['x']).max():.3f}")
