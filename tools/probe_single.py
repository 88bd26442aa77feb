"""Diagnostic: BASELINE configs[1] (one spectrum 256 x 512) and configs[4] (one joint fit, 5120 x 1078) with the coneqp launches
on the one-workgroup batch kernel (group 0) and on the group kernel with the given member counts.
python tools/probe_single.py [members ...]   (default: 0 1 2 4 8 -1; -1 = the library's own choice)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt import synth, _ffi
from hipdrt.models import DRT

ctx = _ffi.get_context(0)
c2 = synth.config_c2()
z1 = synth.zarc2_batch(c2["freq"], 1, first_seed=0)
meas = synth.hybrid_measurement(seed=0, n_pre=96, n_post=4000, nf=512)
groups = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 8, -1]
ref1 = ref5 = None
for G in groups:
    ctx.debug_qp_group(G)
    d = DRT(fixed_basis_tau=c2["tau"])
    d.fit_eis_batch(c2["freq"], z1)
    t0 = time.perf_counter(); r = d.fit_eis_batch(c2["freq"], z1); t1 = time.perf_counter() - t0
    tm = d._plan.timings()[0]
    if ref1 is None:
        ref1 = r["x"].copy()
    print(f"configs[1] group {G}: {t1 * 1e3:.1f} ms wall, device loop {tm['total']:.1f} ms (qp {tm['qp']:.1f}), outer {int(r['outer_iters'][0])}, "
          f"ipm {int(r['qp_iters_total'][0])}, max |dx|/peak vs first {np.abs(r['x'] - ref1).max() / np.abs(ref1).max():.1e}", flush=True)
for G in groups:
    if G == 0 or G == 1 and len(groups) > 3:
        pass
    ctx.debug_qp_group(G)
    d5 = DRT(fixed_basis_tau=np.logspace(-7, 3, 1024), fit_dop=True, warn=False)
    d5.fit_hybrid(*meas, max_iter=2)
    t0 = time.perf_counter(); fp = d5.fit_hybrid(*meas); t5 = time.perf_counter() - t0
    tm = d5._plan.timings()[0]
    x = np.asarray(d5.cvx_result["x"])
    if ref5 is None:
        ref5 = x.copy()
    print(f"configs[4] group {G}: {t5:.3f} s wall, device loop {tm['total'] / 1e3:.3f} s (qp {tm['qp'] / 1e3:.3f}, gram {tm['gram'] / 1e3:.3f}, hyper {tm['hyper'] / 1e3:.3f}), outer "
          f"{int(d5.qphb_params['outer_iterations'])}, max |dx|/peak vs first {np.abs(x - ref5).max() / np.abs(ref5).max():.1e}", flush=True)
ctx.debug_qp_group(-1)
