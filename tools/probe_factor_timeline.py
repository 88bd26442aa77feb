"""Time line of ONE factorisation of the coneqp batch kernel (factor64) under the real load: needs HIPDRT_LIB=.../libhipdrt_prof.so.
Workgroup 0 of the last QP launch of a 1024-spectrum fit (the launch's longest problem under the longest-first order), its last
interior-point iteration: per super column J and wavefront, cycles relative to wavefront 0's start of that super column --
  start of the super column | arrival at barrier (A) | tiles stored (arrival at (B))
and wavefront 0's own sequence (chain a, look-ahead history, look-ahead solve, chain b = arrival at (A), W21 / y, everybody arrived)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt import synth, _ffi
from hipdrt.models import DRT

cfg = synth.config_c2()
z = synth.zarc2_batch(cfg["freq"], 1024)
ctx = _ffi.Context(0)
drt = DRT(fixed_basis_tau=cfg["tau"], context=ctx)
plan = drt.stage_batch(cfg["freq"], z)
plan.set_subbatches(1)
drt.fit_staged()
drt.fit_staged()
ctx.qp_timeline_mean(reset=True)
drt.fit_staged()
mean, cnt = ctx.qp_timeline_mean(reset=True)
tl = ctx.qp_timeline().astype(np.int64)
if not tl.any():
    raise SystemExit("no time line: not a PROFILE build")
if "--last" not in sys.argv:
    print("MEAN over the", cnt, "factorisations of workgroup 0 in one 1024-spectrum fit (--last: the last one only)")
    tl = np.rint(mean).astype(np.int64)
nsup = int(np.count_nonzero(tl[0, :, 0]))
print("super columns", nsup, "| cycles, relative to wavefront 0's start of the super column (k = 1000 cycles)")
k = lambda v: "%6.1f" % (v / 1e3)
for J in range(nsup):
    t0 = tl[0, J, 0]
    w0 = tl[0, J]
    print("J = %d  wavefront 0: chain a done %s | look-ahead history %s | its solve %s | at (A) %s | W21, y %s | all arrived %s | total %s" % (
        J, k(w0[4] - t0), k(w0[5] - t0) if w0[5] else "     -", k(w0[6] - t0) if w0[6] else "     -", k(w0[1] - t0), k(w0[7] - t0), k(w0[3] - t0),
        k((tl[0, J + 1, 0] if J + 1 < nsup else w0[3]) - t0)))
    fat = os.environ.get("HIPDRT_QP_WAVES") == "4"
    for w in range(1, 8):
        r = tl[w, J]
        if not r[0]:
            continue
        if fat and w >= 4:       # fat form, PROFILE build: wavefront 2's panel solves (pass, group) in the rows of the absent wavefronts
            print("       wavefront 2, pass %d group %d: entry %s | solved a +%s | stored, a's update +%s | y_a +%s | rhs +%s | solved b +%s | stored, y_b +%s | rhs +%s" % (
                (w - 4) // 2, (w - 4) % 2, k(r[0] - t0), *[k(r[i] - r[i - 1]) for i in range(1, 8)]))
            continue
        print("       wavefront %d: starts %s | at (A) %s (waits %s) | tiles stored %s (%s behind (A))" % (
            w, k(r[0] - t0), k(r[1] - t0), k(r[2] - r[1]), k(r[3] - t0), k(r[3] - r[2])))
