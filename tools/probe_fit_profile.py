"""In-kernel phase breakdown of the coneqp kernel in the REAL workload (BASELINE configs[2]: 1024 desynchronised spectra,
one plan): needs HIPDRT_LIB=.../libhipdrt_prof.so.  The counters come from workgroup 0 of every launch, which under the
longest-first dispatch order runs the launch's longest problem, i.e. the CU that sets the launch time."""
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
drt.fit_staged()
ctx.qp_profile(reset=True)
drt.fit_staged()
prof = ctx.qp_profile(reset=True)
tms, launches = plan.timings()
names = ["-", "wait_rank_k", "-", "w21_fwd", "wait_stores", "fwd_diag", "fwd_upd", "bwd_diag", "bwd_upd", "matvec", "total"]
tot = prof[10]
print("timings ms", {k: round(v, 1) for k, v in tms.items()}, "qp launches", launches["qp"])
print("WG0 ticks total", tot, "factorisations", prof[11], "ticks per factorisation", tot // max(prof[11], 1))
print("shares:", {n: round(prof[i] / tot, 3) for i, n in enumerate(names) if n != "-"})
if prof[13]:
    print("shader clock while WG0 ran: %.2f GHz (s_memtime ticks / 100 MHz s_memrealtime ticks)" % (tot / prof[13] * 0.1))
chain = prof[12] + prof[14] + prof[15]
print("diagonal chain", round(chain / tot, 3), "(cholinv1", prof[12], "l21+d2", prof[14], "cholinv2", prof[15], ")",
      "=> rank-k phase", round((chain + prof[1]) / tot, 3))
# factor64 (round 4): wavefront 0's timeline per super column -- chain a, look-ahead history of block b, its solve, chain b, barrier (A),
# W21 / y behind (A), barrier (B)
f = lambda i: round(prof[i] / tot, 3)
print("factor64, wavefront 0: chain a", round(f(12) + f(14) + f(15), 3), "| look-ahead history (rows tA+2, tA+3)", f(40), "| its solve + diagonal update", f(41),
      "| chain b", f(42), "| wait at (A)", f(1), "| W21, y_a, y_b behind (A)", f(43), "| rest of (A) -> (B): everybody's panel solves", f(4))
nf = max(prof[11], 1)
print("(A) -> everybody has arrived at (B), as wavefront 0 sees it, by super column, ticks per factorisation:", [int(v / nf) for v in prof[16:25]])
print("whole super column by J, ticks per factorisation:", [int(v / nf) for v in prof[26:40] if v])
if prof[44] or prof[45] or prof[46]:
    t = prof[44] + prof[45] + prof[46]
    print("forward sweep, wavefronts 1..7 (sum over the seven, ticks per factorisation %d): arithmetic + requests %.2f | barrier %.2f | waiting for the block's tiles %.2f"
          % (t / nf, prof[44] / t, prof[45] / t, prof[46] / t))
