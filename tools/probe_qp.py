"""Diagnostic: time hipdrt_qp_batch on B copies of a C2-sized QP and (with HIPDRT_LIB=.../libhipdrt_prof.so)
print the in-kernel phase breakdown of workgroup 0."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt import synth, _ffi
from oracle import drt_oracle as orc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
c2 = synth.config_c2()
d = orc.OracleDRT(fixed_basis_tau=c2["tau"])
m = d.prepare(c2["freq"])
z = synth.zarc2_spectrum(c2["freq"], 0)
cs = (z.real.max() - z.real.min()) / 14
rv = np.concatenate([z.real, z.imag]) / cs
hyp = orc.get_default_hypers()
l2 = orc.calculate_qp_l2_matrix(hyp, np.ones(3), m["pen"], [np.ones(514)] * 3, 2)
w = np.full(512, 50.0)
wa = w[:, None] * m["rzm"]
P = wa.T @ wa + l2
q = -wa.T @ (w * rv)
h = np.zeros(514)
ctx = _ffi.get_context(0)
qs = np.tile(q, (B, 1))
Ps = np.tile(P, (B, 1, 1))
for rep in range(3):
    ctx.qp_profile(reset=True)
    t = time.time(); res = ctx.qp_batch(Ps, qs, h); dt = time.time() - t
    prof = ctx.qp_profile(reset=True)
    it = int(res["iterations"][0])
    print(f"B={B} iters={it} wall {dt*1e3:.1f} ms (incl. H2D of {Ps.nbytes/1e6:.0f} MB)")
# slots as seen by wavefront 0 of workgroup 0: [1] waiting at barrier A for the other wavefronts' rank-k update (beyond its
# own diagonal chain, slots 12/14/15), [3] W21 + fused forward step, [4] waiting at barrier B (stores drained)
names = ["-", "wait_rank_k", "-", "w21_fwd", "wait_stores", "fwd_diag", "fwd_upd", "bwd_diag", "bwd_upd", "matvec", "total"]
if prof[10]:
    tot = prof[10]
    print("in-kernel ticks of WG0:", {n: prof[i] for i, n in enumerate(names)}, "factorizations:", prof[11])
    print("diagonal chain: cholinv1", prof[12], "l21+d2", prof[14], "cholinv2", prof[15]); print("shares:", {n: round(prof[i] / tot, 3) for i, n in enumerate(names)})
if prof[10] and not any(prof[44:48]) and any(prof[16:48]):
    nf = max(prof[11], 1)
    print("wavefront 0's wait at barrier (A) by block column, ticks per factorisation:", [int(v / nf) for v in prof[16:48] if v])
if prof[10] and prof[13]:
    print("shader clock while WG0 ran: %.2f GHz (s_memtime ticks / 100 MHz s_memrealtime ticks)" % (prof[10] / prof[13] * 0.1))
if prof[10] and any(prof[16:44]):
    # role timelines of the super-column kernel (qp_super.hpp), ticks summed over all factorisations of WG0
    f = lambda i: round(prof[i] / tot, 3)
    print("super-column kernel, shares of WG0's total:")
    print("  total ticks", tot, "per factorisation", tot // max(prof[11], 1))
    print("  wave0 chain", f(16), "wait A", f(17), "A->B", f(18))
    print("    inside the chain: cholinv16 x4", f(44), "tile solves / updates", f(45), "Wba + stores", f(46), "fused forward", f(47))
    print("  wave1 rank-k", f(20), "wait A", f(21), "wait rows", f(22), "last 4 chunks + stage", f(23), "wait B", f(24))
    for name, b0 in (("wave2", 26), ("wave7", 32), ("waves3-6 (sum of 4)", 38)):
        print(f"  {name} rank-k", f(b0), "wait A", f(b0 + 1), "solve row 0 + publish", f(b0 + 2), "solve rows 1-3", f(b0 + 3))
from oracle.coneqp import coneqp_boxlow
r = coneqp_boxlow(P, q, h)
print("oracle iters", r["iterations"], "max rel err", np.max(np.abs(res["x"][0] - r["x"])) / np.abs(r["x"]).max())
