"""Diagnostic: one QP launch on the group kernel (few problems / n > 2048) against the batch kernel and the CPU checker, with
timings.  python tools/probe_group.py [n] [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt import _ffi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 514
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(n)
A = rng.standard_normal((n + 50, n)) / np.sqrt(n)
xt = np.maximum(rng.standard_normal(n), 0)
P = A.T @ A + 1e-3 * np.eye(n)
q = -A.T @ (A @ xt)
h = np.zeros(n)
ctx = _ffi.get_context(0)
Ps, qs = np.stack([P] * B), np.stack([q] * B)
ref = None
if n <= 2048:
    ctx.debug_qp_group(0)
    ref = ctx.qp_batch(Ps, qs, h)
    t = time.time(); ref = ctx.qp_batch(Ps, qs, h); dt = time.time() - t
    print(f"batch kernel: status {ref['status'].tolist()} iters {ref['iterations'].tolist()} wall {dt * 1e3:.1f} ms", flush=True)
for G in [int(a) for a in sys.argv[3:]] or [1, 2, 4, 8, -1]:
    ctx.debug_qp_group(G)
    res = ctx.qp_batch(Ps, qs, h)
    ctx.qp_profile(reset=True)
    t = time.time(); res = ctx.qp_batch(Ps, qs, h); dt = time.time() - t
    prof = ctx.qp_profile(reset=True)
    if prof[10]:
        nf = max(prof[11], 1)
        names = {2: "wait_staged", 12: "chain", 1: "wait_A", 3: "A_to_A2", 4: "group_sync", 5: "fwd_diag", 6: "fwd_upd", 7: "bwd_diag", 8: "bwd_upd"}
        print(f"  WG0 wavefront 0, k cycles per factorisation (total {prof[10] // nf // 1000}k, {nf} factorisations):",
              {v: prof[k] // nf // 1000 for k, v in names.items()},
              "clock %.2f GHz" % (prof[10] / max(prof[13], 1) * 0.1))
        w = {20: "w1 drain+publish", 21: "w1 own tile", 22: "w1 wait tiles", 23: "w1 wait (A)", 24: "w1 (A)->(A2)",
             26: "w2 drain+publish", 27: "w2 la tile", 28: "w2 own rows", 29: "w2 wait (A)", 30: "w2 (A)->end"}
        if prof[31] or prof[33]:
            print("  wavefront 1 of WG0 as look-ahead owner, k cycles per factorisation: source tiles + segments %d, wait for own row stores %d, new range %d, publish %d"
                  % (prof[31] // nf // 1000, prof[32] // nf // 1000, prof[33] // nf // 1000, prof[34] // nf // 1000))
        if prof[36]:
            print("  wavefront 2's staged loop: %d super-steps per factorisation; cycles per super-step: wait loads %d, barrier %d, issue %d, compute %d"
                  % (prof[36] // nf, prof[32] // prof[36], prof[33] // prof[36], prof[34] // prof[36], prof[35] // prof[36]))
        if any(prof[k] for k in w):
            print("  wavefronts 1 / 2 of WG0, k cycles per factorisation:", {v: prof[k] // nf // 1000 for k, v in w.items()})
        else:
            print("  wait at (A) by block column:", [int(v / nf) // 1000 for v in prof[16:48] if v])
    same = None if ref is None else bool(np.array_equal(ref["x"], res["x"]))
    if ref is not None:
        print(f"  max |x - x_batch| / peak = {np.abs(res['x'] - ref['x']).max() / np.abs(ref['x']).max():.2e}")
    print(f"group G={G}: status {res['status'].tolist()} iters {res['iterations'].tolist()} wall {dt * 1e3:.1f} ms (incl. transfers), "
          f"bit-identical to the batch kernel: {same}", flush=True)
ctx.debug_qp_group(-1)
if n <= 1100:
    from oracle.coneqp import coneqp_boxlow
    r = coneqp_boxlow(P, q, h)
    print("oracle iters", r["iterations"], "max rel err", float(np.max(np.abs(res["x"][0] - r["x"])) / np.abs(r["x"]).max()))
