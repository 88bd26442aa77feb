"""Diagnostic: one large QP (n > 528: tile-packed kernel, inverse diagonal blocks in global memory) -- wall time per interior-point iteration and, with
HIPDRT_LIB=.../libhipdrt_prof.so, the in-kernel phase breakdown.  python tools/probe_qp_large.py [n] [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt import _ffi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1078
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(0)
A = rng.standard_normal((2 * n, n)) / np.sqrt(n)
xt = np.maximum(rng.standard_normal(n), 0)
P = A.T @ A + 1e-3 * np.eye(n)
q = -A.T @ (A @ xt)
h = np.zeros(n)
ctx = _ffi.get_context(0)
Ps, qs = np.tile(P, (B, 1, 1)), np.tile(q, (B, 1))
for rep in range(3):
    ctx.qp_profile(reset=True)
    t = time.time(); res = ctx.qp_batch(Ps, qs, h); dt = time.time() - t
    prof = ctx.qp_profile(reset=True)
    it = int(res["iterations"][0])
    print(f"n={n} B={B} iters={it} wall {dt*1e3:.1f} ms -> {dt*1e3/(it+1):.2f} ms per IPM iteration (incl. transfers)")
names = ["rank_k", "load_c", "diag", "trsm", "store", "fwd_diag", "fwd_upd", "bwd_diag", "bwd_upd", "matvec", "total"]
if prof[10]:
    print("shares:", {nm: round(prof[i] / prof[10], 3) for i, nm in enumerate(names)}, "ticks total", prof[10])
from oracle.coneqp import coneqp_boxlow
r = coneqp_boxlow(P, q, h)
print("oracle iters", r["iterations"], "max rel err", np.max(np.abs(res["x"][0] - r["x"])) / np.abs(r["x"]).max())
