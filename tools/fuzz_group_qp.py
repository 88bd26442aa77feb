"""Randomised check of the several-workgroups-per-problem coneqp kernel against the CPU checker (oracle/coneqp.py): random sizes
257 <= n <= 1100, 1 <= B <= 16 problems per launch, random conditioning, nonneg / box-low constraint vectors; same iteration
counts, x within 1e-9 of the peak, and the batch kernel on the same problems within 1e-11.  python tools/fuzz_group_qp.py [count] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt import _ffi
from oracle.coneqp import coneqp_boxlow

count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
ctx = _ffi.get_context(0)
bad, worst, worst_b = [], 0.0, 0.0
t0 = time.time()
for case in range(count):
    n = int(rng.integers(257, 1101))
    B = int(rng.integers(1, 17))
    cond = 10.0 ** rng.uniform(-6, -1)
    Ps, qs = [], []
    for b in range(B):
        A = rng.standard_normal((n + int(rng.integers(5, 80)), n)) / np.sqrt(n)
        Ps.append(A.T @ A + cond * np.eye(n))
        qs.append(-A.T @ (A @ np.maximum(rng.standard_normal(n), 0)) * 10.0 ** rng.uniform(-2, 2))
    Ps, qs = np.array(Ps), np.array(qs)
    h = np.zeros(n) if rng.random() < 0.7 else np.full(n, 10.0 ** rng.uniform(0, 5))
    h[:int(rng.integers(0, 5))] = 1000.0
    ctx.debug_qp_group(-1)
    res = ctx.qp_batch(Ps, qs, h)
    ctx.debug_qp_group(0)
    ref = ctx.qp_batch(Ps, qs, h)
    ctx.debug_qp_group(-1)
    for b in (0, B - 1):
        r = coneqp_boxlow(Ps[b], qs[b], h)
        err = np.abs(res["x"][b] - r["x"]).max() / max(np.abs(r["x"]).max(), 1e-300)
        errb = np.abs(res["x"][b] - ref["x"][b]).max() / max(np.abs(ref["x"]).max(), 1e-300)
        worst, worst_b = max(worst, err), max(worst_b, errb)
        if res["iterations"][b] != r["iterations"] or res["iterations"][b] != ref["iterations"][b] or err > 1e-9 or res["status"][b] != 0:
            bad.append((case, n, B, b, int(res["iterations"][b]), int(r["iterations"]), int(ref["iterations"][b]), float(err), int(res["status"][b])))
print(f"{count} launches (n 257..1100, B 1..16, two problems of each checked on the CPU) in {time.time() - t0:.0f} s: "
      f"{len(bad)} mismatches; max |x - x_cpu| / peak {worst:.2e}, max |x - x_batch_kernel| / peak {worst_b:.2e}")
for b in bad:
    print("MISMATCH (case, n, B, b, iters group / cpu / batch, err, status)", b)
