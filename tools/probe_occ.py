"""Experiment: throughput of the resident QP kernel on n=250 problems with 1 vs 2 workgroups per CU
(HIPDRT_QP_LDS_MIN pads the LDS request to force one workgroup per CU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt import _ffi
n, B = 250, 2048
rng = np.random.default_rng(0)
M = rng.standard_normal((n + 20, n)); P = M.T @ M + 0.05 * np.eye(n)
q = rng.standard_normal((B, n)) * 3; h = np.zeros(n)
ctx = _ffi.get_context(0)
for rep in range(3):
    t = time.time(); res = ctx.qp_batch(P, q, h); dt = time.time() - t
print(f"LDS_MIN={os.environ.get('HIPDRT_QP_LDS_MIN')}: n={n} B={B} iters mean {res['iterations'].mean():.2f} wall {dt*1e3:.1f} ms")
