"""Diagnostic: stand-alone weighted Gram (no L2 epilogue terms) for 1024 spectra at C2 size; run under rocprofv3 --stats
to read the kernel time, compare with the fit loop's gram_kernel launches (which carry the L2 epilogue)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt import _ffi
rng = np.random.default_rng(0)
B, m, n = 1024, 512, 514
A = rng.standard_normal((m, n))
w = rng.uniform(0.5, 2.0, (B, m))
b = rng.standard_normal((B, m))
ctx = _ffi.get_context(0)
for rep in range(3):
    t = time.time(); out = ctx.weighted_gram(A, w, b); dt = time.time() - t
    print(f"weighted_gram B={B}: {dt*1e3:.1f} ms wall (incl. transfers)")
