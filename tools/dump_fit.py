"""Fits a fixed set of spectra (C1 golden-size and C2-size) with whatever library HIPDRT_LIB points to and saves the raw
results; two dumps compared bit for bit tell whether two builds compute the same thing.
python tools/dump_fit.py out.npz   /   python tools/dump_fit.py --cmp a.npz b.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
    print("identical" if not bad else f"DIFFERENT: {bad}")
    for k in bad:       # how far apart: iteration counts as a mismatch count, everything else relative to the largest entry
        x, y = np.asarray(a[k], dtype=float), np.asarray(b[k], dtype=float)
        if "iters" in k:
            print(f"  {k}: {int((x != y).sum())} of {x.size} differ")
        else:
            sc = np.abs(x).max(axis=-1, keepdims=True) if x.ndim > 1 else np.abs(x).max()
            print(f"  {k}: max |a - b| / peak = {np.max(np.abs(x - y) / np.maximum(sc, 1e-300)):.2e}")
    sys.exit(1 if bad else 0)
from hipdrt import synth
from hipdrt.models import DRT
out = {}
c2 = synth.config_c2()
z = synth.zarc2_batch(c2["freq"], 96)
d = DRT(fixed_basis_tau=c2["tau"])
r = d.fit_eis_batch(c2["freq"], z)
for k in ("x", "weights", "rho", "outer_iters", "qp_iters_total"):
    out["c2_" + k] = r[k]
v = d.estimate_distribution_var_batch(c2["tau"][::4])
out["c2_var"] = np.asarray(v[0] if isinstance(v, tuple) else v)
c1 = synth.config_c1()
z1 = synth.zarc2_batch(c1["freq"], 16)
r1 = DRT(fixed_basis_tau=c1["tau"]).fit_eis_batch(c1["freq"], z1)
for k in ("x", "weights", "outer_iters", "qp_iters_total"):
    out["c1_" + k] = r1[k]
f = np.logspace(5.5, -0.5, 60)
r3 = DRT(basis_tau_ppd=8).fit_eis_batch(f, synth.zarc2_batch(f, 8))      # n = 61: odd number of block columns etc.
for k in ("x", "outer_iters", "qp_iters_total"):
    out["d_" + k] = r3[k]
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1], {k: v.shape for k, v in out.items() if k.endswith("_x")})
