"""One draw of tests/test_gpu_hybrid.py::test_randomised_joint_fits_with_option_combinations in detail: the deviation of the device's
iterates from the CPU checker's per outer iteration, next to the CPU checker's OWN sensitivity (the same fit with the data vector
scaled by 1 + 1e-13).  python tools/probe_joint_fuzz.py <seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_hybrid as t
from oracle import drt_oracle as orc

seed = int(sys.argv[1])
captured = {}
real = orc.qphb_fit_prepared
def spy(*a, **k):
    captured["args"], captured["kw"] = a, k
    r = real(*a, **k)
    captured["ref"] = r
    return r
orc.qphb_fit_prepared = spy
dev = {}
real_pc = t.parity_close
def pc(name, a, b, *rest, **kw):
    if name.endswith("hist_x"):
        dev["dx"], dev["hx"] = np.asarray(a), np.asarray(b)
t.parity_close = pc
t.parity = lambda *a, **k: None
fn = getattr(t.test_randomised_joint_fits_with_option_combinations, "__wrapped__", t.test_randomised_joint_fits_with_option_combinations)
fn(seed)
orc.qphb_fit_prepared = real
dx, hx = dev["dx"], dev["hx"]
peak = np.abs(hx).max()
print("device vs CPU checker, max |dx| / peak per recorded iterate:", ["%.1e" % v for v in np.abs(dx - hx).max(axis=1) / peak])
a, k = captured["args"], captured["kw"]
a2 = list(a); a2[1] = np.asarray(a[1]) * (1 + 1e-13)
r2 = real(*a2, **k)
h2 = np.array([h["x"] for h in r2["history"]])
print("CPU checker vs itself with the data scaled by 1 + 1e-13:          ", ["%.1e" % v for v in np.abs(h2 - hx).max(axis=1) / peak])
print("interior-point iterations per QP:", [l["iterations"] for l in captured["ref"]["qp_log"]])
