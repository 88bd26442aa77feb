import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from hipdrt import synth, _ffi
from hipdrt.models import DRT
hip = C.CDLL("libamdhip64.so")
def free():
    f, t = C.c_size_t(), C.c_size_t(); hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value, t.value
c2 = synth.config_c2()
_ffi.get_context(0)
f0, tot = free()
for B in (2000, 10000):
    z = synth.zarc2_batch(c2["freq"], 16)
    drt = DRT(fixed_basis_tau=c2["tau"])
    zz = np.tile(z, (B // 16, 1))
    plan = drt.stage_batch(c2["freq"], zz)
    drt.fit_staged()
    f1, _ = free()
    print(f"B = {B}: {(f0 - f1) / 1e9:.2f} GB in use = {(f0 - f1) / B / 1e6:.3f} MB per spectrum (device total {tot / 1e9:.0f} GB)")
    plan.close(); drt._plan = None
