"""Config 5 (512 f + 4096 t x 1024 tau, DOP): per-iteration distance between the device loop and the oracle loop on the
same device-built matrices, conditioning of the QPs, timing.  python tools/probe_c5.py [max_iter]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt.models import DRT
from hipdrt import synth
from oracle import drt_oracle as orc

max_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dop = (sys.argv[2] != "nodop") if len(sys.argv) > 2 else True
meas = synth.hybrid_measurement(seed=0, n_pre=96, n_post=4000, nf=512)
drt = DRT(fixed_basis_tau=np.logspace(-7, 3, 1024), fit_dop=dop, warn=False)
t0 = time.time(); drt.fit_hybrid(*meas, max_iter=max_iter); print("device fit wall", time.time() - t0)
t0 = time.time(); drt.fit_hybrid(*meas, max_iter=max_iter); print("device fit wall (2nd)", time.time() - t0)
print("timings", drt._plan.timings())
qp, special = drt.qphb_params, drt.special_qp_params
rzm0 = qp["rm"].copy(); vi = special["vz_offset"]["index"]; rzm0[:, vi] = 0
vb = special["v_baseline"]
vz = dict(index=vi, strength=qp["vz_strength_vec"], num_chrono=qp["num_chrono"], vb=(vb["index"], vb["index"] + vb["size"]))
hyp = orc.get_default_hypers()
if dop: hyp.update(orc.get_default_dop_hypers())
ref = orc.qphb_fit_prepared(rzm0, qp["rv"], [qp["penalty_matrices"][f"m{k}"] for k in range(3)], qp["vmm"], special, hyp,
                            vz=vz, max_iter=max_iter)
hx = np.array([h["x"] for h in ref["history"]]); dx = np.array([h["x"] for h in drt.qphb_history])
print("qp iters dev", qp["qp_iterations"].tolist()); print("qp iters ref", [l["iterations"] for l in ref["qp_log"]])
for i in range(min(len(hx), len(dx))):
    d = np.abs(hx[i] - dx[i]); j = int(np.argmax(d))
    print(i, "max|dx| %.3e at %d (x=%.3e) rel-to-max %.3e  ||dx||/||x|| %.3e" % (d.max(), j, hx[i][j], d.max() / np.abs(hx[i]).max(), np.linalg.norm(hx[i]-dx[i])/np.linalg.norm(hx[i])))
P = ref["qp_log"][1]["P"]; w = np.linalg.eigvalsh(P); print("cond(P) of QP 1: %.3e" % (w[-1] / w[0]))
