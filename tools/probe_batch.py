import numpy as np, time, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hipdrt import synth
from hipdrt.models import DRT
c2 = synth.config_c2()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
z = synth.zarc2_batch(c2["freq"], B)
drt = DRT(fixed_basis_tau=c2["tau"])
for rep in range(2):
    t = time.time(); res = drt.fit_eis_batch(c2["freq"], z); dt = time.time() - t
    print(f"rep{rep}: B={B} wall {dt:.3f}s -> {B/dt:.1f} fits/s; timings {res['timings_ms']} launches {res['launches']}")
st = res["status"]
print("status counts", {int(k): int((st == k).sum()) for k in np.unique(st)})
oi = res["outer_iters"]; qi = res["qp_iters_total"]
print("outer iters: min/mean/max", oi.min(), oi.mean(), oi.max(), " qp iters total mean", qi.mean(), "per-QP", (qi/(oi+1)).mean())
print("nonconverged idx", np.where(st != 0)[0][:20])
