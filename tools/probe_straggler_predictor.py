"""CPU study (oracle fits of BASELINE configs[2] spectra, no GPU): can the spectra that run 42 ... 50 outer iterations be told EARLY,
so that they could be moved to a faster-paced range of their own?  Records max |dx / x| per outer iteration
of every fit, then tests a decay-rate predictor (iterations left = log(xtol / metric) / log(rate over the last four iterations))
at several decision points k0.  python tools/probe_straggler_predictor.py 256  ->  profiles/r06_straggler_predictor.txt"""
import os, sys
for v in ("OMP_NUM_THREADS","OPENBLAS_NUM_THREADS","MKL_NUM_THREADS"): os.environ[v]="1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multiprocessing import Pool
from hipdrt import synth
from oracle import drt_oracle as orc
cfg=synth.config_c2()
def work(seed):
    z=synth.zarc2_batch(cfg["freq"],1,first_seed=seed)[0]
    d=orc.OracleDRT(fixed_basis_tau=cfg["tau"])
    d.fit_eis(cfg["freq"], z, keep_history=True)
    hx=np.array([h["x"] for h in d.qphb_history])
    # metric per iteration: max|dx/(x_in+1e-15)| ; x_in of iteration 0 is the init x (not in history) -> start at it 1
    met=[np.max(np.abs((hx[i]-hx[i-1])/(hx[i-1]+1e-15))) for i in range(1,len(hx))]
    return seed, len(hx), d.qphb_params["converged"], met
if __name__=="__main__":
    N=int(sys.argv[1])
    with Pool(8) as p: res=p.map(work, range(N))
    its=np.array([r[1] for r in res])
    print("iterations histogram:", np.bincount(its))
    print("mean", its.mean(), "frac at 50:", (its>=50).mean())
    for k0 in (6, 8, 10, 12, 16, 20, 24):
        rows = []
        for seed, n, conv, met in res:
            met = np.array(met)
            if len(met) < k0:
                continue
            m, m4 = met[k0 - 1], met[max(0, k0 - 5)]
            rows.append((n, m, (m / m4) ** 0.25 if m4 > 0 else 1.0))
        n_, m_, r_ = np.array(rows).T
        with np.errstate(all='ignore'):
            pred = np.clip(np.where(r_ < 1, k0 + np.log(1e-2 / m_) / np.log(r_), 99), k0, 99)
        lab = n_ >= 42
        for thr in (35, 45):
            sel = pred >= thr
            print(f"decision at iteration {k0}, lane = predicted total >= {thr}: lane holds {sel.mean():.3f} of the spectra, catches "
                  f"{100 * (sel & lab).sum() / max(lab.sum(), 1):.0f} % of the long runners (>= 42 iterations), "
                  f"{100 * (sel & lab).sum() / max(sel.sum(), 1):.0f} % of the lane are long runners")
