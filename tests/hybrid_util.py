"""Helpers shared by the CPU and GPU tests of the config-5 family (DOP / joint chrono + EIS fits)."""
import os

import numpy as np

from conftest import GOLDEN

CASES = ("golden71x91_dop", "hybrid_s0", "hybrid_s0_dop")


def load_case(name):
    g = np.load(os.path.join(GOLDEN, f"refrun_{name}.npz"), allow_pickle=False)
    special = {str(k): dict(index=int(i), size=int(s), nonneg=bool(nn))
               for k, i, s, nn in zip(g["special_names"], g["special_index"], g["special_size"], g["special_nonneg"])}
    return g, special


def initial_rzm_and_vz(g, special):
    """The matrix the loop starts from (vz_offset column still zero, drt1d.py:5808) and the column-rewrite description"""
    rzm0 = g["rm"].copy()
    vz = None
    if "vz_offset" in special:
        vi = special["vz_offset"]["index"]
        rzm0[:, vi] = 0
        vb = special["v_baseline"]
        vz = dict(index=vi, strength=g["vz_strength_vec"], num_chrono=int(g["num_chrono"]),
                  vb=(vb["index"], vb["index"] + vb["size"]))
    return rzm0, vz


def random_eis_problem(seed):
    """One draw of the randomised EIS differential test (tests/test_gpu_fit.py::test_randomized_fits_vs_oracle):
    frequency range / count, basis density, noise level, circuit parameters, error structure and sign constraint."""
    rng = np.random.default_rng(1000 + seed)
    nf = int(rng.integers(30, 90))
    f_hi, f_lo = 10 ** rng.uniform(4, 6.5), 10 ** rng.uniform(-2, 0.5)
    freq = np.logspace(np.log10(f_hi), np.log10(f_lo), nf)
    ppd = int(rng.choice([6, 8, 10, 12]))
    nonneg = bool(rng.random() < 0.75)
    err = None if rng.random() < 0.7 else 'uniform'
    z = []
    for b in range(4):
        r_inf, r1, r2 = rng.uniform(0.1, 5), rng.uniform(0.2, 3), rng.uniform(0.1, 2)
        t1, t2 = 10 ** rng.uniform(-5, -2), 10 ** rng.uniform(-2, 0.5)
        b1, b2 = rng.uniform(0.6, 1.0), rng.uniform(0.6, 1.0)
        w = 2j * np.pi * freq
        zz = r_inf + r1 / (1 + (w * t1) ** b1) + r2 / (1 + (w * t2) ** b2) + w * 10 ** rng.uniform(-8, -6)
        sig = 10 ** rng.uniform(-4, -2)
        z.append(zz + sig * np.abs(zz) * (rng.standard_normal(nf) + 1j * rng.standard_normal(nf)))
    return freq, np.array(z), ppd, err, dict(nonneg=nonneg)
