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
