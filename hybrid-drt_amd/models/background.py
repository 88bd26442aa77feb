"""Voltage-baseline columns of a chrono fit (hybdrt/models/background.py:23-37)."""
import numpy as np


def get_baseline_matrix(times, deg, normalize=False, sqrt=False):
    """Polynomial (and optional square-root) baseline features of the elapsed time, each scaled to a maximum of 1 when
    ``normalize``; returns (matrix, scales) then."""
    times = np.asarray(times, dtype=float)
    vb_mat = np.zeros((len(times), deg + 1 + int(sqrt)))
    for k in range(deg + 1):
        vb_mat[:, k] = (times - times[0]) ** k
    if sqrt:
        vb_mat[:, -1] = (times - times[0]) ** 0.5
    if normalize:
        scales = np.max(vb_mat, axis=0)
        return vb_mat / scales[None, :], scales
    return vb_mat
