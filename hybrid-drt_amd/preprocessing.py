"""Host-side grid / scale rules of hybdrt/preprocessing.py that sit on the EIS fit path."""
import numpy as np


def get_tau_lim(frequencies, times=None, step_times=None):
    """preprocessing.py:953-972 (EIS branch; chrono data is a later scope row)."""
    if times is not None:
        raise NotImplementedError("chrono/hybrid data are not in this build's scope yet (SURVEY.md 8, C5)")
    return 1 / (2 * np.pi * np.max(frequencies)), 1 / (2 * np.pi * np.min(frequencies))


def get_num_decades(frequencies, times=None, step_times=None):
    tau_min, tau_max = get_tau_lim(frequencies, times, step_times)
    return np.log10(tau_max) - np.log10(tau_min)


def get_basis_tau(frequencies, times=None, step_times=None, ppd=10, extend_decades=1, tau_grid=None):
    """preprocessing.py:982-1013: 10 points per decade, +-extend_decades beyond the measurement range,
    or the matching slice of a tau supergrid."""
    tau_min, tau_max = get_tau_lim(frequencies, times, step_times)
    log_tau_min = np.log10(tau_min) - extend_decades
    log_tau_max = np.log10(tau_max) + extend_decades
    if tau_grid is not None:
        tau_grid = np.asarray(tau_grid)
        if 10 ** log_tau_min < np.min(tau_grid):
            left = 0
        else:
            left = _nearest_index(tau_grid, 10 ** log_tau_min, -1)
        if 10 ** log_tau_max > np.max(tau_grid):
            right = len(tau_grid)
        else:
            right = _nearest_index(tau_grid, 10 ** log_tau_max, 1) + 1
        return tau_grid[left:right]
    exact = (log_tau_max - log_tau_min) * ppd + 1
    num = int(np.ceil(exact))
    pad = 0.5 * (num - exact) / ppd
    return np.logspace(log_tau_min - pad, log_tau_max + pad, num)


def _nearest_index(x_array, x_val, constraint):
    """hybdrt/utils/array.py nearest_index with constraint -1 (<= x_val) / +1 (>= x_val)."""
    if constraint == -1:
        mask = x_array <= x_val
    else:
        mask = x_array >= x_val
    idx = np.where(mask)[0]
    return int(idx[np.argmin(np.abs(x_array[mask] - x_val))])


def get_epsilon_from_ppd(ppd, factor=1):
    """preprocessing.py:1016-1017."""
    return factor / np.log(10 ** (1 / ppd))


def estimate_rp(times, step_times, input_step_sizes, response_signal, step_model, z):
    """preprocessing.py:764-841, EIS branch: span of the real part."""
    if times is not None:
        raise NotImplementedError("chrono/hybrid data are not in this build's scope yet")
    return np.max(z.real) - np.min(z.real)
