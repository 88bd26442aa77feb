"""Kramers-Kronig validity test on top of the DRT fit (hybdrt/models/kk.py and DRT.kk_test / kk_fit,
hybdrt/models/drt1d.py:1370-1491): residual statistics and frequency limits are O(nf) host arithmetic; the fits run on the
device."""
import numpy as np
from scipy import ndimage
from scipy.special import erf
from scipy.stats import chi2


def normalize_residuals(z_meas, z_pred, norm="modulus"):
    """kk.normalize_residuals (kk.py:9-19): percent of |Z| by default."""
    z_err = z_meas - z_pred
    if isinstance(norm, str) and norm == "modulus":
        return 100 * z_err / np.abs(z_meas)
    return z_err / norm


def _std_normal_quantile(q):
    """utils.stats.std_normal_quantile (stats.py:108-116): inverse of a 2000-point table of the normal CDF"""
    s_grid = np.linspace(0, 14, 2000)
    cdf = 0.5 * (1 + erf(s_grid / np.sqrt(2)))
    q = np.asarray(q, dtype=float)
    return np.interp(np.abs(q - 0.5) + 0.5, cdf, s_grid) * np.sign(q - 0.5)


def robust_std(x, sample_fraction=0.5):
    """utils.stats.robust_std (stats.py:124-134): standard deviation from a central quantile range"""
    if sample_fraction > 1:
        raise ValueError("sample_fraction must be no greater than 1")
    q_lo = np.percentile(x, 50 - 100 * sample_fraction / 2)
    q_hi = np.percentile(x, 50 + 100 * sample_fraction / 2)
    return (q_hi - q_lo) / (2 * _std_normal_quantile(0.5 + sample_fraction / 2))


def get_outliers(z_err_norm, n_iter=2, p_thresh=1e-4, n_sigma=None, std_sample_fraction=0.6):
    """kk.get_outliers (kk.py:21-53): squared error modulus against a chi-squared(2) law with a robust scale, iterated"""
    mask = np.zeros(len(z_err_norm), dtype=bool)
    for _ in range(n_iter):
        kept = z_err_norm[~mask]
        std = robust_std(np.concatenate([kept.real, kept.imag]), sample_fraction=std_sample_fraction)
        if n_sigma is None:
            prob = 1 - chi2.cdf(np.abs(z_err_norm) ** 2, 2, loc=0, scale=std ** 2)
            mask = prob < p_thresh
        else:
            mask = np.abs(z_err_norm) > std * n_sigma
    return np.where(mask)[0]


def get_limits(f_fit, outlier_index, max_num_outliers=2, return_index=False):
    """kk.get_limits (kk.py:56-123): widest frequency window whose ends are clean points with a clean neighbour and which
    holds at most max_num_outliers flagged points"""
    order = np.argsort(f_fit)[::-1]
    f_sorted = np.asarray(f_fit)[order]
    pos = [order.tolist().index(i) for i in outlier_index]
    is_outlier = np.zeros(len(f_sorted))
    is_outlier[pos] = 1
    badness = ndimage.uniform_filter1d(is_outlier, size=3)
    clean = np.where(badness == 0)[0]
    i_left, i_right = clean[0], clean[-1]
    num_bad = np.sum(is_outlier[i_left:i_right])
    if num_bad > max_num_outliers:
        need = num_bad - max_num_outliers
        from_left = np.cumsum(is_outlier[i_left:i_right + 1])
        from_right = np.cumsum(is_outlier[i_left:i_right + 1][::-1])
        ll, rr = np.meshgrid(from_left, from_right)
        index = np.argwhere(ll + rr >= need)
        r, l = index[np.argmin(np.sum(index, axis=1))]
        i_left, i_right = i_left + l, i_right - r
    if is_outlier[i_left] == 1:
        i_left = np.min(clean[clean >= i_left])
    if is_outlier[i_right] == 1:
        i_right = np.max(clean[clean <= i_right])
    f_max, f_min = f_sorted[i_left], f_sorted[i_right]
    if return_index:
        return (f_min, f_max), (i_left, i_right)
    return f_min, f_max


def trim_data(frequencies, z, f_min, f_max):
    """kk.trim_data (kk.py:125-127)"""
    mask = (frequencies <= f_max) & (frequencies >= f_min)
    return frequencies[mask], z[mask]
