"""TEST INFRASTRUCTURE ONLY -- CPU restatement of filters.nonuniform_gaussian_filter1d (hybdrt/filters/_filters.py:261-343,
order 0, mode 'reflect', empty=False) on scipy.ndimage.gaussian_filter1d, pinned through the reference-run fixture
tests/golden/refrun_hybrid_downsample.npz (the down-sampled, anti-alias-filtered voltage record)."""
import numpy as np
from scipy import ndimage


def nonuniform_gaussian_filter1d(a, sigma, truncate=4, sigma_node_factor=1.5, min_sigma=0.25):
    a = np.asarray(a, dtype=float)
    sigma = np.array(sigma, dtype=float)
    if not np.max(sigma) > 0:
        return a
    sigma = np.maximum(sigma, 1e-8)
    min_ls = max(np.min(np.log10(sigma)), np.log10(min_sigma))
    max_ls = max(np.max(np.log10(sigma)), np.log10(min_sigma))
    nodes = np.logspace(min_ls, max_ls, int(np.ceil((max_ls - min_ls) / np.log10(sigma_node_factor))) + 1)
    if np.min(sigma) < min_sigma:
        factor = nodes[-1] / nodes[-2] if len(nodes) > 1 else sigma_node_factor
        sigma[sigma < min_sigma / factor ** 2] = min_sigma / factor ** 2
        while nodes[0] > np.min(sigma) * 1.001:
            nodes = np.insert(nodes, 0, nodes[0] / factor)
    delta = np.log(nodes[-1] / nodes[-2]) if len(nodes) > 1 else 1
    outs = np.array([a if nd < min_sigma else ndimage.gaussian_filter1d(a, sigma=nd, mode='reflect', truncate=truncate)
                     for nd in nodes])
    nw = np.abs(np.log(sigma[None, :] / nodes[:, None])) / delta
    nw[nw >= 1] = 1
    return np.sum(outs * (1 - nw), axis=0)


def filter_chrono_signal(times, y, step_index, decimate_index, sigma_factor=0.01, truncate=4.0):
    """preprocessing.filter_chrono_signal (507-572) with sigma_from_decimate_index (575-589), no outlier handling."""
    times, y = np.asarray(times, dtype=float), np.asarray(y, dtype=float)
    bounds = np.array(step_index)
    if bounds[0] > 0:
        bounds = np.insert(bounds, 0, 0)
    if bounds[-1] < len(y):
        bounds = np.append(bounds, len(y))
    t_sample = np.median(np.diff(times))
    max_sigma = sigma_factor / t_sample
    diff = np.diff(decimate_index)
    min_diff = np.minimum(np.insert(diff, 0, diff[0]), np.append(diff, diff[-1]))
    sd = min_diff / (2 * truncate)
    sd[min_diff < 2] = 0
    dec = np.zeros(len(y))
    dec[decimate_index] = sd
    out = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        t = times[a:b]
        sg = sigma_factor * ((np.exp(1) * (t - (t[0] - t_sample)) / 2) / t_sample)
        sg[sg > max_sigma] = max_sigma
        out.append(nonuniform_gaussian_filter1d(y[a:b], np.minimum(dec[a:b], sg)))
    return np.concatenate(out)
