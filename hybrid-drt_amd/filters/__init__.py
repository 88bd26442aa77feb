"""Drop-in for the part of hybdrt/filters that the chrono down-sampling uses: the blended, per-sample-width Gaussian filter.
The correlations run on the device (csrc/matrices.hip: nonuniform_gauss_kernel); the sigma nodes and kernel tables are
derived here exactly as the reference / scipy derive them."""
import numpy as np

from .. import _ffi


def _gaussian_kernel_half(sigma, radius):
    """scipy.ndimage._filters._gaussian_kernel1d(sigma, 0, radius), centre and right half (the kernel is symmetric)"""
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    phi = phi / phi.sum()
    return phi[radius:]


def _sigma_nodes(sigma, sigma_node_factor=1.5, min_sigma=0.25):
    """node grid of filters.nonuniform_gaussian_filter1d (hybdrt/filters/_filters.py:264-295); modifies `sigma` in place
    the way the reference does (floor at 1e-8, then at two increments below min_sigma)"""
    np.maximum(sigma, 1e-8, out=sigma)
    min_ls = max(np.min(np.log10(sigma)), np.log10(min_sigma))
    max_ls = max(np.max(np.log10(sigma)), np.log10(min_sigma))
    num_nodes = int(np.ceil((max_ls - min_ls) / np.log10(sigma_node_factor))) + 1
    nodes = np.logspace(min_ls, max_ls, num_nodes)
    if np.min(sigma) < min_sigma:
        factor = nodes[-1] / nodes[-2] if len(nodes) > 1 else sigma_node_factor
        sigma[sigma < min_sigma / (factor ** 2)] = min_sigma / (factor ** 2)
        while nodes[0] > np.min(sigma) * 1.001:
            nodes = np.insert(nodes, 0, nodes[0] / factor)
    node_delta = np.log(nodes[-1] / nodes[-2]) if len(nodes) > 1 else 1
    return nodes, node_delta


def nonuniform_gaussian_filter1d_segments(a, sigma, seg, truncate=4, sigma_node_factor=1.5, min_sigma=0.25, device=0):
    """filters.nonuniform_gaussian_filter1d applied to every segment [seg[s], seg[s+1]) of `a` on its own (what
    preprocessing.filter_chrono_signal does step by step), in one device launch."""
    a = np.ascontiguousarray(a, dtype=float)
    sigma = np.array(sigma, dtype=float)
    nseg = len(seg) - 1
    per_seg = []
    for s in range(nseg):
        sg = sigma[seg[s]:seg[s + 1]]               # view: _sigma_nodes clips it in place like the reference
        if len(sg) and np.max(sg) > 0:
            per_seg.append(_sigma_nodes(sg, sigma_node_factor, min_sigma))
        else:
            per_seg.append(None)
    if all(p is None for p in per_seg):
        return a.copy()
    K = max(len(p[0]) for p in per_seg if p is not None)
    nodes = np.zeros((nseg, K))
    node_delta = np.ones(nseg)
    radius = -np.ones((nseg, K), dtype=np.int32)
    woff = np.zeros((nseg, K), dtype=np.int32)
    filtered = np.zeros(nseg, dtype=np.int32)
    tables, pos = [], 0
    for s, p in enumerate(per_seg):
        if p is None:
            continue
        filtered[s] = 1
        nodes[s, :len(p[0])] = p[0]
        node_delta[s] = p[1]
        for k, nd in enumerate(p[0]):
            if nd < min_sigma:
                continue                            # below the minimum effective width: the node returns the input
            r = int(truncate * float(nd) + 0.5)
            radius[s, k], woff[s, k] = r, pos
            tables.append(_gaussian_kernel_half(float(nd), r))
            pos += r + 1
    weights = np.concatenate(tables) if tables else np.zeros(1)
    return _ffi.get_context(device).nonuniform_gaussian_filter1d(a, sigma, seg, filtered, nodes, node_delta, weights, woff,
                                                                 radius)


def nonuniform_gaussian_filter1d(a, sigma, axis=-1, empty=False, mode='reflect', cval=0.0, truncate=4, order=0,
                                 sigma_node_factor=1.5, min_sigma=0.25, device=0):
    """filters.nonuniform_gaussian_filter1d (hybdrt/filters/_filters.py:261-343) for a 1-D array."""
    if empty or mode != 'reflect' or order != 0 or np.ndim(a) != 1:
        raise NotImplementedError("only the 1-D, order-0, reflect-mode filter of the down-sampling path is built")
    return nonuniform_gaussian_filter1d_segments(a, sigma, np.array([0, len(a)]), truncate, sigma_node_factor, min_sigma,
                                                 device)
