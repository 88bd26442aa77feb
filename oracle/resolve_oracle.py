"""TEST INFRASTRUCTURE ONLY -- CPU restatement of hybdrt/mapping/resolve.py:11-341 (coherent multi-observation
re-optimisation), pinned by tests/golden/refrun_resolve_*.npz (reference runs).  Observations are plain dicts holding
what the reference reads from each fitted DRT object.  Only tests may import this module."""
import numpy as np
from scipy.ndimage import gaussian_filter1d, median_filter

from .coneqp import coneqp_boxlow


def offset_pq(obs, special):
    """resolve.get_offset_pq (11-64): drop the data-dependent unknowns (v_baseline, vz_offset -- always the leading
    ones), folding their fitted values into q."""
    num_remove = special['v_baseline']['size'] + special['vz_offset']['size']
    x_remove = np.empty(num_remove)
    vb = special['v_baseline']
    scaled = np.array(obs['v_baseline'], dtype=float) / obs['response_signal_scale']
    scaled[0] += obs['scaled_response_offset']
    scaled *= obs['v_baseline_scale']
    x_remove[vb['index']:vb['index'] + vb['size']] = scaled
    x_remove[special['vz_offset']['index']] = obs['vz_offset']
    p, q = obs['p_matrix'], obs['q_vector']
    return p[num_remove:, num_remove:], q[num_remove:] + x_remove @ p[:num_remove, num_remove:]


def shifted_special(special):
    """resolve.offset_special_dict (137-159)"""
    gone = {k: special[k] for k in ('v_baseline', 'vz_offset') if k in special}
    out = {}
    for k, v in special.items():
        if k in gone:
            continue
        shift = sum(g.get('size', 1) for g in gone.values() if g['index'] < v['index'])
        out[k] = dict(v, index=v['index'] - shift)
    return out


def resolve_observations(obs_list, special, nonneg=True, sigma=1, lambda_psi=1):
    """resolve.resolve_observations (189-341) for observations on one tau grid (no resize), no filters.
    Returns (x_opt (nr, nc), coneqp result)."""
    sp = shifted_special(special)
    pq = [offset_pq(o, special) for o in obs_list]
    nr, nc = len(pq), len(pq[0][1])
    ly = gaussian_filter1d(np.eye(nr), sigma=sigma, mode='reflect', order=2)
    scale_vec = np.array([o['coefficient_scale'] for o in obs_list])
    scale_smooth = gaussian_filter1d(median_filter(scale_vec, 3), 2)
    lys = ly @ np.diag(scale_vec / scale_smooth)
    my = lys.T @ lys
    param_scale = np.ones(nc)
    if 'R_inf' in sp:
        x_inf = np.array([o['R_inf'] / o['coefficient_scale'] for o in obs_list])
        param_scale[sp['R_inf']['index']] = (5 * np.std(x_inf)) ** -2
    if 'x_dop' in sp:
        x_dop = np.array([o['x_dop'] / (o['coefficient_scale'] * o['dop_scale_vector']) for o in obs_list])
        a = sp['x_dop']['index']
        param_scale[a:a + sp['x_dop']['size']] = (np.std(x_dop, axis=0) + 0.1 * np.std(x_dop)) ** -2
    p_full = np.kron(my, np.diag(param_scale)) * lambda_psi
    for i, (p, _) in enumerate(pq):
        p_full[i * nc:(i + 1) * nc, i * nc:(i + 1) * nc] += p
    q_full = np.concatenate([q for _, q in pq])
    h = np.zeros(nr * nc) if nonneg else 10 * np.ones(nr * nc)
    for v in sp.values():
        if v['nonneg']:
            for i in range(nr):
                h[v['index'] + i * nc:v['index'] + v.get('size', 1) + i * nc] = 0
    res = coneqp_boxlow(p_full, q_full, h)
    return res['x'].reshape(nr, nc), res, (p_full, q_full, h)
