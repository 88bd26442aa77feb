"""Drop-in for hybdrt/mapping/resolve.py: coherent re-optimisation of neighbouring observations.

The single fits' P matrices sit on the block diagonal of one large QP; a second-derivative Gaussian filter across the
observation index couples equal coefficients of neighbouring observations.  The assembly is O((nr nc)^2) bookkeeping on
the host exactly as in the reference; the QP itself (n = nr * nc up to 4096 unknowns, same coneqp trajectory) is solved
by the device kernel behind ``hipdrt_qp_batch``.  Like the reference, this needs hybrid fits: the data-dependent unknowns
``v_baseline`` and ``vz_offset`` are eliminated first (resolve.py:23-24)."""
from copy import deepcopy

import numpy as np
from scipy.ndimage import gaussian_filter1d, median_filter

from .. import _ffi
from ..matrices.basis import construct_func_eval_matrix


def get_offset_pq(drt):
    """resolve.get_offset_pq (hybdrt/mapping/resolve.py:11-64): P, q of a fitted DRT without the v_baseline / vz_offset
    unknowns, their fitted values folded into q."""
    p, q = drt.fit_parameters['p_matrix'], drt.fit_parameters['q_vector']
    sp = drt.special_qp_params
    num_remove = sum(sp[k]['size'] for k in ('v_baseline', 'vz_offset'))     # KeyError on non-hybrid fits, as upstream
    x_remove = np.empty(num_remove)
    vb = sp['v_baseline']
    # inverse of extract_qphb_parameters' baseline scaling (drt1d.py:6244-6259)
    coef = np.array(drt.fit_parameters['v_baseline'], dtype=float) / drt.response_signal_scale
    coef[0] += drt.scaled_response_offset
    x_remove[vb['index']:vb['index'] + vb['size']] = coef * drt.v_baseline_scale
    x_remove[sp['vz_offset']['index']] = drt.fit_parameters['vz_offset']
    return p[num_remove:, num_remove:], q[num_remove:] + x_remove @ p[:num_remove, num_remove:]


def resize_pq(p, q, special_offset, tau_indices, match_tau_indices):
    """resolve.resize_pq (67-134): embed (or crop) an observation's DRT block into the common supergrid slice; the
    special-parameter block and its couplings keep their place.  One offset rule serves all four expand / truncate
    combinations; it equals the reference wherever the reference is self-consistent (both expanding branches, which is
    all `truncate=False` ever reaches; its two left-truncating branches drop `special_offset` from some indices,
    resolve.py:107-108, 124-129, and misplace or fail to broadcast when special parameters exist)."""
    left = tau_indices[0] - match_tau_indices[0]
    right = tau_indices[1] - match_tau_indices[1]
    ntau_new = match_tau_indices[1] - match_tau_indices[0]
    so = int(special_offset)
    size = so + ntau_new
    # source slice of the observation's own DRT index range, destination slice in the common range
    src0, dst0 = (0, left) if left >= 0 else (-left, 0)
    ntau_old = tau_indices[1] - tau_indices[0]
    src1, dst1 = (ntau_old, ntau_new + right) if right <= 0 else (ntau_old - right, ntau_new)
    p_out, q_out = np.zeros((size, size)), np.zeros(size)
    p_out[:so, :so] = p[:so, :so]
    q_out[:so] = q[:so]
    s, d = slice(so + src0, so + src1), slice(so + dst0, so + dst1)
    p_out[d, d] = p[s, s]
    q_out[d] = q[s]
    p_out[d, :so] = p[s, :so]
    p_out[:so, d] = p[:so, s]
    return p_out, q_out


def offset_special_dict(special_qp_params):
    """resolve.offset_special_dict (137-159): the special-parameter table after removing v_baseline / vz_offset."""
    gone = {k: special_qp_params[k] for k in ('v_baseline', 'vz_offset') if k in special_qp_params}
    out = deepcopy({k: v for k, v in special_qp_params.items() if k not in gone})
    for key, v in out.items():
        v['index'] = v['index'] - int(sum(g.get('size', 1) for g in gone.values() if g['index'] < v['index']))
    return out


def get_tau_indices(obs_tau_indices, truncate=False):
    """resolve.get_tau_indices (162-173)"""
    lefts, rights = [t[0] for t in obs_tau_indices], [t[1] for t in obs_tau_indices]
    return (max(lefts), min(rights)) if truncate else (min(lefts), max(rights))


def resolve_observations(obs_drt_list, obs_tau_indices, nonneg, obs_psi=None, truncate=False, sigma=1, lambda_psi=1,
                         unpack=False, tau_filter_sigma=0, special_filter_sigma=0, device=0, _assemble_only=False):
    """resolve.resolve_observations (189-341).  Returns (x_opt (nr, nc), match_tau_indices), or the unpacked
    (x_drt, x_special, match_tau_indices).  Raises ValueError when the QP breaks down (cvxopt's error)."""
    match = get_tau_indices(obs_tau_indices, truncate=truncate)
    special = offset_special_dict(obs_drt_list[0].special_qp_params)
    so = int(sum(v.get('size', 1) for v in special.values()))
    pq = [resize_pq(*get_offset_pq(drt), so, obs_tau_indices[i], match) for i, drt in enumerate(obs_drt_list)]
    nr, nc = len(pq), len(pq[0][1])

    # smoothness across observations, applied to the coefficients at their true scale (resolve.py:232-245, 271-272)
    ly = gaussian_filter1d(np.eye(nr), sigma=sigma, mode='reflect', order=2)
    scale_vec = np.array([drt.coefficient_scale for drt in obs_drt_list])
    lys = ly @ np.diag(scale_vec / gaussian_filter1d(median_filter(scale_vec, 3), 2))
    my = lys.T @ lys
    param_scale = np.ones(nc)
    if 'R_inf' in special:
        x_inf = np.array([drt.fit_parameters['R_inf'] / drt.coefficient_scale for drt in obs_drt_list])
        param_scale[special['R_inf']['index']] = (5 * np.std(x_inf)) ** -2
    dop = None
    if 'x_dop' in special:
        x_dop = np.array([drt.fit_parameters['x_dop'] / (drt.coefficient_scale * drt.dop_scale_vector)
                          for drt in obs_drt_list])
        dop = (special['x_dop']['index'], special['x_dop']['index'] + special['x_dop'].get('size', 1))
        param_scale[dop[0]:dop[1]] = (np.std(x_dop, axis=0) + 0.1 * np.std(x_dop)) ** -2
    m_full = np.zeros((nr * nc, nr * nc))
    diag = np.arange(nc)
    for i in range(nr):
        for j in range(nr):
            m_full[i * nc + diag, j * nc + diag] = param_scale * my[i, j] * lambda_psi
    if tau_filter_sigma > 0 or special_filter_sigma > 0:       # resolve.py:279-299, 319-320
        filt = np.eye(nc)
        if special_filter_sigma > 0 and dop is not None:
            filt[dop[0]:dop[1], dop[0]:dop[1]] = construct_func_eval_matrix(
                np.arange(dop[0], dop[1]), epsilon=1 / (np.sqrt(2) * special_filter_sigma), order=0)
        if tau_filter_sigma > 0:
            filt[so:, so:] = construct_func_eval_matrix(np.arange(nc - so), epsilon=1 / (np.sqrt(2) * tau_filter_sigma),
                                                        order=0)
        full = np.kron(np.eye(nr), filt)
        m_full = full @ m_full @ full
    p_matrix = m_full
    for i, (p, _) in enumerate(pq):
        p_matrix[i * nc:(i + 1) * nc, i * nc:(i + 1) * nc] += p
    q_vector = np.concatenate([q for _, q in pq])
    h = np.zeros(nr * nc) if nonneg else 10 * np.ones(nr * nc)
    for v in special.values():
        if v['nonneg']:
            for i in range(nr):
                h[v['index'] + i * nc:v['index'] + v.get('size', 1) + i * nc] = 0

    if _assemble_only:
        return p_matrix, q_vector, h, special, match, nr, nc
    res = _ffi.get_context(device).qp_batch(p_matrix[None], q_vector[None], h)
    if res['status'][0] < 0:
        raise ValueError("Rank(A) < p or Rank([P; A; G]) < n")
    resolve_observations.last_qp = dict(iterations=int(res['iterations'][0]), status=int(res['status'][0]))
    x_opt = res['x'][0].reshape(nr, nc)
    if unpack:
        x_drt, x_special = unpack_resolved_x(x_opt, obs_drt_list, special)
        return x_drt, x_special, match
    return x_opt, match


def unpack_resolved_x(x, obs_drt_list, special_dict):
    """resolve.unpack_resolved_x (344-376): coefficients back in data units."""
    so = int(sum(v.get('size', 1) for v in special_dict.values()))
    cs = np.array([drt.coefficient_scale for drt in obs_drt_list])
    x_drt = x[:, so:] * cs[:, None]
    x_special = {}
    for key, info in special_dict.items():
        xk = x[:, info['index']:info['index'] + info.get('size', 1)] * cs[:, None]
        if key == 'x_dop':
            xk = xk * np.array([drt.dop_scale_vector for drt in obs_drt_list])
        elif key == 'inductance':
            xk = xk * np.array([drt.inductance_scale for drt in obs_drt_list])[:, None]
        x_special[key] = xk.flatten() if info.get('size', 1) == 1 else xk
    return x_drt, x_special


def resolve_group(obs_drt_list, obs_tau_indices, nonneg, num_tau_super, batch_size=7, overlap=2, truncate=False, sigma=1,
                  lambda_psi=1, tau_filter_sigma=0, special_filter_sigma=0, device=0):
    """DRTMD.resolve_group (hybdrt/mapping/drtmd.py:486-559) for observations already sorted along psi: overlapping batches
    of `batch_size` observations are re-optimised coherently (resolve_observations each) and the overlaps averaged with
    weights growing with the distance from the batch edge.  The batches are independent QPs: all of one size go to the device as
    ONE batched launch (one size when every observation was fitted on the same tau range).  Returns (obs_x_resolved (num_obs, num_tau_super), obs_special_resolved dict)."""
    num_obs = len(obs_drt_list)
    batch_size = min(batch_size, num_obs)
    stride = max(batch_size - overlap, 1)
    num_batches = 1 + int(np.ceil((num_obs - batch_size) / stride))
    starts = []
    for start in range(0, num_obs, stride):
        if num_obs - start < batch_size:
            start = max(0, num_obs - batch_size)          # a full batch for the last one
        starts.append(start)
        if start + batch_size >= num_obs:
            break
    if num_obs == 1:
        raise ValueError("Only one observation included in resolution group")
    probs = [resolve_observations(obs_drt_list[a:a + batch_size], obs_tau_indices[a:a + batch_size], nonneg,
                                  truncate=truncate, sigma=sigma, lambda_psi=lambda_psi, tau_filter_sigma=tau_filter_sigma,
                                  special_filter_sigma=special_filter_sigma, _assemble_only=True) for a in starts]
    # batches whose common tau ranges have the same length are QPs of one size: one batched launch per size (observations
    # fitted on different slices of the supergrid give batches of different sizes, drtmd.py:513-528 + resolve.py:219-232)
    by_size = {}
    for i, pr in enumerate(probs):
        by_size.setdefault(pr[0].shape, []).append(i)
    x_sol, iterations = [None] * len(probs), [0] * len(probs)
    for members in by_size.values():
        res = _ffi.get_context(device).qp_batch(np.stack([probs[i][0] for i in members]), np.stack([probs[i][1] for i in members]),
                                                np.stack([probs[i][2] for i in members]))
        if np.any(res['status'] < 0):
            raise ValueError("Rank(A) < p or Rank([P; A; G]) < n")
        for j, i in enumerate(members):
            x_sol[i], iterations[i] = res['x'][j], int(res['iterations'][j])
    resolve_group.last_qp = dict(iterations=iterations, launches=len(by_size))
    special = probs[0][3]
    x_batch = np.zeros((len(starts), num_obs, num_tau_super))
    sp_batch = {k: np.zeros((len(starts), num_obs) + ((v.get('size', 1),) if v.get('size', 1) > 1 else ()))
                for k, v in special.items()}
    margins = -np.ones((len(starts), num_obs))
    # The reference runs the batches one after the other on ONE array: a batch overwrites its observations' rows inside its own
    # common tau range only (drtmd.py:476) and then records a copy of those rows (drtmd.py:531) -- where a later batch's range is
    # shorter, the copy still holds what the earlier batch wrote outside it.  Same here, on the solutions of the batched launch.
    rows = np.zeros((num_obs, num_tau_super))
    for i, (a, pr) in enumerate(zip(starts, probs)):
        _, _, _, sp_, match, nr, nc = pr
        x_drt, x_special = unpack_resolved_x(x_sol[i].reshape(nr, nc), obs_drt_list[a:a + batch_size], sp_)
        rows[a:a + batch_size, match[0]:match[1]] = x_drt
        x_batch[i, a:a + batch_size] = rows[a:a + batch_size]
        for k, v in x_special.items():
            sp_batch[k][i, a:a + batch_size] = v
        margins[i, a:a + batch_size] = np.minimum(np.arange(batch_size), np.arange(batch_size)[::-1])
    if overlap > 0 and num_obs > 1:
        w = margins + 0.1                          # drtmd.py:541-546: edge observations still count a little
        w[w < 0] = 0
        x_res = np.average(x_batch, axis=0, weights=np.repeat(w[:, :, None], num_tau_super, axis=2))
        sp_res = {k: np.average(v, axis=0, weights=w if v.ndim == 2 else np.repeat(w[:, :, None], v.shape[-1], axis=2))
                  for k, v in sp_batch.items()}
    else:
        # without overlap every observation is written by exactly one batch
        x_res = x_batch.sum(axis=0)
        sp_res = {k: v.sum(axis=0) for k, v in sp_batch.items()}
    return x_res, sp_res
