"""Drop-in for the QP seam of hybdrt/models/qphb.py: same names / argument meaning / result keys, arithmetic on
the GPU (batched coneqp kernel).  ``iterate_qphb`` is one pass of the device loop's body (Gram + QP + the s / rho / weight
updates of csrc/hyper.hip) behind the reference's signature; the pieces inside it (solve_s, solve_rho,
estimate_weights) are not re-exposed one by one because a host round trip per update would defeat the point of
keeping the loop on the device.  Whole fits go through ``DRT.fit_eis`` / ``fit_eis_batch`` / ``_qphb_fit_core``."""
import zlib

import numpy as np

from .. import _ffi


def get_default_hypers(eff_hp=True, fit_dop=False, nu_basis_type='gaussian'):
    """qphb.get_default_hypers (hybdrt/models/qphb.py:208-255)."""
    if eff_hp:
        s_alpha = np.array([5, 10, 25])
        rho_alpha = np.array([0.15, 0.2, 0.25])
    else:
        s_alpha = np.array([1.05, 1.15, 2.5])
        rho_alpha = np.array([0.05, 0.1, 0.05])
    hypers = dict(rp_scale=14, derivative_weights=np.array([1.5, 1.0, 0.5]), sigma_ds=np.array([1, 1000, 1000]),
                  l1_lambda_0=0, l2_lambda_0=142, iw_alpha=None, iw_beta=None, s_alpha=s_alpha, s_0=np.ones(3),
                  rho_alpha=rho_alpha, rho_0=np.ones(3), outlier_p=None)
    if fit_dop:      # qphb.py:243-253
        hypers.update(dop_l2_lambda_0=10, dop_l1_lambda_0=0, dop_derivative_weights=np.array([0.5, 1.0, 0.5]),
                      dop_s_alpha=np.array([5, 10, 25]), dop_rho_alpha=np.array([0.15, 0.2, 0.25]),
                      dop_s_0=np.ones(3), dop_rho_0=np.ones(3), dop_sigma_ds=np.array([1, 1000, 1000]))
    return hypers


def get_num_special(special_qp_params):
    """qphb.py:44-48."""
    if len(special_qp_params) == 0:
        return 0
    return int(np.sum([qp.get('size', 1) for qp in special_qp_params.values()]))


def make_h_constraint(wrm, n, special_params, nonneg, nonlin=False, neg_allowed_indices=None):
    """qphb.make_h_constraint (hybdrt/models/qphb.py:521-557): x >= -h."""
    if nonlin:
        raise NotImplementedError("nonlin is an experimental branch outside the hot path")
    if nonneg:
        h = np.zeros(n)
        for sp in special_params.values():
            if not sp['nonneg']:
                h[sp['index']:sp['index'] + sp.get('size', 1)] = 1000
    else:
        if neg_allowed_indices is not None:
            h = make_h_constraint(wrm, n, special_params, nonneg=True)
            h[neg_allowed_indices] = 1e5
        else:
            h = 1e5 * np.ones(n)
        for sp in special_params.values():
            if sp['nonneg']:
                h[sp['index']:sp['index'] + sp.get('size', 1)] = 0
    return h


_STATUS = {_ffi.QP_OPTIMAL: 'optimal', _ffi.QP_MAXITER: 'unknown', _ffi.QP_SINGULAR_LATE: 'unknown'}


def solve_convex_opt(wrv, wrm, l2_matrix, l1v, nonneg, special_params, init_vals=None, fixed_x_index=None,
                     fixed_x_values=None, include_fixed_cov=True, curvature_constraint=None, nonlin=False,
                     neg_allowed_indices=None, device=0):
    """qphb.solve_convex_opt (hybdrt/models/qphb.py:426-519): P = wrm'wrm + l2_matrix, q = -wrm'wrv + l1v,
    G = -I, h from make_h_constraint, solved with cvxopt.coneqp's trajectory on the GPU.

    Returns a dict with the cvxopt result keys the reference reads ('x', 'primal objective') plus 'status' and
    'iterations'.  A singular KKT system at the start point raises ValueError like cvxopt does."""
    if fixed_x_index is not None or curvature_constraint is not None or nonlin or init_vals is not None:
        raise NotImplementedError("fixed_x_index / curvature_constraint / nonlin / init_vals are experimental "
                                  "branches never used by the fit methods (drt1d.py:942)")
    wrm = np.asarray(wrm, dtype=float)
    wrv = np.asarray(wrv, dtype=float)
    n = wrm.shape[1]
    ctx = _ffi.get_context(device)
    l1 = np.broadcast_to(np.asarray(l1v, dtype=float), (n,)).copy()
    # weights are already folded into wrm / wrv by the caller, as in the reference
    P, q = ctx.weighted_gram(wrm, np.ones(wrm.shape[0]), wrv, l2=np.asarray(l2_matrix, dtype=float), l1=l1)
    h = make_h_constraint(wrm, n, special_params, nonneg, neg_allowed_indices=neg_allowed_indices)
    res = ctx.qp_batch(P[0], q, h)
    if res['status'][0] == _ffi.QP_SINGULAR:
        raise ValueError("Rank(A) < p or Rank([P; A; G]) < n")
    if res['status'][0] == _ffi.QP_ABORTED:
        raise RuntimeError("hipdrt: the QP's workgroups were not co-resident (device shared with another process?)")
    return {'x': res['x'][0], 'primal objective': float(res['pcost'][0]), 'status': _STATUS[int(res['status'][0])],
            'iterations': int(res['iterations'][0])}


def solve_qp_batch(P, q, h, device=0):
    """B independent cvxopt.solvers.qp(P, q, -I, h) calls in one launch (the batched form of qphb.py:512-519)."""
    return _ffi.get_context(device).qp_batch(P, q, h)


_iter_plan = {}      # one cached prepared plan: successive iterate_qphb calls of a loop share their matrices


def _is_sym_toeplitz(a):
    a = np.asarray(a)
    n = a.shape[0]
    if n < 2 or not np.array_equal(a, a.T):
        return False
    return all(np.all(np.diagonal(a, k) == a[0, k]) for k in range(n))


def _iterate_plan(device, rm, vmm, penalty_matrices, l1v, h, hypers, eff_hp, x_rtol, special_qp_params, batch):
    ns = get_num_special(special_qp_params)
    m, n = rm.shape[-2:]
    mats = [np.ascontiguousarray(penalty_matrices[f'm{k}'], dtype=float) for k in range(3)]
    vmm = np.ascontiguousarray(vmm, dtype=float)
    hyp_key = tuple((k, tuple(np.ravel(v).tolist()) if v is not None else None) for k, v in sorted(hypers.items()))
    key = (device, m, n, ns, batch, bool(eff_hp), float(x_rtol), hyp_key, h.tobytes(), l1v.tobytes(),
           tuple(sorted((k, v['index'], v.get('size', 1)) for k, v in special_qp_params.items())),
           tuple(zlib.crc32(a.tobytes()) for a in mats + [vmm]))
    if _iter_plan.get('key') == key:
        return _iter_plan['plan']
    d = _ffi.PreparedDesc()
    d.m, d.n, d.ns = m, n, ns
    d.vz_index = -1
    dop = special_qp_params.get('x_dop')
    if dop is not None:
        d.dop_start, d.dop_size = int(dop['index']), int(dop.get('size', 1))
        d.dop_l2_lambda_0 = float(hypers['dop_l2_lambda_0'])
        for name in ('dop_derivative_weights', 'dop_s_alpha', 'dop_rho_alpha', 'dop_s_0', 'dop_rho_0'):
            vals = np.broadcast_to(np.asarray(hypers[name], dtype=float), (3,))
            for k in range(3):
                getattr(d, name)[k] = float(vals[k])
    d.toeplitz_m = int(all(_is_sym_toeplitz(a[ns:, ns:]) for a in mats))
    o = _ffi.default_fit_opts()
    for name in ('derivative_weights', 'sigma_ds', 's_alpha', 's_0', 'rho_alpha', 'rho_0'):
        vals = np.broadcast_to(np.asarray(hypers[name], dtype=float), (3,))
        for k in range(3):
            getattr(o, name)[k] = float(vals[k])
    o.l2_lambda_0 = float(hypers['l2_lambda_0'])
    o.outlier_p = -1.0 if hypers.get('outlier_p') is None else float(hypers['outlier_p'])
    o.eff_hp, o.xtol, o.max_iter = int(bool(eff_hp)), float(x_rtol), 1
    plan = _ffi.PreparedPlan(_ffi.get_context(device), d, mats, vmm, h, l1v, opts=o, capacity=batch)
    _iter_plan.clear()
    _iter_plan.update(key=key, plan=plan)
    return plan


def iterate_qphb(x_in, s_vectors, rho_vector, dop_rho_vector, rv, weights, est_weights, out_tvt,
                 rm, vmm, penalty_matrices, penalty_type, l1_lambda_vector,
                 hypers, eff_hp,
                 xmx_norms, dop_xmx_norms, fixed_x_index, fixed_x_values, curvature_constraint,
                 nonneg, special_qp_params, x_rtol, max_hp_iter, history, nonlin=False,
                 neg_allowed_indices=None, device=0):
    """qphb.iterate_qphb (hybdrt/models/qphb.py:606-972), same arguments and the same nine results
    (x, s_vectors, rho_vector, dop_rho_vector, weights, outlier_t, out_tvt, cvx_result, converged), computed by one
    hipdrt_plan_iterate: Gram + q, the coneqp QP, solve_s / solve_rho for the DRT block and the DOP block, estimate_weights
    and is_converged all on the device.

    Batched form: give rv as (B, m) with x_in (B, n), s_vectors (B, 3, n), rho_vector (B, 3), weights / est_weights (B, m),
    xmx_norms (B, 3) and rm either shared (m, n) or (B, m, n); every result then carries the leading B and cvx_result is a
    list.  The branches _qphb_fit_core never takes (drt1d.py:940-943) are not built: fixed_x_index, curvature_constraint,
    nonlin, penalty_type 'discrete', max_hp_iter != 1."""
    if fixed_x_index is not None or fixed_x_values is not None or curvature_constraint or nonlin:
        raise NotImplementedError("fixed_x_index / curvature_constraint / nonlin are experimental branches never used "
                                  "by the fit methods (drt1d.py:942)")
    if penalty_type != 'integral':
        raise NotImplementedError("penalty_type 'discrete' is deprecated in the reference and not built")
    if max_hp_iter != 1:
        raise NotImplementedError("max_hp_iter != 1: the fit methods always pass 1 (drt1d.py:943)")
    if est_weights is None:
        raise NotImplementedError("est_weights=None: _qphb_fit_core always passes the initial estimate (drt1d.py:940)")
    rv = np.asarray(rv, dtype=float)
    single = rv.ndim == 1
    B = 1 if single else rv.shape[0]
    rm = np.asarray(rm, dtype=float)
    m, n = rm.shape[-2:]
    lead = (lambda a: None if a is None else np.asarray(a, dtype=float)[None]) if single else \
        (lambda a: None if a is None else np.asarray(a, dtype=float))
    l1v = np.broadcast_to(np.asarray(l1_lambda_vector, dtype=float), (n,)).copy()
    h = make_h_constraint(rm, n, special_qp_params, nonneg, neg_allowed_indices=neg_allowed_indices)
    plan = _iterate_plan(device, rm, vmm, penalty_matrices, l1v, np.asarray(h, dtype=float), hypers, eff_hp, x_rtol,
                         special_qp_params, B)
    plan.upload(rm, rv if not single else rv[None])
    has_dop = 'x_dop' in special_qp_params
    res = plan.iterate(x_in=lead(x_in), s_vectors=lead(np.asarray(s_vectors, dtype=float)), rho=lead(rho_vector),
                       dop_rho=lead(dop_rho_vector) if has_dop else None, weights=lead(weights),
                       est_weights=lead(est_weights), xmx_norms=lead(xmx_norms),
                       dop_xmx_norms=lead(dop_xmx_norms) if has_dop else None)
    if np.any(res['qp_status'] == _ffi.QP_SINGULAR):
        raise ValueError("Rank(A) < p or Rank([P; A; G]) < n")
    out = plan.download(s_vectors=True)
    x, w_out, rho_out, s_out = out['x'], out['weights'], out['rho'], out['s_vectors']
    dop_out = plan.get('dop_rho') if has_dop else None
    if hypers.get('outlier_p') is not None:
        outlier_t = plan.get('outlier_t')
        sq = outlier_t ** 0.5
        tvt = sq[:, :, None] * np.asarray(vmm, dtype=float)[None] * sq[:, None, :]     # qphb.py:1522-1539
        idx = np.arange(m)
        tvt[:, idx, idx] += 1 - outlier_t
    else:
        outlier_t, tvt = np.ones((B, m)), None
    cvx = [{'x': x[b], 'primal objective': float(res['primal_objective'][b]),
            'status': _STATUS[int(res['qp_status'][b])], 'iterations': int(res['qp_iters'][b])} for b in range(B)]
    conv = res['converged']
    if history is not None:          # qphb.py:947-964
        for b in range(B):
            history.append({'x': x[b].copy(), 's_vectors': s_out[b].copy(), 'rho_vector': rho_out[b].copy(),
                            'dop_rho_vector': None if dop_out is None else dop_out[b].copy(),
                            'weights': w_out[b].copy(), 'outlier_t': outlier_t[b], 'fun': cvx[b]['primal objective'],
                            'cvx_result': cvx[b]})
    if single:
        return (x[0], s_out[0], rho_out[0], None if dop_out is None else dop_out[0], w_out[0], outlier_t[0],
                None if tvt is None else tvt[0], cvx[0], bool(conv[0]))
    return x, s_out, rho_out, dop_out, w_out, outlier_t, tvt, cvx, conv
