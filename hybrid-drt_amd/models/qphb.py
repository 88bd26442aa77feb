"""Drop-in for the QP seam of hybdrt/models/qphb.py: same names / argument meaning / result keys, arithmetic on
the GPU (batched coneqp kernel).  The per-iteration hyper-parameter updates (solve_s, solve_rho,
estimate_weights, iterate_qphb) live inside the device-resident fit loop (csrc/hyper.hip) and are reached
through ``DRT.fit_eis`` / ``fit_eis_batch``; they are not re-exposed one by one because a host round trip per
update would defeat the point of keeping the loop on the device."""
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
