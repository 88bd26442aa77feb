"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement of ``cvxopt.solvers.qp`` -> ``cvxopt.coneqp`` for the only problem shape the
reference's hot path ever hands it (hybdrt/models/qphb.py:472-519):

    minimise  1/2 x'Px + q'x   s.t.  G x + s = h,  s >= 0,   with  G = -I  (one 'l' cone of size n,
    no equality constraints, ``initvals=None``, default options).

cvxopt is an *un-vendored, un-pinned* third-party dependency of the reference (``requirements.txt:4``,
``setup.py:16``: bare "cvxopt"); it is not installable in the build container.  What is restated here is
its published algorithm (cvxopt 1.3.x ``coneprog.coneqp`` with the default ``kktsolver='chol2'`` and
``misc.compute_scaling / update_scaling / scale / scale2 / sinv / max_step`` restricted to the 'l' block):
Mehrotra predictor-corrector, Nesterov-Todd scaling W = diag(d), STEP = 0.99, EXPON = 3, default start
point, default tolerances, the exact stopping test.  The *operation order* of cvxopt is kept (scaled
variables ds~, dz~, lambda update by sqrt(ds)*sqrt(dz), d update d*sqrt(ds)/sqrt(dz)) so that the
trajectory -- which is what pins the reference's results, because termination happens after 2-8
iterations -- is reproduced, not just the optimum.

Pinned by: the reference's own known-answer test ``tests/test_drt_fit.py`` (7 consecutive coneqp solves
feed its golden x / R_inf / inductance / z_sigma_tot / q_vector); see tests/test_oracle_golden.py.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.linalg import cho_solve, cholesky, solve_triangular

# cvxopt.solvers.options defaults (the reference only sets show_progress=False, qphb.py:25)
ABSTOL = 1e-7
RELTOL = 1e-6
FEASTOL = 1e-7
MAXITERS = 100
STEP = 0.99
EXPON = 3


class KKTError(ArithmeticError):
    """Cholesky breakdown of P + diag(di^2) (cvxopt raises ArithmeticError / ValueError)."""


def _factor(P: np.ndarray, di: np.ndarray) -> np.ndarray:
    """kkt_chol2 'factor' for G=-I: S = P + Gs'Gs with Gs = -diag(di); lower Cholesky of S."""
    S = P.copy()
    idx = np.arange(P.shape[0])
    S[idx, idx] += di * di
    try:
        return cholesky(S, lower=True, check_finite=False)
    except np.linalg.LinAlgError as err:  # pragma: no cover - exercised through status path
        raise KKTError(str(err))


def _kkt_solve(L: np.ndarray, di: np.ndarray, bx: np.ndarray, bz: np.ndarray):
    """kkt_chol2 'solve' for G=-I, no equalities.  Returns (ux, W*uz) like cvxopt's f3."""
    zz = bz * di                      # z := W^{-T} bz
    xx = bx - di * zz                 # x := bx + Gs' z          (Gs = -diag(di))
    ux = cho_solve((L, True), xx, check_finite=False)
    zs = -di * ux - zz                # W*uz := Gs ux - z
    return ux, zs


def coneqp_boxlow(P: np.ndarray, q: np.ndarray, h: np.ndarray, *, abstol=ABSTOL, reltol=RELTOL,
                  feastol=FEASTOL, maxiters=MAXITERS, trace: list | None = None) -> dict:
    """Solve min 1/2 x'Px + q'x s.t. -x <= h with cvxopt.coneqp's trajectory.

    Returns a dict with the cvxopt result keys the reference consumes ('x', 'primal objective',
    qphb.py:676,960) plus 'status', 'iterations', 's', 'z', 'gap'.
    """
    P = np.ascontiguousarray(P, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64).ravel()
    h = np.asarray(h, dtype=np.float64).ravel()
    n = q.size

    resx0 = max(1.0, math.sqrt(float(q @ q)))
    resz0 = max(1.0, math.sqrt(float(h @ h)))

    # ---- default starting point: KKT solve with W = I -------------------------------------------
    ones = np.ones(n)
    try:
        L = _factor(P, ones)
    except KKTError:
        raise ValueError("Rank(A) < p or Rank([P; A; G]) < n")
    x, z = _kkt_solve(L, ones, -q, h.copy())
    s = -z
    nrms = math.sqrt(float(s @ s))
    ts = float(np.max(-s))
    if ts >= -1e-8 * max(nrms, 1.0):
        s = s + (1.0 + ts)
    nrmz = math.sqrt(float(z @ z))
    tz = float(np.max(-z))
    if tz >= -1e-8 * max(nrmz, 1.0):
        z = z + (1.0 + tz)

    gap = float(s @ z)
    d = di = lmbda = None
    status = "unknown"

    for iters in range(maxiters + 1):
        # residuals and costs
        Px = P @ x
        rx = Px + q
        f0 = 0.5 * (float(x @ rx) + float(x @ q))
        rx = rx - z                                  # + G'z
        resx = math.sqrt(float(rx @ rx))
        rz = s - h - x                               # s + Gx - h
        resz = math.sqrt(float(rz @ rz))

        pcost = f0
        dcost = f0 + float(z @ rz) - gap
        if pcost < 0.0:
            relgap = gap / -pcost
        elif dcost > 0.0:
            relgap = gap / dcost
        else:
            relgap = None
        pres = resz / resz0
        dres = resx / resx0
        if trace is not None:
            trace.append(dict(it=iters, pcost=pcost, dcost=dcost, gap=gap, pres=pres, dres=dres))

        if (pres <= feastol and dres <= feastol and
                (gap <= abstol or (relgap is not None and relgap <= reltol))) or iters == maxiters:
            status = "optimal" if iters < maxiters or (
                pres <= feastol and dres <= feastol and
                (gap <= abstol or (relgap is not None and relgap <= reltol))) else "unknown"
            break

        if iters == 0:
            d = np.sqrt(s / z)                       # misc.compute_scaling, 'l' block
            di = 1.0 / d
            lmbda = np.sqrt(s * z)
        lmbdasq = lmbda * lmbda

        try:
            L = _factor(P, di)
        except KKTError:
            if iters == 0:
                raise ValueError("Rank(A) < p or Rank([P; A; G]) < n")
            status = "unknown"
            break

        mu = gap / n
        sigma, eta = 0.0, 0.0
        dsdz_a = None
        for i in (0, 1):
            # cvxopt order: ds := 0; (i==1) ds -= ws3; ds -= lmbdasq; ds += sigma*mu
            ds = -dsdz_a - lmbdasq if i == 1 else -lmbdasq
            ds = ds + sigma * mu
            # f4_no_ir
            sv = ds / lmbda                           # misc.sinv
            bz = (-1.0 + eta) * rz - d * sv           # z := bz - W'(lmbda o\ bs)
            bx = (-1.0 + eta) * rx
            dx, dz = _kkt_solve(L, di, bx, bz)
            ds = sv - dz
            dsdz = float(ds @ dz)
            if i == 0:
                dsdz_a = ds * dz                      # Mehrotra correction term (ws3)
            ds = ds / lmbda                           # misc.scale2
            dz = dz / lmbda
            ts = float(np.max(-ds))
            tz = float(np.max(-dz))
            t = max(0.0, ts, tz)
            if t == 0.0:
                step = 1.0
            elif i == 0:
                step = min(1.0, 1.0 / t)
            else:
                step = min(1.0, STEP / t)
            if i == 0:
                sigma = min(1.0, max(0.0, 1.0 - step + dsdz / gap * step ** 2)) ** EXPON
                eta = 0.0

        x = x + step * dx
        # updated iterates in the current scaling, then misc.update_scaling
        ds = (1.0 + step * ds) * lmbda
        dz = (1.0 + step * dz) * lmbda
        sq_s = np.sqrt(ds)
        sq_z = np.sqrt(dz)
        d = d * sq_s / sq_z
        di = 1.0 / d
        lmbda = sq_s * sq_z
        s = lmbda * d
        z = lmbda * di
        gap = float(lmbda @ lmbda)

    return {
        "x": x, "s": s, "z": z, "status": status, "gap": gap, "iterations": iters,
        "primal objective": pcost, "dual objective": dcost, "relative gap": relgap,
        "primal infeasibility": pres, "dual infeasibility": dres,
    }
