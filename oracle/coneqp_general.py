"""ORACLE (test infrastructure, never shipped, never on the product path).

SECOND, structurally independent restatement of ``cvxopt.solvers.qp`` -> ``cvxopt.coneprog.coneqp`` with the default KKT
solver ``misc.kkt_chol2``, for a GENERAL DENSE inequality matrix G (one componentwise 'l' cone of size m, no second-order or
semidefinite blocks, no equality constraints, ``initvals=None``, default options):

    minimise  1/2 x'Px + q'x    subject to    G x + s = h,   s >= 0.

Purpose: oracle/coneqp.py is specialised to G = -I (the only G the reference ever builds: hybdrt/models/qphb.py:472,
hybdrt/mapping/resolve.py:314, hybdrt/matrices/basis.py:54) -- it never forms G, scales no matrix and adds di^2 to a
diagonal.  This module shares no code with it and keeps the GENERAL structure of the library: G is a dense m x n array that
is really multiplied, ``kkt_chol2.factor`` scales its rows by W^-1 and forms  S = P + Gs'Gs  by a symmetric rank-m product
followed by a Cholesky factorisation, ``solve`` runs the two triangular substitutions one after the other, and the iterates
are updated in place in cvxopt's statement order (blas.scal / axpy / tbmv / tbsv on the 'l' block).  Feeding both modules the
reference's problems (G = -I handed over as a dense matrix, exactly as the reference hands it to cvxopt) must give the same
iteration counts and the same x to rounding: tests/test_oracle_general.py, oracle/check_general_shim.py.

Restated from the published sources of cvxopt 1.3.x (un-vendored, un-pinned dependency of the reference:
requirements.txt:4): ``coneprog.coneqp`` (start point, residuals, stopping test, Mehrotra predictor-corrector with STEP 0.99
and EXPON 3, ``f4_no_ir``), ``misc.kkt_chol2`` (p = 0 branch), ``misc.compute_scaling / update_scaling / scale / scale2 /
sinv / sprod / ssqr / sdot / snrm2 / max_step`` restricted to dims = {'l': m, 'q': [], 's': []}.
"""
from __future__ import annotations

import math

import numpy as np
from numpy.linalg import LinAlgError
from scipy.linalg import solve_triangular

OPTIONS = dict(abstol=1e-7, reltol=1e-6, feastol=1e-7, maxiters=100)
_STEP = 0.99
_EXPON = 3


# ---- misc.* on the 'l' block ---------------------------------------------------------------------------------------------
def _scale(x, W, trans="N", inverse="N"):
    """x := W x (inverse 'N') or W^-1 x (inverse 'I'); W = diag(d) is symmetric, so trans does not matter here.  x may be a
    vector or a matrix whose ROWS are scaled (misc.scale works on the columns of a matrix with m rows)."""
    w = W["di"] if inverse == "I" else W["d"]
    if x.ndim == 1:
        x *= w
    else:
        x *= w[:, None]


def _max_step(x):
    """misc.max_step, 'l' block: the smallest t with x + t e >= 0, i.e. -min(x)."""
    return -float(x.min())


def _compute_scaling(s, z, lmbda):
    W = {"d": np.sqrt(s / z)}
    W["di"] = W["d"] ** -1
    lmbda[:] = np.sqrt(s * z)
    return W


def _update_scaling(W, lmbda, s, z):
    """misc.update_scaling, 'l' block.  s, z hold the updated iterates in the current scaling; on return both hold
    their square roots, W is the new scaling and lmbda = W z = W^-T s."""
    np.sqrt(s, out=s)
    np.sqrt(z, out=z)
    W["d"] *= s                 # blas.tbmv(s, W['d'])
    W["d"] /= z                 # blas.tbsv(z, W['d'])
    W["di"] = W["d"] ** -1
    lmbda[:] = s                # blas.copy(s, lmbda)
    lmbda *= z                  # blas.tbmv(z, lmbda)


# ---- misc.kkt_chol2, no equality constraints -----------------------------------------------------------------------------
def _kkt_chol2(G, P):
    m, n = G.shape

    def factor(W):
        Gs = G.copy()
        _scale(Gs, W, trans="T", inverse="I")          # Gs = W^-T G
        S = P + Gs.T @ Gs                               # blas.syrk(Gs, S, trans='T', beta=1.0) on a copy of P
        try:
            L = np.linalg.cholesky(S)                   # lapack.potrf
        except LinAlgError as err:
            raise ArithmeticError(str(err))

        def solve(x, z):
            """[P G'W^-1; W^-T G -I] [ux; W uz] = [bx; W^-T bz], in place: x <- ux, z <- W uz"""
            _scale(z, W, trans="T", inverse="I")        # z := W^-T bz
            x += Gs.T @ z                               # blas.gemv(Gs, z, x, beta=1.0, trans='T')
            x[:] = solve_triangular(L, x, lower=True, check_finite=False)             # lapack.trsv(S, x)
            x[:] = solve_triangular(L, x, lower=True, trans="T", check_finite=False)   # lapack.trsv(S, x, trans='T')
            z *= -1.0                                   # blas.gemv(Gs, x, z, beta=-1.0)
            z += Gs @ x

        return solve

    return factor


def coneqp_dense(P, q, G, h, *, abstol=None, reltol=None, feastol=None, maxiters=None):
    """cvxopt.solvers.qp(P, q, G, h) with every inequality in one 'l' cone.  Returns the result keys the reference reads
    ('x', 'primal objective') plus 'status', 'iterations', 's', 'z', 'gap'."""
    abstol = OPTIONS["abstol"] if abstol is None else abstol
    reltol = OPTIONS["reltol"] if reltol is None else reltol
    feastol = OPTIONS["feastol"] if feastol is None else feastol
    maxiters = OPTIONS["maxiters"] if maxiters is None else maxiters
    P = np.array(P, dtype=np.float64)
    q = np.array(q, dtype=np.float64).ravel()
    G = np.array(G, dtype=np.float64)
    h = np.array(h, dtype=np.float64).ravel()
    m, n = G.shape
    if P.shape != (n, n) or q.size != n or h.size != m:
        raise ValueError("inconsistent dimensions")

    resx0 = max(1.0, math.sqrt(float(np.dot(q, q))))
    resz0 = max(1.0, math.sqrt(float(np.dot(h, h))))          # misc.snrm2(h, dims)
    kktsolver = _kkt_chol2(G, P)

    # ---- default starting point: the KKT system with W = I and right-hand side (-q, h) ---------------------------------
    x = -q.copy()
    z = h.copy()
    s = np.empty(m)
    lmbda = np.empty(m)
    try:
        f = kktsolver({"d": np.ones(m), "di": np.ones(m)})
    except ArithmeticError:
        raise ValueError("Rank(A) < p or Rank([P; A; G]) < n")
    f(x, z)
    s[:] = z
    s *= -1.0
    nrms = math.sqrt(float(np.dot(s, s)))
    ts = _max_step(s)
    if ts >= -1e-8 * max(nrms, 1.0):
        s += 1.0 + ts
    nrmz = math.sqrt(float(np.dot(z, z)))
    tz = _max_step(z)
    if tz >= -1e-8 * max(nrmz, 1.0):
        z += 1.0 + tz

    rx = np.empty(n)
    rz = np.empty(m)
    dx = np.empty(n)
    dz = np.empty(m)
    ds = np.empty(m)
    ws3 = np.empty(m)
    lmbdasq = np.empty(m)
    W = None
    gap = float(np.dot(s, z))                                  # misc.sdot
    status, pcost, dcost, relgap, pres, dres = "unknown", 0.0, 0.0, None, 0.0, 0.0

    for iters in range(maxiters + 1):
        # f0 = 1/2 x'Px + q'x,  rx = P x + q + G'z
        rx[:] = q
        rx += P @ x                                            # fP(x, rx, beta=1.0)
        f0 = 0.5 * (float(np.dot(x, rx)) + float(np.dot(x, q)))
        rx += G.T @ z                                          # fG(z, rx, beta=1.0, trans='T')
        resx = math.sqrt(float(np.dot(rx, rx)))
        # rz = s + G x - h
        rz[:] = s
        rz -= h                                                # blas.axpy(h, rz, alpha=-1.0)
        rz += G @ x                                            # fG(x, rz, beta=1.0)
        resz = math.sqrt(float(np.dot(rz, rz)))

        pcost = f0
        dcost = f0 + float(np.dot(z, rz)) - gap
        if pcost < 0.0:
            relgap = gap / -pcost
        elif dcost > 0.0:
            relgap = gap / dcost
        else:
            relgap = None
        pres = resz / resz0
        dres = resx / resx0
        done = pres <= feastol and dres <= feastol and (gap <= abstol or (relgap is not None and relgap <= reltol))
        if done or iters == maxiters:
            status = "optimal" if done else "unknown"
            break

        if iters == 0:
            W = _compute_scaling(s, z, lmbda)
        np.multiply(lmbda, lmbda, out=lmbdasq)                 # misc.ssqr
        try:
            f3 = kktsolver(W)
        except ArithmeticError:
            if iters == 0:
                raise ValueError("Rank(A) < p or Rank([P; A; G]) < n")
            status = "unknown"                                 # "Terminated (singular KKT matrix)."
            break

        def f4_no_ir(x_, z_, s_):
            s_ /= lmbda                                        # misc.sinv: s := lmbda o\ bs
            ws3[:] = s_
            _scale(ws3, W, trans="T")                          # W'(lmbda o\ bs)
            z_ -= ws3                                          # blas.axpy(ws3, z, alpha=-1.0)
            f3(x_, z_)
            s_ -= z_                                           # blas.axpy(z, s, alpha=-1.0)

        mu = gap / m
        sigma, eta = 0.0, 0.0
        step = 1.0
        for i in (0, 1):
            ds[:] = 0.0                                        # blas.scal(0.0, ds)
            if i == 1:
                ds -= ws3                                      # (ws3 = ds o dz of the predictor, saved below)
            ds -= lmbdasq
            ds += sigma * mu
            dx[:] = 0.0
            dx += (-1.0 + eta) * rx                            # xaxpy(rx, dx, alpha=-1.0 + eta)
            dz[:] = 0.0
            dz += (-1.0 + eta) * rz
            f4_no_ir(dx, dz, ds)
            dsdz = float(np.dot(ds, dz))
            if i == 0:
                ws3[:] = ds                                    # blas.copy(ds, ws3); misc.sprod(ws3, dz)
                ws3 *= dz
            ds /= lmbda                                        # misc.scale2(lmbda, ds)
            dz /= lmbda
            t = max(0.0, _max_step(ds), _max_step(dz))
            if t == 0.0:
                step = 1.0
            elif i == 0:
                step = min(1.0, 1.0 / t)
            else:
                step = min(1.0, _STEP / t)
            if i == 0:
                sigma = min(1.0, max(0.0, 1.0 - step + dsdz / gap * step ** 2)) ** _EXPON
                eta = 0.0

        x += step * dx                                         # xaxpy(dx, x, alpha=step)
        # the 'l' blocks of ds, dz become the updated iterates in the current scaling
        ds *= step
        dz *= step
        ds += 1.0
        dz += 1.0
        ds *= lmbda                                            # misc.scale2(lmbda, ds, inverse='I')
        dz *= lmbda
        _update_scaling(W, lmbda, ds, dz)
        s[:] = lmbda                                           # unscaled s = W' lmbda, z = W^-1 lmbda
        _scale(s, W, trans="T")
        z[:] = lmbda
        _scale(z, W, inverse="I")
        gap = float(np.dot(lmbda, lmbda))

    return {
        "x": x, "s": s, "z": z, "status": status, "gap": gap, "iterations": iters,
        "primal objective": pcost, "dual objective": dcost, "relative gap": relgap,
        "primal infeasibility": pres, "dual infeasibility": dres,
    }
