"""CONTAINER-ONLY SHIM (test infrastructure): lets the read-only reference at /root/reference be imported
where the real ``cvxopt`` wheel is unavailable.  ``solvers.qp`` is routed to the oracle's restatement of
``coneqp`` (oracle/coneqp.py) for the G = -I problems the reference builds (qphb.py:472, basis.py:54,
resolve.py:314).  Used only by oracle/make_golden.py; never imported by the product or on the GPU box.

ORACLE_CVXOPT_SHIM=general routes ``solvers.qp`` to the second restatement instead (oracle/coneqp_general.py: cvxopt's
coneqp + kkt_chol2 for a GENERAL dense G, which is handed the reference's own G as it comes, no G = -I shortcut):
oracle/check_general_shim.py re-runs the reference through it and compares with the committed fixtures."""
import os

import numpy as np

from oracle.coneqp import coneqp_boxlow
from oracle.coneqp_general import coneqp_dense


def matrix(a, *args, **kw):
    return np.array(a, dtype=float)


class _Solvers:
    options = {}

    @staticmethod
    def qp(P, q, G=None, h=None, A=None, b=None, solver=None, kktsolver=None, initvals=None, **kw):
        P = np.asarray(P, dtype=float)
        q = np.asarray(q, dtype=float).ravel()
        G = np.asarray(G, dtype=float)
        h = np.asarray(h, dtype=float).ravel()
        n = q.size
        if os.environ.get("ORACLE_CVXOPT_SHIM") == "general":
            if initvals is not None or A is not None:
                raise NotImplementedError("oracle cvxopt shim: initvals / equality constraints not restated")
            res = coneqp_dense(P.T, q, G, h)            # (the reference hands over p_matrix.T and G itself, qphb.py:512-515)
            log = _Solvers.options.get("_oracle_log")
            if log is not None:
                log.append(dict(P=P.T.copy(), q=q.copy(), h=h.copy(), x=res["x"].copy(),
                                iterations=res["iterations"], pcost=res["primal objective"]))
            return res
        if G.shape != (n, n) or not np.array_equal(G, -np.eye(n)):
            raise NotImplementedError("oracle cvxopt shim only restates coneqp for G = -I")
        if initvals is not None or A is not None:
            raise NotImplementedError("oracle cvxopt shim: initvals / equality constraints not restated")
        log = _Solvers.options.get("_oracle_log")
        res = coneqp_boxlow(P.T, q, h)
        if log is not None:
            log.append(dict(P=P.T.copy(), q=q.copy(), h=h.copy(), x=res["x"].copy(),
                            iterations=res["iterations"], pcost=res["primal objective"]))
        return res


solvers = _Solvers()
