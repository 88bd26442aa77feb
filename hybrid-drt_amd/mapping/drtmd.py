"""Batch driver: the fit-loop contract of hybdrt.mapping.drtmd.DRTMD (drtmd.py:186-329, 1136-1158) for
observations that share one frequency grid, with optional sharding over ranks (one process per GPU)."""
import numpy as np


def shard_bounds(num_obs, world_size, rank):
    """Contiguous block split of the observations over ranks: ceil-sized leading blocks."""
    base, rem = divmod(num_obs, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def fit_observations(drt, frequencies, z_obs, tau_supergrid=None, drt_var=False, **fit_kw):
    """Fit every observation (rows of z_obs) and scatter the coefficients into supergrid slots like
    DRTMD.fit_observation does (drtmd.py:263-275): returns obs_x (B, len(supergrid)), obs_special dict,
    and the raw result dict.  With ``drt_var=True`` the result dict also carries ``obs_drt_var`` (B, len(supergrid)),
    the diagonal of estimate_distribution_cov(tau=tau_supergrid, extend_var=True) of every observation
    (drtmd.py:278-279), and ``obs_drt_var_ok``."""
    res = drt.fit_eis_batch(frequencies, z_obs, **fit_kw)
    basis_tau = res['basis_tau']
    if tau_supergrid is None:
        tau_supergrid = basis_tau
    tau_supergrid = np.asarray(tau_supergrid)
    left = int(np.argmin(np.abs(np.log(tau_supergrid) - np.log(basis_tau[0]))))
    right = left + len(basis_tau)
    obs_x = np.zeros((z_obs.shape[0], len(tau_supergrid)))
    obs_x[:, left:right] = res['fit_x']
    obs_special = {'R_inf': res['R_inf'], 'inductance': res['inductance']}
    if drt_var:
        res['obs_drt_var'], res['obs_drt_var_ok'] = drt.estimate_distribution_var_batch(tau=tau_supergrid,
                                                                                       extend_var=True)
    return obs_x, obs_special, res
