"""Drop-in for the hot-path part of hybdrt/matrices/basis.py: the integral lookup tables, built on the GPU."""
import numpy as np

from .. import _ffi


def generate_impedance_lookup(basis_type, epsilon, grid_points=2000, zga_params=None, device=0):
    """basis.generate_impedance_lookup (hybdrt/matrices/basis.py:648-669).

    Returns ((log_wt_re, z_re), (log_wt_im, z_im)).  The abscissae are the reference's
    ``np.logspace(-2.7, 2.7, n)`` / ``np.logspace(-5.4, 5.4, n)``; the 1000-point trapezoid integrals over
    ``y = linspace(-20, 20, 1000)`` run on the device (one wavefront per table entry)."""
    if basis_type != 'gaussian':
        raise NotImplementedError("only the default gaussian basis is on the hot path (Cole-Cole/zga need mitlef)")
    re_lim = 2.7
    im_lim = re_lim * 2
    wt_re = np.logspace(-re_lim, re_lim, grid_points)
    wt_im = np.logspace(-im_lim, im_lim, grid_points)
    z_re, z_im = _ffi.get_context(device).impedance_lookup(epsilon, wt_re, wt_im, ny=1000)
    return (np.log(wt_re), z_re), (np.log(wt_im), z_im)


def generate_response_lookup(basis_type, op_mode, step_model, epsilon, grid_points=2000, tau_rise=None,
                             zga_params=None, device=0):
    """basis.generate_response_lookup (hybdrt/matrices/basis.py:672-689).

    Returns (log_td_grid, response_grid): the step response of one basis function as a function of
    ``(t - t_step) / tau`` on the reference's ``np.logspace(-6, 2, n)`` grid; the 1000-point trapezoid integrals
    (integrand basis.py:616-618) run on the device, one wavefront per table entry."""
    if basis_type != 'gaussian':
        raise NotImplementedError("only the default gaussian basis is on the hot path (Cole-Cole/zga need mitlef)")
    if op_mode != 'galv' or step_model != 'ideal':
        raise NotImplementedError("only galvanostatic ideal steps (the fit_chrono / fit_hybrid defaults) are built")
    td_grid = np.logspace(-6, 2, grid_points)
    response_grid = _ffi.get_context(device).response_lookup(epsilon, td_grid, ny=1000)
    return np.log(td_grid), response_grid


def construct_func_eval_matrix(basis_grid, eval_grid=None, basis_type='gaussian', epsilon=1, order=0, zga_params=None):
    """basis.construct_func_eval_matrix (hybdrt/matrices/basis.py:488-514) for the gaussian basis, order 0:
    em[i, j] = exp(-(epsilon (eval_i - basis_j))^2); a (neval x nbasis) host array (it is an input of the device
    posterior-variance kernel, not a hot loop)."""
    if basis_type != 'gaussian' or order != 0:
        raise NotImplementedError("only the gaussian basis, order 0, is built")
    basis_grid = np.asarray(basis_grid, dtype=float)
    eval_grid = basis_grid.copy() if eval_grid is None else np.asarray(eval_grid, dtype=float)
    xx_basis, xx_eval = np.meshgrid(basis_grid, eval_grid)
    return np.exp(-(epsilon * (xx_eval - xx_basis)) ** 2)
