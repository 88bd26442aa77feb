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
