"""Drop-in for hybdrt/matrices/phasance.py (distribution of phasances), gaussian nu basis, normalize=False: the DOP
impedance / time-response columns are built on the device."""
import numpy as np

from .. import _ffi


def _check(nu_basis_type, normalize):
    if nu_basis_type != 'gaussian':
        raise NotImplementedError("only the fit's gaussian nu basis is built (delta is a plain power law)")
    if normalize:
        raise NotImplementedError("normalize=True (tau_c scaling) is not built")


def construct_phasor_z_matrix(frequencies, basis_nu, nu_basis_type, nu_epsilon, normalize=False, tau_c=None, device=0):
    """phasance.construct_phasor_z_matrix (hybdrt/matrices/phasance.py:108-118): complex (nf, n_nu)."""
    _check(nu_basis_type, normalize)
    return _ffi.get_context(device).phasor_z_matrix(frequencies, basis_nu, nu_epsilon)


def construct_phasor_v_matrix(times, basis_nu, nu_basis_type, nu_epsilon, step_model, step_times, step_sizes,
                              op_mode='galv', normalize=False, tau_c=None, device=0):
    """phasance.construct_phasor_v_matrix (phasance.py:121-144): (rm, rm_layered)."""
    _check(nu_basis_type, normalize)
    if op_mode != 'galv' or step_model != 'ideal':
        raise ValueError("Phasance response is only supported for op_mode='galv' and step_model='ideal'. "
                         f"Received op_mode {op_mode}, step_model {step_model}")
    return _ffi.get_context(device).phasor_v_matrix(times, basis_nu, nu_epsilon, step_times, step_sizes)


def phasor_scale_vector(nu, basis_tau, quantiles=(0.25, 0.75)):
    """phasance.phasor_scale_vector (phasance.py:165-184): O(n_nu) host arithmetic."""
    nu = np.asarray(nu, dtype=float)
    lt = np.log(basis_tau)
    lt_min, lt_range = np.min(lt), np.max(lt) - np.min(lt)
    tau_q1, tau_q3 = np.exp(lt_min + quantiles[0] * lt_range), np.exp(lt_min + quantiles[1] * lt_range)
    out = np.empty(len(nu))
    out[nu <= 0] = tau_q3 ** nu[nu <= 0]
    out[nu > 0] = tau_q1 ** nu[nu > 0]
    return out
