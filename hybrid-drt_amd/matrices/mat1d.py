"""Drop-in for the hot-path builders of hybdrt/matrices/mat1d.py (same names, argument meaning and error
behaviour); the arithmetic runs in libhipdrt.so."""
import numpy as np

from .. import _ffi
from ..utils.array import is_log_uniform, is_uniform, rel_round


def impedance_matrix_is_toeplitz(frequencies, tau, frequency_precision=10):
    """The reference's Toeplitz decision (hybdrt/matrices/mat1d.py:229-295): f log-uniform and tau equal to
    1/omega, or one a contiguous run of the other with tau log-uniform.  Pure host logic."""
    omega = np.asarray(frequencies) * 2 * np.pi
    tau = np.asarray(tau)
    rnd = lambda v: rel_round(v, frequency_precision)
    tau_eq_omega = len(tau) == len(omega) and np.array_equal(rnd(tau), rnd(1 / omega))
    subset = False
    match = rnd(1 / omega[0]) == rnd(tau)
    if np.sum(match) == 1:
        start = int(np.where(match)[0][0])
        seg = tau[start:start + len(omega)]
        subset = len(seg) == len(omega) and np.array_equal(rnd(seg), rnd(1 / omega))
    elif np.sum(match) > 1:
        raise Exception('Repeated tau values')
    if not subset:
        match = rnd(1 / omega) == rnd(tau[0])
        if np.sum(match) == 1:
            start = int(np.where(match)[0][0])
            seg = omega[start:start + len(tau)]
            subset = len(seg) == len(tau) and np.array_equal(rnd(seg), rnd(1 / tau))
        elif np.sum(match) > 1:
            raise Exception('Repeated omega values')
    if is_log_uniform(frequencies):
        return bool(tau_eq_omega or (subset and is_log_uniform(tau)))
    return False


def construct_impedance_matrices(frequencies, tau, epsilon, integrate_method='trapz', integrate_points=1000,
                                 interpolate_grids=None, frequency_precision=10, device=0):
    """Both parts in one device pass.  ``interpolate_grids`` = ((log_wt_re, z_re), (log_wt_im, z_im)).
    ``frequencies`` may be (B, nf) for a batched build (then never Toeplitz)."""
    frequencies = np.asarray(frequencies, dtype=float)
    tau = np.asarray(tau, dtype=float)
    if integrate_method == 'interp':
        if interpolate_grids is None:
            raise ValueError("interpolate_grids must be provided to use integrate_method 'interp'")
        mode = _ffi.MODE_INTERP
    elif integrate_method == 'trapz':
        mode = _ffi.MODE_TRAPZ
    else:
        raise NotImplementedError("integrate_method 'quad' is not on the hot path")
    tpl = frequencies.ndim == 1 and impedance_matrix_is_toeplitz(frequencies, tau, frequency_precision)
    return _ffi.get_context(device).impedance_matrix(frequencies, tau, epsilon, mode=mode, toeplitz=tpl,
                                                     lookups=interpolate_grids, ny=integrate_points)


def construct_impedance_matrix(frequencies, part, tau=None, basis_type='gaussian', epsilon=1, frequency_precision=10,
                               integrate_method='trapz', integrate_points=1000, zga_params=None,
                               interpolate_grids=None, device=0):
    """mat1d.construct_impedance_matrix (hybdrt/matrices/mat1d.py:212-374).  ``interpolate_grids`` is the
    (log_wt_grid, z_grid) pair of the requested part, as in the reference."""
    if basis_type != 'gaussian':
        raise NotImplementedError("only the default gaussian basis is on the hot path")
    if part not in ('real', 'imag'):
        raise ValueError(f'Invalid part {part}. Options: real, imag')
    frequencies = np.asarray(frequencies, dtype=float)
    if tau is None:
        tau = 1 / (frequencies * 2 * np.pi)
    grids = None
    if integrate_method == 'interp':
        if interpolate_grids is None:
            raise ValueError("interpolate_grids must be provided to use integrate_method 'interp'")
        # the device kernel always produces both parts; feed the given table on both sides and keep one
        grids = (interpolate_grids, interpolate_grids)
    a_re, a_im = construct_impedance_matrices(frequencies, tau, epsilon, integrate_method, integrate_points, grids,
                                              frequency_precision, device)
    if integrate_method == 'interp':
        return a_re      # both outputs interpolate the table that was passed in
    return a_re if part == 'real' else a_im


def construct_response_matrix(basis_tau, times, step_model, step_times, step_sizes, basis_type='gaussian',
                              epsilon=0.975, tau_rise=None, op_mode='galv', integrate_method='trapz',
                              integrate_points=1000, zga_params=None, interpolate_grids=None, device=0):
    """mat1d.construct_response_matrix (hybdrt/matrices/mat1d.py:16-122): returns (A, A_layered) with
    ``A @ x`` the time response to the current steps; ``A_layered[k]`` is the contribution of step k.
    Gaussian basis; galvanostatic with ideal steps ('interp' needs ``interpolate_grids`` = (log_td_grid, response_grid) from
    ``basis.generate_response_lookup``; 'trapz') or with exponentially rising steps (``step_model='expdecay'``, ``tau_rise`` per step,
    'trapz'), and the potentiostatic delta-function response (``op_mode='pot'``).  'quad' is not built."""
    if step_model not in ('ideal', 'expdecay'):
        raise ValueError(f'Invalid step_model {step_model}. Options: ideal, expdecay')
    if op_mode not in ('galv', 'pot'):
        raise ValueError(f'Invalid op_mode {op_mode}. Options: galv, pot')
    if basis_type != 'gaussian':
        raise NotImplementedError("only the gaussian basis is built")
    step_times = np.asarray(step_times, dtype=float)
    step_sizes = np.asarray(step_sizes, dtype=float)
    times = np.asarray(times, dtype=float)
    basis_tau = np.asarray(basis_tau, dtype=float)
    if step_times.size == 0:
        return np.zeros((times.size, basis_tau.size)), np.zeros((0, times.size, basis_tau.size))
    if op_mode == 'pot':
        # mat1d.py:114-118: the basis is a delta function whatever basis_type / step_model / integrate_method say
        return _ffi.get_context(device).response_matrix_variant(times, basis_tau, step_times, step_sizes, _ffi.RESPONSE_POT)
    if integrate_method == 'interp':
        # (the lookup holds the ideal step's response; the reference interpolates it for either step model, mat1d.py:108-112)
        if interpolate_grids is None:
            raise ValueError("interpolate_grids must be provided for integrate_method 'interp'")
        mode = _ffi.MODE_INTERP
    elif integrate_method == 'trapz':
        mode = _ffi.MODE_TRAPZ
    else:
        raise NotImplementedError("integrate_method 'quad' is not built (scipy.integrate.quad on the host in the reference)")
    if step_model == 'expdecay' and mode == _ffi.MODE_TRAPZ:
        if tau_rise is None:            # mat1d.py:44-45: zeros -- the integrand's tau_rise / (tau_rise - T) factor is 0 / (-T) then
            tau_rise = np.zeros(step_times.size)
        return _ffi.get_context(device).response_matrix_variant(times, basis_tau, step_times, step_sizes, _ffi.RESPONSE_EXPDECAY,
                                                                tau_rise=tau_rise, epsilon=epsilon, ny=integrate_points)
    return _ffi.get_context(device).response_matrix(times, basis_tau, step_times, step_sizes, epsilon, mode=mode,
                                                    lookup=interpolate_grids, ny=integrate_points, layered=True)


def construct_integrated_derivative_matrix(basis_grid, basis_type='gaussian', order=1, epsilon=1, zga_params=None,
                                           integration_limits=None, device=0):
    """mat1d.construct_integrated_derivative_matrix (hybdrt/matrices/mat1d.py:125-209), orders 0-2."""
    if basis_type != 'gaussian' or integration_limits is not None:
        raise NotImplementedError("only the gaussian basis with infinite limits is on the hot path")
    if order not in (0, 1, 2):
        raise ValueError(f'Invalid order {order}. Order must be between 0 and 2')
    basis_grid = np.asarray(basis_grid, dtype=float)
    mats = _ffi.get_context(device).penalty_matrices(basis_grid, epsilon, is_uniform(basis_grid))
    return mats[order]


def construct_ohmic_response_vector(times, step_model, step_times, step_sizes, tau_rise, input_signal, smooth,
                                    op_mode='galv'):
    """mat1d.construct_ohmic_response_vector (hybdrt/matrices/mat1d.py:398-420): the response of R_inf is the input
    signal itself -- the ideal steps when ``smooth``, else the measured signal minus its pre-step mean."""
    from .. import preprocessing as pp
    if op_mode != 'galv':
        raise ValueError('Ohmic response vector not implemented for potentiostatic mode')
    times = np.asarray(times, dtype=float)
    if smooth:
        return pp.generate_model_signal(times, step_times, step_sizes, tau_rise, step_model)
    input_signal = np.asarray(input_signal, dtype=float)
    return input_signal - np.mean(input_signal[times < step_times[0]])


def construct_inductance_response_vector(times, step_model, step_times, step_sizes, tau_rise, op_mode='galv'):
    """mat1d.construct_inductance_response_vector (mat1d.py:377-395): zero for ideal steps; an exponentially rising current
    induces (step / tau_rise) exp(-(t - t_step) / tau_rise) after every step."""
    if step_model not in ('ideal', 'expdecay'):
        raise ValueError(f'Invalid step_model {step_model}. Options: ideal, expdecay')
    times = np.asarray(times, dtype=float)
    irv = np.zeros(len(times))
    if step_model == 'expdecay':
        if op_mode != 'galv':
            raise ValueError('Inductance response vector not implemented for potentiostatic mode')
        for st, sa, tr in zip(step_times, step_sizes, tau_rise):
            after = times >= st
            irv[after] += (sa / tr) * np.exp(-(times[after] - st) / tr)
    return irv


def construct_capacitance_response_vector(times, step_model, step_times, step_sizes, tau_rise, op_mode='galv'):
    """mat1d.construct_capacitance_response_vector (mat1d.py:423-443): a series capacitance integrates the current --
    step_size * (t - t_step) after every ideal step."""
    if step_model != 'ideal':
        raise ValueError('Capacitance response not implemented for non-ideal steps')
    if op_mode != 'galv':
        raise ValueError('Capacitance response vector not implemented for potentiostatic mode')
    times = np.asarray(times, dtype=float)
    crv = np.zeros(len(times))
    for st, sa in zip(step_times, step_sizes):
        after = times >= st
        crv[after] += sa * (times[after] - st)
    return crv


def construct_inductance_impedance_vector(frequencies):
    """mat1d.py:446-447."""
    return 1j * 2 * np.pi * frequencies


def construct_capacitance_impedance_vector(frequencies):
    """mat1d.py:450-451."""
    return 1 / (1j * 2 * np.pi * frequencies)


def get_step_indices_from_step_times(times, step_times):
    """preprocessing.get_step_indices_from_step_times (hybdrt/preprocessing.py:161-178): index of the first sample at
    or after each step time."""
    times = np.asarray(times, dtype=float)
    return np.array([int(np.argmin(np.where(times < st, np.inf, times - st))) for st in step_times], dtype=int)


def chrono_time_transform(times, step_times):
    """utils.chrono.get_time_transforms(...)[1] (hybdrt/utils/chrono.py:5-44): linear before the first step,
    log(time since step) within each step segment, segments laid end to end (O(nt) host work feeding the kernel)."""
    times = np.atleast_1d(np.asarray(times, dtype=float))
    t_sample = np.min(np.diff(times))
    start_times = np.array(step_times, dtype=float)
    trans_base = np.log(t_sample / 4)
    trans_offsets = np.concatenate([[0], np.cumsum(np.log(start_times[1:] - start_times[:-1]) - trans_base)])
    tt = np.zeros_like(times)
    before = times < start_times[0]
    tt[before] = times[before] - start_times[0]
    for i, start_time in enumerate(start_times):
        end_time = np.inf if i == len(start_times) - 1 else start_times[i + 1]
        idx = np.where((times >= start_time) & (times < end_time))
        if len(idx[0]) > 0:
            tt[idx] = trans_offsets[i] + np.log(np.maximum(times[idx] - start_time, t_sample / 2)) - trans_base
    return tt


def construct_chrono_var_matrix(times, step_times, vmm_epsilon, error_structure=None, device=0):
    """mat1d.construct_chrono_var_matrix (hybdrt/matrices/mat1d.py:457-490): (nt, nt) variance-estimation weights,
    built on the device from the transformed times and the step segments."""
    times = np.asarray(times, dtype=float)
    if error_structure is None:
        tt = chrono_time_transform(times, step_times)
        seg = np.concatenate(([0], get_step_indices_from_step_times(times, step_times), [len(times)]))
        return _ffi.get_context(device).chrono_var_matrix(tt, seg, vmm_epsilon, uniform=False)
    if error_structure == 'uniform':
        return _ffi.get_context(device).chrono_var_matrix(np.zeros(len(times)), [0, len(times)], vmm_epsilon, uniform=True)
    raise ValueError(f'Invalid error_structure {error_structure}. Options: None, uniform')


def construct_eis_var_matrix(frequencies, vmm_epsilon, reim_cor, error_structure, device=0):
    """mat1d.construct_eis_var_matrix (hybdrt/matrices/mat1d.py:493-515)."""
    if error_structure not in (None, 'uniform'):
        raise ValueError(f'Invalid error_structure {error_structure}')
    return _ffi.get_context(device).eis_var_matrix(frequencies, vmm_epsilon, reim_cor, error_structure == 'uniform')
