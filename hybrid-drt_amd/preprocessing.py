"""Host-side grid / scale / step-detection rules of hybdrt/preprocessing.py that sit on the fit paths (O(samples)
numpy bookkeeping that decides grids and scales; the matrices and the optimisation run on the device)."""
import warnings

import numpy as np


# ---- chrono signals: step detection (preprocessing.py:17-158), ideal step model ---------------------------------
def identify_steps(y, allow_consecutive=True, rthresh=50, athresh=1e-10):
    """preprocessing.identify_steps (17-37): indices where |diff| exceeds rthresh x its median (and athresh)."""
    dy = np.abs(np.diff(y))
    step_idx = np.where((dy >= np.median(dy) * rthresh) & (dy >= athresh))[0] + 1
    if not allow_consecutive:
        gap = np.concatenate(([2], np.diff(step_idx)))
        step_idx = step_idx[gap > 1]
    return step_idx


def get_step_indices_from_step_times(times, step_times):
    """preprocessing.py:161-178: first sample at or after each step time."""
    times = np.asarray(times)
    out = []
    for st in step_times:
        delta = np.where(times >= st, times - st, np.inf)
        out.append(int(np.argmin(delta)))
    return np.array(out, dtype=int)


def get_step_sizes(times, y, step_times, step_index=None):
    """preprocessing.get_step_sizes (109-133): mean level after minus mean level before every step."""
    if step_index is None:
        step_index = get_step_indices_from_step_times(times, step_times)
    n_steps = len(step_times)
    sizes = np.zeros(n_steps)
    for k in range(n_steps):
        end = len(y) if k == n_steps - 1 else step_index[k + 1]
        start = 0 if k == 0 else step_index[k - 1]
        sizes[k] = np.mean(y[step_index[k]:end]) - np.mean(y[start:step_index[k]])
    return sizes


def get_step_info(times, y, allow_consecutive=True, offset_step_times=False, offset_size=None, rthresh=50,
                  athresh=1e-10):
    """preprocessing.get_step_info (59-106): the observed step is assumed to have happened one (minimum) sample period
    earlier, less a hair so that it never coincides with the previous sample."""
    step_idx = identify_steps(y, allow_consecutive, rthresh, athresh)
    step_times = np.array(times)[step_idx].copy()
    if offset_step_times:
        if offset_size is None:
            offset_size = -np.min(np.diff(times)) * (1 - 1e-8)
        step_times += offset_size
    return step_times, get_step_sizes(times, y, step_times, step_index=step_idx)


def process_input_signal(times, input_signal, step_model, offset_steps, offset_size=None, rthresh=50):
    """preprocessing.process_input_signal (136-158) for step_model='ideal'."""
    if step_model != 'ideal':
        # (the matrix level of the expdecay model IS built -- mat1d.construct_response_matrix / construct_inductance_response_vector;
        # a FIT with it does not exist upstream either: _prep_chrono_fit_matrix calls construct_capacitance_response_vector
        # unconditionally, drt1d.py:5581, which raises 'Capacitance response not implemented for non-ideal steps', mat1d.py:440)
        raise NotImplementedError("only the ideal step model can be fitted (upstream's own expdecay fit stops in "
                                  "construct_capacitance_response_vector)")
    step_times, step_sizes = get_step_info(times, input_signal, True, offset_steps, offset_size, rthresh)
    return step_times, step_sizes, None


def generate_model_signal(times, step_times, step_sizes, tau_rise=None, step_model='ideal'):
    """preprocessing.generate_model_signal (181-207), ideal steps: sum of step_size * unit_step(t - step_time)."""
    if step_model != 'ideal':
        raise NotImplementedError("only the ideal step model is built")
    signal = np.zeros(len(times))
    for st, sa in zip(step_times, step_sizes):
        signal += sa * (np.asarray(times) >= st)
    return signal


def get_time_since_step(times, step_times, prestep_value=None):
    """preprocessing.get_time_since_step (918-950): time since the last step, floored at the sample period; samples
    before the first step are dropped unless prestep_value is given."""
    times = np.asarray(times)
    t_sample = np.min(np.diff(times)) if len(times) > 1 else times[0]
    parts = []
    if prestep_value is not None:
        parts.append(np.tile(prestep_value, int(np.sum(times < step_times[0]))))
    for i, start in enumerate(step_times):
        end = np.inf if i == len(step_times) - 1 else step_times[i + 1]
        sel = (times >= start) & (times < end)
        if np.any(sel):
            parts.append(np.maximum(times[sel] - start, t_sample))
    return np.concatenate(parts)


def get_tau_lim(frequencies, times=None, step_times=None):
    """preprocessing.get_tau_lim (953-972)."""
    tau_min, tau_max = np.inf, -np.inf
    if frequencies is not None:
        tau_min, tau_max = 1 / (2 * np.pi * np.max(frequencies)), 1 / (2 * np.pi * np.min(frequencies))
    if times is not None:
        deltas = get_time_since_step(times, step_times)
        tau_min, tau_max = min(tau_min, np.min(deltas)), max(tau_max, np.max(deltas))
    return tau_min, tau_max


def get_num_decades(frequencies, times=None, step_times=None):
    tau_min, tau_max = get_tau_lim(frequencies, times, step_times)
    return np.log10(tau_max) - np.log10(tau_min)


def get_basis_tau(frequencies, times=None, step_times=None, ppd=10, extend_decades=1, tau_grid=None):
    """preprocessing.py:982-1013: 10 points per decade, +-extend_decades beyond the measurement range,
    or the matching slice of a tau supergrid."""
    tau_min, tau_max = get_tau_lim(frequencies, times, step_times)
    log_tau_min = np.log10(tau_min) - extend_decades
    log_tau_max = np.log10(tau_max) + extend_decades
    if tau_grid is not None:
        tau_grid = np.asarray(tau_grid)
        if 10 ** log_tau_min < np.min(tau_grid):
            left = 0
        else:
            left = _nearest_index(tau_grid, 10 ** log_tau_min, -1)
        if 10 ** log_tau_max > np.max(tau_grid):
            right = len(tau_grid)
        else:
            right = _nearest_index(tau_grid, 10 ** log_tau_max, 1) + 1
        return tau_grid[left:right]
    exact = (log_tau_max - log_tau_min) * ppd + 1
    num = int(np.ceil(exact))
    pad = 0.5 * (num - exact) / ppd
    return np.logspace(log_tau_min - pad, log_tau_max + pad, num)


def _nearest_index(x_array, x_val, constraint):
    """hybdrt/utils/array.py nearest_index with constraint -1 (<= x_val) / +1 (>= x_val)."""
    if constraint == -1:
        mask = x_array <= x_val
    else:
        mask = x_array >= x_val
    idx = np.where(mask)[0]
    return int(idx[np.argmin(np.abs(x_array[mask] - x_val))])


def get_epsilon_from_ppd(ppd, factor=1):
    """preprocessing.py:1016-1017."""
    return factor / np.log(10 ** (1 / ppd))


def estimate_rp(times, step_times, input_step_sizes, response_signal, step_model, z):
    """preprocessing.estimate_rp (764-841): span between the smallest and largest apparent resistance seen by either
    data set (per-step (v - v_before) / step size for chrono data, the real part for EIS)."""
    r_min, r_max = np.inf, 0.0
    if times is not None:
        times = np.asarray(times)
        step_times = np.asarray(step_times, dtype=float)
        input_step_sizes = np.asarray(input_step_sizes, dtype=float)
        if step_model == 'ideal':
            # consecutive "steps" less than 20 us apart are one real step (finite rise time)
            first = np.concatenate(([0], np.where(np.diff(step_times) > 2e-5)[0] + 1))
            if len(first) < len(step_times):
                bounds = list(first) + [len(input_step_sizes)]
                input_step_sizes = np.array([np.sum(input_step_sizes[a:b]) for a, b in zip(bounds[:-1], bounds[1:])])
                step_times = step_times[first]
        step_index = [int(np.argmin(np.where(times >= st, times - st, np.inf))) for st in step_times]
        lo = np.full(len(step_index), np.nan)
        hi = np.full(len(step_index), np.nan)
        for i, a in enumerate(step_index):
            b = len(times) if i == len(step_index) - 1 else step_index[i + 1]
            if a == b:
                continue          # truncated step
            r = (response_signal[a:b] - response_signal[a - 1]) / input_step_sizes[i]
            lo[i], hi[i] = np.min(r), np.max(r)
        r_min, r_max = np.nanmean(lo), np.nanpercentile(hi, 99)
    if z is not None:
        r_min, r_max = min(r_min, np.min(z.real)), max(r_max, np.max(z.real))
    return r_max - r_min


def get_quantile_limits(y, qr_size=0.5, qr_thresh=1.5):
    """preprocessing.get_quantile_limits (844-851): the central quantile range stretched by qr_thresh on both sides."""
    q_lo = np.percentile(y, 50 - 100 * qr_size / 2)
    q_hi = np.percentile(y, 50 + 100 * qr_size / 2)
    return q_lo - (q_hi - q_lo) * qr_thresh, q_hi + (q_hi - q_lo) * qr_thresh


def identify_extreme_values(y, qr_size=0.5, qr_thresh=1.5):
    """preprocessing.identify_extreme_values (854-857)."""
    y_min, y_max = get_quantile_limits(y, qr_size, qr_thresh)
    return (y < y_min) | (y > y_max)


def discard_first_n_chrono(times, i_signal, v_signal, n, op_mode='galv', step_indices=None):
    """preprocessing.discard_first_n_chrono (471-504): drop the first n samples of every segment (the pre-step segment
    included) -- for instruments whose first points after a step are perturbed.  Returns (kept indices, (t, i, v))."""
    if step_indices is None:
        step_indices = identify_steps(i_signal if op_mode == 'galv' else v_signal, False)
    bounds = np.concatenate(([0], step_indices, [len(times)]))
    keep = np.concatenate([np.arange(a + n, b) for a, b in zip(bounds[:-1], bounds[1:])])
    return keep, (np.asarray(times)[keep], np.asarray(i_signal)[keep], np.asarray(v_signal)[keep])


# ---- down-sampling with anti-aliasing (preprocessing.py:335-470, 507-589) ---------------------------------------------
def nearest_index(x_array, x_val):
    """utils.array.nearest_index without a constraint"""
    return int(np.argmin(np.abs(np.asarray(x_array) - x_val)))


def sigma_from_decimate_index(y, decimate_index, truncate=4.0):
    """preprocessing.sigma_from_decimate_index (575-589): filter width that reaches halfway to the nearest kept sample
    at `truncate` standard deviations; kept samples with kept neighbours are not filtered."""
    sigmas = np.zeros(len(y))
    diff = np.diff(decimate_index)
    min_diff = np.minimum(np.insert(diff, 0, diff[0]), np.append(diff, diff[-1]))
    sigma_dec = min_diff / (2 * truncate)
    sigma_dec[min_diff < 2] = 0
    sigmas[decimate_index] = sigma_dec
    return sigmas


def filter_chrono_signal(times, y, step_index=None, input_signal=None, decimate_index=None, sigma_factor=0.01,
                         max_sigma=None, device=0):
    """preprocessing.filter_chrono_signal (507-572; no outlier removal, no median pre-filter): per step segment a Gaussian
    filter whose width grows with the time since the step (sigma ~ e (t - t_step) / 2, in samples, times sigma_factor),
    capped by max_sigma and by the decimation widths; the correlations run on the device."""
    from . import filters
    if step_index is None and input_signal is None:
        raise ValueError('Either step_index or input_signal must be provided')
    if step_index is None:
        step_index = identify_steps(input_signal, allow_consecutive=False)
    times = np.asarray(times, dtype=float)
    step_index = np.array(step_index)
    bounds = step_index
    if bounds[0] > 0:
        bounds = np.insert(bounds, 0, 0)
    if bounds[-1] < len(y):
        bounds = np.append(bounds, len(y))
    t_sample = np.median(np.diff(times))
    if max_sigma is None:
        max_sigma = sigma_factor / t_sample
    dec = sigma_from_decimate_index(y, decimate_index) if decimate_index is not None else None
    sigmas = np.empty(len(y))
    for a, b in zip(bounds[:-1], bounds[1:]):
        t_step = times[a:b]
        sg = sigma_factor * ((np.exp(1) * (t_step - (t_step[0] - t_sample)) / 2) / t_sample)
        sg[sg > max_sigma] = max_sigma
        if dec is not None:
            sg = np.minimum(dec[a:b], sg)
        sigmas[a:b] = sg
    return filters.nonuniform_gaussian_filter1d_segments(y, sigmas, bounds, device=device)


def get_decimation_index(times, step_times, t_sample, prestep_points, decimation_interval, decimation_factor,
                         max_t_sample):
    """preprocessing.get_decimation_index (620-689): indices kept by progressive decimation.  ``prestep_points`` evenly
    spaced samples of the pre-step record; after every step the first ``decimation_interval`` + 1 samples at full rate,
    then stretches of ``decimation_interval`` samples at strides ``int(decimation_factor ** j)``, j = 1, 2, ... (capped at
    ``int(max_t_sample / t_sample)``; the capped stretch runs to the end of the step), always ending on the last sample
    before the next step."""
    times = np.asarray(times, dtype=float)
    n = len(times)
    n_pre = int(np.count_nonzero(times < np.min(step_times)))
    keep = [np.linspace(0, n_pre - 1, prestep_points).round(0).astype(int)]
    # first sample at or after each step time (the reference's argmin over non-negative delays)
    starts = [int(np.flatnonzero(times >= st)[np.argmin(times[times >= st])]) for st in step_times]
    stride_cap = np.inf if max_t_sample is None else int(max_t_sample / t_sample)
    for k, start in enumerate(starts):
        stop = n if start == starts[-1] else starts[k + 1]
        head = np.arange(start, min(start + decimation_interval + 1, stop), dtype=int)
        keep.append(head)
        last, j = head[-1], 1
        while last < stop - 1:
            stride = min(int(decimation_factor ** j), stride_cap)
            end = stop if stride == stride_cap else min(last + decimation_interval * stride + 1, stop)
            part = np.arange(last + stride, end, stride, dtype=int)
            if len(part) == 0:
                part = np.array([end - 1])
            if end == stop and part[-1] < stop - 1:
                part = np.append(part, stop - 1)
            keep.append(part)
            last = part[-1]
            j += 1
    return np.unique(np.concatenate(keep))


def select_decimation_interval(times, step_times, t_sample, prestep_points, decimation_factor, max_t_sample, target_size):
    """preprocessing.select_decimation_interval (603-617): interval whose decimated size interpolates to ``target_size``
    over twelve trial intervals between 2 and 1000."""
    intervals = np.logspace(np.log10(2), np.log10(1000), 12).astype(int)
    sizes = [len(get_decimation_index(times, step_times, t_sample, prestep_points, iv, decimation_factor, max_t_sample))
             for iv in intervals]
    if target_size > sizes[-1]:
        warnings.warn(f'Cannot achieve target size of {target_size} with selected decimation factor of '
                      f'{decimation_factor}. Decrease the decimation factor and/or decrease the maximum period')
    if target_size < sizes[0]:
        warnings.warn(f'Cannot achieve target size of {target_size} with selected decimation factor of '
                      f'{decimation_factor}. Increase the decimation factor and/or increase the maximum period')
    return int(np.interp(target_size, sizes, intervals))


def downsample_data(times, i_signal, v_signal, target_times=None, target_size=None, stepwise_sample_times=True,
                    step_times=None, step_model=None, method='match', decimation_interval=10, decimation_factor=2,
                    decimation_max_period=None, antialiased=True, filter_kw=None, discard_first_n_points=None,
                    discard_only=False, op_mode='galv', prestep_samples=20, device=0):
    """preprocessing.downsample_data (335-470).  method='match': keep the samples closest to `target_times` after every
    step (all post-step samples when None) and the whole pre-step record; method='decimate': progressive decimation
    (:func:`get_decimation_index`; `target_size` picks the interval).  Either way after an anti-aliasing filter matched to
    the local decimation (device kernel).  `discard_first_n_points` then drops that many samples after every step of the
    down-sampled record (and at its start); `discard_only` skips the down-sampling.
    Returns (sample_times, sample_i, sample_v, sample_index)."""
    if method not in ('match', 'decimate'):
        raise ValueError(f"Invalid downsample method {method}. Options: 'match', 'decimate'")
    times, i_signal, v_signal = (np.asarray(a, dtype=float) for a in (times, i_signal, v_signal))
    if discard_only:
        sample_index = np.arange(len(times))
    else:
        if not stepwise_sample_times:       # the whole record as one step starting at t = 0
            step_times, step_indices = [0], [0]
        elif step_times is None:
            if step_model not in ('ideal', 'expdecay'):
                raise ValueError(f"Invalid step_model {step_model}. Options: ['ideal', 'expdecay']")
            step_indices = identify_steps(i_signal if op_mode == 'galv' else v_signal, step_model == 'ideal')
            step_times = times[step_indices]
        else:
            step_indices = get_step_indices_from_step_times(times, step_times)
        if method == 'match':
            if target_times is not None:
                target = np.unique(np.concatenate([np.asarray(target_times) + ts for ts in step_times]))
                sample_index = np.unique(np.array([nearest_index(times, tt) for tt in target]))
            else:
                sample_index = np.arange(step_indices[0], len(times), dtype=int)
            if step_indices[0] > 0 and prestep_samples > 0:
                sample_index = np.unique(np.concatenate((np.arange(0, step_indices[0], dtype=int), sample_index)))
        else:
            t_sample = np.min(np.diff(times))
            if target_size is not None:
                decimation_interval = select_decimation_interval(times, step_times, t_sample, prestep_samples,
                                                                 decimation_factor, decimation_max_period, target_size)
            sample_index = get_decimation_index(times, step_times, t_sample, prestep_samples, decimation_interval,
                                                decimation_factor, decimation_max_period)
        if antialiased and stepwise_sample_times:
            fkw = filter_kw or {}
            step_index = identify_steps(i_signal if op_mode == 'galv' else v_signal, allow_consecutive=False)
            i_signal = filter_chrono_signal(times, i_signal, step_index=step_index, decimate_index=sample_index, device=device, **fkw)
            v_signal = filter_chrono_signal(times, v_signal, step_index=step_index, decimate_index=sample_index, device=device, **fkw)
    sample_times, sample_i, sample_v = times[sample_index], i_signal[sample_index], v_signal[sample_index]
    if discard_first_n_points is not None:
        starts = np.insert(identify_steps(sample_i if op_mode == 'galv' else sample_v, False), 0, 0)
        stops = [len(sample_times) if a == starts[-1] else starts[k + 1] for k, a in enumerate(starts)]
        sel = np.concatenate([np.arange(a + discard_first_n_points, b) for a, b in zip(starts, stops)])
        sample_times, sample_i, sample_v, sample_index = sample_times[sel], sample_i[sel], sample_v[sel], sample_index[sel]
    return sample_times, sample_i, sample_v, sample_index
