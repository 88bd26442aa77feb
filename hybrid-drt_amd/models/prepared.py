"""Host orchestration of the fits that go through a prepared-matrix plan (config-5 family): ``DRT.fit_hybrid`` and any
fit with the distribution of phasances (``fit_dop=True``).

Mirrors the data-type agnostic part of hybdrt.models.DRT: ``_prep_for_fit`` (drt1d.py:5439-5558),
``process_chrono_signals`` / ``scale_data`` (drtbase.py:285-514), ``_format_qp_matrices`` (drt1d.py:5736-5963), the
set-up half of ``_qphb_fit_core`` (drt1d.py:365-636) and ``extract_qphb_parameters`` (6228-6289).  Every matrix is
built by a device kernel (response / impedance / phasance / penalty / variance matrices); this module only lays the
blocks side by side, applies the O(m) scalings the reference applies and hands them to ``_ffi.PreparedPlan``, which
runs initialize_weights and the whole iterate_qphb loop on the GPU."""
import warnings

import numpy as np

from .. import _ffi, preprocessing as pp
from ..matrices import mat1d, phasance
from ..utils.array import is_uniform
from . import background, qphb

_CHRONO_KW_DEFAULTS = dict(  # _qphb_fit_core chrono / hybrid keyword defaults (drt1d.py:102-129)
    step_times=None, step_sizes=None, offset_steps=True, step_offset_size=None, offset_baseline=True,
    smooth_inf_response=True, v_baseline_penalty=1e-6, vz_offset=True, vz_offset_scale=1, vz_offset_eps=1,
    chrono_error_structure='uniform', chrono_vmm_epsilon=4, solve_rp=False, v_baseline_deg=0, v_baseline_sqrt=False,
    eis_weight_factor=None, chrono_weight_factor=None, hybrid_weight_factor_method=None, remove_outliers=False,
    outlier_thresh=0.75, remove_extremes=False, extreme_kw=None, neg_allowed_tau_range=None,
    init_weights_separately=False, series_neg=False, discard_first_n=None, downsample=False, downsample_kw=None)

_UNSUPPORTED = dict(subtract_background=False,
                    peak_locations=None)


def combine_status(so_far, step):
    """per-spectrum status of a chain of fits (a full fit and its warm restarts): a failure (< 0: QP breakdown, singular KKT
    system) in ANY step stays -- its iterate went into every later step --, otherwise the last step's verdict (0 converged,
    1 stopped at max_iter)"""
    so_far, step = np.asarray(so_far), np.asarray(step)
    return np.where(step < 0, step, np.where(so_far < 0, so_far, step))


class PreparedFitMixin:
    """Methods of DRT for chrono + EIS (+ DOP) fits."""

    basis_nu = None
    nu_epsilon = None
    nu_basis_type = 'gaussian'
    normalize_dop = True
    step_model = 'ideal'
    chrono_mode = 'galv'

    # ---- drt1d.py:365-408 ---------------------------------------------------------------------------------------
    def _general_special_params(self, has_chrono, has_eis, vz_offset, n_baseline=1):
        sp = {}

        def add(name, nonneg, size=1):
            sp[name] = {'index': int(sum(v['size'] for v in sp.values())), 'nonneg': nonneg, 'size': size}
        if has_chrono:
            add('v_baseline', False, n_baseline)
        if vz_offset and has_chrono and has_eis:
            add('vz_offset', False)
        if self.fit_ohmic:
            add('R_inf', True)
        if self.fit_inductance:
            add('inductance', True)
        if self.fit_capacitance:
            add('C_inv', True)
        if self.fit_dop:
            if self.basis_nu is None:
                self.basis_nu = np.concatenate([np.linspace(-1, -0.4, 25), np.linspace(0.4, 1, 25)])
            if self.nu_epsilon is None:
                self.nu_epsilon = 1 / np.median(np.diff(np.sort(self.basis_nu)))
            add('x_dop', True, len(self.basis_nu))
        return sp

    def _lookups(self, ctx):
        """drtbase.py:138-156: impedance and response lookup tables of the instance's epsilon (device kernels)."""
        key = float(self.tau_epsilon)
        if getattr(self, '_lut_key', None) != key:
            z_re, z_im = ctx.impedance_lookup(key, self._wt_re, self._wt_im)
            td = np.logspace(-6, 2, 2000)
            self._luts = dict(z=((np.log(self._wt_re), z_re), (np.log(self._wt_im), z_im)),
                              response=(np.log(td), ctx.response_lookup(key, td)))
            self._lut_key = key
        return self._luts

    def _memo(self, name, fn, *args):
        """Members of a batch share grids: identical device builds (same inputs, byte for byte) are done once per fit call -- and,
        as upstream keeps its fit matrices while the sampling does not change (the `_recalc_*` flags of drt1d.py:5540-5660), what a
        call used is still there for the NEXT call on this object: a map's loop over observations of one protocol builds its
        penalty / variance / impedance blocks once.  Only what the last call used is kept (one protocol's matrices)."""
        key = (name,) + tuple(a.tobytes() if isinstance(a, np.ndarray) else a for a in args)
        memo = self.__dict__.setdefault('_build_memo', {})
        if key not in memo:
            kept = self.__dict__.get('_build_memo_kept', {})
            memo[key] = kept[key] if key in kept else fn()
        return memo[key]

    def _vz_strength(self, sample_times, frequencies, step_times, vz_offset_eps):
        """DRT._get_vz_strength_vec (drt1d.py:6173-6226): 1 where the two data sets overlap in time scale, Gaussian
        decay in log time scale away from the overlap, 0 before the first step."""
        rbf = lambda y, eps: np.exp(-(eps * y) ** 2)
        deltas = pp.get_time_since_step(sample_times, step_times, prestep_value=-1)
        chrono_tau_min = np.min(deltas[deltas > 0])
        f_inv = 1 / (2 * np.pi * frequencies)
        eis_tau_max = np.max(f_inv)
        cs = np.ones(len(deltas))
        far = deltas >= eis_tau_max
        cs[far] = rbf(np.log(deltas[far] / eis_tau_max), vz_offset_eps)
        cs[deltas == -1] = 0
        es = np.ones(len(frequencies))
        fast = f_inv <= chrono_tau_min
        es[fast] = rbf(np.log(f_inv[fast] / chrono_tau_min), vz_offset_eps)
        return cs, es

    # ---- _prep_for_fit + _format_qp_matrices + the set-up half of _qphb_fit_core --------------------------------------
    def _prepare_measurement(self, ctx, times, i_signal, v_signal, frequencies, z, kw, ckw, hypers):
        has_chrono, has_eis = times is not None, frequencies is not None
        integrate_mode = _ffi.MODE_INTERP if self.integrate_method == 'interp' else _ffi.MODE_TRAPZ
        prep = {}
        # process_chrono_signals (drtbase.py:285-373), no downsampling
        if has_chrono:
            times, i_signal, v_signal = (np.array(a, dtype=float) for a in (times, i_signal, v_signal))
            if not (len(times) == len(i_signal) == len(v_signal)):
                raise ValueError('times, i_signal, and v_signal must have same length')   # validation.check_chrono_data
            step_times, step_sizes = ckw['step_times'], ckw['step_sizes']
            if step_times is None:
                step_times, step_sizes, _ = pp.process_input_signal(times, i_signal, self.step_model,
                                                                    ckw['offset_steps'], ckw['step_offset_size'])
            else:
                step_times = np.asarray(step_times, dtype=float)
                if step_sizes is None:
                    step_sizes = pp.get_step_sizes(times, i_signal, step_times)
            step_sizes = np.asarray(step_sizes, dtype=float)
            if len(step_times) > 1:
                gap = np.diff(step_times) > 1.1 * np.min(np.diff(times))
                nonconsec = np.insert(step_times[1:][gap], 0, step_times[0])
            else:
                nonconsec = step_times
            prep.update(sample_times=times, step_times=step_times, step_sizes=step_sizes, nonconsec_step_times=nonconsec,
                        raw_input_signal=i_signal)
        else:
            step_times = step_sizes = None
        if has_eis:
            frequencies = np.array(frequencies, dtype=float)
            z = np.array(z, dtype=complex)
            if len(frequencies) != len(z):
                raise ValueError('Length of frequencies and z must be equal')

        # basis grid and epsilon (drt1d.py:5471-5487)
        if self.fixed_basis_tau is not None:
            basis_tau = self.fixed_basis_tau
        else:
            basis_tau = pp.get_basis_tau(frequencies, times, step_times, tau_grid=self.tau_supergrid,
                                         extend_decades=self.extend_basis_decades)
        if self.tau_epsilon is None:
            self.tau_epsilon = 1 / np.mean(np.diff(np.log(basis_tau)))
        eps = float(self.tau_epsilon)
        ntau = len(basis_tau)
        if has_chrono and ckw['downsample']:
            # drtbase.py:324-340: anti-aliased down-sampling (device filter) once the steps and the basis grid are known;
            # everything below works on the kept samples
            dkw = ckw['downsample_kw'] if ckw['downsample_kw'] is not None else {'prestep_samples': 10, 'target_times': None}
            times, i_signal, v_signal, sample_index = pp.downsample_data(
                times, i_signal, v_signal, stepwise_sample_times=True, step_times=prep['nonconsec_step_times'],
                op_mode=self.chrono_mode, device=self.device, **dkw)
            prep.update(sample_times=times, sample_index=sample_index, raw_input_signal=i_signal)
        luts = self._lookups(ctx) if integrate_mode == _ffi.MODE_INTERP else dict(z=None, response=None)
        sp = self._general_special_params(has_chrono, has_eis, ckw['vz_offset'],
                                          int(ckw['v_baseline_deg']) + 1 + int(bool(ckw['v_baseline_sqrt'])))
        ns = int(sum(v['size'] for v in sp.values()))
        series_neg = bool(ckw['series_neg'])
        if series_neg and not kw['nonneg']:
            raise ValueError('Only one of series_neg and nonneg may be True')
        ndrt = 2 * ntau if series_neg else ntau         # series_neg: a second, sign-flipped copy of the basis (drt1d.py:5497-5530)
        n = ns + ndrt
        dop = (sp['x_dop']['index'], sp['x_dop']['index'] + sp['x_dop']['size']) if self.fit_dop else None

        # scale_data (drtbase.py:439-514)
        rp_est = pp.estimate_rp(times, step_times, step_sizes, v_signal, self.step_model, z) if kw['scale_data'] else 1.0
        coefficient_scale = rp_est / hypers['rp_scale'] if kw['scale_data'] else 1.0
        input_scale = response_scale = None
        if has_chrono:
            input_scale = np.max(np.abs(step_sizes)) if kw['scale_data'] else 1.0
            response_scale = input_scale * rp_est / hypers['rp_scale'] if kw['scale_data'] else 1.0
            v_scaled = v_signal / response_scale
            response_baseline = np.median(v_scaled[times < step_times[0]])
        impedance_scale = coefficient_scale

        # DOP column scaling (drt1d.py:5767-5788)
        if self.fit_dop:
            dop_scale = phasance.phasor_scale_vector(self.basis_nu,
                                                     self.tau_supergrid if self.tau_supergrid is not None else basis_tau)
            dop_scale = dop_scale / (np.sqrt(np.pi) / self.nu_epsilon)        # basis.get_basis_func_area, gaussian
        else:
            dop_scale = None

        # ---- matrices, each built on the device; blocks laid out as in _format_qp_matrices -------------------
        blocks, rows = [], []
        num_chrono = 0
        if has_chrono:
            num_chrono = len(times)
            rm = np.zeros((num_chrono, n))
            a, _ = ctx.response_matrix(times, basis_tau, step_times, step_sizes, eps, mode=integrate_mode,
                                       lookup=luts['response'], layered=False)
            rm[:, ns:ns + ntau] = a / input_scale
            if series_neg:
                rm[:, ns + ntau:] = -(a / input_scale)
            vb, vb_scale = background.get_baseline_matrix(times, int(ckw['v_baseline_deg']), normalize=True,
                                                          sqrt=bool(ckw['v_baseline_sqrt']))
            rm[:, sp['v_baseline']['index']:sp['v_baseline']['index'] + sp['v_baseline']['size']] = vb
            if 'inductance' in sp:
                rm[:, sp['inductance']['index']] = (mat1d.construct_inductance_response_vector(
                    times, self.step_model, step_times, step_sizes, None) / input_scale) * kw['inductance_scale']
            if 'C_inv' in sp:
                rm[:, sp['C_inv']['index']] = (mat1d.construct_capacitance_response_vector(
                    times, self.step_model, step_times, step_sizes, None) / input_scale) * kw['capacitance_scale']
            if 'R_inf' in sp:
                rm[:, sp['R_inf']['index']] = mat1d.construct_ohmic_response_vector(
                    times, self.step_model, step_times, step_sizes, None, i_signal, ckw['smooth_inf_response']) / input_scale
            if self.fit_dop:
                rm_dop, _ = ctx.phasor_v_matrix(times, self.basis_nu, self.nu_epsilon, step_times, step_sizes)
                rm[:, dop[0]:dop[1]] = (rm_dop / input_scale) * dop_scale
            blocks.append(rm)
            scaled_response_offset = -response_baseline if ckw['offset_baseline'] else 0.0
            rows.append(v_scaled + scaled_response_offset)
            prep.update(v_baseline_scale=vb_scale, scaled_response_offset=scaled_response_offset,
                        response_matrix=a, inf_response=rm[:, sp['R_inf']['index']] * input_scale if 'R_inf' in sp else None)
        if has_eis:
            nf = len(frequencies)

            def build_eis_block():
                tpl_a = mat1d.impedance_matrix_is_toeplitz(frequencies, basis_tau, self.frequency_precision)
                a_re, a_im = ctx.impedance_matrix(frequencies, basis_tau, eps, mode=integrate_mode, toeplitz=tpl_a,
                                                  lookups=luts['z'])
                zm = np.zeros((nf, n), dtype=complex)
                if 'inductance' in sp:
                    zm[:, sp['inductance']['index']] = mat1d.construct_inductance_impedance_vector(frequencies) * kw['inductance_scale']
                if 'R_inf' in sp:
                    zm[:, sp['R_inf']['index']] = 1
                if 'C_inv' in sp:
                    zm[:, sp['C_inv']['index']] = mat1d.construct_capacitance_impedance_vector(frequencies) * kw['capacitance_scale']
                if self.fit_dop:
                    zm[:, dop[0]:dop[1]] = ctx.phasor_z_matrix(frequencies, self.basis_nu, self.nu_epsilon) * dop_scale
                zm[:, ns:ns + ntau] = a_re + 1j * a_im
                if series_neg:
                    zm[:, ns + ntau:] = -(a_re + 1j * a_im)
                return np.vstack([zm.real, zm.imag])
            # independent of the measured values: one build per batch
            blocks.append(self._memo('eis_block', build_eis_block, frequencies, basis_tau, eps, integrate_mode, n,
                                     float(kw['inductance_scale']), float(kw['capacitance_scale']), series_neg,
                                     dop_scale if dop_scale is not None else 0, str(sorted(sp.items())),
                                     self.basis_nu if self.fit_dop else 0, float(self.nu_epsilon) if self.fit_dop else 0.0,
                                     int(self.frequency_precision)))
            z_scaled = z / impedance_scale
            rows.append(np.concatenate([z_scaled.real, z_scaled.imag]))
        rzm = blocks[0] if len(blocks) == 1 else np.vstack(blocks)
        rzv = np.concatenate(rows)
        m = len(rzv)

        # penalty matrices (drt1d.py:5673-5734, 5863-5910)
        ln_tau = np.log(basis_tau)
        tpl_m = is_uniform(ln_tau)

        def build_penalties():
            m_drt = self._memo('mdrt', lambda: ctx.penalty_matrices(ln_tau, eps, tpl_m), ln_tau, eps)
            m_dop = self._memo('mdop', lambda: ctx.penalty_matrices(self.basis_nu, self.nu_epsilon, is_uniform(self.basis_nu)),
                               self.basis_nu, float(self.nu_epsilon)) if self.fit_dop else None
            pen = []
            for k in range(3):
                mk = np.zeros((n, n))
                if 'v_baseline' in sp:      # scalar or one penalty per baseline coefficient (drt1d.py:5872-5886)
                    a0, nb = sp['v_baseline']['index'], sp['v_baseline']['size']
                    pens = np.broadcast_to(np.asarray(ckw['v_baseline_penalty'], dtype=float), (nb,)) \
                        if np.ndim(ckw['v_baseline_penalty']) == 0 or len(ckw['v_baseline_penalty']) == nb else None
                    if pens is None:
                        raise ValueError("If v_baseline_penalty is iterable, it must match the number of v_baseline "
                                         f"parameters. Number of v_baseline parameters is {nb}")
                    mk[np.arange(a0, a0 + nb), np.arange(a0, a0 + nb)] = pens
                if 'inductance' in sp:
                    mk[sp['inductance']['index'], sp['inductance']['index']] = kw['inductance_penalty']
                if 'R_inf' in sp:
                    mk[sp['R_inf']['index'], sp['R_inf']['index']] = kw['ohmic_penalty']
                if 'C_inv' in sp:
                    mk[sp['C_inv']['index'], sp['C_inv']['index']] = kw['capacitance_penalty']
                if 'vz_offset' in sp:
                    mk[sp['vz_offset']['index'], sp['vz_offset']['index']] = 1 / ckw['vz_offset_scale']
                if self.fit_dop:
                    mk[dop[0]:dop[1], dop[0]:dop[1]] = m_dop[k]
                mk[ns:, ns:] = np.kron(np.eye(2), m_drt[k]) if series_neg else m_drt[k]
                pen.append(mk)
            return pen
        pen = self._memo('pen', build_penalties, ln_tau, eps, n, str(sorted(sp.items())), series_neg, str(ckw['v_baseline_penalty']),
                         float(kw['inductance_penalty']), float(kw['ohmic_penalty']), float(kw['capacitance_penalty']),
                         float(ckw['vz_offset_scale']))

        # variance-estimation matrix (drt1d.py:614-636)
        def build_vmm():
            v = np.zeros((m, m))
            if has_chrono:
                v[:num_chrono, :num_chrono] = mat1d.construct_chrono_var_matrix(
                    times, prep['nonconsec_step_times'], ckw['chrono_vmm_epsilon'], ckw['chrono_error_structure'])
            if has_eis:
                v[num_chrono:, num_chrono:] = ctx.eis_var_matrix(frequencies, kw['eis_vmm_epsilon'], kw['eis_reim_cor'],
                                                                kw['eis_error_structure'] == 'uniform')
            return v
        vmm = self._memo('vmm', build_vmm, times if has_chrono else 0, prep.get('nonconsec_step_times', 0),
                         frequencies if has_eis else 0, ckw['chrono_vmm_epsilon'], str(ckw['chrono_error_structure']),
                         kw['eis_vmm_epsilon'], kw['eis_reim_cor'], str(kw['eis_error_structure']))

        # vz_offset strength (drt1d.py:500-522), l1 vector (552-556), h (qphb.py:521-557)
        vz_strength = None
        if 'vz_offset' in sp:
            cs, es = self._vz_strength(times, frequencies, prep['nonconsec_step_times'], ckw['vz_offset_eps'])
            vz_strength = np.concatenate([cs, np.tile(es, 2)])
        l1 = np.zeros(n)
        l1[ns:] = hypers['l1_lambda_0']
        if self.fit_dop:
            l1[dop[0]:dop[1]] = hypers['dop_l1_lambda_0']
        # DRT._get_neg_allowed_indices (drt1d.py:82-91): the loop's QPs allow negative coefficients only inside the window,
        # initialize_weights is called without it (drt1d.py:657-660)
        h_init = None
        if ckw['neg_allowed_tau_range'] is not None:
            if kw['nonneg']:
                raise ValueError("If nonneg==True, neg_allowed_tau_range cannot be specified")
            lo, hi = ckw['neg_allowed_tau_range']
            idx = np.where((basis_tau >= lo) & (basis_tau <= hi))[0] + ns
            h = qphb.make_h_constraint(None, n, sp, False, neg_allowed_indices=idx)
            h_init = qphb.make_h_constraint(None, n, sp, False)
        else:
            h = qphb.make_h_constraint(None, n, sp, kw['nonneg'])

        prep.update(rzm=rzm, rzv=rzv, pen=pen, vmm=vmm, special=sp, ns=ns, n=n, m=m, dop=dop, l1=l1, h=h, h_init=h_init,
                    vz_strength=vz_strength, num_chrono=num_chrono, chrono_uniform=ckw['chrono_error_structure'] == 'uniform', num_eis=len(frequencies) if has_eis else 0,
                    basis_tau=basis_tau, toeplitz_m=tpl_m and not series_neg, coefficient_scale=coefficient_scale,
                    impedance_scale=impedance_scale, input_signal_scale=input_scale, response_signal_scale=response_scale,
                    dop_scale_vector=dop_scale, frequencies=frequencies)
        return prep

    def _solve_data_scale(self, ctx, prep, hypers, kw):
        """DRT._solve_data_scale + the rescaling block of _qphb_fit_core (drt1d.py:568-606, 5421-5437;
        qphb.estimate_x_rp 1684-1717): a lightly penalised QP (l2_lambda_0 = 1e-4 with the DOP / DRT ratio kept, scalar
        l1 = 1e-3, unit weights; Gram and QP on the device) gives Rp = basis_area * sum|x_drt|; the data are rescaled to
        rp_scale / Rp and the DOP columns to the DRT magnitude."""
        n, ns, dop = prep['n'], prep['ns'], prep['dop']
        l2 = np.zeros((n, n))
        for k, dw in enumerate(hypers['derivative_weights']):
            if dw > 0:          # s = s_0 = 1 and rho = rho_0 = 1 at this point (drt1d.py:559-566)
                mk = prep['pen'][k].copy()
                mk[ns:, ns:] *= 1e-4 * dw * hypers['rho_0'][k]
                if dop:
                    mk[dop[0]:dop[1], dop[0]:dop[1]] *= (hypers['dop_l2_lambda_0'] / hypers['l2_lambda_0'] * 1e-4
                                                         * hypers['dop_derivative_weights'][k] * hypers['dop_rho_0'][k])
                sq = np.sqrt(np.full(n, float(hypers['s_0'][k])))
                l2 += (sq[:, None] * mk) * sq[None, :]
        p_mat, q_vec = ctx.weighted_gram(prep['rzm'], np.ones(prep['m']), prep['rzv'], l2=l2, l1=np.full(n, 1e-3))
        res = ctx.qp_batch(p_mat, q_vec, prep['h'])
        if res['status'][0] < 0:
            raise ValueError("Rank(A) < p or Rank([P; A; G]) < n")
        x_rp = res['x'][0]
        rp_est = np.sum(np.abs(x_rp[ns:])) * (np.sqrt(np.pi) / self.tau_epsilon)        # predict_r_p(absolute, raw)
        factor = hypers['rp_scale'] / rp_est
        prep['rzv'] = prep['rzv'] * factor
        # update_data_scale (drtbase.py:516-536), galvanostatic
        prep['coefficient_scale'] /= factor
        prep['impedance_scale'] /= factor
        if prep['response_signal_scale'] is not None:
            prep['response_signal_scale'] /= factor
            prep['scaled_response_offset'] *= factor
        if dop:
            dop_factor = np.max(np.abs(x_rp[ns:])) / np.max(np.abs(x_rp[dop[0]:dop[1]]))
            prep['dop_scale_vector'] = prep['dop_scale_vector'] / dop_factor
            rzm = prep['rzm'].copy()           # the block may be shared between batch members
            rzm[:, dop[0]:dop[1]] /= dop_factor
            prep['rzm'] = rzm
        prep['rp_qp_iterations'] = int(res['iterations'][0])

    def _hybrid_weight_factors(self, prep, meas, hypers, ckw):
        """drt1d.py:743-803: (chrono, eis) weight factors of a hybrid fit -- given, or from the two data sets' apparent
        polarisation resistances ('rp'), or 1."""
        ef, cf = ckw['eis_weight_factor'], ckw['chrono_weight_factor']
        if prep['num_chrono'] == 0 or prep['num_eis'] == 0:
            return 1.0, 1.0            # only hybrid fits apply them
        if ef is not None and cf is not None:
            return float(cf), float(ef)
        if ef is not None or cf is not None:
            warnings.warn("Both eis_weight_factor and chrono_weight_factor must be provided. If only one is provided, "
                          "it will be ignored.")
        method = ckw['hybrid_weight_factor_method']
        if method is None:
            return 1.0, 1.0
        if method == 'rp':
            times, i_signal, v_signal, frequencies, z = meas
            rp_eis = pp.estimate_rp(None, None, None, None, None, np.asarray(z))
            rp_chrono = pp.estimate_rp(prep['sample_times'], prep['step_times'], prep['step_sizes'],
                                       np.asarray(v_signal, dtype=float), self.step_model, None)
            rp_tot = prep['coefficient_scale'] * hypers['rp_scale']
            ef_ = rp_eis ** 0.75 / (rp_chrono ** 0.25 * rp_tot ** 0.5) if ef is None else ef
            cf_ = rp_chrono ** 0.75 / (rp_eis ** 0.25 * rp_tot ** 0.5) if cf is None else cf
            return float(cf_), float(ef_)
        if method == 'weight':
            # decided on the device after initialize_weights (weight_method_kernel); a given factor stays fixed
            return ('weight', cf, ef)
        raise ValueError(f"Invalid hybrid_weight_factor_method argument {method}. Options: 'weight', 'rp', None")

    def _prepared_desc(self, prep, hypers):
        d = _ffi.PreparedDesc()
        sp = prep['special']
        d.m, d.n, d.ns = prep['m'], prep['n'], prep['ns']
        d.dop_start, d.dop_size = (prep['dop'][0], prep['dop'][1] - prep['dop'][0]) if prep['dop'] else (0, 0)
        d.vz_index = sp['vz_offset']['index'] if 'vz_offset' in sp else -1
        d.vb_start, d.vb_size = (sp['v_baseline']['index'], sp['v_baseline']['size']) if 'v_baseline' in sp else (0, 0)
        d.num_chrono = prep['num_chrono']
        d.toeplitz_m = int(prep['toeplitz_m'])
        d.basis_area = float(np.sqrt(np.pi) / self.tau_epsilon)
        d.chrono_vmm_uniform = int(prep['num_chrono'] > 0 and prep['chrono_uniform'])
        if prep['dop']:
            d.dop_l2_lambda_0 = float(hypers['dop_l2_lambda_0'])
            for name in ('dop_derivative_weights', 'dop_s_alpha', 'dop_rho_alpha', 'dop_s_0', 'dop_rho_0'):
                vals = np.broadcast_to(np.asarray(hypers[name], dtype=float), (3,))
                for k in range(3):
                    getattr(d, name)[k] = float(vals[k])
        return d

    def _split_kwargs(self, fit_kw):
        """chrono / hybrid keywords of _qphb_fit_core that this build honours; the rest goes to DRT._make_opts"""
        ckw = dict(_CHRONO_KW_DEFAULTS)
        rest = {}
        for key, val in fit_kw.items():
            if key in ckw:
                ckw[key] = val
            elif key in _UNSUPPORTED:
                if val != _UNSUPPORTED[key]:
                    raise NotImplementedError(f"{key}={val!r} is not built (only the default {_UNSUPPORTED[key]!r})")
            else:
                rest[key] = val
        if ckw['chrono_error_structure'] not in (None, 'uniform'):
            raise ValueError(f"Invalid error_structure {ckw['chrono_error_structure']}")
        return ckw, rest

    # keywords _qphb_fit_core does not hand to its outlier-detection pass (drt1d.py:216-247): they take their defaults there
    _NOT_IN_OUTLIER_PASS = ('vz_offset', 'vz_offset_scale', 'vz_offset_eps', 'eis_weight_factor', 'chrono_weight_factor',
                            'hybrid_weight_factor_method', 'weight_factor', 'xtol', 'max_iter', 'iw_l1_lambda_0',
                            'iw_l2_lambda_0', 'remove_outliers', 'outlier_thresh', 'step_sizes')

    def _drop_extremes(self, meas, extreme_kw=None):
        """drt1d.py:187-212: points whose value lies far outside the central quantile range of the raw signal are removed
        (a chrono sample if the current or the voltage is extreme, an impedance point if either part is)."""
        ekw = extreme_kw if extreme_kw is not None else {'qr_size': 0.8, 'qr_thresh': 1.5}
        times, i_signal, v_signal, frequencies, z = meas
        if times is not None:
            times, i_signal, v_signal = (np.asarray(a) for a in (times, i_signal, v_signal))
            flag = pp.identify_extreme_values(i_signal, **ekw) | pp.identify_extreme_values(v_signal, **ekw)
            if np.any(flag):
                if self.warn:
                    warnings.warn('Identified extreme values in chrono data at the following '
                                  f'indices: {np.where(flag)[0].tolist()}. These data points will be removed before fitting')
                times, i_signal, v_signal = times[~flag], i_signal[~flag], v_signal[~flag]
        if frequencies is not None:
            frequencies, z = np.asarray(frequencies), np.asarray(z)
            flag = pp.identify_extreme_values(z.real, **ekw) | pp.identify_extreme_values(z.imag, **ekw)
            if np.any(flag):
                if self.warn:
                    warnings.warn('Identified extreme values in EIS data at the following '
                                  f'indices: {np.where(flag)[0].tolist()}. These data points will be removed before fitting')
                frequencies, z = frequencies[~flag], z[~flag]
        return times, i_signal, v_signal, frequencies, z

    def _remove_outliers(self, meas, fit_kw, ckw):
        """drt1d.py:214-302: an initialize_weights-only pass with the outlier-aware weights (device: max_iter = 0) gives
        outlier_t; points with 1 - outlier_t above the threshold are dropped (an impedance point if either part is),
        the step times found before the removal are kept."""
        cleaned, step_times, masks = self._remove_outliers_batch([meas], fit_kw, ckw)
        self.chrono_outlier_index, self.eis_outlier_index = masks[0]
        return cleaned[0], step_times

    def _remove_outliers_batch(self, measurements, fit_kw, ckw):
        """the detection pass of _remove_outliers for measurements of ONE protocol as one device batch (mapping.fit_observations):
        returns (cleaned measurements, the protocol's step times, [(chrono mask | None, eis mask | None)])"""
        pass_kw = {k: v for k, v in fit_kw.items() if k not in self._NOT_IN_OUTLIER_PASS}
        preps, plan = self._fit_prepared(list(measurements), dict(pass_kw, max_iter=0), _init_only=True)
        outlier_t = plan.get('outlier_t')
        step_times = preps[0].get('step_times')
        cleaned, masks = [], []
        self.chrono_outliers = self.eis_outliers = None
        for b, (meas, pr) in enumerate(zip(measurements, preps)):
            times, i_signal, v_signal, frequencies, z = meas
            nc = pr['num_chrono']
            flagged = (1 - outlier_t[b]) > ckw['outlier_thresh']
            chrono_idx = flagged[:nc] if times is not None else None
            eis_idx = None
            if frequencies is not None:
                nf = len(frequencies)
                eis_idx = flagged[nc:nc + nf] | flagged[nc + nf:]
            if times is not None and np.any(chrono_idx):
                if self.warn:
                    warnings.warn('Found outliers in chrono data at the following '
                                  f'indices: {np.where(chrono_idx)[0].tolist()}. These data points will be removed before fitting')
                times, i_signal, v_signal = (np.asarray(a) for a in (times, i_signal, v_signal))
                self.chrono_outliers = (times[chrono_idx], i_signal[chrono_idx], v_signal[chrono_idx])
                times, i_signal, v_signal = times[~chrono_idx], i_signal[~chrono_idx], v_signal[~chrono_idx]
            if frequencies is not None and np.any(eis_idx):
                if self.warn:
                    warnings.warn('Found outliers in EIS data at the following '
                                  f'indices: {np.where(eis_idx)[0].tolist()}. These data points will be removed before fitting')
                frequencies, z = np.asarray(frequencies), np.asarray(z)
                self.eis_outliers = (frequencies[eis_idx], z[eis_idx])
                frequencies, z = frequencies[~eis_idx], z[~eis_idx]
            cleaned.append((times, i_signal, v_signal, frequencies, z))
            masks.append((chrono_idx, eis_idx))
        return cleaned, step_times, masks

    def _fit_prepared(self, measurements, fit_kw, history_of=-1, _init_only=False):
        """measurements: list of (times, i_signal, v_signal, frequencies, z) of identical shapes (one protocol)."""
        ckw, rest = self._split_kwargs(fit_kw)
        if ckw['discard_first_n'] is not None and any(meas[0] is not None for meas in measurements):
            # drt1d.py:167-178: drop the first samples of every step; the step is then assumed to have happened that much
            # earlier than the first kept sample
            nd = int(ckw['discard_first_n'])
            cleaned, offset = [], ckw['step_offset_size']
            for times, i_signal, v_signal, frequencies, z in measurements:
                dt_short = np.min(np.diff(times))
                _, (times, i_signal, v_signal) = pp.discard_first_n_chrono(times, i_signal, v_signal, nd, self.chrono_mode)
                if ckw['step_offset_size'] is None:
                    offset = -(dt_short + np.min(np.diff(times)) * (nd - 1e-8))
                cleaned.append((times, i_signal, v_signal, frequencies, z))
            measurements = cleaned
            fit_kw = dict(fit_kw, discard_first_n=None, step_offset_size=offset)
            ckw, rest = self._split_kwargs(fit_kw)
        if ckw['remove_extremes']:
            # drt1d.py:187-212: rough pre-filter on the raw signals (quantile-range rule), before anything else
            if len(measurements) != 1:
                raise NotImplementedError("remove_extremes changes the data size per measurement: single fits only "
                                          "(mapping.fit_observations filters every observation before it forms batches)")
            measurements = [self._drop_extremes(measurements[0], ckw['extreme_kw'])]
            fit_kw = dict(fit_kw, remove_extremes=False)
            ckw, rest = self._split_kwargs(fit_kw)
        if ckw['remove_outliers']:
            if rest.get('outlier_p') is None:
                raise ValueError('If remove_outliers is True, the prior probability of outlier presence, outlier_p, '
                                 'must be specified. A good starting value might be 0.01-0.05')
            if len(measurements) != 1:
                raise NotImplementedError("remove_outliers changes the data size per measurement: single fits only "
                                          "(mapping.fit_observations runs the detection pass per group and forms new batches)")
            meas, step_times = self._remove_outliers(measurements[0], fit_kw, ckw)
            measurements = [meas]
            fit_kw = dict(fit_kw, remove_outliers=False, outlier_p=None)
            if step_times is not None:
                fit_kw.update(step_times=step_times, step_sizes=None)
            ckw, rest = self._split_kwargs(fit_kw)
        opts, hypers, kw = self._make_opts(rest)
        ctx = self._context if self._context is not None else _ffi.get_context(self.device)
        self._build_memo = {}
        preps = [self._prepare_measurement(ctx, *meas, kw, ckw, hypers) for meas in measurements]
        self._build_memo_kept, self._build_memo = self._build_memo, {}        # (what this call used, for the next one)
        for pr in preps:        # the loop's inputs before any host- or device-side rescale (diagnostics, tests)
            pr['rzv_initial'], pr['rzm_initial'] = pr['rzv'], pr['rzm']
        if ckw['solve_rp'] and kw['scale_data']:
            for pr in preps:
                self._solve_data_scale(ctx, pr, hypers, kw)
        p0 = preps[0]
        for pr in preps[1:]:
            if pr['rzm'].shape != p0['rzm'].shape or pr['special'] != p0['special']:
                raise ValueError('all measurements of a batch must share one protocol (same m, n, special parameters)')
            # one plan = one set of shared penalty / variance matrices (memoised builds: identical inputs -> same object)
            if pr['vmm'] is not p0['vmm'] or not np.array_equal(pr['basis_tau'], p0['basis_tau']):
                raise ValueError('all measurements of a batch must share the basis grid and the sampling grids')
        shared = 'vz_offset' not in p0['special'] and all(pr['rzm'] is p0['rzm'] or np.array_equal(pr['rzm'], p0['rzm'])
                                                         for pr in preps[1:])
        desc = self._prepared_desc(p0, hypers)
        hybrid = p0['num_chrono'] > 0 and p0['num_eis'] > 0
        desc.init_weights_separately = int(bool(ckw['init_weights_separately']) and hybrid)
        weight_on_device = (hybrid and ckw['hybrid_weight_factor_method'] == 'weight'
                            and (ckw['eis_weight_factor'] is None or ckw['chrono_weight_factor'] is None))
        if weight_on_device:
            desc.weight_method = 1
            desc.fixed_chrono_factor = -1.0 if ckw['chrono_weight_factor'] is None else float(ckw['chrono_weight_factor'])
            desc.fixed_eis_factor = -1.0 if ckw['eis_weight_factor'] is None else float(ckw['eis_weight_factor'])
        if self._plan is not None:
            self._plan.close()
            self._plan_key = None
        plan = _ffi.PreparedPlan(ctx, desc, p0['pen'], p0['vmm'], p0['h'], p0['l1'], vz_strength=p0['vz_strength'],
                                 opts=opts, capacity=len(preps))
        self._plan = plan
        rows = []
        for pr, meas in zip(preps, measurements):
            cf, ef = self._hybrid_weight_factors(pr, meas, hypers, ckw)[-2:] if weight_on_device else \
                self._hybrid_weight_factors(pr, meas, hypers, ckw)
            pr['chrono_weight_factor'], pr['eis_weight_factor'] = cf, ef
            if not weight_on_device:
                rows.append(np.concatenate([np.full(pr['num_chrono'], cf), np.full(pr['m'] - pr['num_chrono'], ef)]))
        rows = np.array(rows) if rows else np.ones((1, 1))
        plan.set_weight_factors(kw['weight_factor'], None if np.all(rows == 1.0) else rows)
        plan.set_init_h(p0['h_init'])
        plan.upload(p0['rzm'] if shared else np.stack([pr['rzm'] for pr in preps]), np.stack([pr['rzv'] for pr in preps]))
        plan.record_history(-1 if _init_only else history_of)
        plan.fit()
        if _init_only:
            return preps, plan
        out = plan.download(s_vectors=True)
        if weight_on_device:
            wf = plan.get('weight_factors')
            for b, pr in enumerate(preps):
                pr['chrono_weight_factor'], pr['eis_weight_factor'] = float(wf[b, 0]), float(wf[b, 1])
        if opts.update_scale:
            # the device divided its scale (started at 1) by every update's factor: fold it into the host-side scales
            # (update_data_scale, drtbase.py:516-536) and take the rescaled data vector back
            rv_dev = plan.get('rv')
            for b, pr in enumerate(preps):
                f = float(out['coefficient_scale'][b])
                pr['coefficient_scale'] *= f
                pr['impedance_scale'] *= f
                if pr['response_signal_scale'] is not None:
                    pr['response_signal_scale'] *= f
                    pr['scaled_response_offset'] /= f
                pr['rzv'] = rv_dev[b]
        self.basis_tau = p0['basis_tau']
        self.special_qp_params = p0['special']
        self.fit_kwargs = dict(hypers, **kw, **ckw)
        return preps, out, hypers, kw, ckw

    # ---- extract_qphb_parameters (drt1d.py:6228-6289) + the sigma vectors (1071-1081) -------------------------------
    def _extract(self, prep, x, weights, kw, ckw):
        sp, ns, nc = prep['special'], prep['ns'], prep['num_chrono']
        cs = prep['coefficient_scale']
        fp = {'x': x[ns:] * cs,
              'R_inf': x[sp['R_inf']['index']] * cs if 'R_inf' in sp else 0}
        if 'v_baseline' in sp:
            a = sp['v_baseline']['index']
            vbx = x[a:a + sp['v_baseline']['size']] * (1.0 / prep['v_baseline_scale'])
            vbx[0] -= prep['scaled_response_offset']
            fp['v_baseline'] = vbx * prep['response_signal_scale']
        if 'vz_offset' in sp:
            fp['vz_offset'] = x[sp['vz_offset']['index']]
        fp['inductance'] = x[sp['inductance']['index']] * (cs * kw['inductance_scale']) if 'inductance' in sp else 0
        fp['C_inv'] = x[sp['C_inv']['index']] * (cs * kw['capacitance_scale']) if 'C_inv' in sp else 0
        if prep['dop']:
            fp['x_dop'] = x[prep['dop'][0]:prep['dop'][1]] * (prep['dop_scale_vector'] * cs)
        sigma = 1.0 / weights
        nf = prep['num_eis']
        fp['v_sigma_tot'] = sigma[:nc] * prep['response_signal_scale'] if nc else None
        fp['v_sigma_res'] = None
        fp['z_sigma_tot'] = (sigma[nc:nc + nf] + 1j * sigma[nc + nf:]) * prep['impedance_scale'] if nf else None
        fp['vz_offset_eps'] = ckw['vz_offset_eps']
        return fp

    def _store_single(self, preps, out, hypers, kw, ckw, fit_type):
        b, prep, plan = 0, preps[0], self._plan
        if out['status'][b] < 0:
            raise ValueError("Rank(A) < p or Rank([P; A; G]) < n")
        if out['status'][b] == 1 and self.warn:
            warnings.warn(f"Solution did not converge within {kw['max_iter']} iterations. This is usually not an issue.")
        fp = self._extract(prep, out['x'][b], out['weights'][b], kw, ckw)
        fp['p_matrix'] = plan.p_matrix(b)
        fp['q_vector'] = out['q_vector'][b]
        self.fit_parameters = fp
        self.coefficient_scale = prep['coefficient_scale']
        self.impedance_scale = prep['impedance_scale']
        self.input_signal_scale, self.response_signal_scale = prep['input_signal_scale'], prep['response_signal_scale']
        self.step_times, self.step_sizes = prep.get('step_times'), prep.get('step_sizes')
        self.sample_index = prep.get('sample_index')            # drtbase.py:335: kept samples of a down-sampled record
        self.dop_scale_vector = prep['dop_scale_vector']
        self.scaled_response_offset = prep.get('scaled_response_offset')
        self.v_baseline_scale = prep.get('v_baseline_scale')
        self.inductance_scale = kw['inductance_scale']
        self.series_neg = bool(ckw['series_neg'])
        hist = plan.history()
        self.qphb_history = [dict(x=hist['x'][i], rho_vector=hist['rho'][i], weights=hist['weights'][i],
                                  dop_rho_vector=hist['dop_rho'][i] if 'dop_rho' in hist else None)
                             for i in range(len(hist['x']))]
        rzm_final = plan.get('rzm')
        scaled = out['weights'][b] * np.concatenate([np.full(prep['num_chrono'], prep['chrono_weight_factor']),
                                                     np.full(prep['m'] - prep['num_chrono'], prep['eis_weight_factor'])])
        self.qphb_params = {'weights': scaled, 'true_weights': out['weights'][b], 'rho_vector': out['rho'][b],
                            'dop_rho_vector': plan.get('dop_rho')[b] if prep['dop'] else None,
                            's_vectors': list(out['s_vectors'][b]), 'p_matrix': fp['p_matrix'], 'q_vector': fp['q_vector'],
                            'rm': rzm_final[b] if rzm_final.ndim == 3 else rzm_final, 'rv': prep['rzv'], 'vmm': prep['vmm'],
                            'penalty_matrices': {f'm{k}': prep['pen'][k] for k in range(3)},
                            'l1_lambda_vector': prep['l1'], 'num_eis': prep['num_eis'], 'num_chrono': prep['num_chrono'],
                            'vz_strength_vec': prep['vz_strength'] if prep['vz_strength'] is not None else 1,
                            'xmx_norms': plan.get('xmx')[b], 'est_weights': plan.get('est_weights')[b],
                            'qp_iterations': hist['qp_iterations'], 'outer_iterations': int(out['outer_iters'][b]),
                            'hypers': hypers, 'chrono_weight_factor': prep['chrono_weight_factor'],
                            'eis_weight_factor': prep['eis_weight_factor']}
        self.cvx_result = {'x': out['x'][b]}
        self.fit_type = fit_type
        self._prep = prep
        self._last_prepared = None
        return fp

    # ---- warm restarts on prepared plans (drt1d.py:1270-1365): candidates (1497-1632) and PFRT (2558-2715) ------------
    def _last_preps(self):
        lp = getattr(self, '_last_prepared', None)
        if lp:
            return lp[0]
        if getattr(self, '_prep', None) is None:
            raise Exception('continue_from_init needs a finished qphb fit')
        return [self._prep]

    def _collect_prepared(self):
        """the state a warm restart leaves on the device, as arrays with a leading measurement axis"""
        plan = self._plan
        out = plan.download(s_vectors=True)
        res = {k: out[k] for k in ('x', 'rho', 'weights', 's_vectors', 'q_vector', 'outer_iters', 'qp_iters_total', 'status')}
        if plan.desc.dop_size > 0:
            res['dop_rho'] = plan.get('dop_rho')
        res['timings_ms'], res['launches'] = plan.timings()
        return res

    def _continue_prepared(self, x_init=None, rho_vector=None, s_vectors=None, weights=None, dop_rho_vector=None,
                           weight_factor=1, eis_weight_factor=None, chrono_weight_factor=None, xtol=1e-2, max_iter=10,
                           min_iter=2, history_of=-1, **kw):
        """DRT._continue_from_init (drt1d.py:1270-1365) for the last chrono / joint / DOP fit (single or batch): arrays
        carry a leading measurement axis, None keeps what the device holds.  As upstream: the chrono / eis weight
        factors multiply the weights at the top of every iteration together with ``weight_factor`` -- a factor that is
        not given falls back to the FIT's chrono factor, for both blocks (1284-1287) --, the vz_offset column is rewritten
        after every iteration from a copy of the matrix frozen at entry (1295-1298), est_weights, xmx / dop_xmx norms
        and the data scale stay."""
        plan, preps = self._plan, self._last_preps()
        fit_kw = dict(self.fit_kwargs)
        fit_kw.update(kw)
        fit_kw.update(xtol=xtol, max_iter=max_iter)
        _, rest = self._split_kwargs(fit_kw)
        opts, _, _ = self._make_opts(rest)
        rows = []
        for pr in preps:
            nc, m = pr['num_chrono'], pr['m']
            if nc > 0 and pr['num_eis'] > 0:
                cf = pr['chrono_weight_factor'] if chrono_weight_factor is None else chrono_weight_factor
                ef = pr['chrono_weight_factor'] if eis_weight_factor is None else eis_weight_factor
            else:
                cf = ef = 1.0
            rows.append(np.concatenate([np.full(nc, float(cf)), np.full(m - nc, float(ef))]))
        rows = np.array(rows)
        plan.set_weight_factors(1.0, None if np.all(rows == 1.0) else rows)
        plan.set_state(x=x_init, rho=rho_vector, s=s_vectors, weights=weights, dop_rho=dop_rho_vector)
        plan.record_history(history_of)
        plan.continue_fit(opts, weight_factor=weight_factor, min_iter=min_iter)
        res = self._collect_prepared()
        if history_of >= 0:
            res['history'] = plan.history()
        return res

    def _pfrt_prepared(self, measurements, factors, max_iter_per_step, max_init_iter, xtol, nonneg, kw, after_init=None):
        """DRT._pfrt_fit_core (drt1d.py:2558-2700) on a prepared plan: the full fit at factors[0], one warm restart per
        further factor with the fit's own chrono / eis weight factors (2660-2668); step log-likelihoods from weights
        re-estimated on the current iterate alone (2618-2622), all on the device."""
        from . import qphb
        base = qphb.get_default_hypers(True, self.fit_dop, self.nu_basis_type)
        base.update({k: v for k, v in kw.items() if k in base})
        if factors is None:
            factors = np.logspace(-1, 1, 11)
        factors = np.asarray(factors, dtype=float)
        s_0 = np.broadcast_to(np.asarray(base['s_0'], dtype=float), (3,))

        def step_hypers(f):
            return dict(s_0=s_0 * f, l2_lambda_0=base['l2_lambda_0'] / f)

        single = len(measurements) == 1
        init_kw = dict(kw, nonneg=nonneg, max_iter=max_init_iter, xtol=xtol, **step_hypers(factors[0]))
        fitted = self._fit_prepared(measurements, init_kw, history_of=0 if single else -1)
        preps, out = fitted[0], fitted[1]
        if single:          # fit_parameters / qphb_params / qphb_history of the first step, as fit_hybrid leaves them upstream
            self._store_single(*fitted, 'qphb_hybrid' if preps[0]['num_eis'] and preps[0]['num_chrono'] else
                               ('qphb_chrono' if preps[0]['num_chrono'] else 'qphb_eis'))
        else:
            self._last_prepared = (preps, None)
        step_x, step_llh, step_iters = [out['x'].copy()], [self.evaluate_step_llh_batch()], [out['outer_iters'].copy()]
        status = np.array(out['status']).copy()
        history = [self._plan.history()] if single else None
        if after_init is not None:
            after_init(out)
        for f in factors[1:]:
            cf = np.array([pr['chrono_weight_factor'] for pr in preps])
            ef = np.array([pr['eis_weight_factor'] for pr in preps])
            same = np.all(cf == cf[0]) and np.all(ef == ef[0])
            if not same:
                raise NotImplementedError('per-measurement chrono / eis weight factors in a PFRT batch')
            res = self._continue_prepared(xtol=xtol, max_iter=max_iter_per_step, history_of=0 if single else -1,
                                          chrono_weight_factor=float(cf[0]), eis_weight_factor=float(ef[0]),
                                          **step_hypers(f))
            step_x.append(res['x'].copy())
            step_llh.append(self.evaluate_step_llh_batch())
            step_iters.append(res['outer_iters'].copy())
            status = combine_status(status, res['status'])
            if single:
                history.append(res['history'])
        self.pfrt_result = {'factors': factors, 'step_x': np.array(step_x), 'step_llh': np.array(step_llh),
                            'step_iters': np.array(step_iters), 'status': status}
        if single:
            self.pfrt_history = [dict(x=h['x'][i], rho_vector=h['rho'][i], weights=h['weights'][i],
                                      dop_rho_vector=h['dop_rho'][i] if 'dop_rho' in h else None)
                                 for h in history for i in range(len(h['x']))]
        return fitted

    def pfrt_fit_hybrid(self, times, i_signal, v_signal, frequencies, z, factors=None, max_iter_per_step=10,
                        max_init_iter=20, xtol=1e-2, nonneg=True, **kw):
        """DRT.pfrt_fit_hybrid (drt1d.py:2705-2715): leaves fit_parameters / qphb_params of the LAST step's state as a fit
        does, pfrt_result {'factors', 'step_x' (S, 1, n), 'step_llh' (S, 1), 'step_iters' (S, 1)} and pfrt_history."""
        fitted = self._pfrt_prepared([(times, i_signal, v_signal, frequencies, z)], factors, max_iter_per_step,
                                     max_init_iter, xtol, nonneg, kw)
        self.fit_type = 'qphb_hybrid'
        return self.pfrt_result

    def pfrt_fit_chrono(self, times, i_signal, v_signal, factors=None, max_iter_per_step=10, max_init_iter=20,
                        xtol=1e-2, nonneg=True, error_structure='uniform', vmm_epsilon=4, **kw):
        """DRT.pfrt_fit_chrono (drt1d.py:2699-2703)"""
        self._pfrt_prepared([(times, i_signal, v_signal, None, None)], factors, max_iter_per_step, max_init_iter, xtol,
                            nonneg, dict(kw, chrono_error_structure=error_structure, chrono_vmm_epsilon=vmm_epsilon))
        self.fit_type = 'qphb_chrono'
        return self.pfrt_result

    def pfrt_fit_hybrid_batch(self, times, i_batch, v_batch, frequencies, z_batch, factors=None, max_iter_per_step=10,
                              max_init_iter=20, xtol=1e-2, nonneg=True, **kw):
        """pfrt_fit_hybrid for B joint measurements of one protocol at once (what DRTMD with fit_type='pfrt' loops over,
        drtmd.py:98-100, 1338): step_x (S, B, n), step_llh (S, B), step_iters (S, B)."""
        meas = [(times, i_batch[b], v_batch[b], frequencies, z_batch[b]) for b in range(len(z_batch))]
        fitted = self._pfrt_prepared(meas, factors, max_iter_per_step, max_init_iter, xtol, nonneg, kw)
        self._last_prepared = (fitted[0], None)
        return self.pfrt_result

    # ---- public fits --------------------------------------------------------------------------------------------------
    def fit_hybrid(self, times, i_signal, v_signal, frequencies, z, **kw):
        """DRT.fit_hybrid (drt1d.py:1244-1268): joint fit of one chrono measurement and one impedance spectrum."""
        res = self._fit_prepared([(times, i_signal, v_signal, frequencies, z)], kw, history_of=0)
        return self._store_single(*res, 'qphb_hybrid')

    def fit_chrono(self, times, i_signal, v_signal, error_structure='uniform', vmm_epsilon=4, **kw):
        """DRT.fit_chrono (drt1d.py:1195-1213)."""
        res = self._fit_prepared([(times, i_signal, v_signal, None, None)],
                                 dict(kw, chrono_error_structure=error_structure, chrono_vmm_epsilon=vmm_epsilon),
                                 history_of=0)
        return self._store_single(*res, 'qphb_chrono')

    def fit_hybrid_batch(self, times, i_batch, v_batch, frequencies, z_batch, **kw):
        """B joint measurements of one protocol (same sample times / frequencies), fitted concurrently: the hybrid
        counterpart of fit_eis_batch.  Returns a dict of arrays (x in data units, per-measurement specials)."""
        meas = [(times, i_batch[b], v_batch[b], frequencies, z_batch[b]) for b in range(len(z_batch))]
        return self._fit_prepared_batch(meas, kw)

    def batch_fits(self):
        """Per-observation views of the last prepared batch fit, carrying what the mapping step reads from a fitted DRT
        (fit_parameters incl. p_matrix / q_vector, special_qp_params, the scale attributes): feed them to
        hipdrt.mapping.resolve.resolve_observations / resolve_group."""
        import types
        preps, fps = self._last_prepared
        fits = []
        for b, (pr, fp) in enumerate(zip(preps, fps)):
            fp = dict(fp, p_matrix=self._plan.p_matrix(b))
            fits.append(types.SimpleNamespace(
                fit_parameters=fp, special_qp_params=pr['special'], coefficient_scale=pr['coefficient_scale'],
                impedance_scale=pr['impedance_scale'], response_signal_scale=pr['response_signal_scale'],
                scaled_response_offset=pr.get('scaled_response_offset'), v_baseline_scale=pr.get('v_baseline_scale'),
                dop_scale_vector=pr['dop_scale_vector'], inductance_scale=self.inductance_scale, basis_tau=pr['basis_tau']))
        return fits

    def _fit_prepared_batch(self, meas, kw):
        preps, out, hypers, fkw, ckw = self._fit_prepared(meas, kw)
        self.inductance_scale = fkw['inductance_scale']
        fps = [self._extract(pr, out['x'][b], out['weights'][b], fkw, ckw) for b, pr in enumerate(preps)]
        for b, fp in enumerate(fps):
            fp['q_vector'] = out['q_vector'][b]
        self._last_prepared = (preps, fps)
        res = {key: np.array([fp[key] for fp in fps]) for key in fps[0]
               if fps[0][key] is not None and key not in ('vz_offset_eps', 'q_vector')}
        res.update(x_scaled=out['x'], fit_x=res['x'], outer_iters=out['outer_iters'], status=out['status'],
                   qp_iters_total=out['qp_iters_total'], rho=out['rho'], weights=out['weights'],
                   coefficient_scale=np.array([pr['coefficient_scale'] for pr in preps]), basis_tau=preps[0]['basis_tau'])
        return res
