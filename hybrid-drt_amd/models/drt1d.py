"""Drop-in for the EIS fit path of hybdrt.models.DRT (hybdrt/models/drt1d.py + drtbase.py).

``DRT(**ctor).fit_eis(frequencies, z, **hypers)`` leaves ``fit_parameters``, ``qphb_params``,
``qphb_history`` populated like the reference; ``fit_eis_batch`` is the batched form the reference runs as a
serial loop in DRTMD.fit_observations (hybdrt/mapping/drtmd.py:303-319).  All arithmetic -- lookup tables,
Z'/Z'' and penalty matrices, the QPHB loop -- runs in libhipdrt.so on one MI355X; this module only decides
grids and options and rescales results (host logic mirrored from the reference, cited per method)."""
import warnings

import numpy as np

from .. import _ffi, preprocessing as pp
from ..matrices import mat1d
from ..utils.array import is_uniform
from . import qphb
from .prepared import PreparedFitMixin, combine_status

_FIT_KW_DEFAULTS = dict(  # DRT._qphb_fit_core keyword defaults (drt1d.py:102-137) that the device loop honours
    nonneg=True, scale_data=True, ohmic_penalty=1e-6, inductance_penalty=1e-6, inductance_scale=1e-5,
    capacitance_penalty=1e-6, capacitance_scale=1e-3, update_scale=False,
    penalty_type='integral', eis_error_structure=None, eis_vmm_epsilon=0.25, eis_reim_cor=0.25,
    iw_l1_lambda_0=1e-4, iw_l2_lambda_0=1e-4, eff_hp=True, weight_factor=1, xtol=1e-2, max_iter=50)


class DRT(PreparedFitMixin):
    def __init__(self, fixed_basis_tau=None, tau_supergrid=None, tau_basis_type='gaussian', tau_epsilon=None,
                 basis_tau_ppd=10, extend_basis_decades=1, interpolate_integrals=True, fit_dop=False,
                 fit_inductance=True, fit_ohmic=True, fit_capacitance=False, frequency_precision=10,
                 fixed_basis_nu=None, nu_basis_type='gaussian', nu_epsilon=None, normalize_dop=True,
                 step_model='ideal', chrono_mode='galv',
                 print_diagnostics=False, warn=True, device=0, context=None):
        """DRTBase.__init__ (hybdrt/models/drtbase.py:21-159): epsilon rule and the lookup tables."""
        if tau_basis_type != 'gaussian':
            raise NotImplementedError("only the default gaussian basis is on the hot path")
        if nu_basis_type != 'gaussian' or not normalize_dop:
            raise NotImplementedError("only the default gaussian, normalised distribution of phasances is built")
        if step_model != 'ideal' or chrono_mode != 'galv':
            raise NotImplementedError("only ideal galvanostatic steps are built")
        self.basis_nu = None if fixed_basis_nu is None else np.asarray(fixed_basis_nu, dtype=float)
        self.nu_epsilon = nu_epsilon
        if fixed_basis_tau is not None and tau_supergrid is not None:
            warnings.warn('If fixed_basis_tau is provided, tau_supergrid will be ignored')
        self.fixed_basis_tau = None if fixed_basis_tau is None else np.asarray(fixed_basis_tau, dtype=float)
        self.tau_supergrid = None if tau_supergrid is None else np.asarray(tau_supergrid, dtype=float)
        self.tau_basis_type = tau_basis_type
        self.tau_epsilon = tau_epsilon
        self.extend_basis_decades = extend_basis_decades
        self.fit_inductance, self.fit_ohmic, self.fit_capacitance, self.fit_dop = fit_inductance, fit_ohmic, bool(fit_capacitance), bool(fit_dop)
        self.frequency_precision = frequency_precision
        self.print_diagnostics, self.warn = print_diagnostics, warn
        self.device = device
        self._context = context          # optional private hipdrt context (own HIP stream): lets several
                                         # DRT instances keep batches in flight concurrently on one GPU
        if self.tau_epsilon is None:
            if self.fixed_basis_tau is not None:
                self.tau_epsilon = 1 / np.mean(np.diff(np.log(self.fixed_basis_tau)))
            elif self.tau_supergrid is not None:
                self.tau_epsilon = 1 / np.mean(np.diff(np.log(self.tau_supergrid)))
            elif basis_tau_ppd is not None:
                self.tau_epsilon = pp.get_epsilon_from_ppd(basis_tau_ppd)
        self.integrate_method = 'interp' if interpolate_integrals else 'trapz'
        # lookup abscissae (basis.py:653-657); the ordinates are produced on the device inside the plan
        self._wt_re = np.logspace(-2.7, 2.7, 2000)
        self._wt_im = np.logspace(-5.4, 5.4, 2000)
        self._plan = None
        self._plan_key = None
        self._last_batch = None
        self.basis_tau = None
        self.special_qp_params = {}
        self.fit_parameters = None
        self.qphb_params = None
        self.qphb_history = None
        self.cvx_result = None
        self.fit_kwargs = None
        self.fit_type = None
        self.coefficient_scale = 1.0
        self.impedance_scale = 1.0
        self.inductance_scale = None
        self.f_fit = []

    # ---- plan management (the counterpart of the reference's matrix recalc cache, drtbase.py:1008-1032) ----
    @property
    def interpolate_lookups(self):
        if self._plan is None or self.integrate_method != 'interp':
            return {'z_real': None, 'z_imag': None}
        p = self._plan
        return {'z_real': (p.log_wt_re, p.get('lut_z_re')), 'z_imag': (p.log_wt_im, p.get('lut_z_im'))}

    def _special_params(self):
        """drt1d.py:375-408 / drtbase.py:538-547 for an EIS fit."""
        sp = {}
        if self.fit_ohmic:
            sp['R_inf'] = {'index': len(sp), 'nonneg': True, 'size': 1}
        if self.fit_inductance:
            sp['inductance'] = {'index': len(sp), 'nonneg': True, 'size': 1}
        return sp

    def _get_plan(self, frequencies, opts, capacity):
        if self.fixed_basis_tau is not None:
            basis_tau = self.fixed_basis_tau
        else:
            basis_tau = pp.get_basis_tau(frequencies, None, None, tau_grid=self.tau_supergrid,
                                         extend_decades=self.extend_basis_decades)
        if self.tau_epsilon is None:
            self.tau_epsilon = 1 / np.mean(np.diff(np.log(basis_tau)))
        key = (np.asarray(frequencies).tobytes(), basis_tau.tobytes(), float(self.tau_epsilon), bytes(opts))
        if self._plan is not None and self._plan_key == key and self._plan.capacity >= capacity:
            return self._plan
        if self._plan is not None:
            self._plan.close()
        mode = _ffi.MODE_INTERP if self.integrate_method == 'interp' else _ffi.MODE_TRAPZ
        tpl_a = mat1d.impedance_matrix_is_toeplitz(frequencies, basis_tau, self.frequency_precision)
        tpl_m = is_uniform(np.log(basis_tau))
        ctx = self._context if self._context is not None else _ffi.get_context(self.device)
        self._plan = _ffi.Plan(ctx, frequencies, basis_tau, self.tau_epsilon,
                               wt_re=self._wt_re, wt_im=self._wt_im, mode=mode, toeplitz_a=tpl_a, toeplitz_m=tpl_m,
                               opts=opts, capacity=capacity)
        self._plan_key = key
        self.basis_tau = basis_tau
        sub = getattr(self, 'plan_subbatches', None)      # None: the library's choice (hipdrt_plan_set_subbatches(0))
        if sub is not None:
            self._plan.set_subbatches(sub)
        if getattr(self, '_luts_installed', False) and getattr(self, '_lut_key', None) == float(self.tau_epsilon):
            (_, z_re), (_, z_im) = self._luts['z']          # tables received from another rank (install_lookup_tables)
            self._plan.set_lookup(z_re, z_im)
        return self._plan

    def lookup_tables(self):
        """(z_re, z_im, response) ordinates of the three lookup tables of this instance's epsilon (drtbase.py:138-156),
        built on the device (PreparedFitMixin._lookups)."""
        ctx = self._context if self._context is not None else _ffi.get_context(self.device)
        luts = self._lookups(ctx)
        return luts['z'][0][1], luts['z'][1][1], luts['response'][1]

    def install_lookup_tables(self, z_re, z_im, response):
        """Use tables built elsewhere (rank 0 of a sharded map, mapping.share_lookup_tables) instead of building them:
        prepared-matrix fits read them from here, EIS plans receive them right after they are created."""
        td = np.logspace(-6, 2, 2000)
        self._luts = dict(z=((np.log(self._wt_re), np.asarray(z_re, dtype=float)), (np.log(self._wt_im), np.asarray(z_im, dtype=float))),
                          response=(np.log(td), np.asarray(response, dtype=float)))
        self._lut_key = float(self.tau_epsilon)
        self._luts_installed = True
        if self._plan is not None and hasattr(self._plan, 'set_lookup'):
            self._plan.set_lookup(z_re, z_im)

    def _make_opts(self, fit_kw):
        kw = dict(_FIT_KW_DEFAULTS)
        hypers = qphb.get_default_hypers(bool(fit_kw.get('eff_hp', True)), self.fit_dop, 'gaussian')
        for key, val in fit_kw.items():
            if key in kw:
                kw[key] = val
            elif key in hypers:
                hypers[key] = val
            else:
                raise ValueError(f'Invalid keyword argument {key}')     # drt1d.py:415-419
        if kw['penalty_type'] != 'integral':
            raise NotImplementedError("penalty_type 'discrete' is deprecated in the reference and not built")
        if (hypers['iw_alpha'] is None) != (hypers['iw_beta'] is None):
            raise ValueError('iw_alpha and iw_beta must be given together')
        if kw['eis_error_structure'] not in (None, 'uniform'):
            raise ValueError(f"Invalid eis_error_structure {kw['eis_error_structure']}")
        o = _ffi.default_fit_opts()
        o.rp_scale = float(hypers['rp_scale'])
        for name in ('derivative_weights', 'sigma_ds', 's_alpha', 's_0', 'rho_alpha', 'rho_0'):
            vals = np.broadcast_to(np.asarray(hypers[name], dtype=float), (3,))
            for k in range(3):
                getattr(o, name)[k] = float(vals[k])
        o.l1_lambda_0, o.l2_lambda_0 = float(hypers['l1_lambda_0']), float(hypers['l2_lambda_0'])
        # optional branches of the weight estimation (None <-> -1)
        o.outlier_p = -1.0 if hypers['outlier_p'] is None else float(hypers['outlier_p'])
        o.iw_alpha = -1.0 if hypers['iw_alpha'] is None else float(hypers['iw_alpha'])
        o.iw_beta = -1.0 if hypers['iw_beta'] is None else float(hypers['iw_beta'])
        o.iw_l1_lambda_0, o.iw_l2_lambda_0 = float(kw['iw_l1_lambda_0']), float(kw['iw_l2_lambda_0'])
        o.ohmic_penalty, o.inductance_penalty = float(kw['ohmic_penalty']), float(kw['inductance_penalty'])
        o.inductance_scale = float(kw['inductance_scale'])
        o.eis_vmm_epsilon, o.eis_reim_cor = float(kw['eis_vmm_epsilon']), float(kw['eis_reim_cor'])
        o.xtol, o.max_iter = float(kw['xtol']), int(kw['max_iter'])
        o.nonneg, o.scale_data = int(bool(kw['nonneg'])), int(bool(kw['scale_data']))
        o.fit_ohmic, o.fit_inductance = int(self.fit_ohmic), int(self.fit_inductance)
        o.eis_error_uniform = int(kw['eis_error_structure'] == 'uniform')
        o.update_scale = int(bool(kw['update_scale']))
        o.eff_hp = int(bool(kw['eff_hp']))
        return o, hypers, kw

    # ---- the fits ------------------------------------------------------------------------------------------
    def _qphb_fit_core(self, times, i_signal, v_signal, frequencies, z, **kw):
        """DRT._qphb_fit_core(times, i_signal, v_signal, frequencies, z, **fit_kw) (drt1d.py:102-137), the call
        DRTMD.fit_observation makes as ``drt1d._qphb_fit_core(*chrono_data, *eis_data, **fit_kw)`` (drtmd.py:253): which
        data are None selects the EIS, chrono or joint fit, with the keyword names of _qphb_fit_core itself."""
        has_chrono = times is not None
        has_eis = frequencies is not None
        if not has_chrono and not has_eis:
            raise ValueError('At least one of (times, i_signal, v_signal) and (frequencies, z) must be provided')
        if has_chrono and (i_signal is None or v_signal is None):
            raise ValueError('times, i_signal and v_signal must be provided together')     # utils.validation.check_chrono_data
        if has_eis and z is None:
            raise ValueError('frequencies and z must be provided together')                # utils.validation.check_eis_data
        if not has_chrono:
            return self.fit_eis(frequencies, z, **kw)
        if not has_eis:
            kw = dict(kw)
            for core, own in (('chrono_error_structure', 'error_structure'), ('chrono_vmm_epsilon', 'vmm_epsilon')):
                if core in kw:
                    kw[own] = kw.pop(core)
            for eis_only in ('eis_error_structure', 'eis_vmm_epsilon', 'eis_reim_cor'):
                kw.pop(eis_only, None)
            return self.fit_chrono(times, i_signal, v_signal, **kw)
        return self.fit_hybrid(times, i_signal, v_signal, frequencies, z, **kw)

    def fit_eis(self, frequencies, z, **kw):
        """DRT.fit_eis (drt1d.py:1215-1241) -> _qphb_fit_core (102-1104) for one spectrum."""
        frequencies = np.asarray(frequencies, dtype=float)
        z = np.asarray(z, dtype=complex)
        if len(frequencies) != len(z):
            raise ValueError('Length of frequencies and z must be equal')    # utils/validation.check_eis_data
        kw = dict(kw)
        for old, new in (('error_structure', 'eis_error_structure'), ('vmm_epsilon', 'eis_vmm_epsilon'),
                         ('vmm_reim_cor', 'eis_reim_cor')):     # fit_eis's own keyword names (drt1d.py:1215-1241)
            if old in kw:
                kw[new] = kw.pop(old)
        if self.fit_dop or self.fit_capacitance or kw.get('solve_rp') or kw.get('remove_outliers') \
                or kw.get('remove_extremes') or kw.get('neg_allowed_tau_range') is not None \
                or kw.get('series_neg'):   # prepared-matrix plan
            return self._store_single(*self._fit_prepared([(None, None, None, frequencies, z)], kw, history_of=0),
                                      'qphb_eis')
        res = self._fit(frequencies, z[None, :], kw, history_of=0)
        b = 0
        fp = {'x': res['fit_x'][b], 'R_inf': res['R_inf'][b] if self.fit_ohmic else 0,
              'inductance': res['inductance'][b] if self.fit_inductance else 0, 'C_inv': 0,
              'v_sigma_tot': None, 'v_sigma_res': None, 'z_sigma_tot': res['z_sigma_tot'][b], 'vz_offset_eps': 1,
              'p_matrix': self._plan.p_matrix(b), 'q_vector': res['q_vector'][b]}
        if res['status'][b] < 0:
            raise ValueError("Rank(A) < p or Rank([P; A; G]) < n")          # cvxopt's error at the QP start point
        if res['status'][b] == 1 and self.warn:
            warnings.warn(f"Solution did not converge within {self.fit_kwargs['max_iter']} iterations. "
                          f"This is usually not an issue.")
        self.fit_parameters = fp
        self.coefficient_scale = self.impedance_scale = float(res['coefficient_scale'][b])
        hist = self._plan.history()
        self.qphb_history = [{'x': hist['x'][i], 'rho_vector': hist['rho'][i], 'weights': hist['weights'][i]}
                             for i in range(len(hist['x']))]
        self.qphb_params = {'weights': res['weights'][b], 'true_weights': res['weights'][b],
                            'rho_vector': res['rho'][b], 's_vectors': list(res['s_vectors'][b]),
                            'p_matrix': fp['p_matrix'], 'q_vector': fp['q_vector'], 'rm': self._plan.get('rm'),
                            'vmm': self._plan.get('vmm'), 'num_eis': len(frequencies), 'num_chrono': 0,
                            'qp_iterations': hist['qp_iterations'], 'outer_iterations': int(res['outer_iters'][b]),
                            'est_weights': self._plan.get('est_weights')[b], 'rv': self._plan.get('rv')[b]}
        self.cvx_result = {'x': res['x'][b]}
        self.fit_type = 'qphb_eis'
        return fp

    def fit_eis_batch(self, frequencies, z_batch, **kw):
        """B spectra on one frequency grid, fitted concurrently (the reference's DRTMD loop calls
        _qphb_fit_core once per observation, mapping/drtmd.py:245-319).  Returns a dict of arrays."""
        frequencies = np.asarray(frequencies, dtype=float)
        z_batch = np.asarray(z_batch, dtype=complex)
        if z_batch.ndim != 2 or z_batch.shape[1] != len(frequencies):
            raise ValueError('z_batch must have shape (B, len(frequencies))')
        if self.fit_dop or self.fit_capacitance or kw.get('solve_rp'):
            return self._fit_prepared_batch([(None, None, None, frequencies, zb) for zb in z_batch], kw)
        return self._fit(frequencies, z_batch, kw, history_of=-1)

    # staged form: inputs made resident in HBM once, the fit launched separately (what bench.py times)
    def stage_batch(self, frequencies, z_batch, history_of=-1, **kw):
        frequencies = np.asarray(frequencies, dtype=float)
        z_batch = np.asarray(z_batch, dtype=complex)
        opts, hypers, fkw = self._make_opts(kw)
        plan = self._get_plan(frequencies, opts, z_batch.shape[0])
        self.special_qp_params = self._special_params()
        self.inductance_scale = fkw['inductance_scale']
        self.fit_kwargs = dict(hypers, **fkw)
        self.f_fit = frequencies
        plan.record_history(history_of)
        wf = fkw['weight_factor']
        if np.ndim(wf) > 0:       # vector-valued weight_factor (drt1d.py:889-901): one factor per data row
            plan.set_weight_factors(1.0, np.asarray(wf, dtype=float), late=True)
        else:
            plan.set_weight_factors(wf)
        plan.upload(z_batch)
        self._last_batch = z_batch.shape[0]
        return plan

    def fit_staged(self):
        self._plan.fit()

    def collect_staged(self):
        plan = self._plan
        frequencies = self.f_fit
        if getattr(self, 'collect_fields', None) == 'map':
            # a map keeps, per observation, the distribution, the special parameters, llh / rss and the counts (drtmd.py:245-301):
            # the solution in scaled units, weights, rho, s vectors and q stay on the device (mapping.fit_observations_sharded)
            res = plan.download(lean=True)
        else:
            res = plan.download(s_vectors=True)
            nf = len(frequencies)
            sigma = 1.0 / res['weights']
            res['z_sigma_tot'] = (sigma[:, :nf] + 1j * sigma[:, nf:]) * res['coefficient_scale'][:, None]
        res['basis_tau'] = self.basis_tau
        res['timings_ms'], res['launches'] = plan.timings()
        return res

    # ---- warm restarts of the device loop (drt1d.py:1270-1365) and the candidate generators on top (1497-1632) ---
    def continue_from_init(self, x_init=None, rho_vector=None, s_vectors=None, weights=None, weight_factor=1,
                           xtol=1e-2, max_iter=10, min_iter=2, history_of=-1, dop_rho_vector=None, **kw):
        """DRT._continue_from_init for the last fitted batch: the outer loop re-entered on the device from the given
        state (arrays with a leading batch axis; None = the state left by the previous call) with ``kw`` updating the
        hyper-parameters (e.g. s_0, l2_lambda_0).  est_weights, xmx norms and the data scale stay as fitted.
        Returns the same dict of arrays as fit_eis_batch (outer_iters = iterations of this call)."""
        if isinstance(self._plan, _ffi.PreparedPlan):         # chrono / joint fits, DOP: the same loop on the prepared plan
            return self._continue_prepared(x_init=x_init, rho_vector=rho_vector, s_vectors=s_vectors, weights=weights,
                                           dop_rho_vector=dop_rho_vector, weight_factor=weight_factor, xtol=xtol, max_iter=max_iter, min_iter=min_iter,
                                           history_of=history_of, **kw)
        if self._plan is None or self._last_batch is None:
            raise Exception('continue_from_init needs a finished qphb fit')
        fit_kw = dict(self.fit_kwargs)
        fit_kw.update(kw)
        fit_kw.update(xtol=xtol, max_iter=max_iter)
        opts, _, _ = self._make_opts(fit_kw)
        plan = self._plan
        plan.set_state(x=x_init, rho=rho_vector, s=s_vectors, weights=weights)
        plan.record_history(history_of)
        plan.continue_fit(opts, weight_factor=weight_factor, min_iter=min_iter)
        res = self.collect_staged()
        if history_of >= 0:
            res['history'] = plan.history()
        return res

    def _candidate_baseline(self):
        """What the reference's candidate generators re-read from the finished fit before their first warm restart
        (drt1d.py:1517-1525, 1587-1594): x of the last recorded iterate, rho / dop_rho and the (scaled) weights of
        qphb_params -- NOT the s vectors, which its shallow list copies let earlier warm restarts update in place.  Single
        fits only (a batch fit keeps no per-spectrum qphb_params: its restarts go on from the state on the device)."""
        qp, hist = getattr(self, 'qphb_params', None), getattr(self, 'qphb_history', None)
        if not qp or not hist or self._plan.B != 1 or len(qp['weights']) != self._plan.m:
            return {}
        base = dict(x_init=np.asarray(hist[-1]['x'])[None, :], rho_vector=np.asarray(qp['rho_vector'])[None, :],
                    weights=np.asarray(qp['weights'])[None, :])
        if qp.get('dop_rho_vector') is not None:
            base['dop_rho_vector'] = np.asarray(qp['dop_rho_vector'])[None, :]
        return base

    def generate_candidates_s0(self, multiplier, steps, xtol=1e-2, max_iter=10, history_of=-1):
        """DRT._generate_candidates_s0 (drt1d.py:1497-1565) for the last fit (EIS, chrono or joint; single or batch): step i
        restarts with s_0 * multiplier^i, l2_lambda_0 / multiplier^i and (multiplier > 1) the baseline s vectors
        scaled by multiplier^i; the first step from the fit's x / rho / weights, later ones from their predecessor's.
        Returns the list of per-step result dicts."""
        base = self._collect_prepared() if isinstance(self._plan, _ffi.PreparedPlan) else self.collect_staged()
        s_base = base['s_vectors'].copy()
        s_in = s_base.copy()
        s_0 = np.broadcast_to(np.asarray(self.fit_kwargs['s_0'], dtype=float), (3,)).copy()
        out = []
        start = self._candidate_baseline()
        for i in range(1, steps + 1):
            f = multiplier ** i
            s_in = s_base * f if multiplier > 1 else s_in * multiplier
            res = self.continue_from_init(s_vectors=s_in, xtol=xtol, max_iter=max_iter, history_of=history_of,
                                          s_0=s_0 * f, l2_lambda_0=self.fit_kwargs['l2_lambda_0'] / f, **start)
            start = {}
            s_in = res['s_vectors'].copy()
            out.append(res)
        return out

    def generate_candidates_weights(self, multiplier, steps, xtol=1e-2, max_iter=10, history_of=-1):
        """DRT._generate_candidates_weights (drt1d.py:1567-1632): step i restarts with weight_factor = multiplier^i.
        As in the reference (whose shallow list copy lets iterate_qphb update the stored s vectors in place) every
        step starts from the s vectors the previous step ended with."""
        out = []
        start = self._candidate_baseline()
        for i in range(1, steps + 1):
            out.append(self.continue_from_init(weight_factor=multiplier ** i, xtol=xtol, max_iter=max_iter,
                                               history_of=history_of, **start))
            start = {}
        return out

    def evaluate_obs_llh_rss_batch(self, llh_kw=None, rss_kw=None):
        """(DRT.evaluate_llh(**llh_kw), DRT.evaluate_rss(**rss_kw)) (drt1d.py:4433-4496; x = the last iterate) for every
        spectrum of the last fitted batch -- what DRTMD.fit_observation stores as obs_llh / obs_rss (drtmd.py:259-260).
        Keys as upstream: ``weights`` (None = the fit's est_weights, 'uniform' = per-domain means of them, a scalar),
        ``normalize`` (divide by the number of data rows), and for the likelihood ``marginalize_weights``, ``alpha_0``,
        ``beta_0``.  Residuals and all sums on the device."""
        from scipy.special import loggamma
        llh_kw, rss_kw = dict(llh_kw or {}), dict(rss_kw or {})
        bad = (set(llh_kw) - {'weights', 'normalize', 'marginalize_weights', 'alpha_0', 'beta_0', 'subtract_background'}) | \
              (set(rss_kw) - {'weights', 'normalize'})
        if bad:
            raise TypeError(f"unexpected keyword(s) {sorted(bad)}")
        m = self._plan.m
        terms = {}

        def sums(weights):
            key = weights if (weights is None or isinstance(weights, str)) else float(weights)
            if key not in terms:
                terms[key] = self._plan.llh_terms(stored=True, weights=weights)
            return terms[key]

        rss_l, slw = sums(llh_kw.get('weights'))
        alpha_0, beta_0 = llh_kw.get('alpha_0', 2), llh_kw.get('beta_0', 1)
        if llh_kw.get('marginalize_weights', True):
            alpha_n = alpha_0 - 1 + m / 2
            llh = alpha_0 * np.log(beta_0) - alpha_n * np.log(beta_0 + 0.5 * rss_l) + loggamma(alpha_n) - loggamma(alpha_0)
        else:
            llh = -0.5 * rss_l
        llh = llh + slw
        if llh_kw.get('normalize', False):
            llh = llh / m
        rss = sums(rss_kw.get('weights'))[0].copy()
        if rss_kw.get('normalize', False):
            rss /= m
        return llh, rss

    def evaluate_step_llh_batch(self, alpha_0=2, beta_0=1):
        """evaluate_llh(weights=estimate_weights(x), x) (drt1d.py:2618-2622) for the current x of every spectrum of
        the batch: residuals, re-estimated weights and both sums on the device, the two lgamma constants here."""
        from scipy.special import loggamma
        rss, slw = self._plan.llh_terms()
        alpha_n = alpha_0 - 1 + self._plan.m / 2
        beta_n = beta_0 + 0.5 * rss
        return alpha_0 * np.log(beta_0) - alpha_n * np.log(beta_n) + loggamma(alpha_n) - loggamma(alpha_0) + slw

    def pfrt_fit_eis_batch(self, frequencies, z_batch, factors=None, max_iter_per_step=10, max_init_iter=20,
                           xtol=1e-2, nonneg=True, after_init=None, **kw):
        """DRT.pfrt_fit_eis (drt1d.py:2558-2690) for B spectra at once: a full fit at the first regularisation factor
        (s_0 * f, l2_lambda_0 / f), then one warm restart per further factor on the device.  Returns
        {'factors', 'step_x' (S, B, n) scaled-space solutions, 'step_llh' (S, B), 'step_iters' (S, B)}."""
        base = qphb.get_default_hypers(True, False, 'gaussian')
        base.update({k: v for k, v in kw.items() if k in base})
        if factors is None:
            factors = np.logspace(-1, 1, 11)
        s_0 = np.broadcast_to(np.asarray(base['s_0'], dtype=float), (3,))

        def step_hypers(f):
            return dict(s_0=s_0 * f, l2_lambda_0=base['l2_lambda_0'] / f)

        init_kw = dict(kw)
        init_kw.update(step_hypers(factors[0]))
        res = self.fit_eis_batch(frequencies, z_batch, nonneg=nonneg, max_iter=max_init_iter, xtol=xtol, **init_kw)
        step_x, step_llh, step_iters = [res['x'].copy()], [self.evaluate_step_llh_batch()], [res['outer_iters'].copy()]
        status = np.array(res['status']).copy()
        if after_init is not None:          # (what DRTMD reads from the FIRST step's fit: its P matrix, llh / rss -- mapping)
            after_init(res)
        for f in factors[1:]:
            res = self.continue_from_init(xtol=xtol, max_iter=max_iter_per_step, **step_hypers(f))
            step_x.append(res['x'].copy())
            step_llh.append(self.evaluate_step_llh_batch())
            step_iters.append(res['outer_iters'].copy())
            status = combine_status(status, res['status'])
        self.pfrt_result = {'factors': np.asarray(factors), 'step_x': np.array(step_x), 'step_llh': np.array(step_llh),
                            'step_iters': np.array(step_iters), 'status': status,
                            'coefficient_scale': res['coefficient_scale'], 'basis_tau': res['basis_tau']}
        return self.pfrt_result

    # ---- what DRTMD takes from a finished fit (mapping/drtmd.py:258-279) ----------------------------------------
    def _signed_basis(self, bm, sign):
        """series_neg fits carry 2 ntau coefficients [positive copy | negative copy]: the evaluation rows of
        estimate_distribution_cov's three cases (drt1d.py:3090-3103) as ONE matrix over both copies -- sign=1 the positive
        block, -1 the negative one, 0 their difference (B, -B): B S++ B' + B S-- B' - B (S+- + S-+) B'"""
        if not self.series_neg:
            return bm
        zero = np.zeros_like(bm)
        if sign == 1:
            return np.hstack([bm, zero])
        if sign == -1:
            return np.hstack([zero, bm])
        if sign == 0:
            return np.hstack([bm, -bm])
        raise ValueError('sign must be 1, -1 or 0')

    def estimate_distribution_var_batch(self, tau=None, ppd=20, extend_var=False, sign=1):
        """Diagonal of DRT.estimate_distribution_cov (drt1d.py:3063-3151; order 0, no normalisation) for every
        spectrum of the last fitted batch: diag(B P^-1 B') coefficient_scale^2, computed on the device from the
        Cholesky factor of each final P.  Returns (var (B, len(tau)), ok (B,) bool); ``extend_var`` applies the
        reference's clamp outside the measured tau range (drt1d.py:3126-3143)."""
        from ..matrices import basis
        prepared = isinstance(self._plan, _ffi.PreparedPlan)
        if self._plan is None or (self._last_batch is None and not prepared):
            raise Exception('Parameter covariance estimation is only available for qphb fits')
        if tau is None:
            tau = self.get_tau_eval(ppd)
        tau = np.asarray(tau, dtype=float)
        bm = basis.construct_func_eval_matrix(np.log(self.basis_tau), np.log(tau), self.tau_basis_type,
                                              epsilon=self.tau_epsilon, order=0)
        bm = self._signed_basis(bm, sign)
        if prepared:
            # the device loop of a prepared plan runs at unit scale: estimate_param_cov's coefficient_scale^2 is applied here
            var, status = self._plan.distribution_var(bm, self._plan.batch)
            preps = self._last_prepared[0] if getattr(self, '_last_prepared', None) and \
                len(self._last_prepared[0]) == self._plan.batch else [self._prep]
            var = var * np.array([pr['coefficient_scale'] for pr in preps])[:, None] ** 2
        else:
            var, status = self._plan.distribution_var(bm, self._last_batch)
        if extend_var:
            if prepared:
                pr = preps[0]                       # (the members of a prepared batch share their sampling grids)
                t_left, t_right = pp.get_tau_lim(pr['frequencies'], pr.get('sample_times'), pr.get('nonconsec_step_times'))
            else:
                t_left, t_right = 1 / (2 * np.pi * np.max(self.f_fit)), 1 / (2 * np.pi * np.min(self.f_fit))
            left_index = int(np.argmin(np.abs(tau - t_left))) + 1
            right_index = int(np.argmin(np.abs(tau - t_right)))
            var[:, :left_index] = np.maximum(var[:, :left_index], var[:, left_index][:, None])
            var[:, right_index:] = np.maximum(var[:, right_index:], var[:, right_index][:, None])
        return var, status == 0

    def estimate_param_var_batch(self):
        """np.diag(DRT.estimate_param_cov()) (drt1d.py:4116-4138) for every spectrum of the last fitted batch, from the
        Cholesky factor of each final P on the device.  Returns (var (B, n), ok (B,) bool)."""
        prepared = isinstance(self._plan, _ffi.PreparedPlan)
        if self._plan is None or (self._last_batch is None and not prepared):
            raise Exception('Parameter covariance estimation is only available for qphb fits')
        if prepared:     # the prepared loop runs at unit scale: coefficient_scale^2 of estimate_param_cov applied here
            var, status = self._plan.param_var(self._plan.batch)
            preps = self._last_prepared[0] if getattr(self, '_last_prepared', None) and \
                len(self._last_prepared[0]) == self._plan.batch else [self._prep]
            return var * np.array([pr['coefficient_scale'] for pr in preps])[:, None] ** 2, status == 0
        var, status = self._plan.param_var(self._last_batch)
        return var, status == 0

    def _cov_scale(self, b):
        """(coefficient_scale of fitted measurement b where the device loop ran at unit scale, else 1; its prep or None)"""
        if isinstance(self._plan, _ffi.PreparedPlan):
            preps = self._last_prepared[0] if getattr(self, '_last_prepared', None) and \
                len(self._last_prepared[0]) == self._plan.batch else [self._prep]
            return preps[b]['coefficient_scale'], preps[b]
        return 1.0, None

    def estimate_param_cov(self, b=0):
        """DRT.estimate_param_cov (drt1d.py:4116-4138): inv(P) * coefficient_scale^2 with the DOP block rescaled by
        dop_scale_vector, from the Cholesky factor of the final P on the device (hipdrt_plan_param_cov); ``b`` picks a
        member of the last batch.  None (with upstream's warning) when P is not positive definite."""
        if self._plan is None or (self._last_batch is None and not isinstance(self._plan, _ffi.PreparedPlan)):
            raise Exception('Parameter covariance estimation is only available for qphb fits')
        cov, ok = self._plan.param_cov(b)
        if not ok:
            warnings.warn('Singular P matrix - could not obtain covariance estimate')
            return None
        cs, prep = self._cov_scale(b)
        cov = cov * cs ** 2
        if prep is not None and prep['dop']:
            a, e = prep['dop']
            cov[:, a:e] *= prep['dop_scale_vector'][None, :]
            cov[a:e, :] *= prep['dop_scale_vector'][:, None]
        return cov

    def estimate_distribution_cov(self, tau=None, ppd=20, extend_var=False, var_floor=0.0, b=0, sign=1):
        """DRT.estimate_distribution_cov (drt1d.py:3063-3151; order 0, sign 1, no normalisation): basis_matrix @ x_cov @
        basis_matrix.T of the DRT block, formed on the device (hipdrt_plan_distribution_cov), then upstream's ``extend_var``
        clamp of the diagonal outside the measured tau range (3126-3143) and ``var_floor``."""
        from ..matrices import basis
        if self._plan is None or (self._last_batch is None and not isinstance(self._plan, _ffi.PreparedPlan)):
            raise Exception('Parameter covariance estimation is only available for qphb fits')
        if tau is None:
            tau = self.get_tau_eval(ppd)
        tau = np.asarray(tau, dtype=float)
        bm = basis.construct_func_eval_matrix(np.log(self.basis_tau), np.log(tau), self.tau_basis_type,
                                              epsilon=self.tau_epsilon, order=0)
        cov, ok = self._plan.distribution_cov(self._signed_basis(bm, sign), b)
        if not ok:
            warnings.warn('Singular P matrix - could not obtain covariance estimate')
            return None
        cs, prep = self._cov_scale(b)
        cov = cov * cs ** 2
        if extend_var:
            if prep is not None:
                t_left, t_right = pp.get_tau_lim(prep['frequencies'], prep.get('sample_times'), prep.get('nonconsec_step_times'))
            else:
                t_left, t_right = 1 / (2 * np.pi * np.max(self.f_fit)), 1 / (2 * np.pi * np.min(self.f_fit))
            left_index = int(np.argmin(np.abs(tau - t_left))) + 1
            right_index = int(np.argmin(np.abs(tau - t_right)))
            var = np.diag(cov).copy()
            var[:left_index] = np.maximum(var[:left_index], var[left_index])
            var[right_index:] = np.maximum(var[right_index:], var[right_index])
            cov[np.diag_indices(cov.shape[0])] = var
        if var_floor > 0:
            var = np.diag(cov).copy()
            var[var < var_floor] = var_floor
            np.fill_diagonal(cov, var)
        return cov

    def get_tau_eval(self, ppd):
        """drtbase.get_tau_eval (drtbase.py:263-285): one decade beyond the basis grid on each side."""
        basis_tau = self.basis_tau
        log_min, log_max = np.log10(np.min(basis_tau)) - 1, np.log10(np.max(basis_tau)) + 1
        return np.logspace(log_min, log_max, int((log_max - log_min) * ppd) + 1)

    series_neg = False

    def _llh_weights(self, weights):
        """the `weights` argument of evaluate_rss / evaluate_llh (drt1d.py:4434-4443, 4459-4472)"""
        est = self.qphb_params['est_weights']
        if weights is None:
            return est
        if isinstance(weights, str):
            if weights != 'uniform':
                raise ValueError(f"weights must be None, 'uniform', a scalar or an array, got {weights!r}")
            nc = int(self.qphb_params.get('num_chrono', 0) or 0)
            w = np.empty(len(est))
            if nc:
                w[:nc] = np.mean(est[:nc])
            w[nc:] = np.mean(est[nc:])
            return w
        if np.isscalar(weights):
            return np.ones_like(est) * weights
        w = np.asarray(weights, dtype=float)
        if w.shape != est.shape:
            raise ValueError('Expected weights array of shape {}, but received shape {}'.format(est.shape, w.shape))
        return w

    def evaluate_rss(self, weights=None, x=None, normalize=False):
        """drt1d.evaluate_rss (drt1d.py:4433-4455) -> qphb.evaluate_rss (qphb.py:1347-1352)."""
        w = self._llh_weights(weights)
        x = self.qphb_history[-1]['x'] if x is None else x
        rm, rv = self.qphb_params['rm'], self.qphb_params['rv']
        wrm, wrv = w[:, None] * rm, w * rv
        rss = x @ wrm.T @ wrm @ x - 2 * wrv.T @ wrm @ x + wrv.T @ wrv
        return rss / len(rv) if normalize else rss

    def evaluate_llh(self, weights=None, x=None, marginalize_weights=True, alpha_0=2, beta_0=1, normalize=False):
        """drt1d.evaluate_llh (drt1d.py:4457-4496) -> qphb.evaluate_llh (qphb.py:1355-1377)."""
        from scipy.special import loggamma
        w = self._llh_weights(weights)
        rss = self.evaluate_rss(w, x)
        if marginalize_weights:
            alpha_n = alpha_0 - 1 + len(w) / 2
            beta_n = beta_0 + 0.5 * rss
            llh = alpha_0 * np.log(beta_0) - alpha_n * np.log(beta_n) + loggamma(alpha_n) - loggamma(alpha_0)
        else:
            llh = -0.5 * rss
        llh = llh + np.sum(np.log(w))
        return llh / len(w) if normalize else llh

    def _fit(self, frequencies, z_batch, kw, history_of):
        self.stage_batch(frequencies, z_batch, history_of=history_of, **kw)
        self.fit_staged()
        return self.collect_staged()
