"""ctypes binding of libhipdrt.so (C-ABI declared in include/hipdrt.h).  numpy in, numpy out.

There is deliberately no CPU fallback: if the shared library is missing, or no gfx950 device is visible,
every compute entry point raises ``HipDrtError``.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HIPDRT_LIB", os.path.join(_HERE, "libhipdrt.so"))

MODE_INTERP, MODE_TRAPZ = 0, 1
RESPONSE_POT, RESPONSE_EXPDECAY = 0, 1
QP_OPTIMAL, QP_MAXITER, QP_SINGULAR_LATE, QP_SINGULAR, QP_ABORTED = 0, 1, 2, -1, -2


class HipDrtError(RuntimeError):
    pass


class QpOpts(C.Structure):
    _fields_ = [("abstol", C.c_double), ("reltol", C.c_double), ("feastol", C.c_double), ("maxiters", C.c_int)]


class FitOpts(C.Structure):
    _fields_ = [
        ("rp_scale", C.c_double), ("derivative_weights", C.c_double * 3), ("sigma_ds", C.c_double * 3),
        ("l1_lambda_0", C.c_double), ("l2_lambda_0", C.c_double), ("s_alpha", C.c_double * 3),
        ("s_0", C.c_double * 3), ("rho_alpha", C.c_double * 3), ("rho_0", C.c_double * 3),
        ("iw_l1_lambda_0", C.c_double), ("iw_l2_lambda_0", C.c_double),
        ("ohmic_penalty", C.c_double), ("inductance_penalty", C.c_double), ("inductance_scale", C.c_double),
        ("eis_vmm_epsilon", C.c_double), ("eis_reim_cor", C.c_double), ("xtol", C.c_double),
        ("max_iter", C.c_int), ("nonneg", C.c_int), ("scale_data", C.c_int), ("fit_ohmic", C.c_int),
        ("fit_inductance", C.c_int), ("eis_error_uniform", C.c_int), ("update_scale", C.c_int),
        ("eff_hp", C.c_int),
        ("outlier_p", C.c_double), ("iw_alpha", C.c_double), ("iw_beta", C.c_double), ("qp", QpOpts),
    ]


class PreparedDesc(C.Structure):
    """hipdrt_prepared_desc (include/hipdrt.h)"""
    _fields_ = [
        ("m", C.c_int), ("n", C.c_int), ("ns", C.c_int), ("dop_start", C.c_int), ("dop_size", C.c_int),
        ("vz_index", C.c_int), ("vb_start", C.c_int), ("vb_size", C.c_int), ("num_chrono", C.c_int),
        ("toeplitz_m", C.c_int), ("chrono_vmm_uniform", C.c_int), ("basis_area", C.c_double), ("init_weights_separately", C.c_int),
        ("weight_method", C.c_int), ("fixed_chrono_factor", C.c_double), ("fixed_eis_factor", C.c_double),
        ("dop_l2_lambda_0", C.c_double),
        ("dop_derivative_weights", C.c_double * 3),
        ("dop_s_alpha", C.c_double * 3), ("dop_rho_alpha", C.c_double * 3), ("dop_s_0", C.c_double * 3),
        ("dop_rho_0", C.c_double * 3),
    ]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
class IterateState(C.Structure):
    """hipdrt_iterate_state (include/hipdrt.h): the arrays iterate_qphb takes; NULL keeps the device value"""
    _fields_ = [(name, C.POINTER(C.c_double)) for name in
                ("x_in", "s_vectors", "rho", "dop_rho", "weights", "est_weights", "xmx_norms", "dop_xmx_norms")]


_vp = C.c_void_p

# name -> argtypes (all return int unless listed in _RESTYPES).  Mirrors include/hipdrt.h one-to-one;
# tests/test_cabi_symbols.py checks the header and this table against the built library.
SIGNATURES = {
    "hipdrt_create": [C.c_int, C.POINTER(_vp)],
    "hipdrt_destroy": [_vp],
    "hipdrt_last_error": [],
    "hipdrt_stream": [_vp],
    "hipdrt_synchronize": [_vp],
    "hipdrt_device_info": [_vp, C.c_char_p, C.c_int, _ip, C.POINTER(C.c_longlong)],
    "hipdrt_impedance_lookup": [_vp, C.c_double, C.c_int, C.c_int, _dp, _dp, _dp, _dp],
    "hipdrt_impedance_matrix": [_vp, C.c_int, C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_double,
                                C.c_int, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp],
    "hipdrt_impedance_matrix_dev": [_vp, C.c_int, C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_double,
                                    C.c_int, _dp, _dp, _dp, _dp, C.c_int, _vp, _vp, C.c_int, C.POINTER(C.c_float)],
    "hipdrt_phasor_z_matrix": [_vp, _dp, C.c_int, _dp, C.c_int, C.c_double, _dp, _dp],
    "hipdrt_phasor_v_matrix": [_vp, _dp, C.c_int, _dp, C.c_int, C.c_double, _dp, _dp, C.c_int, _dp, _dp],
    "hipdrt_chrono_var_matrix": [_vp, _dp, C.c_int, _ip, C.c_int, C.c_double, C.c_int, _dp],
    "hipdrt_response_lookup": [_vp, C.c_double, C.c_int, C.c_int, _dp, _dp],
    "hipdrt_response_matrix": [_vp, _dp, C.c_int, _dp, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_int, _dp, _dp,
                               C.c_int, _dp, _dp],
    "hipdrt_plan_bytes_per_spectrum": [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong)],
    "hipdrt_response_matrix_variant": [_vp, _dp, C.c_int, _dp, C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_int,
                                       _dp, _dp],
    "hipdrt_nonuniform_gaussian_filter1d": [_vp, _dp, C.c_int, _dp, _ip, C.c_int, _ip, _dp, C.c_int, _dp, _dp, C.c_longlong,
                                            _ip, _ip, _dp],
    "hipdrt_penalty_matrices": [_vp, _dp, C.c_int, C.c_double, C.c_int, _dp, _dp, _dp],
    "hipdrt_eis_var_matrix": [_vp, _dp, C.c_int, C.c_double, C.c_double, C.c_int, _dp],
    "hipdrt_qp_batch": [_vp, C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_int, _dp, C.POINTER(QpOpts), _dp, _ip, _dp, _ip],
    "hipdrt_qp_profile": [_vp, C.POINTER(C.c_ulonglong), C.c_int, C.c_int],
    "hipdrt_debug_qp_occupancy": [_vp, C.c_int, C.c_int],
    "hipdrt_debug_qp_group": [_vp, C.c_int],
    "hipdrt_debug_exact_zero_shortcuts": [_vp, C.c_int],
    "hipdrt_debug_qp_waves": [_vp, C.c_int],
    "hipdrt_debug_stream_pool": [_vp, C.c_int, C.POINTER(C.c_void_p), _ip, _ip, _ip],
    "hipdrt_comm_unique_id": [C.c_char_p],
    "hipdrt_comm_create": [C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(_vp)],
    "hipdrt_comm_destroy": [_vp],
    "hipdrt_comm_info": [_vp, _ip, _ip, _ip],
    "hipdrt_comm_broadcast_dev": [_vp, _vp, C.c_longlong, C.c_int],
    "hipdrt_comm_gather_dev": [_vp, _vp, C.c_longlong, _vp, C.c_int],
    "hipdrt_comm_broadcast": [_vp, _dp, C.c_longlong, C.c_int],
    "hipdrt_comm_gather": [_vp, _dp, C.c_longlong, _dp, C.c_int],
    "hipdrt_comm_allreduce_max": [_vp, _dp],
    "hipdrt_comm_barrier": [_vp],
    "hipdrt_device_alloc": [_vp, C.c_longlong, C.POINTER(_vp)],
    "hipdrt_device_free": [_vp, _vp],
    "hipdrt_device_synchronize": [_vp],
    "hipdrt_device_probe": [C.c_int],
    "hipdrt_weighted_gram": [_vp, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp],
    "hipdrt_default_fit_opts": [C.POINTER(FitOpts)],
    "hipdrt_plan_create": [_vp, _dp, C.c_int, _dp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                           _dp, _dp, _dp, _dp, C.POINTER(FitOpts), C.c_int, C.POINTER(_vp)],
    "hipdrt_plan_create_prepared": [_vp, C.POINTER(PreparedDesc), _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(FitOpts),
                                    C.c_int, C.POINTER(_vp)],
    "hipdrt_plan_upload_prepared": [_vp, C.c_int, C.c_int, _dp, _dp],
    "hipdrt_plan_set_weight_factors": [_vp, C.c_double, _dp, C.c_int],
    "hipdrt_plan_set_init_h": [_vp, _dp],
    "hipdrt_plan_destroy": [_vp],
    "hipdrt_plan_dims": [_vp, _ip, _ip, _ip],
    "hipdrt_plan_get": [_vp, C.c_char_p, _dp, C.c_longlong],
    "hipdrt_plan_set_lookup": [_vp, _dp, _dp],
    "hipdrt_plan_upload": [_vp, C.c_int, _dp, _dp],
    "hipdrt_plan_fit": [_vp],
    "hipdrt_plan_set_subbatches": [_vp, C.c_int],
    "hipdrt_plan_download": [_vp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _ip],
    "hipdrt_plan_get_p_matrix": [_vp, C.c_int, _dp],
    "hipdrt_plan_distribution_var": [_vp, _dp, C.c_int, _dp, _ip],
    "hipdrt_plan_llh_terms": [_vp, _dp, _dp],
    "hipdrt_plan_obs_llh_terms": [_vp, _dp, _dp],
    "hipdrt_plan_obs_llh_terms_w": [_vp, C.c_int, C.c_double, _dp, _dp],
    "hipdrt_plan_set_state": [_vp, _dp, _dp, _dp, _dp],
    "hipdrt_plan_set_state_dop": [_vp, _dp],
    "hipdrt_plan_continue": [_vp, C.POINTER(FitOpts), C.c_double, C.c_int],
    "hipdrt_plan_iterate": [_vp, C.POINTER(IterateState), _ip, _ip, _ip, _dp],
    "hipdrt_plan_param_var": [_vp, _dp, _ip],
    "hipdrt_plan_param_cov": [_vp, C.c_int, _dp, _ip],
    "hipdrt_plan_distribution_cov": [_vp, C.c_int, _dp, C.c_int, _dp, _ip],
    "hipdrt_plan_record_history": [_vp, C.c_int],
    "hipdrt_plan_get_history": [_vp, _dp, _dp, _dp, _ip, C.c_int, _ip],
    "hipdrt_plan_timings": [_vp, C.POINTER(C.c_float), _ip],
    "hipdrt_fit_eis_batch": [_vp, C.c_int, _dp, C.c_int, _dp, _dp, _dp, C.c_int, C.c_double, C.c_int, C.c_int,
                             C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.POINTER(FitOpts), _dp, _dp, _dp, _dp,
                             _dp, _dp, _dp, _dp, _ip, _ip],
}
_RESTYPES = {"hipdrt_last_error": C.c_char_p, "hipdrt_stream": C.c_void_p, "hipdrt_default_fit_opts": None}

_lib = None
_lock = threading.Lock()


def load_library():
    """dlopen libhipdrt.so and attach the prototypes (no device is touched)."""
    global _lib
    with _lock:
        if _lib is None:
            # The HIP runtime maps streams onto 4 hardware queues by default (one of them the null stream's).  libhipdrt creates
            # its streams once per device, one per remaining queue, and deals them to contexts and to the ranges of a fit by
            # activity and compute pipe (csrc/api.hip: StreamPool; profiles/r06_trace_queue_placement.txt) -- with 8 queues it has
            # seven streams over the four pipes, enough for four ranges or four plans side by side on a pipe each; with 4 it
            # has three.  The variable is read when the runtime starts, i.e. at the first HIP call of the process: set here, as
            # a default the caller's environment overrides, it takes effect unless something else in the process has started HIP.
            # (If torch is loaded and has started HIP already, the runtime runs with whatever it found then: exporting 8 now
            # would only make libhipdrt size its stream pool for queues that do not exist.)
            torch_mod = sys.modules.get("torch")
            hip_started = False
            try:
                hip_started = bool(torch_mod is not None and torch_mod.cuda.is_initialized())
            except Exception:                    # noqa: BLE001 (a torch build without the cuda module)
                hip_started = False
            if not hip_started:
                os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
            if not os.path.exists(LIB_PATH):
                raise HipDrtError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                                  f"g.build()'` (hipdrt has no CPU fallback)")
            lib = C.CDLL(LIB_PATH)
            for name, argtypes in SIGNATURES.items():
                fn = getattr(lib, name, None)
                if fn is None:
                    # (tools/: A/B against an OLDER build through HIPDRT_LIB -- a debug hook it does not have yet is simply not
                    # bound; every other missing symbol is an error, as is any missing symbol of the in-tree library)
                    if name.startswith("hipdrt_debug_") and "HIPDRT_LIB" in os.environ:
                        continue
                    raise HipDrtError(f"{LIB_PATH} does not export {name}")
                fn.argtypes = argtypes
                fn.restype = _RESTYPES.get(name, C.c_int)
            _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise HipDrtError(f"hipdrt error {rc}: {load_library().hipdrt_last_error().decode()}")


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _pi(a):
    return None if a is None else a.ctypes.data_as(_ip)


def default_fit_opts() -> FitOpts:
    o = FitOpts()
    load_library().hipdrt_default_fit_opts(C.byref(o))
    return o


class Context:
    """One hipdrt_ctx (device + stream)."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        h = _vp()
        _check(self._lib.hipdrt_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.hipdrt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return self._lib.hipdrt_stream(self._h)

    def synchronize(self):
        _check(self._lib.hipdrt_synchronize(self._h))

    def plan_bytes_per_spectrum(self, nf, ntau, ns):
        """device bytes one staged spectrum costs an EIS plan of this shape (hipdrt_plan_bytes_per_spectrum)"""
        out = C.c_longlong(0)
        _check(self._lib.hipdrt_plan_bytes_per_spectrum(int(nf), int(ntau), int(ns), C.byref(out)))
        return out.value

    def device_info(self):
        buf = C.create_string_buffer(64)
        ncu = C.c_int()
        hbm = C.c_longlong()
        _check(self._lib.hipdrt_device_info(self._h, buf, 64, C.byref(ncu), C.byref(hbm)))
        return dict(arch=buf.value.decode(), num_cu=ncu.value, hbm_bytes=hbm.value)

    # ---- L1 ------------------------------------------------------------------------------------------
    def impedance_lookup(self, epsilon, wt_re, wt_im, ny=1000):
        wt_re, wt_im = _f64(wt_re), _f64(wt_im)
        z_re, z_im = np.empty_like(wt_re), np.empty_like(wt_im)
        _check(self._lib.hipdrt_impedance_lookup(self._h, float(epsilon), wt_re.size, int(ny), _p(wt_re), _p(wt_im),
                                                 _p(z_re), _p(z_im)))
        return z_re, z_im

    def impedance_matrix(self, freq, tau, epsilon, mode=MODE_INTERP, toeplitz=False, lookups=None, ny=1000):
        freq, tau = _f64(freq), _f64(tau)
        batched = freq.ndim == 2
        B = freq.shape[0] if batched else 1
        nf = freq.shape[-1]
        if lookups is not None:
            (lre, zre), (lim, zim) = lookups
            lre, zre, lim, zim = _f64(lre), _f64(zre), _f64(lim), _f64(zim)
            ng = lre.size
        else:
            lre = zre = lim = zim = None
            ng = 0
        a_re = np.empty((B, nf, tau.size))
        a_im = np.empty((B, nf, tau.size))
        _check(self._lib.hipdrt_impedance_matrix(self._h, B, int(batched), _p(freq), nf, _p(tau), tau.size, int(mode),
                                                 int(bool(toeplitz)), float(epsilon), ng, _p(lre), _p(zre), _p(lim),
                                                 _p(zim), int(ny), _p(a_re), _p(a_im)))
        if not batched:
            return a_re[0], a_im[0]
        return a_re, a_im

    def impedance_matrix_timed(self, freq, tau, epsilon, dev_re, dev_im, mode=MODE_INTERP, toeplitz=False,
                               lookups=None, ny=1000, repeat=1):
        """Device-resident build (dev_re/dev_im: integer device pointers); returns elapsed ms of `repeat` launches."""
        freq, tau = _f64(freq), _f64(tau)
        batched = freq.ndim == 2
        B = freq.shape[0] if batched else 1
        nf = freq.shape[-1]
        (lre, zre), (lim, zim) = lookups if lookups is not None else ((None, None), (None, None))
        arrs = [None if a is None else _f64(a) for a in (lre, zre, lim, zim)]
        ng = 0 if arrs[0] is None else arrs[0].size
        ms = C.c_float()
        _check(self._lib.hipdrt_impedance_matrix_dev(self._h, B, int(batched), _p(freq), nf, _p(tau), tau.size,
                                                     int(mode), int(bool(toeplitz)), float(epsilon), ng, _p(arrs[0]),
                                                     _p(arrs[1]), _p(arrs[2]), _p(arrs[3]), int(ny), _vp(dev_re),
                                                     _vp(dev_im), int(repeat), C.byref(ms)))
        return ms.value

    def phasor_z_matrix(self, freq, nu, nu_epsilon):
        freq, nu = _f64(freq), _f64(nu)
        zr, zi = np.empty((freq.size, nu.size)), np.empty((freq.size, nu.size))
        _check(self._lib.hipdrt_phasor_z_matrix(self._h, _p(freq), freq.size, _p(nu), nu.size, float(nu_epsilon), _p(zr),
                                                _p(zi)))
        return zr + 1j * zi

    def phasor_v_matrix(self, times, nu, nu_epsilon, step_times, step_sizes):
        times, nu, st, sa = _f64(times), _f64(nu), _f64(step_times), _f64(step_sizes)
        rm = np.empty((times.size, nu.size))
        lay = np.empty((st.size, times.size, nu.size))
        _check(self._lib.hipdrt_phasor_v_matrix(self._h, _p(times), times.size, _p(nu), nu.size, float(nu_epsilon), _p(st),
                                                _p(sa), st.size, _p(rm), _p(lay)))
        return rm, lay

    def chrono_var_matrix(self, tt, seg, vmm_epsilon, uniform=False):
        tt = _f64(tt)
        seg = np.ascontiguousarray(seg, dtype=np.int32)
        out = np.empty((tt.size, tt.size))
        _check(self._lib.hipdrt_chrono_var_matrix(self._h, _p(tt), tt.size, _pi(seg), seg.size - 1, float(vmm_epsilon),
                                                  int(bool(uniform)), _p(out)))
        return out

    def response_lookup(self, epsilon, td, ny=1000):
        td = _f64(td)
        v = np.empty_like(td)
        _check(self._lib.hipdrt_response_lookup(self._h, float(epsilon), td.size, int(ny), _p(td), _p(v)))
        return v

    def response_matrix(self, times, tau, step_times, step_sizes, epsilon, mode=MODE_INTERP, lookup=None, ny=1000,
                        layered=True):
        times, tau, st, sa = _f64(times), _f64(tau), _f64(step_times), _f64(step_sizes)
        if st.size != sa.size:
            raise ValueError("step_times and step_sizes must have the same length")
        a = np.empty((times.size, tau.size))
        lay = np.empty((st.size, times.size, tau.size)) if layered else None
        if lookup is not None:
            log_td, v = _f64(lookup[0]), _f64(lookup[1])
            ng, plt, pv = log_td.size, _p(log_td), _p(v)
        else:
            ng, plt, pv = 0, None, None
        _check(self._lib.hipdrt_response_matrix(self._h, _p(times), times.size, _p(tau), tau.size, _p(st), _p(sa), st.size,
                                                int(mode), float(epsilon), ng, plt, pv, int(ny), _p(a),
                                                _p(lay) if layered else None))
        return a, lay

    def response_matrix_variant(self, times, tau, step_times, step_sizes, variant, tau_rise=None, epsilon=1.0, ny=1000,
                                layered=True):
        """the potentiostatic (RESPONSE_POT) and the expdecay-step (RESPONSE_EXPDECAY, trapz) forms of construct_response_matrix"""
        times, tau, st, sa = _f64(times), _f64(tau), _f64(step_times), _f64(step_sizes)
        if st.size != sa.size:
            raise ValueError("step_times and step_sizes must have the same length")
        tr = None
        if tau_rise is not None:
            tr = _f64(tau_rise)
            if tr.size != st.size:
                raise ValueError("tau_rise needs one entry per step")
        a = np.empty((times.size, tau.size))
        lay = np.empty((st.size, times.size, tau.size)) if layered else None
        _check(self._lib.hipdrt_response_matrix_variant(self._h, _p(times), times.size, _p(tau), tau.size, _p(st), _p(sa),
                                                        _p(tr) if tr is not None else None, st.size, int(variant), float(epsilon),
                                                        int(ny), _p(a), _p(lay) if layered else None))
        return a, lay

    def nonuniform_gaussian_filter1d(self, y, sigma, seg, filtered, nodes, node_delta, weights, woff, radius):
        """segment-wise blended Gaussian filter; see hipdrt.filters.nonuniform_gaussian_filter1d for the set-up"""
        y, sigma, nodes, node_delta, weights = _f64(y), _f64(sigma), _f64(nodes), _f64(node_delta), _f64(weights)
        seg, filtered, woff, radius = (np.ascontiguousarray(a, dtype=np.int32) for a in (seg, filtered, woff, radius))
        out = np.empty_like(y)
        _check(self._lib.hipdrt_nonuniform_gaussian_filter1d(self._h, _p(y), y.size, _p(sigma), _pi(seg), seg.size - 1,
                                                             _pi(filtered), _p(nodes), nodes.shape[1], _p(node_delta),
                                                             _p(weights), weights.size, _pi(woff), _pi(radius), _p(out)))
        return out

    def penalty_matrices(self, ln_tau, epsilon, toeplitz):
        ln_tau = _f64(ln_tau)
        n = ln_tau.size
        out = [np.empty((n, n)) for _ in range(3)]
        _check(self._lib.hipdrt_penalty_matrices(self._h, _p(ln_tau), n, float(epsilon), int(bool(toeplitz)),
                                                 _p(out[0]), _p(out[1]), _p(out[2])))
        return out

    def eis_var_matrix(self, freq, vmm_epsilon=0.25, reim_cor=0.25, uniform=False):
        freq = _f64(freq)
        vmm = np.empty((2 * freq.size, 2 * freq.size))
        _check(self._lib.hipdrt_eis_var_matrix(self._h, _p(freq), freq.size, float(vmm_epsilon), float(reim_cor),
                                               int(bool(uniform)), _p(vmm)))
        return vmm

    # ---- L2 ------------------------------------------------------------------------------------------
    def qp_batch(self, P, q, h, opts: QpOpts | None = None):
        P, q, h = _f64(P), _f64(q), _f64(h)
        if q.ndim == 1:
            q = q[None, :]
        B, n = q.shape
        p_batched = P.ndim == 3
        h_batched = h.ndim == 2
        x = np.empty((B, n))
        iters = np.empty(B, dtype=np.int32)
        pcost = np.empty(B)
        status = np.empty(B, dtype=np.int32)
        _check(self._lib.hipdrt_qp_batch(self._h, B, n, int(p_batched), _p(P), _p(q), int(h_batched), _p(h),
                                         C.byref(opts) if opts is not None else None, _p(x), _pi(iters), _p(pcost),
                                         _pi(status)))
        return dict(x=x, iterations=iters, pcost=pcost, status=status)

    def fit_eis_batch(self, freq, z, tau, epsilon, wt_re, wt_im, toeplitz_a=False, toeplitz_m=False, opts=None,
                      ny=1000):
        """the one-shot C entry point hipdrt_fit_eis_batch (interp mode): plan create, upload, fit, download, destroy"""
        freq, tau, wt_re, wt_im = _f64(freq), _f64(tau), _f64(wt_re), _f64(wt_im)
        z = np.atleast_2d(np.asarray(z))
        z_re, z_im = _f64(z.real), _f64(z.imag)
        lre, lim = np.log(wt_re), np.log(wt_im)
        opts = opts if opts is not None else default_fit_opts()
        B, nf, ntau = z.shape[0], freq.size, tau.size
        n = ntau + int(opts.fit_ohmic) + int(opts.fit_inductance)
        out = {"x": np.empty((B, n)), "fit_x": np.empty((B, ntau)), "R_inf": np.empty(B), "inductance": np.empty(B),
               "weights": np.empty((B, 2 * nf)), "coefficient_scale": np.empty(B), "rho": np.empty((B, 3)),
               "q_vector": np.empty((B, n)), "outer_iters": np.empty(B, dtype=np.int32),
               "status": np.empty(B, dtype=np.int32)}
        _check(self._lib.hipdrt_fit_eis_batch(self._h, B, _p(freq), nf, _p(z_re), _p(z_im), _p(tau), ntau, float(epsilon),
                                              MODE_INTERP, int(bool(toeplitz_a)), int(bool(toeplitz_m)), wt_re.size,
                                              int(ny), _p(wt_re), _p(wt_im), _p(lre), _p(lim), C.byref(opts),
                                              _p(out["x"]), _p(out["fit_x"]), _p(out["R_inf"]), _p(out["inductance"]),
                                              _p(out["weights"]), _p(out["coefficient_scale"]), _p(out["rho"]),
                                              _p(out["q_vector"]), _pi(out["outer_iters"]), _pi(out["status"])))
        return out

    def device_alloc(self, nbytes):
        """device memory as an integer pointer (hipdrt_device_alloc): for the *_dev entry points; free with device_free"""
        ptr = _vp()
        _check(self._lib.hipdrt_device_alloc(self._h, int(nbytes), C.byref(ptr)))
        return ptr.value

    def device_free(self, ptr):
        _check(self._lib.hipdrt_device_free(self._h, _vp(ptr)))

    def device_synchronize(self):
        """hipDeviceSynchronize on this context's device (every stream, every context of the process on that GPU)"""
        _check(self._lib.hipdrt_device_synchronize(self._h))

    def debug_qp_group(self, members):
        """tests / diagnostics: force the workgroups per problem of this context's coneqp launches sized from now on
        (hipdrt_debug_qp_group, include/hipdrt_debug.h)"""
        _check(self._lib.hipdrt_debug_qp_group(self._h, int(members)))
        self._qp_group_override = int(members)          # (remembered for callers that switch it temporarily: mapping._one_kernel)

    def debug_qp_waves(self, waves):
        """tests / tools: 4 = this context's batch coneqp launches (n <= 528) use the fat four-wavefront kernel, 8 = the
        eight-wavefront one, -1 = the library's choice (hipdrt_debug_qp_waves, include/hipdrt_debug.h)"""
        _check(self._lib.hipdrt_debug_qp_waves(self._h, int(waves)))

    def debug_stream_pool(self):
        """tests: (streams, holders, running) of the library's own streams on this context's device
        (hipdrt_debug_stream_pool, include/hipdrt_debug.h)"""
        size = C.c_int(0)
        _check(self._lib.hipdrt_debug_stream_pool(self._h, 0, None, None, None, C.byref(size)))
        n = size.value
        st, ho, ru = (C.c_void_p * n)(), (C.c_int * n)(), (C.c_int * n)()
        _check(self._lib.hipdrt_debug_stream_pool(self._h, n, st, ho, ru, C.byref(size)))
        return [st[i] for i in range(n)], [ho[i] for i in range(n)], [ru[i] for i in range(n)]

    def debug_exact_zero_shortcuts(self, on):
        """tests: with on = False this context's fits visit the penalty matrices' exact zeros as well (same bits, slower)
        (hipdrt_debug_exact_zero_shortcuts, include/hipdrt_debug.h)"""
        _check(self._lib.hipdrt_debug_exact_zero_shortcuts(self._h, int(bool(on))))

    def qp_profile(self, reset=True):
        buf = (C.c_ulonglong * 64)()        # 0..47 the QP kernel's phases, 48..63 hyper_kernel's (PROFILE=1 builds)
        _check(self._lib.hipdrt_qp_profile(self._h, buf, 64, int(reset)))
        return [int(v) for v in buf]

    def qp_timeline(self):
        """PROFILE builds: s_memtime stamps [wavefront 8][super column 16][stamp 8] of workgroup 0's last factorisation
        (csrc/qp_common.hpp, g_qp_tl); zeros otherwise"""
        n = 64 + 8 * 16 * 8
        buf = (C.c_ulonglong * n)()
        _check(self._lib.hipdrt_qp_profile(self._h, buf, n, 0))
        return np.array(buf[64:], dtype=np.uint64).reshape(8, 16, 8)

    def qp_timeline_mean(self, reset=True):
        """PROFILE builds: the same stamps as mean cycles since the start of the factorisation, over all factorisations of
        workgroup 0 since the last reset -> (array [8][16][8], number of factorisations)"""
        nt = 8 * 16 * 8
        n = 64 + 2 * nt + 1
        buf = (C.c_ulonglong * n)()
        _check(self._lib.hipdrt_qp_profile(self._h, buf, n, int(reset)))
        cnt = int(buf[64 + 2 * nt])
        return np.array(buf[64 + nt:64 + 2 * nt], dtype=np.float64).reshape(8, 16, 8) / max(cnt, 1), cnt

    def weighted_gram(self, A, w, b, l2=None, l1=None):
        A, w, b = _f64(A), _f64(w), _f64(b)
        if w.ndim == 1:
            w, b = w[None, :], b[None, :]
        B, m = w.shape
        n = A.shape[1]
        l2a = None if l2 is None else _f64(l2)
        l1a = None if l1 is None else _f64(l1)
        P = np.empty((B, n, n))
        q = np.empty((B, n))
        _check(self._lib.hipdrt_weighted_gram(self._h, B, m, n, _p(A), _p(w), _p(b),
                                              int(l2a is not None and l2a.ndim == 3), _p(l2a), _p(l1a), _p(P), _p(q)))
        return P, q


class Plan:
    """hipdrt_plan: shared matrices + work space for `capacity` spectra on one frequency / tau grid."""

    def __init__(self, ctx: Context, freq, tau, epsilon, wt_re=None, wt_im=None, mode=MODE_INTERP,
                 toeplitz_a=False, toeplitz_m=False, opts: FitOpts | None = None, capacity=1, ny=1000):
        self._lib = load_library()
        self.ctx = ctx
        freq, tau = _f64(freq), _f64(tau)
        self.freq, self.tau = freq, tau
        if mode == MODE_INTERP:
            wt_re, wt_im = _f64(wt_re), _f64(wt_im)
            lre, lim = np.log(wt_re), np.log(wt_im)
            ng = wt_re.size
        else:
            wt_re = wt_im = lre = lim = None
            ng = 0
        self.log_wt_re, self.log_wt_im = lre, lim
        self.opts = opts if opts is not None else default_fit_opts()
        h = _vp()
        _check(self._lib.hipdrt_plan_create(ctx._h, _p(freq), freq.size, _p(tau), tau.size, float(epsilon), int(mode),
                                            int(bool(toeplitz_a)), int(bool(toeplitz_m)), ng, int(ny), _p(wt_re),
                                            _p(wt_im), _p(lre), _p(lim), C.byref(self.opts), int(capacity),
                                            C.byref(h)))
        self._h = h
        if os.environ.get("HIPDRT_SUBBATCHES"):           # tools / A-B runs
            self.set_subbatches(int(os.environ["HIPDRT_SUBBATCHES"]))
        n, m, ns = C.c_int(), C.c_int(), C.c_int()
        _check(self._lib.hipdrt_plan_dims(self._h, C.byref(n), C.byref(m), C.byref(ns)))
        self.n, self.m, self.ns = n.value, m.value, ns.value
        self.nf, self.ntau, self.ngrid = freq.size, tau.size, ng
        self.capacity = int(capacity)
        self.B = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.hipdrt_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get(self, which):
        shapes = {"lut_z_re": (self.ngrid,), "lut_z_im": (self.ngrid,), "a_re": (self.nf, self.ntau),
                  "a_im": (self.nf, self.ntau), "rm": (self.m, self.n), "m0": (self.n, self.n),
                  "m1": (self.n, self.n), "m2": (self.n, self.n), "vmm": (self.m, self.m), "h": (self.n,),
                  "est_weights": (self.batch, self.m), "rv": (self.batch, self.m), "xmx": (self.batch, 3),
                  "outlier_t": (self.batch, self.m)}
        out = np.empty(shapes[which])
        _check(self._lib.hipdrt_plan_get(self._h, which.encode(), _p(out), out.size))
        return out

    def set_lookup(self, z_re, z_im):
        z_re, z_im = _f64(z_re), _f64(z_im)
        _check(self._lib.hipdrt_plan_set_lookup(self._h, _p(z_re), _p(z_im)))

    batch = 0      # spectra of the last upload

    def upload(self, z):
        z = np.asarray(z)
        if z.ndim == 1:
            z = z[None, :]
        z_re, z_im = _f64(z.real), _f64(z.imag)
        _check(self._lib.hipdrt_plan_upload(self._h, z.shape[0], _p(z_re), _p(z_im)))
        self.batch = z.shape[0]
        self.B = z.shape[0]

    def fit(self):
        _check(self._lib.hipdrt_plan_fit(self._h))

    def set_subbatches(self, k):
        """ranges the staged batch is fitted in, side by side inside one fit() call (hipdrt_plan_set_subbatches): 0 = the
        library chooses from the batch size (default; env HIPDRT_SUBBATCHES overrides at plan creation), 1 = one launch sequence"""
        _check(self._lib.hipdrt_plan_set_subbatches(self._h, int(k)))

    def llh_terms(self, stored=False, weights=None):
        """(rss, sum(log w)) per spectrum; stored=False: weights re-estimated from the current x (PFRT steps),
        stored=True: the `weights` argument of DRT.evaluate_rss() / evaluate_llh(): None = the fit's own est_weights,
        'uniform' = per-domain means of them (DRTMD's default), a positive scalar = that weight for every row."""
        rss, slw = np.empty(self.batch), np.empty(self.batch)
        if not stored:
            _check(self._lib.hipdrt_plan_llh_terms(self._h, _p(rss), _p(slw)))
        elif weights is None:
            _check(self._lib.hipdrt_plan_obs_llh_terms(self._h, _p(rss), _p(slw)))
        elif isinstance(weights, str):
            if weights != 'uniform':
                raise ValueError(f"weights must be None, 'uniform' or a scalar, got {weights!r}")
            _check(self._lib.hipdrt_plan_obs_llh_terms_w(self._h, 2, 1.0, _p(rss), _p(slw)))
        else:
            _check(self._lib.hipdrt_plan_obs_llh_terms_w(self._h, 3, float(weights), _p(rss), _p(slw)))
        return rss, slw

    def set_state(self, x=None, rho=None, s=None, weights=None, dop_rho=None):
        arrs = [None if a is None else _f64(a) for a in (x, rho, s, weights)]
        _check(self._lib.hipdrt_plan_set_state(self._h, *[None if a is None else _p(a) for a in arrs]))
        if dop_rho is not None:
            _check(self._lib.hipdrt_plan_set_state_dop(self._h, _p(_f64(dop_rho))))

    def continue_fit(self, opts, weight_factor=1.0, min_iter=2):
        _check(self._lib.hipdrt_plan_continue(self._h, C.byref(opts), float(weight_factor), int(min_iter)))

    def set_weight_factors(self, weight_factor=1.0, row_factors=None, late=False):
        """weight_factor / chrono- and EIS-row factors of _qphb_fit_core; row_factors (m,) or (capacity, m); late=True: the
        rows are a vector-valued weight_factor (applied from the second iteration on)"""
        rf = None if row_factors is None else _f64(row_factors)
        if rf is not None and rf.ndim == 2 and rf.shape[0] < self.capacity:     # the C side reads capacity rows
            rf = _f64(np.vstack([rf, np.ones((self.capacity - rf.shape[0], rf.shape[1]))]))
        _check(self._lib.hipdrt_plan_set_weight_factors(self._h, float(weight_factor), _p(rf),
                                                        int(rf is not None and rf.ndim == 2) | (2 if late else 0)))

    def set_init_h(self, h_init):
        h = None if h_init is None else _f64(h_init)
        _check(self._lib.hipdrt_plan_set_init_h(self._h, _p(h)))

    def record_history(self, b):
        _check(self._lib.hipdrt_plan_record_history(self._h, int(b)))

    def download(self, s_vectors=False, lean=False):
        """results of the staged spectra; ``lean``: only what a map records per observation (fit_x, R_inf, inductance, the
        coefficient scale, iteration counts, status) -- 4 kB instead of 29 kB per spectrum at 256 x 512"""
        B, n, m = self.B, self.n, self.m
        out = dict(fit_x=np.empty((B, self.ntau)), R_inf=np.empty(B), inductance=np.empty(B), coefficient_scale=np.empty(B),
                   outer_iters=np.empty(B, dtype=np.int32), qp_iters_total=np.empty(B, dtype=np.int32),
                   status=np.empty(B, dtype=np.int32))
        if not lean:
            out.update(x=np.empty((B, n)), weights=np.empty((B, m)), rho=np.empty((B, 3)), q_vector=np.empty((B, n)))
        sv = np.empty((B, 3, n)) if (s_vectors and not lean) else None
        opt = lambda key: _p(out[key]) if key in out else None          # noqa: E731 (NULL = not wanted, include/hipdrt.h)
        _check(self._lib.hipdrt_plan_download(self._h, opt("x"), _p(out["fit_x"]), _p(out["R_inf"]),
                                              _p(out["inductance"]), opt("weights"), _p(out["coefficient_scale"]),
                                              opt("rho"), _p(sv) if sv is not None else None, opt("q_vector"),
                                              _pi(out["outer_iters"]), _pi(out["qp_iters_total"]), _pi(out["status"])))
        if sv is not None:
            out["s_vectors"] = sv
        return out

    def p_matrix(self, b):
        out = np.empty((self.n, self.n))
        _check(self._lib.hipdrt_plan_get_p_matrix(self._h, int(b), _p(out)))
        return out

    def distribution_var(self, basis_eval, batch):
        """diag(B P^-1 B') cs^2 for every fitted spectrum; basis_eval (neval, ntau)."""
        basis_eval = _f64(basis_eval)
        out = np.empty((int(batch), basis_eval.shape[0]))
        status = np.empty(int(batch), dtype=np.int32)
        _check(self._lib.hipdrt_plan_distribution_var(self._h, _p(basis_eval), basis_eval.shape[0], _p(out), _pi(status)))
        return out, status

    def param_cov(self, b=0):
        """inv(P_b) cs_b^2 (n, n) of fitted spectrum b, and whether P_b was positive definite"""
        out = np.empty((self.n, self.n))
        status = C.c_int()
        _check(self._lib.hipdrt_plan_param_cov(self._h, int(b), _p(out), C.byref(status)))
        return out, status.value == 0

    def distribution_cov(self, basis_eval, b=0):
        """basis_eval inv(P_b)[DRT block] basis_eval' cs_b^2 (neval, neval) of fitted spectrum b"""
        basis_eval = _f64(basis_eval)
        out = np.empty((basis_eval.shape[0], basis_eval.shape[0]))
        status = C.c_int()
        _check(self._lib.hipdrt_plan_distribution_cov(self._h, int(b), _p(basis_eval), basis_eval.shape[0], _p(out),
                                                      C.byref(status)))
        return out, status.value == 0

    def param_var(self, batch):
        out = np.empty((int(batch), self.n))
        status = np.empty(int(batch), dtype=np.int32)
        _check(self._lib.hipdrt_plan_param_var(self._h, _p(out), _pi(status)))
        return out, status

    def history(self):
        cap = int(self.opts.max_iter)
        hx, hr, hw = np.empty((cap, self.n)), np.empty((cap, 3)), np.empty((cap, self.m))
        qi = np.empty(cap + 1, dtype=np.int32)
        rows = C.c_int()
        _check(self._lib.hipdrt_plan_get_history(self._h, _p(hx), _p(hr), _p(hw), _pi(qi), cap, C.byref(rows)))
        r = rows.value
        return dict(x=hx[:r], rho=hr[:r], weights=hw[:r], qp_iterations=qi[:r + 1])

    def timings(self):
        t = (C.c_float * 5)()
        l = (C.c_int * 5)()
        _check(self._lib.hipdrt_plan_timings(self._h, t, l))
        names = ("total", "gram", "qp", "hyper", "other")
        return {k: float(t[i]) for i, k in enumerate(names)}, {k: int(l[i]) for i, k in enumerate(names)}


class PreparedPlan(Plan):
    """hipdrt_plan_create_prepared: the device loop on caller-prepared matrices (any data type; optional x_dop block
    and vz_offset column).  `desc` is a PreparedDesc, penalty = [m0, m1, m2] (n, n), vmm (m, m), h / l1 (n,)."""

    def __init__(self, ctx: Context, desc: PreparedDesc, penalty, vmm, h, l1, vz_strength=None,
                 opts: FitOpts | None = None, capacity=1):
        self._lib = load_library()
        self.ctx = ctx
        self.desc = desc
        self.opts = opts if opts is not None else default_fit_opts()
        mk = [_f64(a) for a in penalty]
        vmm, h, l1 = _f64(vmm), _f64(h), _f64(l1)
        vzs = None if vz_strength is None else _f64(vz_strength)
        hnd = _vp()
        _check(self._lib.hipdrt_plan_create_prepared(ctx._h, C.byref(desc), _p(mk[0]), _p(mk[1]), _p(mk[2]), _p(vmm),
                                                     _p(h), _p(l1), _p(vzs), C.byref(self.opts), int(capacity),
                                                     C.byref(hnd)))
        self._h = hnd
        self.n, self.m, self.ns = desc.n, desc.m, desc.ns
        self.nf, self.ntau, self.ngrid = 0, desc.n - desc.ns, 0
        self.capacity = int(capacity)
        self.B = 0
        self.rm_batched = False

    def upload(self, rzm, rzv):
        """rzm (m, n) shared or (B, m, n) per measurement; rzv (B, m)"""
        rzm, rzv = _f64(rzm), _f64(rzv)
        if rzv.ndim == 1:
            rzv = rzv[None, :]
        batched = rzm.ndim == 3
        _check(self._lib.hipdrt_plan_upload_prepared(self._h, rzv.shape[0], int(batched), _p(rzm), _p(rzv)))
        self.batch = self.B = rzv.shape[0]
        self.rm_batched = batched

    def iterate(self, x_in=None, s_vectors=None, rho=None, dop_rho=None, weights=None, est_weights=None,
                xmx_norms=None, dop_xmx_norms=None):
        """hipdrt_plan_iterate: one qphb.iterate_qphb on every staged measurement; arrays are (B, ...) or None to keep
        what the device holds.  Returns dict(converged, qp_status, qp_iters, primal_objective), each (B,)."""
        B, n, m = self.batch, self.n, self.m
        shapes = dict(x_in=(B, n), s_vectors=(B, 3, n), rho=(B, 3), dop_rho=(B, 3), weights=(B, m),
                      est_weights=(B, m), xmx_norms=(B, 3), dop_xmx_norms=(B, 3))
        given = dict(x_in=x_in, s_vectors=s_vectors, rho=rho, dop_rho=dop_rho, weights=weights,
                     est_weights=est_weights, xmx_norms=xmx_norms, dop_xmx_norms=dop_xmx_norms)
        st, keep = IterateState(), []
        for name, arr in given.items():
            if arr is None:
                continue
            a = _f64(arr)
            if a.shape != shapes[name]:
                raise ValueError(f"{name}: expected shape {shapes[name]}, got {a.shape}")
            keep.append(a)
            setattr(st, name, _p(a))
        conv, status, iters = (np.empty(B, dtype=np.int32) for _ in range(3))
        pobj = np.empty(B)
        _check(self._lib.hipdrt_plan_iterate(self._h, C.byref(st), _pi(conv), _pi(status), _pi(iters), _p(pobj)))
        return dict(converged=conv.astype(bool), qp_status=status, qp_iters=iters, primal_objective=pobj)

    def get(self, which):
        B = self.batch
        shapes = {"m0": (self.n, self.n), "m1": (self.n, self.n), "m2": (self.n, self.n), "vmm": (self.m, self.m),
                  "h": (self.n,), "est_weights": (B, self.m), "rv": (B, self.m), "xmx": (B, 3), "dop_rho": (B, 3),
                  "dop_xmx": (B, 3), "hist_dop_rho": (int(self.opts.max_iter), 3), "outlier_t": (B, self.m),
                  "weight_factors": (B, 2),
                  "rzm": (B, self.m, self.n) if self.rm_batched else (self.m, self.n)}
        out = np.empty(shapes[which])
        _check(self._lib.hipdrt_plan_get(self._h, which.encode(), _p(out), out.size))
        return out

    def history(self):
        h = super().history()
        if self.desc.dop_size > 0:
            h["dop_rho"] = self.get("hist_dop_rho")[:len(h["x"])]
        return h


_default_ctx = {}


def device_usable(device: int = 0) -> bool:
    """can this process open gfx950 device `device`? (no exception, and nothing is created on the device -- not even a stream:
    streams are dealt to the hardware queues in the order they are made; mapping.dist picks its default backend with this)"""
    try:
        lib = load_library()
    except HipDrtError:
        return False
    return lib.hipdrt_device_probe(int(device)) == 0


def comm_unique_id() -> bytes:
    """128 opaque bytes of a fresh RCCL communicator id (rank 0 makes them, every rank passes them to Comm)"""
    buf = C.create_string_buffer(128)
    _check(load_library().hipdrt_comm_unique_id(buf))
    return buf.raw


class Comm:
    """RCCL communicator behind the C ABI (include/hipdrt.h: hipdrt_comm_*): one per process, numpy in / numpy out"""

    def __init__(self, device, rank, world, unique_id):
        self._lib = load_library()
        self._h = _vp()
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        if len(unique_id) != 128:
            raise HipDrtError("a RCCL unique id has 128 bytes")
        _check(self._lib.hipdrt_comm_create(self.device, self.rank, self.world, C.c_char_p(bytes(unique_id)), C.byref(self._h)))

    def close(self):
        if self._h:
            self._lib.hipdrt_comm_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 (interpreter shutdown)
            pass

    def broadcast(self, array, root=0):
        """`array` (contiguous float64) of `root` into every rank's `array`, in place"""
        if not (isinstance(array, np.ndarray) and array.dtype == np.float64 and array.flags.c_contiguous):
            raise HipDrtError("broadcast needs a C-contiguous float64 array (it is filled in place)")
        _check(self._lib.hipdrt_comm_broadcast(self._h, _p(array), array.size, int(root)))
        return array

    def gather(self, block, root=0):
        """every rank's `block` (same size everywhere) to `root`: returns (world, block.size) there, None elsewhere"""
        block = _f64(block).ravel()
        out = np.empty((self.world, block.size)) if self.rank == root else None
        _check(self._lib.hipdrt_comm_gather(self._h, _p(block), block.size, _p(out), int(root)))
        return out

    def allreduce_max(self, value):
        v = C.c_double(float(value))
        _check(self._lib.hipdrt_comm_allreduce_max(self._h, C.byref(v)))
        return v.value

    def barrier(self):
        _check(self._lib.hipdrt_comm_barrier(self._h))


def get_context(device: int = 0) -> Context:
    """Process-wide context per device (created on first use)."""
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
