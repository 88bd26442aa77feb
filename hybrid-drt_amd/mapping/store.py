"""The observation store around the batch driver: what DRTMD keeps between calls (hybdrt/mapping/drtmd.py:102-160, 186-329).

``mapping.fit_observations`` is a function -- it fits what it is handed.  ``DRTMD.fit_all(refit=False)`` is a method of an object
that REMEMBERS: observations are added one by one (``add_observation``, drtmd.py:186-241), each carries a fit status, an ignore
flag and the error that made it fail, and ``fit_all`` fits only the ones that are neither fitted nor ignored (drtmd.py:321-329)
-- the resume-after-interruption path of a long map.  This class is that bookkeeping on top of the device batches: the
observations selected by a call are fitted as ONE list (mapping.fit_observations groups them by data type and sampling grid, one
device batch per group), results are scattered into the per-observation arrays, and the status / ignore / error rules are the
reference's:

    fit succeeded                          obs_fit_status True
    fit failed, ignore_errors=True         obs_fit_status False, obs_ignore_flag True, obs_fit_errors[i] = the error
                                           (drtmd.py:292-299) -- a later fit_all(refit=False) skips it
    fit failed, ignore_errors=False        the error is raised (drtmd.py:300-301); observations of the same call that were
                                           fitted on the device before the failed one keep their results, as in the
                                           reference's serial loop, later ones stay unfitted

Not here (out of scope): file readers, psi interpolation / filtering, resolve bookkeeping, pickling."""
import numpy as np

from . import drtmd as _driver


class DRTMD:
    def __init__(self, tau_supergrid, drt=None, fit_kw=None, fit_type='drt', pfrt_factors=None, drt_var=True, llh_kw=None,
                 rss_kw=None, psi_dim_names=None, **drt_kw):
        from ..models import DRT
        self.tau_supergrid = np.asarray(tau_supergrid, dtype=float)
        self.drt1d = drt if drt is not None else DRT(tau_supergrid=self.tau_supergrid, **drt_kw)      # drtmd.py:52-58
        if fit_type not in ('drt', 'pfrt'):
            raise ValueError(f"Invalid fit_type {fit_type}. Options: ['drt', 'pfrt']")          # drtmd.py:1479-1482
        self.fit_type = fit_type
        defaults = {'nonneg': True}                                                                # drtmd.py:89-96
        defaults.update(fit_kw or {})
        self.fit_kw = defaults
        # (upstream's attribute says logspace(-0.7, 0.7, 11), drtmd.py:98-100, but its fits never receive it and run with
        # _pfrt_fit_core's own logspace(-1, 1, 11) -- mapping.fit_observations_pfrt's docstring; None = that behaviour)
        self.pfrt_factors = None if pfrt_factors is None else np.asarray(pfrt_factors, dtype=float)
        n_steps = 11 if self.pfrt_factors is None else len(self.pfrt_factors)
        self.drt_var = bool(drt_var)
        self.llh_kw, self.rss_kw = dict(llh_kw or {}), dict(rss_kw or {})
        for kw in (self.llh_kw, self.rss_kw):                                                     # drtmd.py:127-129
            kw.setdefault('normalize', True)
            kw.setdefault('weights', 'uniform')
        self.psi_dim_names = psi_dim_names
        self.obs_psi = None if psi_dim_names is None else np.zeros((0, len(psi_dim_names)))
        self.obs_data, self.obs_group_id, self.obs_fit_errors, self.obs_tau_indices = [], [], [], []
        self.obs_ignore_flag = np.zeros(0, dtype=bool)
        self.obs_fit_status = np.zeros(0, dtype=bool)
        nt = len(self.tau_supergrid)
        self.obs_x = np.zeros((0, nt)) if fit_type == 'drt' else np.zeros((0, n_steps, nt))
        self.obs_drt_var = np.zeros_like(self.obs_x)
        self.obs_special = None
        self.obs_llh, self.obs_rss = np.zeros(0), np.zeros(0)
        self.last_fit_index = np.zeros(0, dtype=int)          # what the last fit_observations call sent to the device

    @property
    def num_obs(self):
        return len(self.obs_data)

    def add_observation(self, psi, chrono_data, eis_data, group_id=None, fit=False):
        """drtmd.py:186-241: append one observation (unfitted, not ignored); ``fit=True`` fits it right away"""
        psi = np.atleast_1d(psi).astype(float).flatten()
        if self.obs_psi is None:
            self.obs_psi = np.zeros((0, len(psi)))
        if len(psi) != self.obs_psi.shape[1]:
            raise ValueError(f'psi must have length {self.obs_psi.shape[1]}')
        self.obs_psi = np.vstack([self.obs_psi, psi[None]])
        self.obs_data.append((chrono_data, eis_data))
        self.obs_group_id.append(group_id)
        self.obs_ignore_flag = np.append(self.obs_ignore_flag, False)
        self.obs_fit_status = np.append(self.obs_fit_status, False)
        self.obs_fit_errors.append(None)
        self.obs_tau_indices.append(None)
        self.obs_x = np.concatenate([self.obs_x, np.zeros((1,) + self.obs_x.shape[1:])])
        self.obs_drt_var = np.concatenate([self.obs_drt_var, np.zeros((1,) + self.obs_drt_var.shape[1:])])
        self.obs_llh, self.obs_rss = np.append(self.obs_llh, 0.0), np.append(self.obs_rss, 0.0)
        if self.obs_special is not None:
            for key in self.obs_special:
                val = self.obs_special[key]
                self.obs_special[key] = np.concatenate([val, np.zeros((1,) + val.shape[1:])])
        if fit:
            self.fit_observation(self.num_obs - 1)

    def get_obs_data(self, obs_index):
        return self.obs_data[obs_index]

    def fit_observation(self, obs_index, ignore_errors=False):
        self.fit_observations([obs_index], ignore_errors=ignore_errors)

    def fit_observations(self, obs_index, ignore_errors=False):
        """drtmd.py:303-319: the listed observations, as one device job"""
        idx = np.asarray(obs_index, dtype=int).ravel()
        self.last_fit_index = idx
        if len(idx) == 0:
            return
        observations = [self.obs_data[i] for i in idx]
        kw = dict(self.fit_kw)
        obs_x, obs_special, res = _driver.fit_observations(
            self.drt1d, observations=observations, tau_supergrid=self.tau_supergrid, drt_var=self.drt_var, ignore_errors=True,
            llh_kw=self.llh_kw, rss_kw=self.rss_kw, fit_type=self.fit_type,
            pfrt_factors=self.pfrt_factors if self.fit_type == 'pfrt' else None, **kw)
        ok = np.asarray(res['obs_fit_status'], dtype=bool)
        errors = list(res['obs_fit_errors'])
        first_bad = int(np.argmin(ok)) if not ok.all() else len(idx)
        # without ignore_errors the reference's loop stops at the first failure: what came before it is stored, the rest is not
        keep = np.arange(len(idx)) if ignore_errors else np.arange(first_bad)
        for j in keep:
            i = idx[j]
            if ok[j]:
                self.obs_x[i] = obs_x[j]
                self.obs_llh[i], self.obs_rss[i] = res['obs_llh'][j], res['obs_rss'][j]
                ti = res['obs_tau_indices']
                self.obs_tau_indices[i] = tuple(ti[j]) if isinstance(ti, list) else tuple(ti)
                if self.drt_var and 'obs_drt_var' in res:
                    self.obs_drt_var[i] = res['obs_drt_var'][j]
                if self.obs_special is None:                      # initialize_obs_special, drtmd.py:270-271
                    self.obs_special = {}
                for key, val in obs_special.items():
                    val = np.asarray(val)
                    if key not in self.obs_special:               # "key is new", drtmd.py:281-285
                        self.obs_special[key] = np.zeros((self.num_obs,) + val.shape[1:])
                    self.obs_special[key][i] = val[j]
                self.obs_fit_status[i] = True
                self.obs_fit_errors[i] = None
            else:                                                 # drtmd.py:292-299
                self.obs_fit_status[i] = False
                self.obs_ignore_flag[i] = True
                self.obs_fit_errors[i] = errors[j]
        if not ignore_errors and first_bad < len(idx):
            print(f"Error encountered at obs_index {idx[first_bad]}")
            raise errors[first_bad]

    def fit_all(self, refit=False, ignore_errors=False):
        """drtmd.py:321-329: everything (refit=True) or only what is neither fitted nor ignored"""
        if refit:
            fit_index = np.arange(self.num_obs)
        else:
            fit_index = np.where(~self.obs_fit_status & ~self.obs_ignore_flag)[0]
        self.fit_observations(fit_index, ignore_errors=ignore_errors)
        return fit_index
