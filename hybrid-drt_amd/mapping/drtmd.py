"""Batch driver: the fit-loop contract of hybdrt.mapping.drtmd.DRTMD (drtmd.py:186-329, 1136-1158) for
observations that share one frequency grid, with optional sharding over ranks (one process per GPU)."""
import numpy as np


def shard_bounds(num_obs, world_size, rank):
    """Contiguous block split of the observations over ranks: ceil-sized leading blocks."""
    base, rem = divmod(num_obs, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def difficulty_proxy(z_obs):
    """Cheap stand-in for how many outer iterations a spectrum will take, known before any fit: the relative roughness of
    the spectrum (second differences along frequency against its span).  Noisier spectra run longer hyper-parameter
    loops (25 ... 50 outer iterations at C2 size), smooth ones converge early."""
    z = np.asarray(z_obs)
    d2 = np.abs(np.diff(z, n=2, axis=1)).mean(axis=1)
    span = np.abs(z - z.mean(axis=1, keepdims=True)).max(axis=1)
    return d2 / np.maximum(span, 1e-300)


def shard_indices(num_obs, world_size, rank, scheme='interleave', cost=None):
    """Observation indices owned by `rank`.
    'block'      contiguous blocks (shard_bounds);
    'interleave' rank r takes r, r + W, r + 2W, ... : neighbouring observations of a map (similar difficulty) are dealt
                 to different ranks, and every rank gets ceil/floor(num_obs / W) of them;
    'lpt'        longest-processing-time-first on `cost` (one value per observation, e.g. difficulty_proxy): observations
                 sorted by descending cost, dealt in a serpentine so that the per-rank cost sums stay level and the counts
                 differ by at most one.
    The union over ranks is a permutation of range(num_obs) for every scheme."""
    if scheme == 'block':
        a, b = shard_bounds(num_obs, world_size, rank)
        return np.arange(a, b)
    if scheme == 'interleave':
        return np.arange(rank, num_obs, world_size)
    if scheme == 'lpt':
        if cost is None:
            raise ValueError("scheme 'lpt' needs one cost per observation")
        order = np.argsort(-np.asarray(cost, dtype=float), kind='stable')
        pos = np.arange(num_obs)
        rnd, lane = pos // world_size, pos % world_size
        owner = np.where(rnd % 2 == 0, lane, world_size - 1 - lane)     # serpentine: 0..W-1, W-1..0, ...
        return np.sort(order[owner == rank])
    raise ValueError(f"unknown sharding scheme {scheme!r}")


_NOT_PER_OBS = ('basis_tau', 'timings_ms', 'launches', 'obs_tau_indices', 'obs_fit_errors')


def auto_inflight(num_obs):
    """Plans in flight that serve `num_obs` C2-size observations of ONE rank best.  Round 6: always ONE -- the plan cuts its batch
    into ranges on its own streams (hipdrt_plan_set_subbatches, automatic: four from 1000 spectra on under the loader's
    GPU_MAX_HW_QUEUES=8), which now beats every count of sibling plans with one plan's memory and one caller thread
    (tools/probe_inflight_ranges.py, profiles/r06_inflight_ranges.txt, fits/s through fit_observations for 1 plan x 4 ranges /
    2 plans / 3 plans: 1250 observations 2371 / 2326 / 2263, 2500: 2547 / 2515 / 2459, 10 000: 2651 / 2633 / 2616).  Rounds 3-5
    answered 2 up to 2000 observations and 3 above (four hardware queues: profiles/r04_subbatch_sweep.txt); `inflight=k` is
    still there for callers whose environment pins fewer queues."""
    return 1


def drt_siblings(drt, count):
    """`count` DRT objects with the configuration of `drt` (itself first), each sibling with its own hipdrt context (HIP
    stream) and plan, so that their device loops run side by side; cached on `drt`."""
    import copy
    from .. import _ffi
    clones = getattr(drt, '_sibling_clones', None) or []     # (the clones only: no reference cycle through `drt`)
    while len(clones) < count - 1:
        c = copy.copy(drt)                     # configuration only: the arrays it refers to are read-only
        c._plan = c._plan_key = c._last_batch = None
        c._context = _ffi.Context(drt.device)
        c._sibling_clones = None
        clones.append(c)
    drt._sibling_clones = clones
    return ([drt] + clones)[:count]


_RANGES_PER_INFLIGHT_PLAN = 1      # (tools/probe_inflight_ranges.py sweeps it)


def _fit_observations_inflight(drt, frequencies, z_obs, inflight, tau_supergrid, drt_var, ignore_errors, llh_kw, rss_kw, fit_kw):
    """The observations in `inflight` contiguous chunks, each chunk one device batch on its own sibling plan, the host
    threads overlapping their device loops: spectra finish after 4 ... 50 outer iterations, so one batch alone leaves
    CUs idle in its tail, several side by side fill them ('batches in flight')."""
    import threading
    num = z_obs.shape[0]
    chunks = [c for c in np.array_split(np.arange(num), inflight) if len(c)]
    sibs = drt_siblings(drt, len(chunks))
    # several plans side by side already fill the tails of each other's launches: every plan fits its chunk as ONE range
    # (cutting each of them again inside the library loses: profiles/r04_subbatch_sweep.txt) -- for the duration of this
    # call only: the DRT objects get their own setting back below, so that a later one-plan fit on `drt` sub-batches again
    before = [getattr(sib, 'plan_subbatches', 0) for sib in sibs]

    def pin(sib, k):
        sib.plan_subbatches = k
        if getattr(sib, '_plan', None) is not None:
            sib._plan.set_subbatches(k)

    for sib in sibs:
        pin(sib, _RANGES_PER_INFLIGHT_PLAN)
        sib.collect_fields = getattr(drt, 'collect_fields', None)
    outs, errs = [None] * len(chunks), [None] * len(chunks)

    def work(i):
        try:
            outs[i] = fit_observations(sibs[i], frequencies, z_obs[chunks[i]], tau_supergrid=tau_supergrid, drt_var=drt_var,
                                       ignore_errors=True, llh_kw=llh_kw, rss_kw=rss_kw, **fit_kw)
        except BaseException as exc:            # re-raised in the caller's thread
            errs[i] = exc

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(chunks))]
    try:
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        for sib, k in zip(sibs, before):
            pin(sib, k)
    for exc in errs:
        if exc is not None:
            raise exc
    return _merge_chunk_results(outs, chunks, ignore_errors)


def _merge_chunk_results(outs, chunks, ignore_errors):
    """results of fit_observations on consecutive chunks of one map -> the result for the whole map"""
    obs_x = np.concatenate([o[0] for o in outs])
    obs_special = {k: np.concatenate([o[1][k] for o in outs]) for k in outs[0][1]}
    res = {}
    for k, v in outs[0][2].items():
        if k in _NOT_PER_OBS:
            res[k] = v
        elif isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == len(chunks[0]):
            res[k] = np.concatenate([o[2][k] for o in outs])
        else:
            res[k] = v
    res['obs_fit_errors'] = [e for o in outs for e in o[2]['obs_fit_errors']]
    _raise_first_error(res, ignore_errors)
    return obs_x, obs_special, res


def max_batch_for(drt, frequencies):
    """How many spectra of this shape ONE device batch may hold: 80 % of the device's memory over what a staged spectrum costs the
    plan (hipdrt_plan_bytes_per_spectrum: 4.9 MB at 256 x 512, i.e. about 47 000 spectra on 288 GB).  None when the DRT object
    cannot say (stand-ins in the CPU tests)."""
    try:
        from .. import _ffi
        ctx = drt._context if getattr(drt, '_context', None) is not None else _ffi.get_context(drt.device)
        if drt.fixed_basis_tau is not None:
            tau = drt.fixed_basis_tau
        else:
            from .. import preprocessing as pp
            tau = pp.get_basis_tau(np.asarray(frequencies, dtype=float), None, None, tau_grid=drt.tau_supergrid,
                                   extend_decades=drt.extend_basis_decades)
        ns = int(bool(drt.fit_ohmic)) + int(bool(drt.fit_inductance))
        per = ctx.plan_bytes_per_spectrum(len(frequencies), len(tau), ns)
        return max(1, int(0.8 * ctx.device_info()['hbm_bytes'] / per))
    except (AttributeError, TypeError):
        return None


def _raise_first_error(res, ignore_errors):
    """fit_observation's error contract (drtmd.py:292-301): without ignore_errors the first failed observation raises"""
    if not ignore_errors and not np.all(res['obs_fit_status']):
        bad = int(np.flatnonzero(~np.asarray(res['obs_fit_status']))[0])
        print(f"Error encountered at obs_index {bad}")
        raise res['obs_fit_errors'][bad]


def _metric_kw(llh_kw, rss_kw):
    """DRTMD.__init__ (drtmd.py:121-134): both keyword sets default to normalize=True, weights='uniform'"""
    llh_kw, rss_kw = dict(llh_kw or {}), dict(rss_kw or {})
    for kw in (llh_kw, rss_kw):
        kw.setdefault('normalize', True)
        kw.setdefault('weights', 'uniform')
    return llh_kw, rss_kw


def _supergrid_slots(tau_supergrid, basis_tau):
    """drtmd.py:262-267 (utils.array.nearest_index on both ends of the basis grid)"""
    left = int(np.argmin(np.abs(np.log(tau_supergrid) - np.log(basis_tau[0]))))
    right = int(np.argmin(np.abs(np.log(tau_supergrid) - np.log(basis_tau[-1])))) + 1
    return left, right


def observation_groups(observations, tags=None):
    """Observations that can share one device plan: same data type (EIS / chrono / joint), same frequency grid, same sample
    times and current signal -- hence the same matrices and the same slice of the tau supergrid (and the same `tags[k]`, when
    tags are given: what else must agree inside a group).  Returns a list of (kind, [observation indices]) in order of first
    appearance."""
    groups, order = {}, []
    for k, (chrono, eis) in enumerate(observations):
        has_c = chrono is not None and chrono[0] is not None
        has_e = eis is not None and eis[0] is not None
        if not (has_c or has_e):
            raise ValueError(f'observation {k} has neither chrono nor EIS data')
        key = ('hybrid' if has_c and has_e else 'eis' if has_e else 'chrono',)
        if has_c:
            key += (np.asarray(chrono[0], dtype=float).tobytes(), np.asarray(chrono[1], dtype=float).tobytes())
        if has_e:
            key += (np.asarray(eis[0], dtype=float).tobytes(),)
        if tags is not None:
            key += (tags[k],)
        if key not in groups:
            groups[key] = []
            order.append(key)
        groups[key].append(k)
    return [(key[0], groups[key]) for key in order]


_FILTER_KW = ('remove_extremes', 'extreme_kw', 'remove_outliers', 'outlier_thresh')


def _as_measurement(observation):
    chrono, eis = observation
    chrono = chrono if (chrono is not None and chrono[0] is not None) else (None, None, None)
    eis = eis if (eis is not None and eis[0] is not None) else (None, None)
    return chrono[0], chrono[1], chrono[2], eis[0], eis[1]


def _as_observation(meas):
    return (None if meas[0] is None else (meas[0], meas[1], meas[2])), (None if meas[3] is None else (meas[3], meas[4]))


def prefilter_observations(drt, observations, fit_kw):
    """``remove_extremes`` / ``remove_outliers`` among the fit keywords (drt1d.py:187-302) for a LIST of observations.  Both
    change the SIZE of an observation's data, so they run before the batches are formed: the quantile-range pre-filter per
    observation on the host, then the outlier detection -- an initialize_weights-only pass with the outlier-aware weights -- as
    one device batch per group of like observations; every observation loses its own flagged points and the fit keywords come
    back without the two switches (and without outlier_p, as upstream's refit).  Returns (observations, fit_kw, tags, step_times):
    tags[k] = (detection group, which points were dropped) keeps observations apart that must not share a plan, step_times[k] =
    the step times the detection pass found before the removal (drt1d.py:283-302; None without chrono data)."""
    fit_kw = dict(fit_kw)
    num = len(observations)
    tags, step_times = None, [None] * num
    meas = [_as_measurement(o) for o in observations]
    if fit_kw.get('remove_extremes'):
        meas = [drt._drop_extremes(m, fit_kw.get('extreme_kw')) for m in meas]
    if fit_kw.get('remove_outliers'):
        if fit_kw.get('outlier_p') is None:
            raise ValueError('If remove_outliers is True, the prior probability of outlier presence, outlier_p, '
                             'must be specified. A good starting value might be 0.01-0.05')           # drt1d.py:215-218
        pass_kw = dict(fit_kw, remove_extremes=False)
        ckw, _ = drt._split_kwargs(pass_kw)
        tags = [None] * num
        for g, (_, idx) in enumerate(observation_groups([_as_observation(m) for m in meas])):
            cleaned, st, masks = drt._remove_outliers_batch([meas[k] for k in idx], pass_kw, ckw)
            for k, c, (cm, em) in zip(idx, cleaned, masks):
                meas[k], step_times[k] = c, st
                tags[k] = (g, b'' if cm is None else np.packbits(cm).tobytes(), b'' if em is None else np.packbits(em).tobytes())
        fit_kw['outlier_p'] = None
    for key in _FILTER_KW:
        fit_kw.pop(key, None)
    return [_as_observation(m) for m in meas], fit_kw, tags, step_times


def fit_observation_list(drt, observations, tau_supergrid, drt_var=False, ignore_errors=False, llh_kw=None, rss_kw=None,
                         **fit_kw):
    """DRTMD.fit_observations (drtmd.py:245-319) for ANY mix of observations, each given as DRTMD.add_observation takes it:
    (chrono_data, eis_data) with chrono_data = (times, i_signal, v_signal) or None and eis_data = (frequencies, z) or None.
    Observations are grouped by (data type, sampling grids) on the host (observation_groups); every group is ONE device
    batch -- an EIS plan (fit_eis_batch) or a prepared-matrix plan (joint / chrono fits) -- whose results are scattered
    into the observation's own slice of the tau supergrid.  Returns (obs_x, obs_special, res) like fit_observations, with
    per observation: obs_tau_indices (list of (left, right)), obs_group (index into res['groups']), obs_llh, obs_rss,
    obs_fit_status / obs_fit_errors, outer_iters, qp_iters_total, status [, obs_drt_var, obs_drt_var_ok]."""
    tau_supergrid = np.asarray(tau_supergrid, dtype=float)
    num = len(observations)
    llh_kw, rss_kw = _metric_kw(llh_kw, rss_kw)
    obs_x = np.zeros((num, len(tau_supergrid)))
    obs_special = {}
    res = dict(obs_llh=np.zeros(num), obs_rss=np.zeros(num), obs_tau_indices=[None] * num, obs_group=np.zeros(num, dtype=int),
               obs_fit_status=np.zeros(num, dtype=bool), obs_fit_errors=[None] * num, outer_iters=np.zeros(num, dtype=np.int64),
               qp_iters_total=np.zeros(num, dtype=np.int64), status=np.zeros(num, dtype=np.int64), groups=[])
    if drt_var:
        res['obs_drt_var'] = np.zeros((num, len(tau_supergrid)))
        res['obs_drt_var_ok'] = np.zeros(num, dtype=bool)
    tags, step_times = None, [None] * num
    if fit_kw.get('remove_extremes') or fit_kw.get('remove_outliers'):
        observations, fit_kw, tags, step_times = prefilter_observations(drt, observations, fit_kw)
    for g, (kind, idx) in enumerate(observation_groups(observations, tags)):
        idx = np.asarray(idx)
        if kind == 'eis':
            freq = np.asarray(observations[idx[0]][1][0], dtype=float)
            out = drt.fit_eis_batch(freq, np.array([observations[k][1][1] for k in idx]), **fit_kw)
            special_keys = [key for key in ('R_inf', 'inductance') if key in out]
        else:
            meas = [_as_measurement(observations[k]) for k in idx]
            group_kw = fit_kw
            if step_times[idx[0]] is not None:          # found before the outliers were removed (drt1d.py:283-302)
                group_kw = dict(fit_kw, step_times=step_times[idx[0]], step_sizes=None)
            out = drt._fit_prepared_batch(meas, group_kw)
            special_keys = [key for key in drt.special_qp_params if key in out]
        basis_tau = out['basis_tau']
        left, right = _supergrid_slots(tau_supergrid, basis_tau)
        ok = np.asarray(out['status']) >= 0
        obs_x[idx, left:right] = np.where(ok[:, None], out['fit_x'], 0.0)
        for key in special_keys:
            val = np.asarray(out[key], dtype=float)
            if key not in obs_special:          # (initialize_obs_special / the "key is new" branch of drtmd.py:281-285)
                obs_special[key] = np.zeros((num,) + val.shape[1:])
            obs_special[key][idx] = np.where(ok.reshape((-1,) + (1,) * (val.ndim - 1)), val, 0.0)
        llh, rss = drt.evaluate_obs_llh_rss_batch(llh_kw=llh_kw, rss_kw=rss_kw)
        res['obs_llh'][idx], res['obs_rss'][idx] = np.where(ok, llh, 0.0), np.where(ok, rss, 0.0)
        res['obs_fit_status'][idx] = ok
        res['obs_group'][idx] = g
        for key in ('outer_iters', 'qp_iters_total', 'status'):
            res[key][idx] = out[key]
        for j, k in enumerate(idx):
            res['obs_tau_indices'][k] = (left, right)
            if not ok[j]:
                res['obs_fit_errors'][k] = ValueError("Rank(A) < p or Rank([P; A; G]) < n")
        if drt_var:
            var, vok = drt.estimate_distribution_var_batch(tau=tau_supergrid, extend_var=True)
            vok = np.asarray(vok, dtype=bool) & ok
            res['obs_drt_var'][idx] = np.where(vok[:, None], var, 0.0)
            res['obs_drt_var_ok'][idx] = vok
        res['groups'].append(dict(kind=kind, indices=idx, basis_tau=basis_tau, tau_indices=(left, right)))
    _raise_first_error(res, ignore_errors)
    return obs_x, obs_special, res


def fit_observations_pfrt(drt, observations, tau_supergrid, pfrt_factors=None, drt_var=False, ignore_errors=False,
                          llh_kw=None, rss_kw=None, **fit_kw):
    """DRTMD with fit_type='pfrt' (drtmd.py:98-100, 1136-1158, 1338-1342): every observation through _pfrt_fit_core
    (drt1d.py:2558-2700) -- one full fit at the first regularisation factor, one warm restart per further factor -- with one
    solution PER FACTOR recorded.  Factors: ``pfrt_factors`` (or ``factors=`` among the fit keywords, which is where upstream
    reads them); None = logspace(-1, 1, 11), _pfrt_fit_core's own default, which is what an upstream DRTMD actually fits with:
    its ``pfrt_factors`` attribute says logspace(-0.7, 0.7, 11) (drtmd.py:98-100) but its fit_kw never receives them
    (fit_kw is a plain attribute, the setter at drtmd.py:1490-1500 is not bound), checked against its own run
    (tests/golden/refrun_drtmd_pfrt6.npz).  Recorded: obs_x (num, S, len(supergrid)), every special
    parameter (num, S[, size]).  Grouped and batched like fit_observation_list (one device plan per sampling grid).  What the
    reference takes from the drt1d object after the PFRT run refers to its FIRST step: obs_llh / obs_rss are evaluate_llh /
    evaluate_rss of the first step's last iterate (qphb_history[-1], drt1d.py:4433-4496) and obs_drt_var the variance of the
    first step's P matrix (fit_parameters['p_matrix']); the same here.  The reference's own DRTMD stops at a joint observation
    (v_baseline arrives as (S, 1) where (S,) was allocated, drtmd.py:287); here such specials are stored (num, S, size).
    Also returned per observation: step_llh (num, S), step_iters (num, S)."""
    tau_supergrid = np.asarray(tau_supergrid, dtype=float)
    if 'factors' in fit_kw:
        pfrt_factors = fit_kw.pop('factors')
    factors = np.logspace(-1, 1, 11) if pfrt_factors is None else np.asarray(pfrt_factors, dtype=float)
    num, S, nsup = len(observations), len(factors), len(tau_supergrid)
    llh_kw, rss_kw = _metric_kw(llh_kw, rss_kw)
    obs_x = np.zeros((num, S, nsup))
    obs_special = {}
    res = dict(obs_llh=np.zeros(num), obs_rss=np.zeros(num), obs_tau_indices=[None] * num, obs_group=np.zeros(num, dtype=int),
               obs_fit_status=np.zeros(num, dtype=bool), obs_fit_errors=[None] * num, status=np.zeros(num, dtype=np.int64),
               step_llh=np.zeros((num, S)), step_iters=np.zeros((num, S), dtype=np.int64), pfrt_factors=factors, groups=[])
    if drt_var:
        res['obs_drt_var'] = np.zeros((num, S, nsup))
        res['obs_drt_var_ok'] = np.zeros(num, dtype=bool)
    pf_kw = {k: fit_kw.pop(k) for k in ('max_iter_per_step', 'max_init_iter', 'xtol', 'nonneg') if k in fit_kw}
    for g, (kind, idx) in enumerate(observation_groups(observations)):
        idx = np.asarray(idx)
        first = {}

        def after_init(out, first=first):
            # the drt1d object the reference reads from still describes the first step's fit at this point
            first['llh'], first['rss'] = drt.evaluate_obs_llh_rss_batch(llh_kw=llh_kw, rss_kw=rss_kw)
            if drt_var:
                first['var'], first['vok'] = drt.estimate_distribution_var_batch(tau=tau_supergrid, extend_var=True)

        if kind == 'eis':
            freq = np.asarray(observations[idx[0]][1][0], dtype=float)
            pr = drt.pfrt_fit_eis_batch(freq, np.array([observations[k][1][1] for k in idx]), factors=factors,
                                        after_init=after_init, **pf_kw, **fit_kw)
            basis_tau, cs = pr['basis_tau'], pr['coefficient_scale']
            ns = drt._plan.ns
            sp = drt.special_qp_params
            fx = pr['step_x'][:, :, ns:] * cs[None, :, None]                               # (S, B, ntau)
            specials = {}
            if 'R_inf' in sp:
                specials['R_inf'] = pr['step_x'][:, :, sp['R_inf']['index']] * cs[None, :]
            if 'inductance' in sp:
                specials['inductance'] = pr['step_x'][:, :, sp['inductance']['index']] * cs[None, :] * drt.inductance_scale
        else:
            meas = []
            for k in idx:
                chrono, eis = observations[k]
                eis = eis if (eis is not None and eis[0] is not None) else (None, None)
                meas.append((chrono[0], chrono[1], chrono[2], eis[0], eis[1]))
            preps, out, hypers, kw2, ckw = drt._pfrt_prepared(meas, factors, pf_kw.get('max_iter_per_step', 10),
                                                              pf_kw.get('max_init_iter', 20), pf_kw.get('xtol', 1e-2),
                                                              pf_kw.get('nonneg', True), dict(fit_kw), after_init=after_init)
            pr = drt.pfrt_result
            basis_tau = preps[0]['basis_tau']
            fps = [[drt._extract(prep, pr['step_x'][s_, b], out['weights'][b], kw2, ckw) for b, prep in enumerate(preps)]
                   for s_ in range(S)]
            fx = np.array([[fp['x'] for fp in row] for row in fps])
            specials = {key: np.array([[np.asarray(fp[key], dtype=float) for fp in row] for row in fps])
                        for key in drt.special_qp_params if key in fps[0][0]}
        left, right = _supergrid_slots(tau_supergrid, basis_tau)
        ok = np.asarray(pr['status']) >= 0
        obs_x[idx, :, left:right] = np.where(ok[:, None, None], np.swapaxes(fx, 0, 1), 0.0)
        for key, val in specials.items():
            val = np.swapaxes(val, 0, 1)                                                  # (B, S[, size])
            if key not in obs_special:
                obs_special[key] = np.zeros((num,) + val.shape[1:])
            obs_special[key][idx] = np.where(ok.reshape((-1,) + (1,) * (val.ndim - 1)), val, 0.0)
        res['obs_llh'][idx], res['obs_rss'][idx] = np.where(ok, first['llh'], 0.0), np.where(ok, first['rss'], 0.0)
        res['step_llh'][idx], res['step_iters'][idx] = pr['step_llh'].T, pr['step_iters'].T
        res['obs_fit_status'][idx] = ok
        res['status'][idx] = pr['status']
        res['obs_group'][idx] = g
        for j, k in enumerate(idx):
            res['obs_tau_indices'][k] = (left, right)
            if not ok[j]:
                res['obs_fit_errors'][k] = ValueError("Rank(A) < p or Rank([P; A; G]) < n")
        if drt_var:
            vok = np.asarray(first['vok'], dtype=bool) & ok
            res['obs_drt_var'][idx] = np.where(vok[:, None, None], first['var'][:, None, :], 0.0)   # one variance, every factor
            res['obs_drt_var_ok'][idx] = vok
        res['groups'].append(dict(kind=kind, indices=idx, basis_tau=basis_tau, tau_indices=(left, right)))
    _raise_first_error(res, ignore_errors)
    return obs_x, obs_special, res


def fit_observations(drt, frequencies=None, z_obs=None, tau_supergrid=None, drt_var=False, ignore_errors=False, llh_kw=None,
                     rss_kw=None, inflight=1, observations=None, fit_type='drt', pfrt_factors=None, max_batch=None, **fit_kw):
    """Fit every observation and scatter the coefficients into supergrid slots like DRTMD.fit_observation does
    (drtmd.py:245-301): returns obs_x (B, len(supergrid)), obs_special dict, and the raw result dict, which also
    carries what the reference keeps per observation:
      obs_llh, obs_rss       DRT.evaluate_llh(**llh_kw) / evaluate_rss(**rss_kw) of every fit (drtmd.py:259-260), both
                             keyword sets with DRTMD's defaults normalize=True, weights='uniform' (drtmd.py:121-134);
      obs_tau_indices        (left, right) supergrid slots of the basis grid (drtmd.py:262-267);
      obs_fit_status         True where the fit succeeded; obs_fit_errors: None or the exception the reference would have
                             raised for that observation (cvxopt's ValueError at a singular start point).  As upstream
                             (``ignore_errors=False``, drtmd.py:245, 292-301) the first failed observation raises;
                             with ``ignore_errors=True`` failed observations keep zeros everywhere;
      obs_drt_var(+_ok)      with ``drt_var=True``: diagonal of estimate_distribution_cov(tau=tau_supergrid,
                             extend_var=True) (drtmd.py:278-279).
    Two call forms: ``(frequencies, z_obs)`` = impedance spectra on one shared frequency grid (rows of z_obs), or
    ``observations=[(chrono_data, eis_data), ...]`` = any mix of data types and grids (fit_observation_list; needs
    `tau_supergrid`).  ``inflight`` > 1 (shared-grid form) fits the observations as that many batches side by side
    (sibling plans of `drt`, one host thread each): same results, in the same order, at the throughput of several batches
    in flight; 'auto' = auto_inflight(number of observations).  (Afterwards `drt` itself holds the first batch only.)
    A shared-grid map that does not fit the device at once is fitted as consecutive batches of nearly equal size through the same
    plan (``max_batch`` spectra at most; default max_batch_for(drt, frequencies): 80 % of the device's memory, about 47 000 spectra
    of 256 x 512 on 288 GB) -- same results, same order."""
    if fit_type not in ('drt', 'pfrt'):
        raise ValueError(f"Invalid fit_type {fit_type}. Options: ['drt', 'pfrt']")          # drtmd.py:1479-1482
    if fit_type == 'pfrt':
        # DRTMD(fit_type='pfrt') (drtmd.py:98-100, 1338-1342): one solution per regularisation factor and observation
        if observations is None:
            frequencies, z_obs = np.asarray(frequencies, dtype=float), np.asarray(z_obs)
            observations = [(None, (frequencies, zb)) for zb in z_obs]
            if tau_supergrid is None:
                tau_supergrid = drt.fixed_basis_tau if drt.fixed_basis_tau is not None else drt.tau_supergrid
        if tau_supergrid is None:
            raise ValueError("fit_type='pfrt' needs tau_supergrid")
        return fit_observations_pfrt(drt, observations, tau_supergrid, pfrt_factors=pfrt_factors, drt_var=drt_var,
                                     ignore_errors=ignore_errors, llh_kw=llh_kw, rss_kw=rss_kw, **fit_kw)
    if observations is not None:
        if tau_supergrid is None:
            raise ValueError('a heterogeneous observation list needs tau_supergrid')
        return fit_observation_list(drt, observations, tau_supergrid, drt_var=drt_var, ignore_errors=ignore_errors,
                                    llh_kw=llh_kw, rss_kw=rss_kw, **fit_kw)
    z_obs = np.asarray(z_obs)
    inflight = auto_inflight(z_obs.shape[0]) if inflight == 'auto' else int(inflight)
    if inflight > 1 and z_obs.shape[0] >= 2 * inflight:
        return _fit_observations_inflight(drt, frequencies, z_obs, int(inflight), tau_supergrid, drt_var, ignore_errors,
                                          llh_kw, rss_kw, fit_kw)
    limit = max_batch if max_batch is not None else max_batch_for(drt, frequencies)
    if limit is not None and z_obs.shape[0] > limit:
        # a map that does not fit the device at once: consecutive batches of (nearly) equal size through the same plan
        parts = -(-z_obs.shape[0] // limit)
        chunks = [c for c in np.array_split(np.arange(z_obs.shape[0]), parts) if len(c)]
        outs = [fit_observations(drt, frequencies, z_obs[c], tau_supergrid=tau_supergrid, drt_var=drt_var, ignore_errors=True,
                                 llh_kw=llh_kw, rss_kw=rss_kw, inflight=1, max_batch=limit, **fit_kw) for c in chunks]
        return _merge_chunk_results(outs, chunks, ignore_errors)
    llh_kw, rss_kw = _metric_kw(llh_kw, rss_kw)
    res = drt.fit_eis_batch(frequencies, z_obs, **fit_kw)
    num = z_obs.shape[0]
    basis_tau = res['basis_tau']
    if tau_supergrid is None:
        tau_supergrid = basis_tau
    tau_supergrid = np.asarray(tau_supergrid)
    left = int(np.argmin(np.abs(np.log(tau_supergrid) - np.log(basis_tau[0]))))
    right = left + len(basis_tau)
    ok = np.asarray(res['status']) >= 0
    errors = [None if good else ValueError("Rank(A) < p or Rank([P; A; G]) < n") for good in ok]
    res['obs_fit_status'], res['obs_fit_errors'] = ok, errors
    _raise_first_error(res, ignore_errors)
    obs_x = np.zeros((num, len(tau_supergrid)))
    obs_x[:, left:right] = np.where(ok[:, None], res['fit_x'], 0.0)
    obs_special = {'R_inf': np.where(ok, res['R_inf'], 0.0), 'inductance': np.where(ok, res['inductance'], 0.0)}
    llh, rss = drt.evaluate_obs_llh_rss_batch(llh_kw=llh_kw, rss_kw=rss_kw)
    res['obs_llh'], res['obs_rss'] = np.where(ok, llh, 0.0), np.where(ok, rss, 0.0)
    res['obs_tau_indices'] = (left, right)
    if drt_var:
        var, vok = drt.estimate_distribution_var_batch(tau=tau_supergrid, extend_var=True)
        vok = np.asarray(vok, dtype=bool) & ok
        res['obs_drt_var'], res['obs_drt_var_ok'] = np.where(vok[:, None], var, 0.0), vok     # (failed fits keep zeros)
    return obs_x, obs_special, res


_GATHER_KEYS = ('obs_llh', 'obs_rss', 'outer_iters', 'qp_iters_total', 'status')
# every special parameter a fit can report (x layout of drt1d.py:377-408); the gathered rows name them by position here
_SPECIAL_REGISTRY = ('v_baseline', 'vz_offset', 'R_inf', 'inductance', 'C_inv', 'x_dop')


def share_lookup_tables(drt, rank, world, src=0, force=False):
    """SURVEY 8e / north_star: the lookup tables of the shared tau basis (impedance Z', Z'' and the chrono response, 3 x
    2000 doubles) are built once, on rank `src`, and broadcast to the other ranks (RCCL over xGMI on a GPU node) -- ONE
    collective per (DRT instance, epsilon, world, src), not per map: the first call of a DRT instance broadcasts and marks
    the instance (and, through copy, its sibling clones); later calls return False without communicating.  Whether a
    rank takes part is decided from state every rank shares by construction -- tau_epsilon is fixed in DRT.__init__ from the
    constructor arguments, the mark is set by this very function -- so ranks that call the sharded driver with DRT
    instances of the same history agree (an SPMD program does; a rank that swaps in a fresh DRT for a later map must do so on
    every rank, or pass force=True everywhere).  Non-`src` ranks install the received tables BEFORE their plans exist where
    they can (DRT._get_plan hands them to the new plan), so nothing is built twice on them; a plan that already exists is
    re-pointed at the received tables (one matrix rebuild)."""
    from . import dist as hd
    if not (world > 1 or hd.forced()) or drt.tau_epsilon is None or drt.integrate_method != 'interp':
        return False
    key = (float(drt.tau_epsilon), int(world), int(src))
    if not force and getattr(drt, '_lut_shared_key', None) == key:
        return False
    shapes = [(len(drt._wt_re),), (len(drt._wt_im),), (2000,)]
    if rank == src:
        z_re, z_im, resp = drt.lookup_tables()
    else:
        z_re, z_im, resp = (np.zeros(shp) for shp in shapes)
    z_re, z_im, resp = hd.broadcast_arrays([z_re, z_im, resp], src=src)
    if rank != src:
        drt.install_lookup_tables(z_re, z_im, resp)
    drt._lut_shared_key = key
    for clone in getattr(drt, '_sibling_clones', None) or []:
        if rank != src:
            clone.install_lookup_tables(z_re, z_im, resp)
        clone._lut_shared_key = key
    return True


def _pack_rows(obs_x, obs_special, res, drt_var):
    """One rank's results as rows of doubles behind ONE header row that describes them, so that `dst` can unpack blocks from
    ranks whose fits reported other special parameters (or none at all) without a second collective:
        header = [nsup, drt_var, n_specials, (registry index, width, ndim) x n_specials, 0 ...]
        row    = [obs_x (nsup) | specials at their real widths | _GATHER_KEYS | left, right | obs_drt_var (nsup), ok]"""
    num, nsup = obs_x.shape
    unknown = [k for k in obs_special if k not in _SPECIAL_REGISTRY]
    if unknown:
        raise NotImplementedError(f'special parameters {unknown} are not known to the sharded driver')
    cols, head = [obs_x], [float(nsup), float(bool(drt_var)), 0.0]
    for ki, key in enumerate(_SPECIAL_REGISTRY):
        if obs_special.get(key) is None:
            continue
        raw = np.asarray(obs_special[key], dtype=float)
        val = raw.reshape(num, -1)
        cols.append(val)
        head += [float(ki), float(val.shape[1]), float(raw.ndim)]
        head[2] += 1
    cols += [np.asarray(res[k], dtype=float)[:, None] for k in _GATHER_KEYS]
    ti = res.get('obs_tau_indices', (0, nsup))
    ti = np.array(ti, dtype=float) if isinstance(ti, list) else np.tile(np.array(ti, dtype=float), (num, 1))
    cols.append(ti)
    if drt_var:
        cols += [res['obs_drt_var'], np.asarray(res['obs_drt_var_ok'], dtype=float)[:, None]]
    body = np.concatenate(cols, axis=1)
    width = max(body.shape[1], len(head))
    packed = np.zeros((num + 1, width))
    packed[0, :len(head)] = head
    packed[1:, :body.shape[1]] = body
    return packed


def _unpack_block(block):
    """inverse of _pack_rows for one rank's block (header row first): (obs_x, {special: (2-d array, ndim of the original)}, {key: column}, ti, var, vok)"""
    head, body = block[0], block[1:]
    nsup, drt_var, nsp = int(head[0]), bool(head[1]), int(head[2])
    obs_x = body[:, :nsup]
    pos = nsup
    special = {}
    for j in range(nsp):
        key, w, nd = _SPECIAL_REGISTRY[int(head[3 + 3 * j])], int(head[4 + 3 * j]), int(head[5 + 3 * j])
        special[key] = (body[:, pos:pos + w], nd)
        pos += w
    cols = {}
    for k in _GATHER_KEYS:
        cols[k] = body[:, pos]
        pos += 1
    ti = body[:, pos:pos + 2]
    pos += 2
    var = vok = None
    if drt_var:
        var, vok = body[:, pos:pos + nsup], body[:, pos + nsup] > 0.5
    return obs_x, special, cols, ti, var, vok


def _one_kernel(drt, members, saved=None):
    """pin (members = 0: one workgroup per problem, the batch kernel) the coneqp kernel choice of `drt` and its sibling clones,
    or give back what each context held before (``saved`` = the list a pinning call returned).  Only plans made under ANOTHER
    choice are dropped (their scratch layout follows the kernel); the contexts are the instances' own -- an instance on the
    process-wide default context gets a private one first, so that no other DRT object or thread sees the switch."""
    from .. import _ffi
    held = []
    for k, d in enumerate([drt] + list(getattr(drt, '_sibling_clones', None) or [])):
        if d._context is None:                       # never switch the shared default context under other users' feet
            d._context = _ffi.Context(d.device)
            if getattr(d, '_plan', None) is not None:
                d._plan.close()
                d._plan = d._plan_key = None
        ctx = d._context
        before = getattr(ctx, '_qp_group_override', -1)
        want = members if saved is None else saved[k]
        held.append(before)
        if want != before:
            ctx.debug_qp_group(want)
            ctx._qp_group_override = want
            if getattr(d, '_plan', None) is not None:
                d._plan.close()
                d._plan = d._plan_key = None
    return held


def fit_observations_sharded(drt, frequencies=None, z_obs=None, rank=None, world=None, tau_supergrid=None, scheme='interleave',
                             drt_var=False, dst=0, fit=fit_observations, inflight=1, observations=None, ignore_errors=False,
                             reproducible=False, **fit_kw):
    """BASELINE configs[3]: the observations of one map sharded over the ranks of a node (one process per GPU), every
    rank fitting its share as device batches, the results gathered on rank `dst` with ONE collective per map (a gather).
    The lookup tables of rank `dst` are broadcast ONCE per DRT instance (share_lookup_tables: the first map of an instance
    pays one more collective, later maps none); only when some rank owns no observation at all does the row width have to
    be agreed by one more (scalar) all-reduce.

    Every rank calls this with the same (frequencies, z_obs) -- or at least with its own rows valid -- or the same
    ``observations`` list (any mix of data types and grids, see fit_observations), and its own `drt`.  Returns on `dst` the
    same triple as fit_observations for ALL observations in their original order (result dict reduced to the
    per-observation arrays obs_llh, obs_rss, outer_iters, qp_iters_total, status [, obs_tau_indices, obs_drt_var]); None
    elsewhere.  Special parameters travel at their real widths (x_dop: one column per basis_nu point, v_baseline: one per
    coefficient): a key missing on some rank is zero-filled there, as DRTMD's initialize_obs_special does for observations
    that do not report it.  `scheme`: see shard_indices ('lpt' uses difficulty_proxy(z_obs); shared-grid form only).  `fit`
    is the per-rank fit function (the CPU tests inject a stand-in); `inflight` > 1 is handed to it (batches side by side on
    every rank).  Every rank fits with ignore_errors=True, so that all of them reach the collective; a failed observation then
    raises on `dst`, after the gather, unless ``ignore_errors``.
    A fit's bits depend (at the 1e-14 level) on whether its device batch held more or fewer than #CUs / 16 spectra (two QP
    kernels): a map re-sharded over another number of ranks reproduces to that level, not bit for bit --
    unless ``reproducible=True``, which keeps every device batch of this call on the one-workgroup-per-problem kernel whatever
    its size (n <= 2048; small shares then fit slower, ~1.2 x for a single spectrum): the gathered map is then bit-identical
    for every world size, shard scheme and `inflight`."""
    from . import dist as hd
    if reproducible:
        if fit is not fit_observations:
            raise ValueError("reproducible=True pins the device kernel of mapping.fit_observations; it cannot do that for another `fit`")
        saved = _one_kernel(drt, 0)
        try:
            return fit_observations_sharded(drt, frequencies, z_obs, rank=rank, world=world, tau_supergrid=tau_supergrid,
                                            scheme=scheme, drt_var=drt_var, dst=dst, fit=fit, inflight=inflight,
                                            observations=observations, ignore_errors=ignore_errors, **fit_kw)
        finally:
            _one_kernel(drt, 0, saved)
    if rank is None or world is None:
        rank, world = hd.get_rank(), hd.get_world_size()
    general = observations is not None
    if general:
        num = len(observations)
        if scheme == 'lpt':
            raise ValueError("scheme 'lpt' needs the shared-grid form (one z_obs array)")
        cost = None
    else:
        z_obs = np.asarray(z_obs)
        num = z_obs.shape[0]
        cost = difficulty_proxy(z_obs) if scheme == 'lpt' else None
    owned = [shard_indices(num, world, r, scheme, cost) for r in range(world)]
    mine = owned[rank]
    if fit is fit_observations:
        share_lookup_tables(drt, rank, world, src=dst)
    if len(mine):
        kw = dict(fit_kw, ignore_errors=True)
        if inflight != 1:
            kw['inflight'] = inflight
        if general:
            obs_x, obs_special, res = fit(drt, tau_supergrid=tau_supergrid, drt_var=drt_var,
                                          observations=[observations[k] for k in mine], **kw)
        else:
            # the gather below carries obs_x, the special parameters, llh / rss and the counts: nothing else is downloaded
            lean = fit is fit_observations
            before = getattr(drt, 'collect_fields', None)
            if lean:
                drt.collect_fields = 'map'
            try:
                obs_x, obs_special, res = fit(drt, frequencies, z_obs[mine], tau_supergrid=tau_supergrid, drt_var=drt_var, **kw)
            finally:
                if lean:
                    drt.collect_fields = before
        packed = _pack_rows(obs_x, obs_special, res, drt_var)
    else:
        packed = None
    # blocks: one header row + the rank's observations; a rank without observations sends nothing
    counts = [len(o) + 1 if len(o) else 0 for o in owned]
    if min(counts) == 0 and world > 1:
        # the row width is a function of the fit's configuration; a rank that fitted nothing learns it from the others
        width = int(hd.max_over_ranks(0 if packed is None else packed.shape[1]))
        if packed is None:
            packed = np.zeros((0, width))
        elif packed.shape[1] < width:            # (ranks whose fits report different specials: pad to the widest row)
            packed = np.pad(packed, ((0, 0), (0, width - packed.shape[1])))
    elif packed is None:
        packed = np.zeros((0, 1))
    elif (world > 1 or hd.forced()) and general:
        # heterogeneous lists: ranks may report different sets of special parameters, i.e. rows of different widths
        width = int(hd.max_over_ranks(packed.shape[1]))
        if packed.shape[1] < width:
            packed = np.pad(packed, ((0, 0), (0, width - packed.shape[1])))
    full = hd.gather_rows(packed, counts, dst=dst)
    if rank != dst:
        return None
    if num == 0:
        raise ValueError('no observations')
    blocks, pos = [], 0
    for r in range(world):
        if counts[r]:
            blocks.append((owned[r], _unpack_block(full[pos:pos + counts[r]])))
            pos += counts[r]
    nsup = blocks[0][1][0].shape[1]
    obs_x = np.zeros((num, nsup))
    obs_special, res = {}, {}
    for k in _GATHER_KEYS:
        res[k] = np.zeros(num) if k in ('obs_llh', 'obs_rss') else np.zeros(num, dtype=np.int64)
    tis = np.zeros((num, 2), dtype=np.int64)
    if drt_var:
        res['obs_drt_var'], res['obs_drt_var_ok'] = np.zeros((num, nsup)), np.zeros(num, dtype=bool)
    for idx, (bx, bspecial, bcols, bti, bvar, bvok) in blocks:
        obs_x[idx] = bx
        for key, (val, nd) in bspecial.items():
            if key not in obs_special:
                obs_special[key] = np.zeros((num, val.shape[1]) if nd > 1 else (num,))
            obs_special[key][idx] = val if nd > 1 else val[:, 0]
        for k in _GATHER_KEYS:
            res[k][idx] = bcols[k] if k in ('obs_llh', 'obs_rss') else bcols[k].astype(np.int64)
        tis[idx] = bti.astype(np.int64)
        if drt_var:
            res['obs_drt_var'][idx], res['obs_drt_var_ok'][idx] = bvar, bvok
    # (shapes as fit_observations returns them: (num,) for scalar specials, (num, width) for vector-valued ones)
    res['obs_tau_indices'] = [(int(a_), int(b_)) for a_, b_ in tis]
    res['obs_fit_status'] = res['status'] >= 0
    res['obs_fit_errors'] = [None if good else ValueError("Rank(A) < p or Rank([P; A; G]) < n") for good in res['obs_fit_status']]
    _raise_first_error(res, ignore_errors)
    return obs_x, obs_special, res
