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
    """Batches in flight that served `num_obs` C2-size observations best on one MI355X (tools/probe_inflight.py, fits/s with
    1 / 2 / 3 / 4 batches: 1250 obs. 1631 / 1814 / 1718 / 1473; 2500: 1792 / 1954 / 1971 / 1738; 10 000: 1994 / 2035 / 2053 /
    1986): a batch should keep more than ~500 spectra, and more than three host threads get in each other's way."""
    return 1 if num_obs < 512 else (2 if num_obs < 2000 else 3)


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


def _fit_observations_inflight(drt, frequencies, z_obs, inflight, tau_supergrid, drt_var, ignore_errors, llh_kw, fit_kw):
    """The observations in `inflight` contiguous chunks, each chunk one device batch on its own sibling plan, the host
    threads overlapping their device loops: spectra finish after 4 ... 50 outer iterations, so one batch alone leaves
    CUs idle in its tail, several side by side fill them ('batches in flight')."""
    import threading
    num = z_obs.shape[0]
    chunks = [c for c in np.array_split(np.arange(num), inflight) if len(c)]
    sibs = drt_siblings(drt, len(chunks))
    outs, errs = [None] * len(chunks), [None] * len(chunks)

    def work(i):
        try:
            outs[i] = fit_observations(sibs[i], frequencies, z_obs[chunks[i]], tau_supergrid=tau_supergrid, drt_var=drt_var,
                                       ignore_errors=True, llh_kw=llh_kw, **fit_kw)
        except BaseException as exc:            # re-raised in the caller's thread
            errs[i] = exc

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(chunks))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for exc in errs:
        if exc is not None:
            raise exc
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
    if not ignore_errors and not res['obs_fit_status'].all():
        bad = int(np.flatnonzero(~res['obs_fit_status'])[0])
        print(f"Error encountered at obs_index {bad}")
        raise res['obs_fit_errors'][bad]
    return obs_x, obs_special, res


def fit_observations(drt, frequencies, z_obs, tau_supergrid=None, drt_var=False, ignore_errors=True, llh_kw=None,
                     inflight=1, **fit_kw):
    """Fit every observation (rows of z_obs) and scatter the coefficients into supergrid slots like
    DRTMD.fit_observation does (drtmd.py:245-301): returns obs_x (B, len(supergrid)), obs_special dict, and the raw
    result dict, which also carries what the reference keeps per observation:
      obs_llh, obs_rss       DRT.evaluate_llh(**llh_kw) / evaluate_rss() of every fit (drtmd.py:259-260);
      obs_tau_indices        (left, right) supergrid slots of the basis grid (drtmd.py:262-267);
      obs_fit_status         True where the fit succeeded; obs_fit_errors: None or the exception the reference would have
                             raised for that observation (cvxopt's ValueError at a singular start point), which
                             ``ignore_errors=False`` raises for the first failed observation as upstream does
                             (drtmd.py:292-301);
      obs_drt_var(+_ok)      with ``drt_var=True``: diagonal of estimate_distribution_cov(tau=tau_supergrid,
                             extend_var=True) (drtmd.py:278-279).
    ``inflight`` > 1 fits the observations as that many batches side by side (sibling plans of `drt`, one host thread
    each): same results, in the same order, at the throughput of several batches in flight; 'auto' = auto_inflight(number
    of observations).  (Afterwards `drt` itself holds the first batch only.)"""
    z_obs = np.asarray(z_obs)
    inflight = auto_inflight(z_obs.shape[0]) if inflight == 'auto' else int(inflight)
    if inflight > 1 and z_obs.shape[0] >= 2 * inflight:
        return _fit_observations_inflight(drt, frequencies, z_obs, int(inflight), tau_supergrid, drt_var, ignore_errors,
                                          llh_kw, fit_kw)
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
    if not ignore_errors and not ok.all():
        bad = int(np.flatnonzero(~ok)[0])
        print(f"Error encountered at obs_index {bad}")
        raise errors[bad]
    obs_x = np.zeros((num, len(tau_supergrid)))
    obs_x[:, left:right] = np.where(ok[:, None], res['fit_x'], 0.0)
    obs_special = {'R_inf': np.where(ok, res['R_inf'], 0.0), 'inductance': np.where(ok, res['inductance'], 0.0)}
    llh, rss = drt.evaluate_obs_llh_rss_batch(**(llh_kw or {}))
    res['obs_llh'], res['obs_rss'] = np.where(ok, llh, 0.0), np.where(ok, rss, 0.0)
    res['obs_tau_indices'] = (left, right)
    res['obs_fit_status'], res['obs_fit_errors'] = ok, errors
    if drt_var:
        res['obs_drt_var'], res['obs_drt_var_ok'] = drt.estimate_distribution_var_batch(tau=tau_supergrid,
                                                                                       extend_var=True)
    return obs_x, obs_special, res


_GATHER_KEYS = ('obs_llh', 'obs_rss', 'outer_iters', 'qp_iters_total', 'status')


def fit_observations_sharded(drt, frequencies, z_obs, rank=None, world=None, tau_supergrid=None, scheme='interleave',
                             drt_var=False, dst=0, fit=fit_observations, inflight=1, **fit_kw):
    """BASELINE configs[3]: the observations of one map sharded over the ranks of a node (one process per GPU), every
    rank fitting its share in one device batch, the results gathered on rank `dst` with ONE collective.

    Every rank calls this with the same (frequencies, z_obs) -- or at least with its own rows valid -- and its own `drt`.
    Returns on `dst` the same triple as fit_observations for ALL observations in their original order (result dict
    reduced to the per-observation arrays obs_llh, obs_rss, outer_iters, qp_iters_total, status [, obs_drt_var]); None
    elsewhere.  `scheme`: see shard_indices ('lpt' uses difficulty_proxy(z_obs)).  `fit` is the per-rank fit function
    (the CPU tests inject a stand-in); `inflight` > 1 is handed to it (batches side by side on every rank)."""
    from . import dist as hd
    if rank is None or world is None:
        import torch.distributed as tdist
        rank = tdist.get_rank() if tdist.is_initialized() else 0
        world = tdist.get_world_size() if tdist.is_initialized() else 1
    z_obs = np.asarray(z_obs)
    num = z_obs.shape[0]
    cost = difficulty_proxy(z_obs) if scheme == 'lpt' else None
    owned = [shard_indices(num, world, r, scheme, cost) for r in range(world)]
    mine = owned[rank]
    if len(mine):
        if inflight != 1:
            fit_kw = dict(fit_kw, inflight=inflight)
        obs_x, obs_special, res = fit(drt, frequencies, z_obs[mine], tau_supergrid=tau_supergrid, drt_var=drt_var, **fit_kw)
        cols = [obs_x, obs_special['R_inf'][:, None], obs_special['inductance'][:, None]]
        cols += [np.asarray(res[k], dtype=float)[:, None] for k in _GATHER_KEYS]
        if drt_var:
            cols += [res['obs_drt_var'], np.asarray(res['obs_drt_var_ok'], dtype=float)[:, None]]
        packed = np.concatenate(cols, axis=1)
    else:
        packed = None
    # ranks without observations still take part in the collective: the row width comes from a rank that has some
    width = hd.max_over_ranks(0 if packed is None else packed.shape[1])
    if packed is None:
        packed = np.zeros((0, int(width)))
    counts = [len(o) for o in owned]
    full = hd.gather_rows(packed, counts, dst=dst)
    if rank != dst:
        return None
    order = np.concatenate(owned) if world > 1 else mine
    out = np.empty_like(full)
    out[order] = full                                    # back to the original observation order
    # row = [obs_x (nsup) | R_inf | inductance | _GATHER_KEYS | obs_drt_var (nsup) | ok]  (the last two with drt_var)
    nsup = (out.shape[1] - 2 - len(_GATHER_KEYS) - (1 if drt_var else 0)) // (2 if drt_var else 1)
    obs_x = out[:, :nsup]
    pos = nsup
    obs_special = {'R_inf': out[:, pos], 'inductance': out[:, pos + 1]}
    pos += 2
    res = {}
    for k in _GATHER_KEYS:
        res[k] = out[:, pos] if k in ('obs_llh', 'obs_rss') else out[:, pos].astype(np.int64)
        pos += 1
    if drt_var:
        res['obs_drt_var'] = out[:, pos:pos + nsup]
        res['obs_drt_var_ok'] = out[:, pos + nsup] > 0.5
    res['obs_fit_status'] = res['status'] >= 0
    return obs_x, obs_special, res
