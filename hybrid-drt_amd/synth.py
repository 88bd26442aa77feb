"""Synthetic EIS spectra used by bench.py, the tests and the golden generator.

Two-ZARC circuit with series inductance and proportional Gaussian noise; numpy only, seeded, so the same
inputs can be regenerated on the GPU box and in the build container.
"""
import numpy as np

BASE = dict(r_inf=1.0, r1=1.0, tau1=1e-3, beta1=0.8, r2=0.5, tau2=1e-1, beta2=0.9, induc=1e-7, sigma=0.002)


def zarc2_spectrum(freq, seed=0, jitter=False, **overrides):
    """Z(f) = Rinf + R1/(1+(jwt1)^b1) + R2/(1+(jwt2)^b2) + jwL, plus sigma*|Z|*(N+jN) noise.

    ``jitter=True`` draws log-normal perturbations of (R1, R2, tau1, tau2) from ``default_rng(10000+seed)``
    (batch members); the noise always comes from ``default_rng(seed)``.
    """
    p = dict(BASE)
    p.update(overrides)
    if jitter:
        jr = np.random.default_rng(10_000 + seed)
        g = jr.standard_normal(4)
        p["r1"] *= np.exp(0.2 * g[0])
        p["r2"] *= np.exp(0.2 * g[1])
        p["tau1"] *= np.exp(0.5 * g[2])
        p["tau2"] *= np.exp(0.5 * g[3])
    freq = np.asarray(freq, dtype=float)
    jw = 1j * 2 * np.pi * freq
    z = (p["r_inf"] + p["r1"] / (1 + (jw * p["tau1"]) ** p["beta1"])
         + p["r2"] / (1 + (jw * p["tau2"]) ** p["beta2"]) + jw * p["induc"])
    rng = np.random.default_rng(seed)
    noise = rng.standard_normal(len(freq)) + 1j * rng.standard_normal(len(freq))
    return z + p["sigma"] * np.abs(z) * noise


def zarc2_batch(freq, batch, first_seed=0):
    """(batch, nf) complex array; member b uses seed first_seed+b with parameter jitter."""
    return np.stack([zarc2_spectrum(freq, seed=first_seed + b, jitter=True) for b in range(batch)])


# Named configurations of BASELINE.json / SURVEY.md section 8d
def config_c1():
    return dict(freq=np.logspace(6, -1, 71), tau=np.logspace(-9, 3, 121))


def config_c2():
    return dict(freq=np.logspace(6, -1, 256), tau=np.logspace(-8, 2, 512))


def hybrid_measurement(seed=0, n_pre=24, n_post=200, nf=41, f_hi=1e5, f_lo=1e1, t_step=0.05, dt_pre=5e-4,
                       t_lo=1e-4, t_hi=50.0, i_step=1e-3, v0=0.1, v_noise=2e-6, n_rc=57, jitter=False, extra_steps=(), c_series=None, uniform_dt=None):
    """Joint time/frequency-domain measurement of the same 2-ZARC cell (SURVEY.md section 8d, config 5 family):
    a galvanostatic step of ``i_step`` at ``t_step`` (n_pre uniform samples before it, n_post log-uniform after it)
    whose voltage comes from the closed-form response of an RC (Debye) discretisation of the two ZARCs, plus the
    impedance spectrum of :func:`zarc2_spectrum` on ``logspace(f_hi, f_lo, nf)``.

    ``jitter=True`` perturbs (R1, R2, tau1, tau2) exactly as :func:`zarc2_spectrum` does for batch members, in both
    data sets.  ``extra_steps`` = ((delay, current change), ...) appends further current steps, each followed by its own
    n_post log-uniform samples (superposition of the RC responses).  ``c_series`` adds a series (blocking) capacitance to
    both data sets.  ``uniform_dt`` samples every segment uniformly instead (a raw potentiostat record: n_post samples at
    that period).  Returns (times, i_signal, v_signal, freq, z).
    """
    p = dict(BASE)
    if jitter:
        g = np.random.default_rng(10_000 + seed).standard_normal(4)
        p["r1"] *= np.exp(0.2 * g[0])
        p["r2"] *= np.exp(0.2 * g[1])
        p["tau1"] *= np.exp(0.5 * g[2])
        p["tau2"] *= np.exp(0.5 * g[3])
    rng = np.random.default_rng(50_000 + seed)
    freq = np.logspace(np.log10(f_hi), np.log10(f_lo), nf)
    z = zarc2_spectrum(freq, seed, jitter=jitter)
    if c_series is not None:
        z = z + 1.0 / (1j * 2 * np.pi * freq * c_series)
    pre = t_step - (uniform_dt if uniform_dt is not None else dt_pre) * np.arange(n_pre, 0, -1)
    steps = [(t_step, i_step)]
    for delay, di in extra_steps:
        steps.append((steps[-1][0] + delay, di))
    segs = [pre]
    for k, (ts, _) in enumerate(steps):
        t_end = t_hi if k == len(steps) - 1 else 0.999 * (steps[k + 1][0] - ts)
        segs.append(ts + (uniform_dt * np.arange(1, n_post + 1) if uniform_dt is not None
                          else np.logspace(np.log10(t_lo), np.log10(t_end), n_post)))
    times = np.concatenate(segs)
    i_signal = np.zeros(len(times))
    for ts, di in steps:
        i_signal += np.where(times >= ts, di, 0.0)
    lt = np.linspace(-6.0, 1.0, n_rc)
    taus = 10.0 ** lt

    def gamma(r, t0, beta):   # Cole-Cole distribution of relaxation times, integrated over one ln(tau) cell
        u = np.log(taus / t0)
        g = r / (2 * np.pi) * np.sin((1 - beta) * np.pi) / (np.cosh(beta * u) - np.cos((1 - beta) * np.pi))
        return g * np.log(10.0) * (lt[1] - lt[0])

    rk = gamma(p["r1"], p["tau1"], p["beta1"]) + gamma(p["r2"], p["tau2"], p["beta2"])
    v = np.zeros(len(times))
    for ts, di in steps:
        post = times >= ts
        dt = times[post] - ts
        v[post] += di * (p["r_inf"] + (rk[None, :] * (1.0 - np.exp(-dt[:, None] / taus[None, :]))).sum(axis=1))
        if c_series is not None:
            v[post] += di * dt / c_series
    v_signal = v + v0 + v_noise * rng.standard_normal(len(times))
    return times, i_signal, v_signal, freq, z
