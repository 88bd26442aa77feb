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
