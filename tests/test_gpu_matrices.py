"""GPU parity of the kernel-matrix builders (through the C-ABI) against the oracle and the committed
reference-run fixtures.  Floating-point tolerance: 1e-12 relative (+1e-300 abs) -- the kernels evaluate the
same formulas; only libm (exp/log) and summation order differ from numpy."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

RTOL = 1e-12


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="module")
def ctx():
    from hipdrt import _ffi
    return _ffi.get_context(0)


def test_device_is_gfx950(ctx):
    info = ctx.device_info()
    assert info["arch"] == "gfx950" and info["num_cu"] > 0


@pytest.mark.parametrize("eps", [4.342944819032518, 22.19])
def test_impedance_lookup(ctx, eps):
    from hipdrt.matrices import basis
    from oracle import drt_oracle as orc
    (lre, zre), (lim, zim) = basis.generate_impedance_lookup('gaussian', eps, 2000)
    (olre, ozre), (olim, ozim) = orc.generate_impedance_lookup(eps, 2000)
    np.testing.assert_array_equal(lre, olre)
    np.testing.assert_array_equal(lim, olim)
    np.testing.assert_allclose(zre, ozre, rtol=RTOL, atol=1e-300)
    np.testing.assert_allclose(zim, ozim, rtol=RTOL, atol=1e-300)


def test_lookup_vs_reference_fixture(ctx):
    g = load("refrun_golden71x91.npz")
    zre, zim = ctx.impedance_lookup(float(g["tau_epsilon"]), np.exp(g["lut_log_wt_re"]) * 0 + np.logspace(-2.7, 2.7, 2000),
                                    np.logspace(-5.4, 5.4, 2000))
    np.testing.assert_allclose(zre, g["lut_z_re"], rtol=RTOL, atol=1e-300)
    np.testing.assert_allclose(zim, g["lut_z_im"], rtol=RTOL, atol=1e-300)


def test_impedance_interp_toeplitz_vs_reference_fixture(ctx):
    from hipdrt.matrices import mat1d
    g = load("refrun_golden71x91.npz")
    grids = ((g["lut_log_wt_re"], g["lut_z_re"]), (g["lut_log_wt_im"], g["lut_z_im"]))
    assert mat1d.impedance_matrix_is_toeplitz(g["freq"], g["basis_tau"])
    a_re, a_im = mat1d.construct_impedance_matrices(g["freq"], g["basis_tau"], float(g["tau_epsilon"]), 'interp',
                                                    interpolate_grids=grids)
    np.testing.assert_allclose(a_re, g["zm_re"], rtol=RTOL, atol=1e-300)
    np.testing.assert_allclose(a_im, g["zm_im"], rtol=RTOL, atol=1e-300)
    # single-part signature of the reference
    one = mat1d.construct_impedance_matrix(g["freq"], 'imag', tau=g["basis_tau"], epsilon=float(g["tau_epsilon"]),
                                           integrate_method='interp', interpolate_grids=grids[1])
    np.testing.assert_array_equal(one, a_im)


def test_impedance_interp_general_256x512(ctx):
    """Non-Toeplitz build at the C2 size incl. the clamped region (48 % of Z' entries)."""
    from hipdrt import synth
    from hipdrt.matrices import mat1d
    from oracle import drt_oracle as orc
    c2 = synth.config_c2()
    eps = orc.epsilon_from_tau(c2["tau"])
    grids = orc.generate_impedance_lookup(eps)
    assert not mat1d.impedance_matrix_is_toeplitz(c2["freq"], c2["tau"])
    a_re, a_im = mat1d.construct_impedance_matrices(c2["freq"], c2["tau"], eps, 'interp', interpolate_grids=grids)
    o_re = orc.construct_impedance_matrix(c2["freq"], 'real', c2["tau"], eps, 'interp', interpolate_grids=grids[0])
    o_im = orc.construct_impedance_matrix(c2["freq"], 'imag', c2["tau"], eps, 'interp', interpolate_grids=grids[1])
    np.testing.assert_allclose(a_re, o_re, rtol=RTOL, atol=1e-300)
    np.testing.assert_allclose(a_im, o_im, rtol=RTOL, atol=1e-300)
    assert np.mean(a_re == o_re) > 0.45          # the clamped entries are copies of the table ends: bit-exact
    g = load("refrun_c2_256x512_s0.npz")
    stride = int(g["row_stride"])
    np.testing.assert_allclose(a_re[::stride], g["zm_re"], rtol=RTOL, atol=1e-300)
    np.testing.assert_allclose(a_im[::stride], g["zm_im"], rtol=RTOL, atol=1e-300)


def test_impedance_interp_batched_frequency_grids(ctx):
    """Per-spectrum frequency grids (ragged-in-value, odd tau count exercises the scalar-store path)."""
    from hipdrt.matrices import mat1d
    from oracle import drt_oracle as orc
    rng = np.random.default_rng(5)
    tau = np.logspace(-7, 2, 101)
    eps = orc.epsilon_from_tau(tau)
    grids = orc.generate_impedance_lookup(eps)
    freq = np.sort(10 ** rng.uniform(-2, 6.5, size=(3, 37)), axis=1)[:, ::-1].copy()
    a_re, a_im = mat1d.construct_impedance_matrices(freq, tau, eps, 'interp', interpolate_grids=grids)
    assert a_re.shape == (3, 37, 101)
    for b in range(3):
        o_re = orc.construct_impedance_matrix(freq[b], 'real', tau, eps, 'interp', interpolate_grids=grids[0])
        o_im = orc.construct_impedance_matrix(freq[b], 'imag', tau, eps, 'interp', interpolate_grids=grids[1])
        np.testing.assert_allclose(a_re[b], o_re, rtol=RTOL, atol=1e-300)
        np.testing.assert_allclose(a_im[b], o_im, rtol=RTOL, atol=1e-300)


def test_impedance_interp_irregular_tables(ctx):
    """The general build's interpolation must be numpy's for ANY increasing table: knots jittered off the uniform grid (the
    arithmetic bin is then off by one and the search runs), an odd number of knots, different abscissae for Z' and Z'',
    abscissae beyond both table ends, odd tau count -- and the same table on both sides (one shared bin search)."""
    from hipdrt.matrices import mat1d
    rng = np.random.default_rng(11)
    ng = 1999
    def table(lim):
        x = np.linspace(-lim, lim, ng)
        x[1:-1] += rng.uniform(-0.45, 0.45, ng - 2) * (x[1] - x[0])          # still increasing, visibly non-uniform
        return x, np.cumsum(rng.standard_normal(ng)) * 0.01 + np.sin(x)
    t_re, t_im = table(6.2), table(12.4)
    tau = np.logspace(-6, 3, 75)
    freq = np.sort(10 ** rng.uniform(-4, 8, size=(4, 41)), axis=1)[:, ::-1].copy()      # ln(omega tau) in [-21, 26]
    for grids in ((t_re, t_im), (t_im, t_im)):
        a_re, a_im = mat1d.construct_impedance_matrices(freq, tau, 1.0, 'interp', interpolate_grids=grids)
        for b in range(freq.shape[0]):
            x = np.log(2 * np.pi * freq[b])[:, None] + np.log(tau)[None, :]
            for got, (xp, fp) in ((a_re[b], grids[0]), (a_im[b], grids[1])):
                want = np.interp(x, xp, fp)
                np.testing.assert_allclose(got, want, rtol=0, atol=1e-13 * np.abs(fp).max())
                outside = (x < xp[0]) | (x > xp[-1])
                assert outside.any() and np.array_equal(got[outside], want[outside])      # table ends: copies, bit-exact


@pytest.mark.parametrize("name", ["refrun_trapz_32x64.npz", "refrun_trapz_71x91_toeplitz.npz"])
def test_impedance_trapz_vs_reference_fixture(ctx, name):
    from hipdrt.matrices import mat1d
    g = load(name)
    for part in ("real", "imag"):
        a = mat1d.construct_impedance_matrix(g["freq"], part, tau=g["tau"], epsilon=float(g["eps"]),
                                             integrate_method='trapz')
        np.testing.assert_allclose(a, g[f"A_{part}"], rtol=1e-11, atol=1e-300)


def test_penalty_matrices(ctx):
    from hipdrt.matrices import mat1d
    from oracle import drt_oracle as orc
    g = load("refrun_golden71x91.npz")
    ln_tau = np.log(g["basis_tau"])
    for k in range(3):
        m = mat1d.construct_integrated_derivative_matrix(ln_tau, order=k, epsilon=float(g["tau_epsilon"]))
        np.testing.assert_allclose(m, g[f"m{k}"], rtol=RTOL, atol=1e-300)
    grid = np.sort(np.random.default_rng(1).uniform(-10, 3, 57))     # non-uniform grid: full evaluation
    for k in range(3):
        m = mat1d.construct_integrated_derivative_matrix(grid, order=k, epsilon=3.0)
        np.testing.assert_allclose(m, orc.construct_integrated_derivative_matrix(grid, k, 3.0), rtol=RTOL, atol=1e-300)
    with pytest.raises(ValueError):
        mat1d.construct_integrated_derivative_matrix(grid, order=3, epsilon=3.0)


def test_eis_var_matrix(ctx):
    from hipdrt.matrices import mat1d
    from oracle import drt_oracle as orc
    g = load("refrun_golden71x91.npz")
    np.testing.assert_allclose(mat1d.construct_eis_var_matrix(g["freq"], 0.25, 0.25, None), g["vmm"], rtol=RTOL)
    np.testing.assert_allclose(mat1d.construct_eis_var_matrix(g["freq"], 0.25, 0.25, 'uniform'),
                               orc.construct_eis_var_matrix(g["freq"], 0.25, 0.25, 'uniform'), rtol=RTOL)


# ---- chrono response path (survey row a3): basis.generate_response_lookup / mat1d.construct_response_matrix ----

def test_response_lookup_vs_reference_fixture(ctx):
    from hipdrt.matrices import basis
    g = load("refrun_response.npz")
    for tag in ("eps_grid", "eps_4p34"):
        log_td, v = basis.generate_response_lookup('gaussian', 'galv', 'ideal', float(g[f"lookup_{tag}_eps"]), 2000)
        np.testing.assert_array_equal(log_td, g[f"lookup_{tag}_log_td"])
        # the integrand's 1 - exp(-x) cancels for x -> 0 (entries ~1e-7 at td = 1e-6), so the reference's own values
        # carry an absolute rounding error of ~1e-16; beyond that floor the usual 1e-12 relative holds
        np.testing.assert_allclose(v, g[f"lookup_{tag}_v"], rtol=RTOL, atol=1e-15)


@pytest.mark.parametrize("case", ["one_step", "three_steps", "step_after_end"])
def test_response_matrix_interp_vs_reference_fixture(ctx, case):
    from hipdrt.matrices import mat1d
    g = load("refrun_response.npz")
    grids = (g["lookup_eps_grid_log_td"], g["lookup_eps_grid_v"])
    a, lay = mat1d.construct_response_matrix(g["tau"], g["times"], 'ideal', g[f"{case}_step_times"],
                                             g[f"{case}_step_sizes"], epsilon=float(g["epsilon"]),
                                             integrate_method='interp', interpolate_grids=grids)
    ref_a, ref_l = g[f"{case}_A"], g[f"{case}_layered"]
    assert a.shape == ref_a.shape and lay.shape == ref_l.shape
    # rows at or before a step are exactly zero in that layer
    np.testing.assert_array_equal(lay == 0.0, ref_l == 0.0)
    # abscissa ln((t - t_k)/tau) differs from numpy's by <= 1 ulp of libm's log -> ~1e-13 relative in the interpolant
    np.testing.assert_allclose(lay, ref_l, rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(a, ref_a, rtol=1e-11, atol=1e-18)


def test_response_matrix_trapz_vs_reference_fixture(ctx):
    from hipdrt.matrices import mat1d
    g = load("refrun_response.npz")
    a, lay = mat1d.construct_response_matrix(g["trapz_tau"], g["trapz_times"], 'ideal', g["trapz_step_times"],
                                             g["trapz_step_sizes"], epsilon=float(g["trapz_epsilon"]),
                                             integrate_method='trapz', integrate_points=1000)
    # same cancellation floor as the lookup, scaled by the step size (2e-3)
    np.testing.assert_allclose(lay, g["trapz_layered"], rtol=RTOL, atol=2e-18)
    np.testing.assert_allclose(a, g["trapz_A"], rtol=RTOL, atol=4e-18)


def test_response_matrix_potentiostatic_and_expdecay_vs_reference_fixture(ctx):
    """the non-default forms of construct_response_matrix (mat1d.py:96-118) against the reference's own output: the potentiostatic
    delta-function response (rows before a step exactly zero, the step's own row included from t >= t_step on), the expdecay step
    model's trapezoid integrals with one rise time per step, and the inductance response vector that goes with it"""
    from hipdrt.matrices import mat1d
    g = load("refrun_response.npz")
    tau, times, st, sa = g["trapz_tau"], g["trapz_times"], g["trapz_step_times"], g["trapz_step_sizes"]
    a, lay = mat1d.construct_response_matrix(tau, times, 'ideal', st, sa, epsilon=float(g["trapz_epsilon"]), op_mode='pot')
    np.testing.assert_array_equal(lay == 0.0, g["pot_layered"] == 0.0)
    np.testing.assert_allclose(lay, g["pot_layered"], rtol=1e-13, atol=0)
    np.testing.assert_allclose(a, g["pot_A"], rtol=1e-13, atol=1e-18)
    tr = g["expdecay_tau_rise"]
    a, lay = mat1d.construct_response_matrix(tau, times, 'expdecay', st, sa, epsilon=float(g["trapz_epsilon"]), tau_rise=tr,
                                             integrate_method='trapz', integrate_points=1000)
    np.testing.assert_array_equal(lay == 0.0, g["expdecay_layered"] == 0.0)
    # (the integrand subtracts exponentials of nearly equal size where e^y tau passes tau_rise: absolute floor as for 'trapz')
    np.testing.assert_allclose(lay, g["expdecay_layered"], rtol=1e-10, atol=1e-17)
    np.testing.assert_allclose(a, g["expdecay_A"], rtol=1e-10, atol=1e-17)
    assert np.abs(a - g["trapz_A"]).max() > 1e-3 * np.abs(a).max()            # not the ideal step's matrix
    # tau_rise -> 0 is the ideal step
    a0, _ = mat1d.construct_response_matrix(tau, times, 'expdecay', st, sa, epsilon=float(g["trapz_epsilon"]),
                                            integrate_method='trapz', integrate_points=1000)
    np.testing.assert_allclose(a0, g["trapz_A"], rtol=RTOL, atol=4e-18)
    irv = mat1d.construct_inductance_response_vector(times, 'expdecay', st, sa, tr)
    np.testing.assert_allclose(irv, g["expdecay_inductance_rv"], rtol=1e-14, atol=0)
    assert not mat1d.construct_inductance_response_vector(times, 'ideal', st, sa, None).any()
    with pytest.raises(NotImplementedError):
        mat1d.construct_response_matrix(tau, times, 'ideal', st, sa, integrate_method='quad')


def test_response_matrix_c5_size_vs_oracle(ctx):
    """C5-sized chrono grid (4096 samples x 1024 tau, one step): HIP vs the CPU restatement on a row subset, and
    size-independent properties: linearity in the step size, zero rows before the step, monotone saturation."""
    from hipdrt.matrices import basis, mat1d
    from oracle import drt_oracle as orc
    tau = np.logspace(-7, 3, 1024)
    eps = 1 / np.mean(np.diff(np.log(tau)))
    t_step = 0.05
    times = np.concatenate([t_step - 5e-4 * np.arange(96, 0, -1) + 5e-4, t_step + np.logspace(-4, np.log10(50), 4000)])
    grids = basis.generate_response_lookup('gaussian', 'galv', 'ideal', eps, 2000)
    a, lay = mat1d.construct_response_matrix(tau, times, 'ideal', [t_step], [1e-3], epsilon=eps,
                                             integrate_method='interp', interpolate_grids=grids)
    assert a.shape == (4096, 1024)
    np.testing.assert_array_equal(a, lay[0])
    assert not a[:96].any()
    rows = np.r_[90:100, 500:4096:397]
    oa, _ = orc.construct_response_matrix(tau, times[rows], [t_step], [1e-3], eps, 'interp', interpolate_grids=grids)
    np.testing.assert_allclose(a[rows], oa, rtol=1e-11, atol=1e-300)
    a2, _ = mat1d.construct_response_matrix(tau, times, 'ideal', [t_step], [-2e-3], epsilon=eps,
                                            integrate_method='interp', interpolate_grids=grids)
    np.testing.assert_allclose(a2, -2.0 * a, rtol=1e-15, atol=0)
    assert np.all(np.diff(a[96:], axis=0) >= -1e-18)          # response grows with time ...
    assert np.all(np.diff(a[96:], axis=1) <= 1e-18)           # ... and is smaller for slower basis functions


def test_response_matrix_rejects_unbuilt_branches(ctx):
    from hipdrt.matrices import mat1d
    with pytest.raises(ValueError):
        mat1d.construct_response_matrix([1.0], [1.0], 'ideal', [0.0], [1.0], integrate_method='interp')
    with pytest.raises(ValueError):
        mat1d.construct_response_matrix([1.0], [1.0], 'bogus', [0.0], [1.0])
    with pytest.raises(NotImplementedError):
        mat1d.construct_response_matrix([1.0], [1.0], 'ideal', [0.0], [1.0], integrate_method='quad')
    with pytest.raises(NotImplementedError):
        mat1d.construct_response_matrix([1.0], [1.0], 'ideal', [0.0], [1.0], basis_type='Cole-Cole')


@pytest.mark.parametrize("case", ["one_step", "three_steps"])
def test_chrono_var_matrix_vs_reference_fixture(ctx, case):
    """survey row a5 (chrono half): Gaussian in transformed time, block diagonal per step, rows normalised."""
    from hipdrt.matrices import mat1d
    g = load("refrun_response.npz")
    vmm = mat1d.construct_chrono_var_matrix(g["times"], g[f"{case}_step_times"], 0.25, None)
    ref = g[f"{case}_vmm"]
    np.testing.assert_array_equal(vmm == 0.0, ref == 0.0)            # no correlation across steps
    np.testing.assert_allclose(vmm, ref, rtol=RTOL, atol=1e-300)
    np.testing.assert_allclose(vmm.sum(axis=1), 1.0, rtol=1e-14)
    uni = mat1d.construct_chrono_var_matrix(g["times"], g["one_step_step_times"], 0.25, 'uniform')
    np.testing.assert_array_equal(uni, g["uniform_vmm"])


def test_phasance_matrices_vs_reference_fixture(ctx):
    """survey row a18: distribution-of-phasances columns (complex erf on the device, A&S 7.1.29)."""
    from hipdrt.matrices import phasance
    g = load("refrun_response.npz")
    nu, eps = g["dop_nu"], float(g["dop_epsilon"])
    zm = phasance.construct_phasor_z_matrix(g["dop_freq"], nu, 'gaussian', eps)
    ref = g["dop_zm"]
    # each entry is a difference of two erf values times a prefactor of up to 2e4: relative to the row's largest entry
    np.testing.assert_allclose(zm, ref, rtol=1e-11, atol=1e-13 * np.abs(ref).max())
    # a wide basis (eps = 2): both erf values sit near +-1 and their difference loses digits in the reference as well
    # (each erf carries ~1e-16 absolute error); 6e-11 observed
    zm2 = phasance.construct_phasor_z_matrix(g["dop_freq"], nu, 'gaussian', 2.0)
    np.testing.assert_allclose(zm2, g["dop_zm_eps2"], rtol=1e-9, atol=1e-13 * np.abs(g["dop_zm_eps2"]).max())
    vm, vlay = phasance.construct_phasor_v_matrix(g["times"], nu, 'gaussian', eps, 'ideal', g["three_steps_step_times"],
                                                  g["three_steps_step_sizes"])
    np.testing.assert_array_equal(vlay == 0.0, g["dop_vm_layered"] == 0.0)
    np.testing.assert_allclose(vlay, g["dop_vm_layered"], rtol=1e-11, atol=1e-13 * np.abs(g["dop_vm_layered"]).max())
    np.testing.assert_allclose(vm, g["dop_vm"], rtol=1e-11, atol=1e-13 * np.abs(g["dop_vm"]).max())
    np.testing.assert_array_equal(phasance.phasor_scale_vector(nu, g["tau"]), g["dop_scale"])
