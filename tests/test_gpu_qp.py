"""GPU parity of the weighted normal equations and the batched coneqp kernel against the oracle and the QPs
captured from the reference run (P, q, h -> x, iterations)."""
import os
import time

import numpy as np
import pytest

from conftest import GOLDEN, parity

pytestmark = pytest.mark.gpu


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="module")
def ctx():
    from hipdrt import _ffi
    return _ffi.get_context(0)


@pytest.mark.parametrize("m,n", [(7, 5), (64, 64), (142, 93), (512, 514), (100, 131)])
def test_weighted_gram_layout_and_values(ctx, m, n):
    """Asymmetric integer-valued operands: exact in FP64, catches any swapped MFMA fragment / C-D map."""
    rng = np.random.default_rng(m * 1000 + n)
    A = rng.integers(-4, 5, size=(m, n)).astype(float)
    w = rng.integers(1, 4, size=(3, m)).astype(float)
    b = rng.integers(-3, 4, size=(3, m)).astype(float)
    l2 = rng.integers(-2, 3, size=(n, n)).astype(float)
    l2 = l2 + l2.T
    l1 = rng.integers(0, 3, size=n).astype(float)
    P, q = ctx.weighted_gram(A, w, b, l2=l2, l1=l1)
    for i in range(3):
        wa = w[i][:, None] * A
        np.testing.assert_array_equal(P[i], wa.T @ wa + l2)
        np.testing.assert_array_equal(q[i], -wa.T @ (w[i] * b[i]) + l1)


def test_weighted_gram_float_vs_numpy(ctx):
    g = load("refrun_golden71x91.npz")
    rm, rv = g["rm"], g["rv"]
    w = np.stack([g["est_weights"], g["weights"]])
    P, q = ctx.weighted_gram(rm, w, np.stack([rv, rv]))
    for i in range(2):
        wa = w[i][:, None] * rm
        np.testing.assert_allclose(P[i], wa.T @ wa, rtol=1e-13, atol=1e-13 * np.abs(wa.T @ wa).max())
        np.testing.assert_allclose(q[i], -wa.T @ (w[i] * rv), rtol=1e-12, atol=1e-12 * np.abs(q[i]).max())


@pytest.mark.parametrize("name", ["refrun_golden71x91.npz", "refrun_golden71x91_neg.npz", "refrun_c1_71x121.npz"])
def test_qp_vs_reference_run(ctx, name):
    """Every QP of the reference trajectories: same iteration count, x within 1e-7 relative of the peak."""
    g = load(name)
    nqp = len(g["qp_iterations"])
    P = np.stack([g[f"qp{i}_P"] for i in range(nqp)])
    q = np.stack([g[f"qp{i}_q"] for i in range(nqp)])
    h = np.stack([g[f"qp{i}_h"] for i in range(nqp)])
    res = ctx.qp_batch(P, q, h)
    assert res["status"].tolist() == [0] * nqp
    assert res["iterations"].tolist() == g["qp_iterations"].tolist()
    for i in range(nqp):
        ref = g[f"qp{i}_x"]
        parity("x", res["x"][i], ref, default=1e-7)
        assert abs(res["pcost"][i] - float(g[f"qp{i}_pcost"])) <= 1e-9 * abs(float(g[f"qp{i}_pcost"]))


@pytest.mark.parametrize("n", [1, 2, 5, 16, 17, 31, 32, 33, 48, 49, 64, 65, 80, 96, 97, 100, 129, 257, 400, 480, 497, 512, 513,
                               514, 527, 528, 529, 600])
def test_qp_random_spd_vs_oracle(ctx, n):
    from oracle.coneqp import coneqp_boxlow
    rng = np.random.default_rng(n)
    B = 3
    Ps, qs, hs = [], [], []
    for b in range(B):
        M = rng.standard_normal((n + 3, n))
        Ps.append(M.T @ M + 0.1 * np.eye(n))
        qs.append(rng.standard_normal(n) * 3)
        hs.append(np.where(rng.random(n) < 0.8, 0.0, 1000.0))
    res = ctx.qp_batch(np.stack(Ps), np.stack(qs), np.stack(hs))
    for b in range(B):
        ref = coneqp_boxlow(Ps[b], qs[b], hs[b])
        assert res["iterations"][b] == ref["iterations"]
        np.testing.assert_allclose(res["x"][b], ref["x"], rtol=1e-8, atol=1e-9 * max(1.0, np.abs(ref["x"]).max()))


def test_qp_shared_p_and_h(ctx):
    from oracle.coneqp import coneqp_boxlow
    rng = np.random.default_rng(0)
    n = 40
    M = rng.standard_normal((60, n))
    P = M.T @ M
    q = rng.standard_normal((5, n))
    h = np.zeros(n)
    res = ctx.qp_batch(P, q, h)
    for b in range(5):
        ref = coneqp_boxlow(P, q[b], h)
        assert res["iterations"][b] == ref["iterations"]
        np.testing.assert_allclose(res["x"][b], ref["x"], rtol=1e-8, atol=1e-10)


def test_qp_singular_start_point_reports_status(ctx):
    """cvxopt raises ValueError when P + G'G is not positive definite at the start point; the batch API reports
    it per problem and the solve_convex_opt mirror raises."""
    from hipdrt.models import qphb
    n = 6
    P = -2.0 * np.eye(n)
    res = ctx.qp_batch(np.stack([P, np.eye(n)]), np.zeros((2, n)) - 1.0, np.zeros(n))
    assert res["status"][0] == -1 and res["status"][1] == 0
    with pytest.raises(ValueError):
        qphb.solve_convex_opt(np.zeros(3), np.zeros((3, n)), P, 0.0, True, {})


def test_solve_convex_opt_mirror(ctx):
    """Same call the reference makes at qphb.py:673 (first outer iteration of the golden case)."""
    from hipdrt.models import qphb
    from oracle import drt_oracle as orc
    g = load("refrun_golden71x91.npz")
    rm, rv, w = g["rm"], g["rv"], g["est_weights"]
    special = {'R_inf': {'index': 0, 'nonneg': True, 'size': 1}, 'inductance': {'index': 1, 'nonneg': True, 'size': 1}}
    pen = []
    for k in range(3):
        mk = np.zeros((93, 93)); mk[0, 0] = mk[1, 1] = 1e-6; mk[2:, 2:] = g[f"m{k}"]; pen.append(mk)
    l2 = orc.calculate_qp_l2_matrix(orc.get_default_hypers(), np.ones(3), pen, [np.ones(93)] * 3, 2)
    l1 = np.zeros(93)
    out = qphb.solve_convex_opt(w * rv, w[:, None] * rm, l2, l1, True, special)
    ref = g["qp1_x"]
    assert out['iterations'] == int(g["qp_iterations"][1]) and out['status'] == 'optimal'
    parity("x", np.array(list(out['x'])), ref, default=1e-7)


@pytest.mark.parametrize("n", [529, 564, 641, 700, 1078, 1500, 2048])
def test_large_problems_on_the_tile_packed_kernel(n):
    """n > 528: the tile-packed kernel with its inverse diagonal blocks in global memory (qp_kernel_resident<true>).  Against the CPU checker: same iteration count, x within 1e-9 of the peak; members of
    the batch bit-identical to the same problem solved in a batch of another size."""
    from hipdrt import _ffi
    from oracle.coneqp import coneqp_boxlow
    rng = np.random.default_rng(n)
    ctx = _ffi.get_context()
    B = 4
    Ps, qs = [], []
    for b in range(B):
        A = rng.standard_normal((n + 50, n)) / np.sqrt(n)
        xt = np.maximum(rng.standard_normal(n), 0)
        Ps.append(A.T @ A + 1e-3 * np.eye(n))
        qs.append(-A.T @ (A @ xt))
    Ps, qs = np.array(Ps), np.array(qs)
    h = np.zeros(n)
    h[:3] = 1000.0
    res = ctx.qp_batch(Ps, qs, h)
    assert np.all(res["status"] == 0)
    if n <= 1078:
        for b in (0, B - 1):
            r = coneqp_boxlow(Ps[b], qs[b], h)
            assert r["iterations"] == res["iterations"][b]
            parity("x", res["x"][b], r["x"], default=1e-9)
    res6 = ctx.qp_batch(np.concatenate([Ps, Ps[:2]]), np.concatenate([qs, qs[:2]]), h)
    np.testing.assert_array_equal(res6["x"][:4], res["x"])
    np.testing.assert_array_equal(res6["x"][4:], res["x"][:2])


def test_large_problem_reports_breakdown():
    """an indefinite P breaks the factorisation down (n > 528 form of the kernel): status < 0, as for small problems"""
    from hipdrt import _ffi
    ctx = _ffi.get_context()
    n = 800
    P = np.eye(n)
    P[300, 300] = -5.0
    res = ctx.qp_batch(P[None], np.ones((1, n)), np.zeros(n))
    assert res["status"][0] < 0


def test_concurrent_large_launches_from_two_streams():
    """two host threads, two contexts (HIP streams), each launching batches of n = 700 problems: the launches interleave on
    the device without disturbing each other (per-problem scratch only) and give identical results"""
    import threading
    from hipdrt import _ffi
    rng = np.random.default_rng(0)
    n, B = 700, 12
    A = rng.standard_normal((n + 20, n)) / np.sqrt(n)
    P = A.T @ A + 1e-3 * np.eye(n)
    q = -A.T @ (A @ np.maximum(rng.standard_normal(n), 0))
    out = {}

    def work(tag):
        ctx = _ffi.Context(0)
        for rep in range(4):
            out[tag] = ctx.qp_batch(np.tile(P, (B, 1, 1)), np.tile(q, (B, 1)), np.zeros(n))
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
        assert not t.is_alive(), "QP launches hung"
    np.testing.assert_array_equal(out[0]["x"], out[1]["x"])
    assert np.all(out[0]["status"] == 0)


def _random_qp(n, rng):
    A = rng.standard_normal((n + 50, n)) / np.sqrt(n)
    xt = np.maximum(rng.standard_normal(n), 0)
    return A.T @ A + 1e-3 * np.eye(n), -A.T @ (A @ xt)


@pytest.mark.parametrize("n,B", [(514, 1), (514, 8), (1078, 1), (1078, 2), (2049, 1), (2500, 2), (3598, 1), (4096, 1)])
def test_group_kernel_few_large_problems(n, B):
    """Few problems, or n > 2048: every problem on several co-resident workgroups (qp_group.hpp).  Against the CPU checker:
    same iteration count, x within 1e-9 of the peak (n <= 2500); the result does not depend on the number of problems in the
    launch (hence not on the group size either)."""
    from hipdrt import _ffi
    from oracle.coneqp import coneqp_boxlow
    rng = np.random.default_rng(7 * n + B)
    ctx = _ffi.get_context()
    Ps, qs = zip(*[_random_qp(n, rng) for _ in range(B)])
    Ps, qs = np.array(Ps), np.array(qs)
    h = np.zeros(n)
    h[:3] = 1000.0
    res = ctx.qp_batch(Ps, qs, h)
    assert np.all(res["status"] == 0) and np.all(np.isfinite(res["x"]))
    assert np.all(res["x"][:, 3:] > -1e-9)
    if n <= 2500:
        r = coneqp_boxlow(Ps[0], qs[0], h)
        assert r["iterations"] == res["iterations"][0]
        parity("x", res["x"][0], r["x"], default=1e-9)
    else:
        # KKT residual of the returned point instead of a (slow) CPU solve: P x + q - z = 0, z >= 0, z (x + h) ~ 0
        for b in range(B):
            x = res["x"][b]
            z = Ps[b] @ x + qs[b]
            assert z.min() > -1e-6 * np.abs(qs[b]).max() and np.abs(z * (x + h)).max() < 1e-5 * np.abs(qs[b]).max()
    # the same first problem alone / in a launch of another size: bit-identical
    more = ctx.qp_batch(np.concatenate([Ps[:1]] * 3), np.concatenate([qs[:1]] * 3), h)
    for b in range(3):
        np.testing.assert_array_equal(more["x"][b], res["x"][0])
    assert more["iterations"].tolist() == [int(res["iterations"][0])] * 3


def test_group_kernel_result_independent_of_group_size():
    """The same problems on 1, 3, 8 and 16 workgroups each (debug switch of the C-ABI): identical bits whatever the group size
    (the rows' arithmetic does not depend on who owns them); against the batch kernel (one workgroup per problem, forward
    substitution fused into the factorisation, hence another summation order): same iteration counts, x within 1e-12 of the
    peak"""
    from hipdrt import _ffi
    rng = np.random.default_rng(99)
    ctx = _ffi.get_context()
    n = 600
    Ps, qs = zip(*[_random_qp(n, rng) for _ in range(3)])
    Ps, qs = np.array(Ps), np.array(qs)
    h = np.zeros(n)
    try:
        ctx.debug_qp_group(0)                         # batch kernel
        ref = ctx.qp_batch(Ps, qs, h)
        outs = []
        for G in (1, 3, 8, 16):
            ctx.debug_qp_group(G)
            outs.append(ctx.qp_batch(Ps, qs, h))
    finally:
        ctx.debug_qp_group(-1)                        # automatic choice again
    for o in outs:
        assert o["iterations"].tolist() == ref["iterations"].tolist()
        np.testing.assert_allclose(o["x"], ref["x"], rtol=0, atol=1e-12 * np.abs(ref["x"]).max())
        np.testing.assert_array_equal(o["x"], outs[0]["x"])


def test_group_kernel_reports_breakdown():
    from hipdrt import _ffi
    ctx = _ffi.get_context()
    n = 1500
    P = np.eye(n)
    P[700, 700] = -5.0
    res = ctx.qp_batch(P[None], np.ones((1, n)), np.zeros(n))
    assert res["status"][0] < 0


@pytest.mark.timeout(600)
def test_group_launches_beside_a_batch_launch_that_fills_the_chip():
    """Co-residency is a precondition of the several-workgroups-per-problem kernel, and the device is shared: while another
    context's stream keeps every CU busy with 1024-problem batch launches (one workgroup per problem, whole LDS and register
    file of its CU), group launches from this context must neither hang nor trap -- their members either become resident
    together or the launch gives up cleanly and is repeated with one member per problem (qp_group.hpp, start rendezvous) --
    and must return the bits of the uncontended launch (results do not depend on the group size)."""
    import threading
    from hipdrt import _ffi, synth
    from hipdrt.models import DRT
    rng = np.random.default_rng(5)
    n = 600
    Ps, qs = zip(*[_random_qp(n, rng) for _ in range(2)])
    Ps, qs = np.array(Ps), np.array(qs)
    h = np.zeros(n)
    ctx = _ffi.get_context()
    quiet = ctx.qp_batch(Ps, qs, h)                    # nothing else running: the library's own choice (group kernel, 2 problems)
    c2 = synth.config_c2()
    z = synth.zarc2_batch(c2["freq"], 1024, first_seed=100)
    hog = DRT(fixed_basis_tau=c2["tau"], context=_ffi.Context(0))
    hog_plan = hog.stage_batch(c2["freq"], z)
    hog_plan.set_subbatches(1)
    stop, errors, steps = threading.Event(), [], [0]

    def keep_busy():
        try:
            while not stop.is_set():
                hog.fit_staged()                       # ~0.5 s of back-to-back full-chip launches per call
                steps[0] += 1
        except BaseException as exc:                   # noqa: BLE001 -- reported in the main thread
            errors.append(exc)

    t = threading.Thread(target=keep_busy)
    t.start()
    try:
        outs = []
        while steps[0] < 1 and t.is_alive():           # the first batch fit is under way
            time.sleep(0.01)
        for _ in range(6):
            outs.append(ctx.qp_batch(Ps, qs, h))
    finally:
        stop.set()
        t.join()
    assert not errors, errors
    assert steps[0] >= 1
    for o in outs:
        assert o["status"].tolist() == quiet["status"].tolist() and o["iterations"].tolist() == quiet["iterations"].tolist()
        np.testing.assert_array_equal(o["x"], quiet["x"])


@pytest.mark.parametrize("n", [1, 17, 33, 64, 65, 93, 123, 129, 257, 400, 497, 513, 514, 528])
def test_fat_four_wavefront_kernel_same_bits(n):
    """The fat form of the batch coneqp kernel (four wavefronts, one per SIMD, 512 registers each: accumulators in AccVGPRs,
    qp_kernel_resident<false, 256, 1, true, 2>; debug switch hipdrt_debug_qp_waves) against the eight-wavefront kernel: every
    tile receives the same MFMA sequence whoever owns it, the sweeps apply the blocks' contributions in the same order, and the
    interior-point reductions run over two VIRTUAL threads per thread -- the results are the same bits, the failure statuses too."""
    from hipdrt import _ffi
    rng = np.random.default_rng(1000 + n)
    ctx = _ffi.Context(0)
    B = 5
    Ps, qs, hs = [], [], []
    for b in range(B):
        M = rng.standard_normal((n + 3, n))
        Ps.append(M.T @ M + 0.1 * np.eye(n))
        qs.append(rng.standard_normal(n) * 3)
        hs.append(np.where(rng.random(n) < 0.8, 0.0, 1000.0))
    Ps[B - 1] = -np.eye(n)                           # not positive definite: status -1 from both
    Ps, qs, hs = np.stack(Ps), np.stack(qs), np.stack(hs)
    ctx.debug_qp_group(0)
    ctx.debug_qp_waves(8)
    ref = ctx.qp_batch(Ps, qs, hs)
    ctx.debug_qp_waves(4)
    fat = ctx.qp_batch(Ps, qs, hs)
    assert ref["status"].tolist() == [0] * (B - 1) + [-1]
    for key in ("status", "iterations"):
        assert fat[key].tolist() == ref[key].tolist()
    np.testing.assert_array_equal(fat["x"][:B - 1], ref["x"][:B - 1])
    np.testing.assert_array_equal(fat["pcost"][:B - 1], ref["pcost"][:B - 1])


def test_fat_kernel_whole_fits_same_bits():
    """64 config-2 spectra (n = 514) and 8 spectra on the reference's default 91-point grid through the whole QPHB loop with
    either kernel: identical x, weights, rho, iteration counts, posterior variances"""
    from hipdrt import _ffi, synth
    from hipdrt.models import DRT
    out = {}
    for waves in (8, 4):
        ctx = _ffi.Context(0)
        ctx.debug_qp_waves(waves)
        c2 = synth.config_c2()
        d = DRT(fixed_basis_tau=c2["tau"], context=ctx)
        r = d.fit_eis_batch(c2["freq"], synth.zarc2_batch(c2["freq"], 64))
        v = d.estimate_distribution_var_batch(c2["tau"][::8])
        f = np.logspace(6, -1, 71)
        r1 = DRT(context=ctx).fit_eis_batch(f, synth.zarc2_batch(f, 8, first_seed=3))
        out[waves] = [r["x"], r["weights"], r["rho"], r["outer_iters"], r["qp_iters_total"],
                      np.asarray(v[0] if isinstance(v, tuple) else v), r1["x"], r1["qp_iters_total"]]
    for a, b in zip(out[8], out[4]):
        np.testing.assert_array_equal(a, b)
