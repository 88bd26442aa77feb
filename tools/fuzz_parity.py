"""Runs the randomised differential tests of tests/test_gpu_hybrid.py for many more seeds than the suite does and reports the
seeds that fail (diagnostic; needs a GPU).  python tools/fuzz_parity.py [first] [count]
--eis: the EIS differential test instead.
--c2 [--count N] [--procs P]: the FULL-SIZE workload of BASELINE configs[2] (256 x 512, spectra seeds first .. first+N-1 of the
bench's batch) fitted as one device batch and EVERY spectrum compared with the CPU checker (oracle/drt_oracle.py, a forked
process pool on the host, one BLAS thread each, started before the GPU is touched): outer and interior-point iteration
counts, max |dx| / peak, and the spectra that ran into max_iter."""
import os, sys, traceback
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import warnings
warnings.filterwarnings("ignore")
import test_gpu_hybrid as t

def _c2_mode():
    import multiprocessing as mp, time
    import numpy as np
    from hipdrt import synth
    from oracle import drt_oracle as orc
    av = sys.argv[1:]
    count = int(av[av.index("--count") + 1]) if "--count" in av else 256
    first = int(av[av.index("--first") + 1]) if "--first" in av else 0
    procs = int(av[av.index("--procs") + 1]) if "--procs" in av else min(os.cpu_count() or 1, count, 128)
    c2 = synth.config_c2()
    z = synth.zarc2_batch(c2["freq"], count, first_seed=first)
    od = orc.OracleDRT(fixed_basis_tau=c2["tau"])
    od.prepare(c2["freq"])
    glob = globals()
    glob["_C2"] = (od, c2["freq"], z)
    t0 = time.time()
    with mp.get_context("fork").Pool(procs) as pool:        # before anything GPU-side exists in this process
        ref = pool.map(_c2_one, range(count), chunksize=1)
    t_cpu = time.time() - t0
    from hipdrt.models import DRT
    drt = DRT(fixed_basis_tau=c2["tau"])
    t0 = time.time()
    res = drt.fit_eis_batch(c2["freq"], z)
    t_gpu = time.time() - t0
    rx = np.array([r[0] for r in ref]); ro = np.array([r[1] for r in ref]); rq = np.array([r[2] for r in ref])
    err = np.abs(res["x"] - rx).max(axis=1) / np.abs(rx).max(axis=1)
    same_o, same_q = res["outer_iters"] == ro, res["qp_iters_total"] == rq
    print(f"c2 full size: {count} spectra (seeds {first}..{first + count - 1}), oracle pool of {procs}: {t_cpu:.1f} s, device batch: {t_gpu:.2f} s")
    print(f"outer iterations identical: {int(same_o.sum())}/{count}; interior-point totals identical: {int(same_q.sum())}/{count}")
    print(f"max |dx| / peak: median {np.median(err):.2e}, 99th pct {np.quantile(err, 0.99):.2e}, max {err.max():.2e} (spectrum {int(err.argmax())})")
    print(f"spectra at max_iter (50): device {int((res['outer_iters'] >= 50).sum())}, oracle {int((ro >= 50).sum())}")
    worst = np.argsort(-err)[:8]
    for b in worst:
        print(f"  spectrum {int(b) + first}: err {err[b]:.2e}, outer {int(res['outer_iters'][b])} / {int(ro[b])}, ipm {int(res['qp_iters_total'][b])} / {int(rq[b])}")
    bad = np.flatnonzero((err > 1e-7) | ~same_o | ~same_q)
    print(f"beyond 1e-7 or with different counts: {len(bad)} {bad[:20].tolist()}")


def _c2_one(i):
    od, freq, z = globals()["_C2"]
    od.fit_eis(freq, z[i], structure='fast', keep_history=True)
    return (od.qphb_params["x_scaled"].copy(), int(od.qphb_params["outer_iterations"]),
            int(sum(l["iterations"] for l in od.qp_log)))


if "--c2" in sys.argv:
    _c2_mode()
    sys.exit(0)

args = [a for a in sys.argv[1:] if not a.startswith("--")]
first = int(args[0]) if len(args) > 0 else 100
count = int(args[1]) if len(args) > 1 else 100
bad = []
if "--eis" in sys.argv:
    import test_gpu_fit as tf
    fn = getattr(tf.test_randomized_fits_vs_oracle, "__wrapped__", tf.test_randomized_fits_vs_oracle)
    nfail = 0
    for seed in range(first, first + count):
        try:
            fn(seed)
        except Exception as e:          # noqa: BLE001
            nfail += 1
            bad.append(("test_randomized_fits_vs_oracle", seed, str(e).splitlines()[0][:200]))
    print(f"test_randomized_fits_vs_oracle: {count - nfail}/{count} seeds ok", flush=True)
    for b in bad:
        print("FAIL", b)
    sys.exit(0)
for name in ("test_randomised_joint_fits_follow_the_oracle", "test_randomised_option_combinations_follow_the_oracle",
             "test_randomised_joint_fits_with_option_combinations"):
    fn = getattr(t, name)
    fn = getattr(fn, "__wrapped__", fn)
    nfail = 0
    for seed in range(first, first + count):
        try:
            fn(seed)
        except Exception as e:          # noqa: BLE001
            nfail += 1
            bad.append((name, seed, str(e).splitlines()[0][:200]))
    print(f"{name}: {count - nfail}/{count} seeds ok", flush=True)
for b in bad:
    print("FAIL", b)
