"""Runs the randomised differential tests of tests/test_gpu_hybrid.py for many more seeds than the suite does and reports the
seeds that fail (diagnostic; needs a GPU).  python tools/fuzz_parity.py [first] [count]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import warnings
warnings.filterwarnings("ignore")
import test_gpu_hybrid as t

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
