import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the library is a build artefact (git-ignored): if this checkout has none yet, build it the way build() does
    # (a failed build must not take the pure-numpy tests down with it: the tests that need the library fail on their own)
    lib = os.path.join(ROOT, "hybrid-drt_amd", "libhipdrt.so")
    if not os.path.exists(lib) and "HIPDRT_LIB" not in os.environ:
        try:
            import __graft_entry__
            __graft_entry__.build()
        except Exception as e:          # noqa: BLE001  (no hipcc on this box, or a compile error)
            config.hipdrt_build_error = e
            sys.stderr.write(f"conftest: building libhipdrt.so failed ({e}); tests that load it will fail\n")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ---- parity bounds that follow what is measured -----------------------------------------------------------------------------
# parity_close() is assert_allclose(rtol=0, atol=bound * scale) that also RECORDS the deviation it found: every call appends
# "label measured bound" to tests/_parity_measured.txt (and to gpurun_out/ when that directory exists, so the record of a GPU
# run travels back), and DESIGN.md section 2 tabulates bound against measured.  Policy: a bound sits at
# about ten times the measured deviation; where the arithmetic itself needs more (a QP that stops at coneqp's start point is a
# direct solve, cond * eps in any implementation), the reason is stated at the call.
_PARITY_LOG = []


def parity_close(label, actual, desired, bound, scale=None):
    import numpy as np
    actual, desired = np.asarray(actual, dtype=float), np.asarray(desired, dtype=float)
    assert actual.shape == desired.shape, (label, actual.shape, desired.shape)
    sc = float(np.abs(desired).max()) if scale is None else float(scale)
    err = float(np.abs(actual - desired).max()) / max(sc, 1e-300)
    _PARITY_LOG.append((label, err, bound))
    print(f"parity {label}: measured {err:.2e} bound {bound:.1e}")
    assert err <= bound, f"{label}: deviation {err:.3e} of the peak exceeds the bound {bound:.1e}"
    return err


# parity(): the same record-and-assert, for the many per-quantity checks of the fixture tests.  The label is "<test id>:<quantity>";
# its bound comes from tests/parity_bounds.json, a committed table "label -> [measured on the GPU, bound]" that
# tools/update_parity_bounds.py rewrites from a GPU run's record (bound = 20 x measured, rounded up to 1 / 2 / 5 x 10^k, never
# below the noise floor of the metric).  A label the table does not know yet is asserted at `default` (the documented tolerance
# of that quantity) and shows up as "default" in the record, so that the next table update picks it up.
#   rel=False: max |a - d| / max |d|                  (coefficients: fraction of the largest one)
#   rel=True : max |a - d| / max(|d|, floor * max|d|) (weights, s vectors, sigma: element by element, entries below `floor`
#              of the largest measured against that floor)
_BOUNDS = None


def _bounds_table():
    global _BOUNDS
    if _BOUNDS is None:
        import json
        try:
            with open(os.path.join(ROOT, "tests", "parity_bounds.json")) as f:
                _BOUNDS = json.load(f)
        except OSError:
            _BOUNDS = {}
    return _BOUNDS


def current_test_id():
    t = os.environ.get("PYTEST_CURRENT_TEST", "unknown").split(" ")[0]
    return t.split("::")[-1]


def parity(quantity, actual, desired, default=1e-7, rel=False, floor=1e-3, scale=None, bound=None, label=None):
    import numpy as np
    label = label or f"{current_test_id()}:{quantity}"
    actual, desired = np.asarray(actual), np.asarray(desired)
    if np.iscomplexobj(actual) or np.iscomplexobj(desired):     # sigma of an impedance: real and imaginary parts side by side
        actual = np.stack([np.real(actual), np.imag(actual)], axis=-1)
        desired = np.stack([np.real(desired), np.imag(desired)], axis=-1)
    actual, desired = np.asarray(actual, dtype=float), np.asarray(desired, dtype=float)
    assert actual.shape == desired.shape, (label, actual.shape, desired.shape)
    peak = float(np.abs(desired).max()) if scale is None else float(scale)
    if rel:
        den = np.maximum(np.abs(desired), floor * peak)
        err = float((np.abs(actual - desired) / np.maximum(den, 1e-300)).max()) if actual.size else 0.0
    else:
        err = float(np.abs(actual - desired).max()) / max(peak, 1e-300) if actual.size else 0.0
    src = "call"
    if bound is None:
        entry = _bounds_table().get(label)
        bound, src = (float(entry[1]), "table") if entry is not None else (default, "default")
    _PARITY_LOG.append((label, err, bound))
    print(f"parity {label}: measured {err:.2e} bound {bound:.1e} ({src})")
    assert np.isfinite(err) and err <= bound, f"{label}: deviation {err:.3e} exceeds the bound {bound:.1e} ({src})"
    return err


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY_LOG:
        return
    worst = {}
    for label, err, bound in _PARITY_LOG:
        if label not in worst or err > worst[label][0]:
            worst[label] = (err, bound)
    lines = ["# label | worst measured deviation (fraction of the scale) | asserted bound | bound / measured"]
    for label in sorted(worst):
        err, bound = worst[label]
        lines.append(f"{label} | {err:.2e} | {bound:.1e} | {bound / max(err, 1e-300):.0f}x")
    text = "\n".join(lines) + "\n"
    for d in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "_parity_measured.txt"), "w") as f:
                    f.write(text)
            except OSError:
                pass
