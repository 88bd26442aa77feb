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
